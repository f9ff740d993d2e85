"""Delay-pattern bookkeeping of parler_tts_runner on the host (numpy mirror of host/parler_runner.cpp
for bench/tests; the C++ runner is the product implementation).

adjust_output_tokens (/root/reference/src/models/parler/model.cpp:734-760): frame i takes head k's
token from step i+k; frames that would read past the end, or that contain an id >= audio_vocab
(EOS/BOS/pad), are dropped."""
import numpy as np


def undelay(tokens, audio_vocab):
    """tokens [steps][n_out] (still delayed) -> frames [n_frames][n_out] uint32"""
    tokens = np.asarray(tokens, dtype=np.uint32)
    steps, n_out = tokens.shape
    n = steps - n_out + 1
    if n <= 0:
        return np.zeros((0, n_out), dtype=np.uint32)
    idx = np.arange(n)[:, None] + np.arange(n_out)[None, :]
    frames = tokens[idx, np.arange(n_out)[None, :]]
    keep = (frames < audio_vocab).all(axis=1)
    return np.ascontiguousarray(frames[keep])


def next_ids(step, tokens, eos_seen, bos, eos):
    """model.cpp:778-785 for the step `step` (>=1) that just produced `tokens`"""
    n_out = len(tokens)
    return np.array([(eos if eos_seen[i] else tokens[i]) if step > i else bos for i in range(n_out)], dtype=np.uint32)

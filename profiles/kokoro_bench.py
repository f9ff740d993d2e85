#!/usr/bin/env python3
"""Kokoro at the real shapes (hexgrad/Kokoro-82M: ALBERT 768 x 12 recurrences, predictor / text encoder width 512, decoder 1024,
generator 512 -> 256 -> 128 with (10, 6) upsampling, n_fft 20 / hop 5), synthetic weights: BASELINE's configuration — 64 and 400
phoneme ids with the durations forced to fixed integers for shape determinism.  Prints the time of the duration graph and of the
generation graph and the real-time factor (600 samples per duration frame at 24 kHz)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tts_cpp_amd  # noqa: F401
from tts_cpp_amd import hip, synth

t0 = time.perf_counter()
model = synth.build_kokoro(synth.kokoro_82m())
eng = hip.KokoroEngine(model)
print(f"model ready in {time.perf_counter() - t0:.1f}s ({sum(len(t.raw()) for t in model.tensors) / 1e6:.0f} MB)", flush=True)
rng = np.random.default_rng(1)
cfg = model.cfg
for n_ids, dur in ((64, 3), (400, 3)):
    toks = np.concatenate([[0], rng.integers(1, cfg.vocab, n_ids), [0]]).astype(np.uint32)
    eng.durations(toks, cfg.voices[0])
    t0 = time.perf_counter()
    lens, hid = eng.durations(toks, cfg.voices[0])
    t_dur = time.perf_counter() - t0
    forced = np.full(toks.size, float(dur), dtype=np.float32)
    total = int(forced.sum())
    noise = rng.random((cfg.harmonic_num + 1) * total * cfg.up_sampling_factor, dtype=np.float32)
    eng.generate(toks, forced, hid, cfg.voices[0], noise)
    t0 = time.perf_counter()
    pcm = eng.generate(toks, forced, hid, cfg.voices[0], noise)
    t_gen = time.perf_counter() - t0
    audio_s = pcm.size / 24000.0
    print(f"{n_ids} phoneme ids x {dur} frames: durations {t_dur * 1e3:.1f} ms, generation {t_gen * 1e3:.1f} ms for {audio_s:.2f} s of audio "
          f"-> {audio_s / (t_dur + t_gen):.1f}x real time")

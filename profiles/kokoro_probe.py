"""One Kokoro-82M synthesis of 400 phoneme ids (durations forced to 3 frames), a few repetitions: for a kernel trace (rocprofv3 --kernel-trace --stats)."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import tts_cpp_amd
from tts_cpp_amd import hip, synth
model = synth.build_kokoro(synth.kokoro_82m()); cfg = model.cfg
eng = hip.KokoroEngine(model)
rng = np.random.default_rng(1)
toks = np.concatenate([[0], rng.integers(1, cfg.vocab, 400), [0]]).astype(np.uint32)
forced = np.full(toks.size, 3.0, dtype=np.float32)
noise = rng.random((cfg.harmonic_num + 1) * int(forced.sum()) * cfg.up_sampling_factor, dtype=np.float32)
for i in range(4):
    t0 = time.perf_counter(); lens, hid = eng.durations(toks, cfg.voices[0]); t1 = time.perf_counter()
    pcm = eng.generate(toks, forced, hid, cfg.voices[0], noise); t2 = time.perf_counter()
    print(f"durations {1e3 * (t1 - t0):.2f} ms, generation {1e3 * (t2 - t1):.2f} ms, audio {pcm.size / 24000.0:.1f} s")

#!/usr/bin/env python3
"""T5 voice-prompt encoder (flan-t5-large shapes: 24 layers, d_model 1024, d_ff 2816, 16 heads; synthetic weights):
time of one update_conditional_prompt-sized encode on the GPU and with the CPU oracle beside it."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import tts_cpp_amd
from tts_cpp_amd import gguf, hip, synth
import oracle as orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
wt = {"f16": gguf.F16, "f32": gguf.F32, "q8_0": gguf.Q8_0}[sys.argv[2] if len(sys.argv) > 2 else "f16"]
t0 = time.perf_counter()
model = synth.build_t5(synth.t5_flan_large(vocab=4096, weight_type=wt))
print(f"model built in {time.perf_counter() - t0:.1f}s")
eng = hip.T5Engine(model.cfg); eng.load(model)
ids = np.random.default_rng(0).integers(3, 4096, n).astype(np.uint32)
out = eng.encode(ids)
t0 = time.perf_counter()
for _ in range(5): out = eng.encode(ids)
gpu = (time.perf_counter() - t0) / 5
o = orc.T5Oracle(model)
t0 = time.perf_counter(); ref = o.encode(ids); cpu = time.perf_counter() - t0
print(f"tokens={n} gpu {gpu*1e3:.2f} ms/encode  cpu oracle ({orc.default_threads()} threads) {cpu*1e3:.0f} ms  ratio {cpu/gpu:.0f}x  relerr {np.abs(out-ref).max()/np.abs(ref).max():.2e}")

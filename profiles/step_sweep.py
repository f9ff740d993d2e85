#!/usr/bin/env python3
"""SURVEY §8(d) metric 2: ms per decode step of the Parler-TTS-Mini decoder (fp16 weights, fp32 KV cache, batch 1) as the
cache grows, T ~ {128, 512, 1024, 2580}.  Two ways through the boundary:
  loop : tts_hip_parler_generate_greedy, the product path (ids stay on the device, one step = one hipGraph replay)
  step : tts_hip_parler_step per token with the 39 KB of logits copied to the host and arg-max there — the shape of the
         reference's decode() + sampler (parler/model.cpp:648-693)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tts_cpp_amd  # noqa: F401
from tts_cpp_amd import gguf, hip, synth

cfg = synth.parler_mini(weight_type=gguf.F16)
t0 = time.perf_counter()
model = synth.build(cfg)
eng = hip.HipEngine(cfg, device=0, max_seqs=1, kv_type=gguf.F32, kv_positions=min(cfg.ctx, cfg.max_gen))
eng.load(model)
print(f"model ready in {time.perf_counter() - t0:.1f}s", flush=True)
prompt = np.random.default_rng(3).integers(3, cfg.prompt_vocab, 16).astype(np.uint32)
P = len(prompt)
NMAX = cfg.max_gen - P

eng.prefill_batch([prompt]); eng.generate_greedy([P], 32)          # warm-up (graph capture)
print("loop: N steps from T=16 -> mean ms/step over steps 1..N")
prev = (0, 0.0)
for n in (128, 512, 1024, NMAX):
    eng.reset(); eng.prefill_batch([prompt])
    t0 = time.perf_counter()
    eng.generate_greedy([P], n)
    dt = time.perf_counter() - t0
    print(f"  N={n:5d} (T up to {P + n:4d}): {dt / n * 1e3:.3f} ms/step; steps {prev[0]}..{n}: {(dt - prev[1]) / (n - prev[0]) * 1e3:.3f} ms/step", flush=True)
    prev = (n, dt)

eng.reset(); eng.prefill_batch([prompt])
ids = np.full((1, cfg.n_out), cfg.bos, dtype=np.uint32)
ts = np.zeros(NMAX)
for s in range(NMAX):
    t0 = time.perf_counter()
    lg = eng.step(ids, [P + s])
    ids = lg.argmax(-1).astype(np.uint32)
    ts[s] = time.perf_counter() - t0
print("step: per-token call + logits D2H + host arg-max, mean over a 64-step window around T")
for T in (128, 512, 1024, NMAX + P - 33):
    s = T - P
    print(f"  T~{T:4d}: {ts[max(0, s - 32):s + 32].mean() * 1e3:.3f} ms/step")
print(f"  all {NMAX} steps: {ts.mean() * 1e3:.3f} ms/step")

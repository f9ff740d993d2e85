#!/usr/bin/env python3
"""Batch-1 Parler-TTS-Mini greedy loop (fp16 weights, fp32 KV) for a kernel trace: 16-token prompt + N steps."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tts_cpp_amd  # noqa: F401
from tts_cpp_amd import gguf, hip, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cfg = synth.parler_mini(weight_type=gguf.F16)
model = synth.build(cfg)
eng = hip.HipEngine(cfg, device=0, max_seqs=1, kv_type=gguf.F32, kv_positions=min(cfg.ctx, cfg.max_gen), flags=hip.FLAG_NO_GRAPH if os.environ.get('B1_NO_GRAPH') else 0)
eng.load(model)
prompt = np.random.default_rng(3).integers(3, cfg.prompt_vocab, 16).astype(np.uint32)
eng.prefill_batch([prompt]); eng.generate_greedy([len(prompt)], 32)
eng.reset(); eng.prefill_batch([prompt])
t0 = time.perf_counter()
eng.generate_greedy([len(prompt)], N)
dt = time.perf_counter() - t0
print(f"N={N}: {dt / N * 1e3:.3f} ms/step = {1 / (dt / N) / 86.13:.2f}x real time")
# per-class times of the eager forward (events around every launch; no graph)
ids = np.full((1, cfg.n_out), cfg.bos, dtype=np.uint32)
P = len(prompt)
for s in range(8): eng.step(ids, [P + N + s])
eng.profile(True)
K = 64
for s in range(K): eng.step(ids, [P + N + 8 + s])
st = eng.profile_get(); eng.profile(False)
tot = sum(v["ms_total"] for v in st.values())
print(f"eager, T~{P + N}: {tot / K * 1e3:.1f} us of kernel time per step")
for k, v in sorted(st.items(), key=lambda kv: -kv[1]["ms_total"]):
    if v["launches"]: print(f"  {k:16s} {v['ms_total'] / v['launches'] * 1e3:7.2f} us x {v['launches'] // K:3d}/step  {v['ms_total'] / tot * 100:5.1f}%  {v['bytes_total'] / v['ms_total'] / 1e6 if v['ms_total'] else 0:7.0f} GB/s")

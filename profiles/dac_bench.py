#!/usr/bin/env python3
"""DAC-only micro-benchmark (full DAC-44k dims, synthetic weights): decode `frames` frames `reps` times."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tts_cpp_amd
from tts_cpp_amd import gguf, hip, synth
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 248
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
flags = hip.FLAG_NO_PARLER | (hip.FLAG_VALU_GEMM if "--valu" in sys.argv else 0)
cfg = synth.parler_mini(layers=1, prompt_vocab=64, ctx=64, dac_f16="--f16" in sys.argv)
model = synth.build(cfg)
tune = {}
for a in sys.argv:
    if a.startswith("--tune="):          # --tune=dac_convt_planes=0,dac_fuse=0
        tune.update({kv.split("=")[0]: int(kv.split("=")[1]) for kv in a[len("--tune="):].split(",")})
eng = hip.HipEngine(cfg, flags=flags, tune=tune)
eng.load(model)
batch = 1
for a in sys.argv:
    if a.startswith("--batch="):
        batch = int(a.split("=")[1])
rng = np.random.default_rng(0)
codes = [rng.integers(0, cfg.cb_size, (frames, cfg.n_out)).astype(np.uint32) for _ in range(batch)]
if "--no-warmup" not in sys.argv:
    eng.dac_decode_batch(codes)
t0 = time.perf_counter()
for _ in range(reps):
    eng.dac_decode_batch(codes)
dt = (time.perf_counter() - t0) / reps
print(f"batch={batch} frames={frames} {dt*1e3:.2f} ms/decode  {1.608e9*frames*batch/dt/1e12:.1f} TFLOP/s  {batch*frames*512/44100/dt:.1f}x real-time")
if "--prof" in sys.argv:
    eng.profile(True); eng.dac_decode_batch(codes); st = eng.profile_get(); eng.profile(False)
    for k, v in st.items():
        if v["launches"]:
            print(k, v["launches"], f'{v["ms_total"]:.3f} ms', f'{v["flops_total"]/max(v["ms_total"],1e-9)/1e9:.2f} TF')

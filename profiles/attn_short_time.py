#!/usr/bin/env python3
"""Cross-attention launch time at 1024 lock-step rows (eager steps, HIP events around every launch) and the logits of the last step:
   python profiles/attn_short_time.py [out.npy]        (the 4-lanes-per-key experiment it was written for has been removed: profiles/r03/attn_short16_rejected.txt)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tts_cpp_amd  # noqa: F401
from tts_cpp_amd import gguf, hip, synth
R, N = 1024, 12
cfg = synth.parler_mini(weight_type=gguf.F16, max_gen=16 + N + 4)
model = synth.build(cfg)
eng = hip.HipEngine(cfg, device=0, max_seqs=R, kv_type=gguf.F32, kv_positions=16 + N + 4, flags=hip.FLAG_NO_GRAPH | hip.FLAG_NO_DAC)
eng.load(model)
rng = np.random.default_rng(3)
eng.prefill_batch([rng.integers(3, cfg.prompt_vocab, 16).astype(np.uint32) for _ in range(R)])
ids = np.full((R, cfg.n_out), cfg.bos, dtype=np.uint32)
for s in range(4): lg = eng.step(ids, [16 + s] * R)
eng.profile(True)
for s in range(4, N): lg = eng.step(ids, [16 + s] * R)
st = eng.profile_get(); eng.profile(False)
for k in ("attn_cross", "attn_self", "ln", "gemm_qkv"):
    v = st[k]
    print(f"{k:12s} {v['ms_total'] / v['launches'] * 1e3:7.2f} us per launch ({v['launches']} launches)")
if len(sys.argv) > 1: np.save(sys.argv[1], np.asarray(lg)[:8])

#!/usr/bin/env python3
"""The decoder loop of the headline workload alone (1024 lock-step utterances, 16-id prompts, 256 audio steps, fp16 weights, fp32 KV cache, the
device-resident greedy loop = hipGraph replays): ms per step under tts_hip_tune keys given as KEY=VALUE arguments.

  python profiles/dec_loop.py [rows] [steps] [key=value ...] [wt=f16|q8_0|q5_0|q4_0] [prof=1]
wt: GGUF type of the decoder matrices; prof=1: one more pass of 8 eager steps with per-launch events, by kernel class."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tts_cpp_amd  # noqa: F401
from tts_cpp_amd import gguf, hip, synth
args = [a for a in sys.argv[1:] if "=" not in a]
kv = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
WT = {"f16": gguf.F16, "q8_0": gguf.Q8_0, "q5_0": gguf.Q5_0, "q4_0": gguf.Q4_0}[kv.pop("wt", "f16")]
PROF = int(kv.pop("prof", "0"))
tune = {k: int(v) for k, v in kv.items()}
R = int(args[0]) if args else 1024
N = int(args[1]) if len(args) > 1 else 256
cfg = synth.parler_mini(weight_type=WT, max_gen=16 + N)
model = synth.build(cfg)
eng = hip.HipEngine(cfg, device=0, max_seqs=R, kv_type=gguf.F32, kv_positions=16 + N, flags=hip.FLAG_NO_DAC, tune=tune)
eng.load(model)
rng = np.random.default_rng(3)
prompts = [rng.integers(3, cfg.prompt_vocab, 16).astype(np.uint32) for _ in range(R)]
best = 1e9
for rep in range(3):
    eng.reset(); eng.prefill_batch(prompts)
    t0 = time.perf_counter()
    toks, _ = eng.generate_greedy([16] * R, N)
    best = min(best, time.perf_counter() - t0)
print(f"{R} rows x {N} steps, tune {tune}: {best / N * 1e3:.3f} ms/step (best of 3); token checksum {int(toks.astype(np.uint64).sum())}")
if PROF:
    eng.reset(); eng.prefill_batch(prompts)
    toks, _ = eng.generate_greedy([16] * R, 64)
    eng.profile(True)
    ids = toks[-1]
    for st in range(8):
        ids = eng.step_greedy(ids, [16 + 64 + st] * R)
    st = eng.profile_get()
    eng.profile(False)
    tot = sum(v["ms_total"] for v in st.values())
    for k, v in sorted(st.items(), key=lambda kv: -kv[1]["ms_total"]):
        if v["launches"]:
            print(f"  {k:18s} {v['ms_total'] / 8 * 1e3:9.1f} us/step  {v['launches'] // 8:4d} launches/step  {v['ms_total'] / v['launches'] * 1e3:8.2f} us each  {100 * v['ms_total'] / tot:5.1f} %")
    print(f"  total {tot / 8:.3f} ms/step (eager, event-timed)")

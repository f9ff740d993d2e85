#!/usr/bin/env python3
"""The decoder of the bench's default workload, launch by launch (no hipGraph: rocprofv3's counter collection cannot follow graph replays):
1024 lock-step utterances, 16-id prompts, 256 audio steps with the fp32 KV cache growing from 16 to 272 positions.  Run under
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); profiles/pmc_summary.py turns the counter CSVs into profiles/pmc_traffic.json."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tts_cpp_amd  # noqa: F401
from tts_cpp_amd import gguf, hip, synth
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg = synth.parler_mini(weight_type=gguf.F16, max_gen=16 + N)
model = synth.build(cfg)
eng = hip.HipEngine(cfg, device=0, max_seqs=R, kv_type=gguf.F32, kv_positions=16 + N, flags=hip.FLAG_NO_GRAPH | hip.FLAG_NO_DAC)
eng.load(model)
rng = np.random.default_rng(3)
prompts = [rng.integers(3, cfg.prompt_vocab, 16).astype(np.uint32) for _ in range(R)]
eng.prefill_batch(prompts)
ids = np.full((R, cfg.n_out), cfg.bos, dtype=np.uint32)
t0 = time.perf_counter()
for s in range(N):
    ids = eng.step_greedy(ids, [16 + s] * R) if hasattr(eng, "step_greedy") else eng.step(ids, [16 + s] * R).argmax(-1).astype(np.uint32)
print(f"{R} rows x {N} eager steps: {(time.perf_counter() - t0) / N * 1e3:.2f} ms/step")

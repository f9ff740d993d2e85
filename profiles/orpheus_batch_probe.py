import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/profiles")
import tts_cpp_amd
from tts_cpp_amd import gguf, hip, synth
import secondary_bench as sb
cfg = synth.orpheus_3b(ctx=1024, weight_type=gguf.Q4_0)
rng = np.random.default_rng(7)
tensors, per_layer = sb.orpheus_tensors(cfg, rng)
NO_STOP = 0xFFFFFFFF
BS = [int(x) for x in os.environ.get("PROBE_B", "8,16").split(",")]
TUNES = [dict(kv.split("=") for kv in t.split("+")) for t in os.environ.get("PROBE_TUNE", "q_stream=0,q_stream=1").split(",")]
for B in BS:
    for tune in TUNES:
        eng = hip.OrpheusEngine(cfg, max_seqs=B)
        for k, v in tune.items(): eng.tune(k, int(v))
        eng.load(sb._Model(cfg, tensors))
        prompts = [rng.integers(0, cfg.vocab, 32).astype(np.uint32) for _ in range(B)]
        eng.generate_batch(prompts, 8, NO_STOP)
        t1 = time.perf_counter(); eng.generate_batch(prompts, 16, NO_STOP); t16 = time.perf_counter() - t1
        t1 = time.perf_counter(); eng.generate_batch(prompts, 80, NO_STOP); tn = time.perf_counter() - t1
        print(B, tune, "ms/step", (tn - t16) / 64 * 1e3, flush=True)
        if len(sys.argv) > 1:
            eng.profile(True) if hasattr(eng, "profile") else None
        eng.close()

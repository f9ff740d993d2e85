#!/usr/bin/env python3
"""Orpheus decoder step at the real shapes (canopylabs/orpheus-3b: 28 layers, hidden 3072, 24:8 heads x 128, ffn 8192,
vocab 156 940; all matrices Q4_0 = BASELINE config 4).  Timing does not depend on the weight values, so the Q4_0
blocks are random bytes with a fixed fp16 scale (minting 3.8 G normals in numpy would cost more box time than the run).
Prints ms per decode step (difference of two greedy runs of different length) and the Q4_0 bytes one step must read."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tts_cpp_amd  # noqa: F401
from tts_cpp_amd import gguf, hip, synth

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 28
cfg = synth.orpheus_3b(layers=layers, ctx=int(os.environ.get("ORPHEUS_BENCH_CTX", "1024")), weight_type=gguf.Q4_0)
rng = np.random.default_rng(7)


def q4(name, rows, cols):
    nb = rows * cols // 32
    b = rng.integers(0, 256, size=(nb, 18), dtype=np.uint8)
    b[:, 0], b[:, 1] = 0x00, 0x1C          # d = 2^-8 as fp16: values in [-8, 7] / 256
    return gguf.Tensor("orpheus." + name, gguf.Q4_0, [cols, rows], b.reshape(-1))


def f32(name, arr):
    return gguf.Tensor.from_array("orpheus." + name, np.asarray(arr, dtype=np.float32))


class Model:
    pass


t0 = time.perf_counter()
H, F, QH, KVH = cfg.hidden, cfg.ffn, cfg.heads * cfg.head_dim, cfg.kv_heads * cfg.head_dim
m = Model()
m.cfg = cfg
m.tensors = [q4("embed_tokens", cfg.vocab, H)]
per_layer = 0
for l in range(cfg.layers):
    p = f"layers.{l}."
    for nm, r, c in (("self_attn.q_proj", QH, H), ("self_attn.k_proj", KVH, H), ("self_attn.v_proj", KVH, H), ("self_attn.o_proj", H, QH),
                     ("mlp.gate_proj", F, H), ("mlp.up_proj", F, H), ("mlp.down_proj", H, F)):
        m.tensors.append(q4(p + nm, r, c))
        per_layer += r * c if l == 0 else 0
    m.tensors += [f32(p + "input_layernorm", np.ones(H)), f32(p + "post_attention_layernorm", np.ones(H))]
m.tensors += [f32("norm", np.ones(H)), q4("lm_head", cfg.vocab, H), f32("rope_frequencies", synth.llama3_rope_factors(cfg.head_dim))]
print(f"{sum(len(t.raw()) for t in m.tensors) / 1e9:.2f} GB of tensors minted in {time.perf_counter() - t0:.1f}s", flush=True)

eng = hip.OrpheusEngine(cfg)
for kv in os.environ.get("ORPHEUS_TUNE", "").split(","):   # e.g. ORPHEUS_TUNE=attn_split=4
    if "=" in kv:
        eng.tune(kv.split("=")[0], int(kv.split("=")[1]))
t0 = time.perf_counter()
eng.load(m)
print(f"loaded in {time.perf_counter() - t0:.1f}s", flush=True)
prompt = rng.integers(0, cfg.vocab, 32).astype(np.uint32)
NO_STOP = 0xFFFFFFFF
eng.generate_greedy(prompt, 16, NO_STOP)       # warm-up (graph capture, attribute calls)
res = {}
for n in (64, 448):
    t0 = time.perf_counter()
    out = eng.generate_greedy(prompt, n, NO_STOP)
    res[n] = time.perf_counter() - t0
    assert len(out) == n
step = (res[448] - res[64]) / (448 - 64)
params = cfg.layers * per_layer + cfg.vocab * H           # matrices one decode step reads (lm_head included, embed row excluded)
q4_bytes = params / 32 * 18
print(f"layers={cfg.layers} prompt 32 + 64 tokens {res[64]*1e3:.1f} ms, + 448 tokens {res[448]*1e3:.1f} ms -> {step*1e3:.3f} ms/step at positions 96..480 "
      f"({1/step:.0f} tokens/s = {1/step/7*2048/24000:.2f}x real time: 7 tokens per 2048-sample SNAC frame at 24 kHz)")
print(f"Q4_0 bytes per step {q4_bytes/1e9:.3f} GB -> {q4_bytes/step/1e9:.0f} GB/s algorithmic ({q4_bytes/step/8e12*100:.1f}% of 8 TB/s); "
      f"HBM floor {q4_bytes/8e12*1e3:.3f} ms/step")

if os.environ.get("ORPHEUS_BENCH_LONG"):   # the back half of a long utterance: histories beyond the first 512 keys of the decode attention
    rl = {}
    for n in (1088, 1536):
        t0 = time.perf_counter()
        out = eng.generate_greedy(prompt, n, NO_STOP)
        rl[n] = time.perf_counter() - t0
    print(f"long: {(rl[1536] - rl[1088]) / 448 * 1e3:.3f} ms/step at positions 1120..1568", flush=True)

if os.environ.get("ORPHEUS_BENCH_GREEDY_ONLY"):   # counter passes (profiles/r04/scripts/r4_pmc_secondary.sh): the arg-max loop only
    sys.exit(0)
# the default generation_configuration samples (top_k 50, temperature 1, top_p 1): sampler::sample over the 156 940 logits on the device
u = rng.random(448, dtype=np.float32)
eng.generate_sampled(prompt, 16, NO_STOP, u[:16], top_k=50, repetition_penalty=1.1)
rs = {}
for n in (64, 448):
    t0 = time.perf_counter()
    out = eng.generate_sampled(prompt, n, NO_STOP, u[:n], top_k=50, temperature=0.6, repetition_penalty=1.1)
    rs[n] = time.perf_counter() - t0
    assert len(out) == n
sstep = (rs[448] - rs[64]) / (448 - 64)
print(f"sampled on the device (top_k 50, temperature 0.6, repetition penalty 1.1): {sstep*1e3:.3f} ms/step ({(sstep - step)*1e6:+.0f} us vs the arg-max loop)")
# nucleus sampling (top_p < 1): + softmax_total_kernel, the full-vocabulary softmax total accumulated in index order by one thread (round 5)
tp = float(os.environ.get("ORPHEUS_BENCH_TOP_P", "0.9"))
eng.generate_sampled(prompt, 16, NO_STOP, u[:16], top_k=50, temperature=0.6, repetition_penalty=1.1, top_p=tp)
rp = {}
for n in (64, 448):
    t0 = time.perf_counter()
    out = eng.generate_sampled(prompt, n, NO_STOP, u[:n], top_k=50, temperature=0.6, repetition_penalty=1.1, top_p=tp)
    rp[n] = time.perf_counter() - t0
pstep = (rp[448] - rp[64]) / (448 - 64)
print(f"sampled on the device with top_p {tp}: {pstep*1e3:.3f} ms/step ({(pstep - sstep)*1e6:+.0f} us vs top_p 1)")
# the per-step host loop it replaces: logits D2H (628 KB) + a full sort of 156 940 values per step
import ctypes as C
sys.path.insert(0, os.path.join(ROOT, "oracle"))
t0 = time.perf_counter()
lg, _ = eng.decode(prompt, 0)
pos = len(prompt)
for s in range(16):
    order = np.argsort(-lg, kind="stable")[:50]        # stand-in for the host sampler's sort: the cost is the point here
    lg, _ = eng.decode([int(order[0])], pos)
    pos += 1
print(f"per-step host loop (decode + logits D2H + numpy sort of the vocabulary): {(time.perf_counter() - t0) / 16 * 1e3:.2f} ms/step")

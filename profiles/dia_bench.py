#!/usr/bin/env python3
"""Dia at the real shapes (nari-labs/Dia-1.6B: encoder 12 x 1024, decoder 18 x 2048 with 16 query heads on 4 k/v groups x 128,
ffn 8192, 9 heads x 1028, 1024 text positions; fp16 matrices): one encoder pass (both streams + cross K/V of 18 layers) and
the guided decoder step.  Timing does not depend on the weight values: every matrix is a slice of one small-normal fp16
buffer.  Prints ms per encode, ms per step and the bytes one step must read."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tts_cpp_amd  # noqa: F401
from tts_cpp_amd import gguf, hip, synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = synth.dia_1_6b(weight_type=gguf.F16)
rng = np.random.default_rng(3)
pool16 = (rng.standard_normal(1 << 25, dtype=np.float32) * np.float32(0.02)).astype(np.float16).view(np.uint8)
pool32 = (rng.standard_normal(1 << 22, dtype=np.float32) * np.float32(0.5)).view(np.uint8)
EH, DH, A, kvH, hd = cfg.enc_hidden, cfg.dec_hidden, cfg.dec_heads * cfg.head_dim, cfg.dec_kv_heads * cfg.head_dim, cfg.head_dim
tensors, n_dec = [], 0


def mat(name, rows, cols):
    global n_dec
    n = rows * cols
    assert n * 2 <= pool16.size
    tensors.append(gguf.Tensor(name, gguf.F16, [cols, rows], pool16[: n * 2]))
    if ".decoder.layers." in name or ".heads." in name:
        n_dec += n


def vec(name, n):
    tensors.append(gguf.Tensor.from_array(name, np.ones(n, dtype=np.float32)))


def table(name, rows, cols):
    tensors.append(gguf.Tensor(name, gguf.F32, [cols, rows], pool32[: rows * cols * 4]))


for i in range(cfg.n_out):
    table(f"dia.decoder.embeddings.{i}", cfg.out_vocab, DH)
    mat(f"dia.decoder.heads.{i}", cfg.out_vocab, DH)
vec("dia.decoder.norm", DH)
for l in range(cfg.dec_layers):
    p = f"dia.decoder.layers.{l}."
    for nm in ("pre_sa_norm", "pre_ca_norm", "pre_mlp_norm"):
        vec(p + nm, DH)
    for nm, r, c in (("self_q_proj", A, DH), ("self_k_proj", kvH, DH), ("self_v_proj", kvH, DH), ("self_o_proj", DH, A), ("cross_q_proj", A, DH),
                     ("cross_k_proj", A, EH), ("cross_v_proj", A, EH), ("cross_o_proj", DH, A), ("gate", cfg.dec_ffn, DH), ("up", cfg.dec_ffn, DH),
                     ("wo", DH, cfg.dec_ffn)):
        mat(p + nm, r, c)
table("dia.encoder.embedding", cfg.enc_vocab, EH)
vec("dia.encoder.norm", EH)
for l in range(cfg.enc_layers):
    p = f"dia.encoder.layers.{l}."
    vec(p + "pre_sa_norm", EH); vec(p + "post_sa_norm", EH)
    for nm, r, c in (("q_proj", A, EH), ("k_proj", A, EH), ("v_proj", A, EH), ("o_proj", EH, A), ("gate", cfg.enc_ffn, EH), ("up", cfg.enc_ffn, EH), ("wo", EH, cfg.enc_ffn)):
        mat(p + nm, r, c)


class M:
    pass


m = M(); m.cfg = cfg; m.tensors = tensors
U = int(os.environ.get('DIA_BENCH_UTTERANCES', '4'))   # BASELINE config 3: 4 utterances per GPU (x 2 guidance rows)
eng = hip.DiaEngine(cfg, max_utterances=U)
for _k, _v in __import__('json').loads(os.environ.get('DIA_TUNE', '{}')).items():   # tts_hip_tune keys of this run
    eng.tune(_k, _v)
t0 = time.perf_counter()
eng.load(m)
print(f"loaded {sum(len(t.raw()) for t in tensors) / 1e9:.2f} GB in {time.perf_counter() - t0:.1f}s", flush=True)
toks = np.zeros(cfg.max_ctx, dtype=np.uint32)
toks[:200] = rng.integers(32, 127, 200)
eng.encode(toks, 200)
t0 = time.perf_counter()
eng.encode(toks, 200)
enc_ms = (time.perf_counter() - t0) * 1e3
ids = np.full(cfg.n_out, cfg.bos, dtype=np.uint32)
for s in range(4):
    eng.step(ids, s)
t0 = time.perf_counter()
for s in range(4, 4 + steps):
    eng.step(ids, s)
step = (time.perf_counter() - t0) / steps
# cross K/V are fp32 [18][2][1024][2048] each and every step reads all of them (dia/model.cpp:609-633)
w_bytes = n_dec * 2
ckv_bytes = cfg.dec_layers * 2 * cfg.max_ctx * A * 4 * 2
print(f"encode (2 x {cfg.max_ctx} positions, 12 layers + cross K/V of {cfg.dec_layers} layers): {enc_ms:.1f} ms")
print(f"decoder step at positions 4..{4 + steps}: {step * 1e3:.3f} ms = {1 / step:.0f} steps/s = {1 / step / 86.13:.2f}x real time (86.13 frames/s)")
print(f"bytes per step: fp16 matrices {w_bytes / 1e9:.3f} GB + cross K/V {ckv_bytes / 1e9:.3f} GB -> {(w_bytes + ckv_bytes) / step / 1e9:.0f} GB/s "
      f"({(w_bytes + ckv_bytes) / step / 8e12 * 100:.1f}% of 8 TB/s; floor {(w_bytes + ckv_bytes) / 8e12 * 1e3:.3f} ms/step)")

# ---- lock-step utterances (BASELINE config 3: 32 utterances over 8 GPUs = 4 per GPU, M = 8 rows per step) ----
for u in range(U):
    eng.encode_slot(u, toks, 200)
idsb = np.full((U, cfg.n_out), cfg.bos, dtype=np.uint32)
for s in range(4):
    eng.step_batch(idsb, np.full(U, s, dtype=np.uint32))
t0 = time.perf_counter()
for s in range(4, 4 + steps):
    eng.step_batch(idsb, np.full(U, s, dtype=np.uint32))
stepb = (time.perf_counter() - t0) / steps
tot = w_bytes + U * ckv_bytes
print(f"{U} utterances in lock-step ({2 * U} rows): {stepb * 1e3:.3f} ms per step = {U / stepb:.0f} utterance-steps/s = {U / stepb / 86.13:.2f}x real time per GPU")
print(f"bytes per step: fp16 matrices {w_bytes / 1e9:.3f} GB + {U} x cross K/V {ckv_bytes / 1e9:.3f} GB -> {tot / stepb / 1e9:.0f} GB/s "
      f"({tot / stepb / 8e12 * 100:.1f}% of 8 TB/s; floor {tot / 8e12 * 1e3:.3f} ms/step)")

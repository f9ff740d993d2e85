#!/usr/bin/env python3
"""Bench lines for BASELINE configs 2-4 (`python bench.py --workload dia|orpheus|kokoro`): the same JSON contract as the headline
workload (metric / value / unit / roofline / cpu_baseline), measured through the C ABI engines at the real model shapes with
synthetic weights.  Timing does not depend on the weight values, so the big matrices are slices of small random pools (minting
billions of normals in numpy would cost more box time than the runs).

roofline here is the HBM roofline of the WHOLE decoder step (a step is a chain of ~250-350 short launches that stream the model once):
achieved = algorithmic bytes one step must read / measured step time.  cpu_baseline = the oracle on a reduced number of layers,
extrapolated linearly in the layer count (stated in `sample`)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tts_cpp_amd  # noqa: E402,F401
from tts_cpp_amd import gguf, hip, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0


class _Model:
    def __init__(self, cfg, tensors):
        self.cfg, self.tensors = cfg, tensors
        self.by_name = {t.name: t for t in tensors}



def _pmc(name):
    """HBM bytes per step / launch from the committed counter passes (profiles/pmc_traffic_secondary.json, profiles/r04/scripts/r4_pmc_secondary.sh); None if absent"""
    path = os.path.join(ROOT, "profiles", "pmc_traffic_secondary.json")
    try:
        return round(json.load(open(path))[name]["hbm_bytes"], 1)
    except (OSError, KeyError, ValueError):
        return None

def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    return orc


# ------------------------------------------------------------------------------------------------------------------------------
# Dia-1.6B fp16 (configs[3]: 32 utterances over 8 GPUs = 4 per GPU in lock-step, 2 guidance rows each)
# ------------------------------------------------------------------------------------------------------------------------------
def dia_tensors(cfg, rng):
    pool16 = (rng.standard_normal(1 << 25, dtype=np.float32) * np.float32(0.02)).astype(np.float16).view(np.uint8)
    pool32 = (rng.standard_normal(1 << 22, dtype=np.float32) * np.float32(0.5)).view(np.uint8)
    EH, DH, A, kvH = cfg.enc_hidden, cfg.dec_hidden, cfg.dec_heads * cfg.head_dim, cfg.dec_kv_heads * cfg.head_dim
    tensors, n_dec = [], [0]

    def mat(name, rows, cols):
        n = rows * cols
        tensors.append(gguf.Tensor(name, gguf.F16, [cols, rows], pool16[: n * 2]))
        if ".decoder.layers." in name or ".heads." in name:
            n_dec[0] += n

    def vec(name, n):
        tensors.append(gguf.Tensor.from_array(name, np.ones(n, dtype=np.float32)))

    def table(name, rows, cols):
        tensors.append(gguf.Tensor(name, gguf.F32, [cols, rows], pool32[: rows * cols * 4]))

    # the rows of the special ids (EOS / PAD / BOS, >= audio_vocab) of every head are zero: random weights then never pick one and
    # every utterance runs the full max_gen steps
    head = np.frombuffer(pool16[: cfg.out_vocab * DH * 2].tobytes(), dtype=np.float16).reshape(cfg.out_vocab, DH).copy()
    head[cfg.audio_vocab:] = 0
    head = head.reshape(-1).view(np.uint8)
    for i in range(cfg.n_out):
        table(f"dia.decoder.embeddings.{i}", cfg.out_vocab, DH)
        tensors.append(gguf.Tensor(f"dia.decoder.heads.{i}", gguf.F16, [DH, cfg.out_vocab], head))
        n_dec[0] += cfg.out_vocab * DH
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
        vec(p + "pre_sa_norm", EH)
        vec(p + "post_sa_norm", EH)
        for nm, r, c in (("q_proj", A, EH), ("k_proj", A, EH), ("v_proj", A, EH), ("o_proj", EH, A), ("gate", cfg.enc_ffn, EH), ("up", cfg.enc_ffn, EH), ("wo", EH, cfg.enc_ffn)):
            mat(p + nm, r, c)
    return tensors, n_dec[0]


def run_dia(args, ranks=None):
    """ranks (bench.py, --gpus N > 1): this process is one of N, one per GPU.  Utterance i -> rank i mod N, i.e. every rank decodes its own U
    utterances; rank 0 uploads the matrices, the other ranks lay the same arena out declare-only and receive it by the one collective of the
    path; timing = barrier, max over ranks; value = the utterances of all ranks / that time (weak scaling)."""
    U = 4
    steps = max(32, int(os.environ.get("TTS_BENCH_DIA_STEPS", "512")))
    cfg = synth.dia_1_6b(weight_type=gguf.F16)
    rng = np.random.default_rng(3)
    tensors, n_dec = dia_tensors(cfg, rng)
    rank, world, dev = (ranks["rank"], ranks["world"], ranks["local_rank"]) if ranks else (0, 1, 0)
    eng = hip.DiaEngine(cfg, device=dev, max_utterances=U)
    eng.load(_Model(cfg, tensors), declare_only=rank != 0)
    bcast = ranks["broadcast"](eng.ctx) if ranks else None
    toks = np.zeros(cfg.max_ctx, dtype=np.uint32)
    toks[:200] = rng.integers(32, 127, 200)
    A = cfg.dec_heads * cfg.head_dim
    # the codec: Dia decodes through the same 44.1 kHz DAC as Parler (dia/model.cpp:892-898); a codec-only context
    pm = synth.build(synth.parler_mini())
    dac = hip.HipEngine(pm.cfg, device=dev, max_seqs=1, flags=hip.FLAG_NO_PARLER)
    for t in pm.tensors:
        if t.name.startswith("audio_encoder."):
            dac.upload(t)
    dac.finalize()
    delay = np.array([0, 8, 9, 10, 11, 12, 13, 14, 15])   # dia/model.h:84

    urng = np.random.default_rng(11 + rank)

    def one_pass(n_steps):
        """4 sentences -> encoder + cross K/V per slot -> the generation loop on the device (tts_hip_dia_generate: check_stopping, guided step,
        sampler::sample with the reference's default top_k 50, delay-pattern feedback as one captured graph; max_gen = n_steps, so the
        countdown starts at position n_steps - 15 and n_steps - 1 sampler calls are made) -> un-delay -> one batched DAC pass"""
        t0 = time.perf_counter()
        for u in range(U):
            eng.encode_slot(u, toks, 200)
        t1 = time.perf_counter()
        uni = urng.random((n_steps, U, cfg.n_out), dtype=np.float32)
        hist = eng.generate(U, n_steps, delay, cfg.bos, cfg.eos, cfg.pad, 15, uniforms=uni, top_k=50)
        t2 = time.perf_counter()
        assert all(h.shape == (n_steps - 1, cfg.n_out) and int(h.max()) < cfg.audio_vocab for h in hist)
        # adjust_output_tokens (:787-808): frame i takes head h from step i + delay[h]; then one batched DAC pass
        nf = n_steps - 1 - 15
        codes = [np.stack([hist[u][np.arange(nf) + delay[h], h] for h in range(cfg.n_out)], axis=1) for u in range(U)]
        pcm = dac.dac_decode_batch(codes)
        assert sum(p.size for p in pcm) == U * nf * 512
        return t1 - t0, t2 - t1, time.perf_counter() - t2

    for _ in range(args.warmup):
        one_pass(24)
    enc_s, dec_s, dac_s = [], [], []

    def barrier():
        eng.synchronize()
        if ranks:
            ranks["torch"].cuda.synchronize()
            ranks["dist"].barrier()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        e, d, k = one_pass(steps)
        enc_s.append(e)
        dec_s.append(d)
        dac_s.append(k)
    barrier()
    elapsed = time.perf_counter() - t0
    step_ms = float(np.mean(dec_s)) / (steps - 1) * 1e3
    frames = steps - 1 - 15                  # max_gen = steps -> steps - 1 sampler calls; un-delay drops max_delay steps (dia/model.cpp:787-808)
    audio_s = U * frames * 512 / 44100.0 * args.steps
    if ranks:
        elapsed, audio_s = ranks["reduce"](elapsed, audio_s)   # max over ranks, sum over ranks
    w_bytes = n_dec * 2
    ckv_bytes = cfg.dec_layers * 2 * cfg.max_ctx * A * 4 * 2
    # self-attention cache rows a step reads: fp32 K and V of the 4 k/v groups, positions 0..t of both guidance rows, t averaged over the loop
    kvH = cfg.dec_kv_heads * cfg.head_dim
    skv_bytes = cfg.dec_layers * 2 * 2 * (steps / 2.0) * kvH * 4
    tot = w_bytes + U * (ckv_bytes + skv_bytes)
    out = {
        "metric": "audio-seconds/sec (Dia-1.6B fp16: encoder + guided decoder + DAC to 44.1 kHz PCM, lock-step utterances)",
        "value": round(audio_s / elapsed, 3), "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16",
        "data": "synthetic (fp16 matrices = slices of one random pool; shapes of nari-labs/Dia-1.6B)",
        "config": {"workload": f"configs[3]: Dia-1.6B fp16, the per-GPU share of batch 32 over 8 GPUs = {U} utterances in lock-step x 2 guidance rows, "
                               f"200-character sentences (encoder over 2 x 1024 positions + cross K/V per utterance), max_generation_size {steps}: {steps - 1} guided decoder "
                               "steps with check_stopping, sampler::sample (top_k 50) and the delay-pattern feedback on the device (tts_hip_dia_generate), "
                               "un-delay, one batched DAC pass to PCM",
                   "utterances_per_gpu": U, "utterances": U * world, "rows_per_step": 2 * U, "decoder_steps": steps,
                   "parallelism": f"dp{world}" + (" of dp8" if world < 8 else "") + " (utterance i -> rank i mod N: one process per GPU, rank 0's weight arena broadcast over RCCL, no per-step collective)"},
        "ranks": world, "rccl_ranks": world if (world > 1 and bcast and "RCCL" in bcast.get("via", "")) else 0, "weight_broadcast": bcast,
        "ms_per_decode_step": round(step_ms, 4), "encode_ms_per_utterance": round(float(np.mean(enc_s)) / U * 1e3, 2),
        "dac_ms_per_pass": round(float(np.mean(dac_s)) * 1e3, 2),
        "x_real_time_per_gpu": round(U / (step_ms * 1e-3) / 86.13, 2),
        "roofline": {"bound": "hbm", "achieved": round(tot / (step_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(tot / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": _pmc("dia_1_6b_lockstep4_step"),
                     "kernel": "whole decoder step (18 layers x 12 launches: gemv_stream_kernel, attn_gqa_split_kernel<128> (self) / attn_gqa_wave_kernel<128, 3, EXT> (cross), rms_fold_rows_kernel, llama_rope_kv_kernel, ... + sample_kernel)",
                     "algorithmic_bytes_per_launch": tot, "note": f"fp16 matrices {w_bytes / 1e9:.3f} GB + {U} x (fp32 cross K/V {ckv_bytes / 1e9:.3f} GB + self-attention K/V at the mean position {skv_bytes / 1e9:.3f} GB) per step"},
    }
    eng.close()
    dac.close()
    if not args.no_cpu_baseline:
        orc = _oracle()
        threads = args.cpu_threads or min(len(os.sched_getaffinity(0)), 32)
        orc.lib().orc_set_threads(threads)
        t = {}
        for nl in (1, 2):
            c2 = synth.dia_1_6b(weight_type=gguf.F16, dec_layers=nl, enc_layers=1, max_ctx=64, max_gen=16)
            ts, _ = dia_tensors(c2, np.random.default_rng(3))
            o = orc.DiaOracle(_Model(c2, ts), act_mode=1)
            tk = np.zeros(c2.max_ctx, dtype=np.uint32)
            tk[:40] = 65
            o.encode(tk, 40)
            ids = np.full(c2.n_out, c2.bos, dtype=np.uint32)
            o.step(ids, 0)
            t0 = time.perf_counter()
            for s in range(1, 4):
                o.step(ids, s)
            t[nl] = (time.perf_counter() - t0) / 3
        t_layer = max(t[2] - t[1], 1e-6)
        t_full = (t[1] - t_layer) + cfg.dec_layers * t_layer
        out["cpu_baseline"] = {"value": round(512 / 44100.0 / t_full, 4), "unit": "audio-seconds/sec", "cores": threads, "kind": "port",
                               "sample": "oracle decoder steps (one utterance = 2 guidance rows) at the 1.6B widths with 1 and 2 decoder layers and a 64-position "
                                         f"encoder context, extrapolated linearly to {cfg.dec_layers} layers; encoder pass and codec excluded",
                               "ms_per_decode_step": round(t_full * 1e3, 2)}
    return out


# ------------------------------------------------------------------------------------------------------------------------------
# Orpheus-3B Q4_0 + SNAC (configs[4]): 7 tokens per 2048-sample frame (orpheus/model.cpp:389-405), the 24 kHz SNAC decoder
# (snac_model.cpp:86-159) inside the timed region
# ------------------------------------------------------------------------------------------------------------------------------
def orpheus_tensors(cfg, rng):
    def q4(name, rows, cols):
        nb = rows * cols // 32
        b = rng.integers(0, 256, size=(nb, 18), dtype=np.uint8)
        b[:, 0], b[:, 1] = 0x00, 0x1C          # d = 2^-8 as fp16
        return gguf.Tensor("orpheus." + name, gguf.Q4_0, [cols, rows], b.reshape(-1))

    def f32(name, arr):
        return gguf.Tensor.from_array("orpheus." + name, np.asarray(arr, dtype=np.float32))

    H, F, QH, KVH = cfg.hidden, cfg.ffn, cfg.heads * cfg.head_dim, cfg.kv_heads * cfg.head_dim
    tensors = [q4("embed_tokens", cfg.vocab, H)]
    per_layer = 0
    for l in range(cfg.layers):
        p = f"layers.{l}."
        for nm, r, c in (("self_attn.q_proj", QH, H), ("self_attn.k_proj", KVH, H), ("self_attn.v_proj", KVH, H), ("self_attn.o_proj", H, QH),
                         ("mlp.gate_proj", F, H), ("mlp.up_proj", F, H), ("mlp.down_proj", H, F)):
            tensors.append(q4(p + nm, r, c))
            per_layer += r * c if l == 0 else 0
        tensors += [f32(p + "input_layernorm", np.ones(H)), f32(p + "post_attention_layernorm", np.ones(H))]
    tensors += [f32("norm", np.ones(H)), q4("lm_head", cfg.vocab, H), f32("rope_frequencies", synth.llama3_rope_factors(cfg.head_dim))]
    return tensors, per_layer


def run_orpheus(args):
    cfg = synth.orpheus_3b(ctx=1024, weight_type=gguf.Q4_0)
    rng = np.random.default_rng(7)
    tensors, per_layer = orpheus_tensors(cfg, rng)
    eng = hip.OrpheusEngine(cfg)
    eng.load(_Model(cfg, tensors))
    prompt = rng.integers(0, cfg.vocab, 32).astype(np.uint32)
    NO_STOP = 0xFFFFFFFF
    n_tok = 448
    # the codec of this config: SNAC 24 kHz at its real shapes (synthetic weights), 7 ids per frame -> 1 / 2 / 4 codes on its three levels
    # (orpheus/model.cpp:389-405); random codes of the right layout stand in for the ids' payload (timing does not depend on their values)
    scfg = synth.snac_24khz(max_frames=4 * (n_tok // 7))
    snac = hip.SnacEngine(scfg)
    snac.load(synth.build_snac(scfg))
    frames = n_tok // 7
    T = 4 * frames

    def codec():
        codes = np.concatenate([rng.integers(0, scfg.cb_size, T // r) for r in scfg.repeats]).astype(np.uint32)   # level-major (snac_runner::set_inputs :161-178)
        pcm = snac.decode(codes, T)
        assert pcm.size == frames * 2048
        return pcm

    for _ in range(max(1, args.warmup)):
        eng.generate_greedy(prompt, 16, NO_STOP)
        codec()
    t0 = time.perf_counter()
    ts, tc = [], []
    for _ in range(args.steps):
        t1 = time.perf_counter()
        out_ids = eng.generate_greedy(prompt, n_tok, NO_STOP)
        t2 = time.perf_counter()
        codec()
        tc.append(time.perf_counter() - t2)
        ts.append(t2 - t1)
        assert len(out_ids) == n_tok
    elapsed = time.perf_counter() - t0
    t64 = time.perf_counter()
    eng.generate_greedy(prompt, 64, NO_STOP)
    t64 = time.perf_counter() - t64
    step = (float(np.mean(ts)) - t64) / (n_tok - 64)
    params = cfg.layers * per_layer + cfg.vocab * cfg.hidden
    q4_bytes = params / 32 * 18
    audio_s = n_tok / 7 * 2048 / 24000.0 * args.steps
    out = {
        "metric": "audio-seconds/sec (Orpheus-3B Q4_0 decoder, greedy, + SNAC codec to 24 kHz PCM)",
        "value": round(audio_s / elapsed, 3), "unit": "audio-seconds/sec", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "i8",
        "dtype_detail": "Q4_0 matrices x Q8_0-quantised activations (ggml's vec_dot_q4_0_q8_0 semantics), integer block dots, fp16 block scales, f32 accumulate",
        "data": "synthetic (random Q4_0 blocks with a fixed scale; shapes of canopylabs/orpheus-3b)",
        "config": {"workload": f"configs[4]: Orpheus (Llama-3-3B backbone) Q4_0 on 1 x MI355X, one utterance: 32-token prompt (prefill) + {n_tok} greedy tokens "
                               "(= 64 SNAC frames of 2048 samples at 24 kHz) through tts_hip_orpheus_generate_greedy (captured step, streaming Q4_0 GEMV kernels), "
                               "then tts_hip_snac_decode of the 64 frames (SNAC 24 kHz shapes), both inside the timed region",
                   "tokens": n_tok, "parallelism": "dp1"},
        "ms_per_decode_step": round(step * 1e3, 4), "snac_ms_per_64_frames": round(float(np.mean(tc)) * 1e3, 3),
        "snac_share_of_step": round(float(np.mean(tc)) / (float(np.mean(ts)) + float(np.mean(tc))), 4),
        "x_real_time_per_gpu": round(audio_s / elapsed, 2),
        "roofline": {"bound": "hbm", "achieved": round(q4_bytes / step / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(q4_bytes / step / 1e9 / HBM_PEAK_GBS, 4),
                     "traffic": _pmc("orpheus_3b_q4_0_step"), "kernel": "whole decoder step (28 layers x 6 launches: gemv_q4_qkv_rope_kernel, attn_gqa_wave_kernel<128> + attn_gqa_combine_kernel, gemv_q4_rows_lds_kernel, gemv_q4_gateup_silu_kernel; lm_head over 156 940 logits, arg-max)",
                     "algorithmic_bytes_per_launch": q4_bytes, "note": "Q4_0 bytes of every matrix one step reads (lm_head included, one embedding row excluded); "
                     "the decoder steps are the dominant part of the timed region (snac_share_of_step is the codec's)"},
    }
    eng.close()
    # lock-step utterances inside the GPU (tts_hip_orpheus_generate_batch, round 6): B cache slots, one row per utterance and step; int8 codes
    # and fp16 scales of every matrix cross HBM once per step whatever B (from 5 rows on the kernels read the int8 expansion: twice the Q4_0 bytes), so the
    # tokens of all utterances share one weight stream
    i8_bytes = params * (1.0 + 2.0 / 32)
    out["lockstep_batches"] = {}
    for B in ((1, 8, 32) if not os.environ.get("TTS_BENCH_ORPHEUS_B") else tuple(int(x) for x in os.environ["TTS_BENCH_ORPHEUS_B"].split(","))):
        engb = hip.OrpheusEngine(cfg, max_seqs=B)
        engb.load(_Model(cfg, tensors))
        prompts = [rng.integers(0, cfg.vocab, 32).astype(np.uint32) for _ in range(B)]
        nb = 112   # 16 SNAC frames per utterance
        engb.generate_batch(prompts, 8, NO_STOP)
        t1 = time.perf_counter(); engb.generate_batch(prompts, 16, NO_STOP); t16 = time.perf_counter() - t1
        t1 = time.perf_counter(); res = engb.generate_batch(prompts, nb, NO_STOP); tn = time.perf_counter() - t1
        assert all(len(r) == nb for r in res)
        stepb = (tn - t16) / (nb - 16)
        wbytes = q4_bytes if B <= 4 else i8_bytes
        out["lockstep_batches"][str(B)] = {"ms_per_decode_step": round(stepb * 1e3, 4), "tokens_per_s": round(B / stepb, 1),
                                           "audio_s_per_s_decoder_only": round(B / stepb / 7 * 2048 / 24000.0, 2),
                                           "weight_stream": "Q4_0 codes (streaming 1-4 row kernels)" if B <= 4 else ("int8 expansion, streamed once per step (qgemv_stream_kernel, 5..64 rows)" if B <= 64 else "int8 expansion (MFMA workgroups)"),
                                           "hbm_frac_of_step": round(wbytes / stepb / 1e9 / HBM_PEAK_GBS, 4)}
        engb.close()
    b1 = out["lockstep_batches"].get("1")
    if b1:
        for B, v in out["lockstep_batches"].items():
            v["throughput_vs_B1"] = round(v["tokens_per_s"] / b1["tokens_per_s"], 2)
    snac.close() if hasattr(snac, "close") else None
    if not args.no_cpu_baseline:
        orc = _oracle()
        threads = args.cpu_threads or min(len(os.sched_getaffinity(0)), 32)
        orc.lib().orc_set_threads(threads)
        t = {}
        for nl in (1, 2):
            c2 = synth.orpheus_3b(layers=nl, ctx=64, vocab=cfg.vocab, weight_type=gguf.Q4_0)
            ts2, _ = orpheus_tensors(c2, np.random.default_rng(7))
            o = orc.OrpheusOracle(_Model(c2, ts2), act_mode=1)
            o.decode([5, 6, 7], 0)
            t0 = time.perf_counter()
            for s in range(3, 6):
                o.decode([9], s)
            t[nl] = (time.perf_counter() - t0) / 3
        t_layer = max(t[2] - t[1], 1e-6)
        t_full = (t[1] - t_layer) + cfg.layers * t_layer
        out["cpu_baseline"] = {"value": round(2048 / 24000.0 / (7 * t_full), 4), "unit": "audio-seconds/sec", "cores": threads, "kind": "port",
                               "sample": f"oracle decode steps at the 3B widths with 1 and 2 layers (full 156 940-row lm_head), extrapolated linearly to {cfg.layers} layers; codec excluded",
                               "ms_per_decode_step": round(t_full * 1e3, 2)}
    return out


# ------------------------------------------------------------------------------------------------------------------------------
# Kokoro-82M (configs[2]): duration graph + generation graph, forced durations for shape determinism
# ------------------------------------------------------------------------------------------------------------------------------
def run_kokoro(args):
    model = synth.build_kokoro(synth.kokoro_82m())
    ktune = dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in os.environ.get("TTS_BENCH_KOKORO_TUNE", "").split(",") if kv)   # e.g. kokoro_split=0: the bf16 x 3 form
    eng = hip.KokoroEngine(model, tune=ktune)
    cfg = model.cfg
    rng = np.random.default_rng(1)
    res = {}
    audio_total = 0.0
    timed = 0.0
    for n_ids in (64, 400):
        toks = np.concatenate([[0], rng.integers(1, cfg.vocab, n_ids), [0]]).astype(np.uint32)
        forced = np.full(toks.size, 3.0, dtype=np.float32)
        noise = rng.random((cfg.harmonic_num + 1) * int(forced.sum()) * cfg.up_sampling_factor, dtype=np.float32)
        for _ in range(max(1, args.warmup)):
            lens, hid = eng.durations(toks, cfg.voices[0])
            eng.generate(toks, forced, hid, cfg.voices[0], noise)
        td, tg = [], []
        for _ in range(args.steps):
            t0 = time.perf_counter()
            lens, hid = eng.durations(toks, cfg.voices[0])
            t1 = time.perf_counter()
            pcm = eng.generate(toks, forced, hid, cfg.voices[0], noise)
            t2 = time.perf_counter()
            td.append(t1 - t0)
            tg.append(t2 - t1)
            audio_total += pcm.size / 24000.0
            timed += t2 - t0
        res[f"{n_ids}_ids"] = {"durations_ms": round(float(np.mean(td)) * 1e3, 2), "generation_ms": round(float(np.mean(tg)) * 1e3, 2),
                               "audio_s": round(pcm.size / 24000.0, 2), "x_real_time": round(pcm.size / 24000.0 / (np.mean(td) + np.mean(tg)), 2)}
    out = {
        "metric": "audio-seconds/sec (Kokoro-82M: duration predictor + iSTFT vocoder path)",
        "value": round(audio_total / timed, 3), "unit": "audio-seconds/sec", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(timed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic (seeded random weights at the shapes of hexgrad/Kokoro-82M)",
        "config": {"workload": "configs[2]: Kokoro-82M on 1 x MI355X, 64 and 400 phoneme ids, durations forced to 3 frames per id for shape determinism; "
                               "tts_hip_kokoro_durations + tts_hip_kokoro_generate (k = 3 / 5 / 7 / 11 convolutions as fp16 hi + lo split products (three MFMAs per product) on the fp16 matrix pipe at fp32-level error, the rest exact-fp32 MFMA; workgroup-split LSTM recurrence)", "parallelism": "dp1"},
        "by_length": res,
    }
    # the dominant family (the stride-1 "same" convolutions on conv1d_mfma_kernel, exact-fp32 MFMA: kokoro/model.cpp:1141-1242) from an
    # event-timed pass of the 400-id synthesis
    hip.engine_profile(eng, 1)
    t0 = time.perf_counter()
    lens, hid = eng.durations(toks, cfg.voices[0])
    eng.generate(toks, forced, hid, cfg.voices[0], noise)
    wall = time.perf_counter() - t0
    st = hip.engine_profile_get(eng)["kokoro_conv_mfma"]
    hip.engine_profile(eng, 0)
    if st["launches"]:
        tf = st["flops_total"] / st["ms_total"] / 1e9
        # round 4: the k = 3 / 5 / 7 / 11 convolutions run as bf16 x 3 split products (six v_mfma_f32_32x32x16_bf16 per product term, tap pairs per k-step):
        # priced like the DAC families — ISSUED bf16 flops (6 x the algorithmic ones; the odd tap slot adds 9-33 % on top, not counted) against the
        # 2.5 PFLOP/s dense bf16 peak, the fp32-equivalent rate beside it (the exact-fp32 MFMA kernel of round 3 reached 65 TF of 157.3)
        # round 6: fp16 hi + lo split, three products (tune kokoro_split = 0: the six-product bf16 form)
        mult = 6 if ktune.get("kokoro_split", 1) == 0 else 3
        out["roofline"] = {"bound": "mfma", "achieved": round(mult * tf, 3), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(mult * tf / 2500.0, 4), "traffic": _pmc("kokoro_conv_mfma"),
                           "fp32_equivalent_TFLOPs": round(tf, 3), "products_per_fp32_product": mult,
                           "kernel": "conv1d_mfma_b3_kernel<2,1,1,8, 3 / 5 / 7 / 11, SplitH2 | SplitB3> (generator + AdaIN + text-encoder convolutions as split products; k = 1 stays on the exact-fp32 MFMA kernel)",
                           "avg_launch_us": round(st["ms_total"] / st["launches"] * 1e3, 2), "launches": st["launches"],
                           "share_of_wall_time": round(st["ms_total"] * 1e-3 / wall, 3),
                           "algorithmic_flops_per_launch": round(st["flops_total"] / st["launches"], 1),
                           "hbm_GBps": round(st["bytes_total"] / st["ms_total"] / 1e6, 1),
                           "note": "400 phoneme ids, one synthesis, HIP events around every launch of the family (eager pass); short sequences: few workgroups per launch"}
    else:
        out["roofline"] = None
    eng.close()
    # utterance-level concurrency, the reference's own model (N independent workers, examples/server/server.cpp:225-321): N contexts on N streams of the one
    # GPU, a host thread each (ctypes releases the GIL inside a call), every context synthesising the 400-id utterance `steps` times.  One synthesis is a
    # chain of short launches (LSTMs, AdaIN, iSTFT): the GPU is far from full with one context.
    import threading
    out["concurrent_contexts"] = {}
    for n_ctx in tuple(int(x) for x in os.environ.get("TTS_BENCH_KOKORO_CONTEXTS", "2,4,8").split(",") if x):
        engs = [hip.KokoroEngine(model, tune=ktune) for _ in range(n_ctx)]
        def work(e, reps, box):
            a = 0.0
            for _ in range(reps):
                lens, hid = e.durations(toks, cfg.voices[0])
                a += e.generate(toks, forced, hid, cfg.voices[0], noise).size / 24000.0
            box.append(a)
        for reps in (1, max(2, args.steps)):   # warm-up, then the timed pass
            box = []
            th = [threading.Thread(target=work, args=(e, reps, box)) for e in engs]
            t0 = time.perf_counter()
            for t in th: t.start()
            for t in th: t.join()
            dt = time.perf_counter() - t0
        out["concurrent_contexts"][str(n_ctx)] = {"audio_seconds_per_sec": round(sum(box) / dt, 1), "syntheses": n_ctx * max(2, args.steps), "seconds": round(dt, 3)}
        for e in engs: e.close()
    # the product's form of the same thing: kokoro_runner::generate_batch (host/kokoro_runner.cpp) — runner_from_file on a GGUF file, n phoneme strings in one
    # call, their clauses through `max_seqs` device contexts on ONE weight arena, the source noise drawn on the host as the reference does (one minstd stream,
    # handed out by jump-ahead): audio bit-equal to generate() calls in a row (tests/test_gpu_kokoro.py).  16 utterances of 400 ids x 3 frames (a duration head
    # that predicts 3 frames per id: synth forced_frames) = 30.15 s of audio each; max_seqs = 1 is the reference's shape (one generate() after the other).
    import tempfile
    from tts_cpp_amd import runner as _runner
    fmodel = synth.build_kokoro(synth.kokoro_82m(forced_frames=3))
    with tempfile.TemporaryDirectory() as td:
        gpath = fmodel.write_gguf(os.path.join(td, "kokoro82m.gguf"))
        texts = ["".join(chr(0x61 + int(v)) for v in rng.integers(0, 26, 398)) for _ in range(16)]
        os.environ["TTS_KOKORO_INPUT_IS_PHONEMES"] = "1"
        out["runner_generate_batch"] = {}
        first = None
        for lanes in tuple(int(x) for x in os.environ.get("TTS_BENCH_KOKORO_LANES", "1,2,4,8").split(",") if x):
            r = _runner.Runner(gpath, voice=cfg.voices[0].encode(), max_seqs=lanes, share_with=first)
            first = first or r
            r.generate_batch_sizes(texts[:max(2, lanes)], voice=cfg.voices[0].encode())      # warm-up: lanes created, planes packed
            t0 = time.perf_counter()
            sizes = r.generate_batch_sizes(texts, voice=cfg.voices[0].encode())
            dt = time.perf_counter() - t0
            out["runner_generate_batch"][str(lanes)] = {"audio_seconds_per_sec": round(sum(sizes) / 24000.0 / dt, 1), "utterances": len(texts), "seconds": round(dt, 3),
                                                        "audio_s_per_utterance": round(sizes[0] / 24000.0, 2)}
            if r is not first:
                r.close()
        first.close()
    if not args.no_cpu_baseline:
        orc = _oracle()
        o = orc.KokoroOracle(model)
        toks = np.concatenate([[0], rng.integers(1, cfg.vocab, 16), [0]]).astype(np.uint32)
        t0 = time.perf_counter()
        lens, hid = o.durations(toks, cfg.voices[0])
        forced = np.full(toks.size, 3.0, dtype=np.float32)
        noise = rng.random(o.noise_len(int(forced.sum())), dtype=np.float32)
        pcm = o.generate(toks, forced, hid, cfg.voices[0], noise)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(pcm.size / 24000.0 / dt, 4), "unit": "audio-seconds/sec", "cores": 1, "kind": "port",
                               "sample": "oracle (oracle/kokoro_oracle.c, scalar C) on 16 phoneme ids x 3 frames, both graphs"}
    return out


RUNNERS = {"dia": run_dia, "orpheus": run_orpheus, "kokoro": run_kokoro}

if __name__ == "__main__":
    import argparse
    import json
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", choices=sorted(RUNNERS))
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    a = ap.parse_args()
    print(json.dumps(RUNNERS[a.workload](a)), flush=True)

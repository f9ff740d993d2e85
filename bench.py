#!/usr/bin/env python3
"""bench.py — audio-seconds/sec of the Parler-TTS-Mini hot path (AR decoder + DAC) on MI355X, through the product path.

A "step" is one pass of the hot path over one batch of synthetic utterances, driven exactly as an application drives it:
  tts_c_runner_from_file(GGUF)  (C++ loader: GGUF reader, weight upload, cross K/V, KV cache)        -- before the timed region
  tts_c_generate_batch(texts)   (C++ runner: unigram tokenizer -> text-prompt prefill -> N greedy audio steps in the device-resident
                                 delay-pattern loop -> adjust_output_tokens -> DAC decode to 44.1 kHz PCM)   -- the timed region
for `--batch` utterances per context decoded in lock-step (`--streams` contexts per GPU).  Python only writes the synthetic GGUF,
picks the sentences and reads the clock: tokenizer, loop control, un-delay and the codec call are the C++ host's
(tts.cpp_amd/host/parler_runner.cpp), the compute is the HIP library's (include/tts_hip.h).

Multi-GPU: `python bench.py --gpus N` starts N ranks itself (re-exec under torch.distributed.run; a launcher that already set WORLD_SIZE is
used as is) and fails loudly when fewer devices exist.  Rank 0's first runner parses and uploads the file, the first runner of every
other rank is laid out declare-only (tts_load_options) and receives the finished weight arena (incl. precomputed cross K/V) by ONE RCCL
broadcast through the C ABI (tts_hip_comm_unique_id + tts_hip_broadcast_weights_rank; examples/server/server.cpp:885-895 is the
reference's per-worker load), further runners of a rank share their rank's arena (tts_load_options::share_with); afterwards utterances
are independent, no data-path collective -> "scaling": "weak".

Prints ONE JSON line on rank 0 (see the task contract): metric/value/unit + roofline + cpu_baseline, plus SURVEY §8(d)'s second metric
(batch-1 ms per decode step at T ~ 128 / 512 / 1024 / 2580, a 1024-step utterance), `long_utterances` (1024 audio steps per utterance,
uniform and ragged batches) and `secondary` (short runs of BASELINE configs 2-4).  The extra sections share a wall-clock budget
(`--time-budget-s`, default 480 s for the whole run; `time_budget` in the line names what was skipped): with the driver's `--steps 20 --warmup 5`
the 25 timed steps take ~280 s on their own, and the contract's part must never be lost to an extra.
"""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import threading
import time

_T_START = time.perf_counter()   # the wall-clock budget of the extra sections counts from here (the first `import torch` on a fresh box can take a minute)

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import tts_cpp_amd  # noqa: E402,F401
from tts_cpp_amd import dist as tdist  # noqa: E402
from tts_cpp_amd import gguf, hip, runner, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
I8_PEAK_TOPS = 5033.6     # dense int8 MFMA = 2 x the fp16 rate (MI355X_MICROARCH.md: >= 3944 TOPS measured on 16x16x64)
F32_PEAK_TFLOPS = 157.3   # fp32 vector == fp32-input MFMA peak
F16_PEAK_TFLOPS = 2516.8  # dense fp16 / bf16 MFMA = 16 x the fp32 matrix rate (MI355X_MICROARCH.md); with random operands the chip sustains
                          # 1830 TFLOP/s of v_mfma_f32_32x32x16_bf16 (power-limited clock, profiles/r03/mfma_rate.txt)
SAMPLE_RATE = 44100.0

# kernel families as rocprofv3 groups them by symbol (profiles/r02/kernel_stats_*.csv): the per-class HIP-event statistics of the shim
# (tts_hip_profile) are summed per family, so "dominant" means what it means in the rocprof table
FAMILIES = {
    "gemm_tile_kernel (decoder GEMMs: qkv, out_proj, cross q with the cross-attention in its epilogue, cross out, fc1, fc2, heads)":
        ["gemm_qkv", "gemm_attn_out", "gemm_cross_q", "gemm_cross_out", "gemm_fc1", "gemm_fc2", "gemm_heads", "gemm_other"],
    "attn_rows_kernel / attn_kernel (self-attention over the fp32 KV cache; one workgroup per row from 1024 rows)": ["attn_self"],
    "attn_short_kernel (cross-attention over the voice prompt)": ["attn_cross"],
    "ln_rows_kernel (LayerNorm + split-K fold)": ["ln"],
    "resunit_t7_kernel (DAC residual units at 96 / 192 channels, one launch each)": ["dac_resunit"],
    "conv_b3p_kernel<7,...> (DAC k=7 convs of the wide classes, split planes)": ["dac_conv7"],
    "conv_b3p_kernel<1,...> + snake_split_kernel (DAC k=1 convs + residual of the wide classes)": ["dac_conv1"],
    "convt_b3_kernel (DAC transposed convs)": ["dac_convt"],
    "other (embed, sampler/feed, DAC quantizer + final conv)": ["embed", "sample", "dac_embed", "dac_final"],
}
MFMA_FP16 = {"gemm_tile_kernel (decoder GEMMs: qkv, out_proj, cross q with the cross-attention in its epilogue, cross out, fc1, fc2, heads)"}
MFMA_FP32 = {"conv_b3p_kernel<1,...> + snake_split_kernel (DAC k=1 convs + residual of the wide classes)"}
# families that run fp32 convolutions as bf16 x 3 split products when the codec arithmetic is on (tts_hip_dac_arith): issued bf16 flops per
# algorithmic flop = six products, and the k = 7 kernels pad their 7 taps to 8 k-slots
B3_ISSUE = {"resunit_t7_kernel (DAC residual units at 96 / 192 channels, one launch each)": 6.0,     # resunit_t7_kernel: one tap per k-step, no padded slot
            "conv_b3p_kernel<7,...> (DAC k=7 convs of the wide classes, split planes)": 6.0,   # conv_b3p_kernel, one tap per k-step: no padded slot
            "convt_b3_kernel (DAC transposed convs)": 6.0,
            "conv_b3p_kernel<1,...> + snake_split_kernel (DAC k=1 convs + residual of the wide classes)": 6.0}
B3_BIT = {"conv_b3p_kernel<1,...> + snake_split_kernel (DAC k=1 convs + residual of the wide classes)": 32,
          "resunit_t7_kernel (DAC residual units at 96 / 192 channels, one launch each)": 2,
          "conv_b3p_kernel<7,...> (DAC k=7 convs of the wide classes, split planes)": 1,
          "convt_b3_kernel (DAC transposed convs)": 4}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


class ArenaView:
    """a runner's weight arena as a CUDA array (for torch.as_tensor): the RCCL broadcast writes straight into it"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def make_sentences(rn, n, target_ids, seed):
    """`n` distinct pseudo-sentences that the runner's own tokenizer (batch_from_sentence: ids + EOS) turns into exactly
    `target_ids` ids, so every utterance of a lock-step batch runs the same number of audio steps"""
    rng = np.random.default_rng(seed)
    letters = "abcdefghijklmnopqrstuvwxyz"
    out, tries = [], 0
    while len(out) < n:
        tries += 1
        if tries > 200000:
            raise RuntimeError("could not find sentences of the requested token length")
        words = ["".join(rng.choice(list(letters), size=int(rng.integers(2, 7)))) for _ in range(int(rng.integers(2, 3 + target_ids // 2)))]
        text = " ".join(words)
        k = len(rn.tokenize(text))
        while k > target_ids and len(text) > 1:   # trim characters until the count fits
            text = text[:-1].rstrip()
            k = len(rn.tokenize(text)) if text else 0
        if k == target_ids and text not in out:
            out.append(text)
    return out


def pmc_traffic(family, args, n_audio, arith=0):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
    runs, FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM); None when the committed measurement was taken on a different workload."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    t = json.load(open(path))
    w = t.get("workload", {})
    key = FAMILIES[family][0]
    if key.startswith("dac_"):  # DAC launches depend only on the utterances per codec pass and the frame count
        if args.dac_wtype != "f32" or w.get("audio_steps") != n_audio or w.get("dac_group") != int(os.environ.get("TTS_HIP_DAC_GROUP", "64")):
            return None
        if w.get("codec_arith", 0) != (arith & 31):   # measured with other codec kernels than the ones this run used
            return None
    elif w.get("batch") != args.batch or w.get("audio_steps") != n_audio:
        return None
    v = t.get("kernels", {}).get(key)
    return None if v is None else round(v["hbm_bytes_per_launch"], 1)


_TAIL = "f32 accumulate and residual stream; KV cache f32; DAC codec f32 (exact-f32 MFMA)"
_B3 = ("DAC codec: F32 tensors, every fp32 operand carried as three bf16 terms and every product as six v_mfma_f32_32x32x16_bf16 partial products "
       "with fp32 accumulation (fp32-level error, the suite's fp32 tolerances); the one-channel final conv on fp32 VALU")
_H2 = ("DAC codec: F32 tensors, every fp32 operand carried as fp16 hi + lo (x = h + l, the low part an fp16 subnormal where x is small) and every product as three "
       "v_mfma_f32_32x32x16_f16 partial products with fp32 accumulation (~2^-22 per product: the suite's fp32 tolerances hold, measured PCM 1.4e-6 / stages "
       "3.4-4.9e-6 against the oracle at the DAC-44k dims); the one-channel final conv on fp32 VALU")
DTYPE_DETAIL = {
    "f16": "decoder: f16 weights, f16 MFMA inputs, " + _TAIL,
    "f32": "decoder: f32 weights and activations (exact-f32 MFMA), " + _TAIL,
    "q8_0": "decoder: Q8_0 weights x Q8_0-quantised activations, int8 MFMA block dots, fp16 block scales, " + _TAIL,
    "q5_0": "decoder: Q5_0 weights x Q8_0-quantised activations, int8 MFMA block dots, fp16 block scales, " + _TAIL,
    "q4_0": "decoder: Q4_0 weights x Q8_0-quantised activations, int8 MFMA block dots, fp16 block scales, " + _TAIL,
}


def cpu_baseline(model, cfg, prompt, threads):
    """The oracle ("port": restated CPU path, ggml unavailable) on the host cores, bounded sample:
    prompt prefill + 128 audio steps (one greedy utterance up to T ~ 144) + the DAC on 64 frames — about 10-15 s of CPU work —
    extrapolated per frame."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["ORACLE_THREADS"] = str(threads)
    import oracle as orc

    orc.lib().orc_set_threads(threads)
    o = orc.ParlerOracle(model, act_mode=1, gelu_mode=1)
    o.decode(prompt, 0, audio=False, want_logits=False)
    ids = np.full(cfg.n_out, cfg.bos, dtype=np.uint32)
    n_steps = min(128, max(1, cfg.max_gen - len(prompt) - 1))
    t0 = time.perf_counter()
    for s in range(n_steps):
        lg, _ = o.decode(ids, len(prompt) + s, audio=True)
        ids = lg[:, 0, :].argmax(-1).astype(np.uint32)
    t_step = (time.perf_counter() - t0) / n_steps
    d = orc.DacOracle(model)
    n_frames = 64
    codes = np.random.default_rng(0).integers(0, cfg.cb_size, (n_frames, cfg.n_out)).astype(np.uint32)
    t0 = time.perf_counter()
    d.decode(codes)
    t_frame = (time.perf_counter() - t0) / n_frames
    frame_s = cfg.hop / SAMPLE_RATE
    return {
        "value": round(frame_s / (t_step + t_frame), 4),
        "unit": "audio-seconds/sec",
        "cores": threads,
        "kind": "port",
        "sample": f"oracle (restated CPU path; ggml absent): {n_steps} decoder steps at T~{len(prompt)}..{len(prompt) + n_steps} "
                  f"+ DAC on {n_frames} frames, batch 1, extrapolated per 11.6 ms frame",
        "ms_per_decode_step": round(t_step * 1e3, 3),
        "dac_ms_per_frame": round(t_frame * 1e3, 3),
    }


def decode_step_sweep(cfg_full, model, arena_ptr, device):
    """SURVEY §8(d) metric 2: batch 1, ms per decode step as the cache grows (device-resident greedy loop, one hipGraph replay per
    step) + a 1024-step utterance.  A one-sequence context with the full 2580-position cache on the resident weights."""
    e = hip.HipEngine(cfg_full, device=device, max_seqs=1, kv_positions=cfg_full.max_gen)
    for t in model.tensors:
        e.upload(t, declare_only=True)
    e.finalize(external_arena=arena_ptr)
    e.arena_filled()
    prompt = np.arange(3, 19, dtype=np.uint32)
    out = {}
    e.prefill(0, prompt)
    e.generate_greedy([len(prompt)], 8)   # graph capture outside the clock
    e.reset()
    e.prefill(0, prompt)
    pos = len(prompt)
    for label, upto in (("T~128", 128), ("T~512", 512), ("T~1024", 1024), ("T~2580", 2580)):
        upto = min(upto, cfg_full.max_gen)
        n = upto - pos
        if n <= 0:
            continue
        t0 = time.perf_counter()
        e.generate_greedy([pos], n)
        out[label] = round((time.perf_counter() - t0) / n * 1e3, 4)
        pos = upto
    # the reference's shape of a step (decode() + get_ggml_node_data + sampler::max on the host, parler/model.cpp:648-693): one call per
    # token, the 39 KB of logits copied back, arg-max here — 64 steps around T = 128 and T = 512
    per_call = {}
    for label, T in (("T~128", 128), ("T~512", 512)):
        e.reset()
        e.prefill(0, prompt)
        e.generate_greedy([len(prompt)], T - 32 - len(prompt))
        ids = np.full((1, cfg_full.n_out), cfg_full.bos, dtype=np.uint32)
        for s in range(4):
            ids = e.step(ids, [T - 32 - 4 + s]).argmax(-1).astype(np.uint32)
        t0 = time.perf_counter()
        for s in range(64):
            ids = e.step(ids, [T - 32 + s]).argmax(-1).astype(np.uint32)
        per_call[label] = round((time.perf_counter() - t0) / 64 * 1e3, 4)
    e.reset()
    e.prefill(0, prompt)
    n_long = min(1024, cfg_full.max_gen - len(prompt))
    t0 = time.perf_counter()
    e.generate_greedy([len(prompt)], n_long)
    t1024 = time.perf_counter() - t0
    e.close()
    return {"ms_per_step_by_cached_positions": out, "ms_per_step_per_call_with_logits_d2h": per_call,
            "steps_1024": {"steps": n_long, "ms_total": round(t1024 * 1e3, 2), "ms_per_step": round(t1024 / n_long * 1e3, 4),
                           "x_real_time": round((n_long - cfg_full.n_out + 1) * cfg_full.hop / SAMPLE_RATE / t1024, 2)},
            "note": "batch 1, fp16 weights, fp32 KV cache, cross-attention on, greedy, device-resident loop (tts_hip_parler_generate_greedy); "
                    "each interval continues the same utterance from the previous one; per_call: tts_hip_parler_step per token + logits D2H + host arg-max"}


def generate_batch1_end_to_end(path, device, both=False):
    """The reference's own protocol and harness for ONE utterance at a time (examples/perf_battery/perf_battery.cpp:100-117: the 30 Harvard
    sentences, one generate() each = tokenizer + prefill + AR loop + un-delay + DAC, wall time per sentence; RTF = generation ms / audio ms):
    oracle/_ref/perf_battery_ref is the reference's perf_battery.cpp compiled UNCHANGED against this engine's libtts.so through the compat/
    overlay (oracle/Makefile; the binary travels, the source does not) — a harness, not an oracle: what it times is the product.  Falls back to
    the engine's own host/perf_battery (same protocol) when the binary is absent.  Run on the headline's model file: every sentence generates
    to max_generation (random weights never emit EOS), 2.8-2.9 s of audio each; the first sentence pays the graph captures."""
    import re
    import subprocess
    ref = os.path.join(ROOT, "oracle", "_ref", "perf_battery_ref")
    exe = ref if os.path.exists(ref) else os.path.join(ROOT, "tts.cpp_amd", "host", "perf_battery")
    if not os.path.exists(exe):
        return {"error": "no perf_battery harness built"}
    out = {"harness": os.path.relpath(exe, ROOT) + (" (the reference's perf_battery.cpp, unchanged, linked against libtts.so)" if exe == ref else " (engine's tool, same protocol)"),
           "protocol": "30 sentences, one tts_generation_runner::generate() each (decoder + DAC), mean over sentences; x_real_time = 1 / real-time factor"}
    env = dict(os.environ, TTS_HIP_DEVICE=str(device), TTS_HIP_MAX_SEQS="1")
    # top-k 50 is the reference's default protocol; the greedy pass measured the same in round 4 (10.78 / 10.77 x) and only runs without a time budget
    for label, topk in ((("top_k_50", 50), ("top_k_1_greedy", 1)) if both else (("top_k_50", 50),)):
        t0 = time.perf_counter()
        r = subprocess.run([exe, "--model-path", path, "--topk", str(topk)], capture_output=True, text=True, timeout=120, env=env)   # 8 s when healthy: a hang must not eat the run's time budget
        m1 = re.search(r"Generation Time \(ms\):\s+([0-9.]+)", r.stdout)
        m2 = re.search(r"Real Time Factor \(ms\):\s+([0-9.]+)", r.stdout)
        if r.returncode != 0 or not m1 or not m2:
            out[label] = {"error": (r.stderr or r.stdout)[-300:]}
            continue
        rtf = float(m2.group(1))
        out[label] = {"mean_generation_ms": round(float(m1.group(1)), 2), "real_time_factor": round(rtf, 5), "x_real_time": round(1.0 / rtf, 2),
                      "wall_s": round(time.perf_counter() - t0, 1)}
    return out


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks here (torch.distributed.run, one process per GPU, rendezvous on
    127.0.0.1).  Fails loudly when the box has fewer devices (the test hook TTS_BENCH_FORCE_DEVICE puts every rank on one device)."""
    import torch
    have = torch.cuda.device_count()
    if os.environ.get("TTS_BENCH_FORCE_DEVICE") is None and have < n:
        raise SystemExit(f"bench.py --gpus {n}: this box has {have} GPU(s); refusing to report an n_gpus={n} line from fewer devices")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    log("bench.py: starting", n, "ranks:", " ".join(cmd))
    os.execv(sys.executable, cmd)


def broadcast_arena_ctx(L, ctx, rank, world, local_rank, backend, dist, torch):
    """The path's one collective: rank 0's finished weight arena (weights + precomputed tables) to every rank's declare-only context, through the
    C ABI (tts_hip_comm_unique_id + tts_hip_broadcast_weights_rank: RCCL over xGMI) — or through torch.distributed under the CPU-collective test
    hook (several ranks on one device, where RCCL cannot form a communicator).  Any model family: the arena is just bytes."""
    nbytes = int(L.tts_hip_arena_bytes(ctx))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if backend == "nccl":
        # rank 0 mints the communicator id, every rank joins with its own context
        ident = torch.zeros(128, dtype=torch.uint8)
        if rank == 0 and L.tts_hip_comm_unique_id(ident.numpy().ctypes.data_as(C.c_void_p)) != 0:
            raise RuntimeError(L.tts_hip_last_error().decode())
        ident = ident.cuda(local_rank)
        dist.broadcast(ident, src=0)
        ident = ident.cpu()
        if L.tts_hip_broadcast_weights_rank(ctx, ident.numpy().ctypes.data_as(C.c_void_p), rank, world, 0) != 0:
            raise RuntimeError(L.tts_hip_last_error().decode())
        via = "tts_hip_comm_unique_id + tts_hip_broadcast_weights_rank (RCCL, C ABI)"
    else:
        arena = torch.as_tensor(ArenaView(L.tts_hip_arena_ptr(ctx), nbytes), device=f"cuda:{local_rank}")
        tdist.broadcast_arena(arena, src=0)
        torch.cuda.synchronize()
        if rank != 0 and L.tts_hip_arena_filled(ctx) != 0:
            raise RuntimeError(L.tts_hip_last_error().decode())
        via = f"torch.distributed.broadcast ({backend})"
    torch.cuda.synchronize()
    dist.barrier()
    return {"bytes": nbytes, "ms": round((time.perf_counter() - t0) * 1e3, 2), "via": via}


def load_runners(path, args, rank, world, local_rank, backend, dist, torch, gen_cfg, batch):
    """`--streams` runners on this rank's device with ONE weight arena per rank: the first runner loads the file (rank 0) or is laid out
    declare-only and filled by the RCCL broadcast (other ranks); the others share its arena."""
    L = hip.load_lib()
    info = None
    first = runner.Runner(path, device=local_rank, max_seqs=batch, declare_only=(rank != 0), **gen_cfg)
    if world > 1:
        info = broadcast_arena_ctx(L, first.device_context(), rank, world, local_rank, backend, dist, torch)
    runners = [first]
    for _ in range(1, args.streams):
        runners.append(runner.Runner(path, device=local_rank, max_seqs=batch, share_with=first, **gen_cfg))
    return runners, info


def run_all(runners, texts, timings_=None, stream=False, reps=1):
    """Every runner (a host thread each) makes `reps` generate calls back to back; returns when all are done.  reps > 1: the runners of a GPU are not
    joined between their calls — a runner that finishes a batch starts its next one at once, as the workers of a server do, so one runner's codec passes
    keep falling on the others' decoder loops instead of all runners idling into a common tail after every batch."""
    res = [None] * len(runners)

    def work(i):
        try:
            n, dt = 0, 0.0
            for _ in range(reps):
                t0 = time.perf_counter()
                sizes = runners[i].generate_stream(texts[i], sizes_only=True) if stream else runners[i].generate_batch_sizes(texts[i])
                n += sum(sizes)
                dt += time.perf_counter() - t0
            res[i] = (n, dt / reps)
        except Exception as e:   # surfaced by the caller: a worker thread must not fail silently
            res[i] = e

    if len(runners) == 1:
        work(0)
    else:
        th = [threading.Thread(target=work, args=(i,)) for i in range(len(runners))]
        for t in th:
            t.start()
        for t in th:
            t.join()
    for r in res:
        if isinstance(r, Exception):
            raise r
    if timings_ is not None:
        timings_.append(float(np.mean([r[1] for r in res])))
    return sum(r[0] for r in res)


def profile_get(L, rn):
    out = {}
    for k in range(64):
        name = L.tts_hip_kclass_name(k).decode()
        if name == "?":
            break
        st = hip.KStat()
        if L.tts_hip_profile_get(rn.device_context(), k, C.byref(st)) == 0:
            out[name] = dict(ms_total=st.ms_total, launches=st.launches, bytes_total=st.bytes_total, flops_total=st.flops_total)
    return out


def family_stats(stats):
    fam = {}
    for name, keys in FAMILIES.items():
        a = dict(ms_total=0.0, launches=0, bytes_total=0.0, flops_total=0.0)
        for k in keys:
            for f in a:
                a[f] += stats.get(k, {}).get(f, 0)
        if a["launches"]:
            fam[name] = a
    return fam


def roof_of(name, st, src, tot, arith, dac_wtype, wtype="f16"):
    """one family against the roofline that bounds it; bf16 x 3 families are priced with the flops they ISSUE against the bf16 pipe,
    the fp32-equivalent (algorithmic) rate beside it.  With quantised decoder matrices (--wtype q*) the decoder GEMMs are qgemm_tile_kernel's
    int8 MFMA blocks over Q8_0-quantised rows, priced against the dense int8 matrix peak."""
    per_launch_ms = st["ms_total"] / max(st["launches"], 1)
    tf = st["flops_total"] / max(st["ms_total"], 1e-9) / 1e9
    gb = st["bytes_total"] / max(st["ms_total"], 1e-9) / 1e6
    b3 = name in B3_ISSUE and dac_wtype == "f32" and (arith & B3_BIT[name]) and not (arith & 24)
    if b3:
        h2 = bool(arith & 64)   # round 6: fp16 hi + lo operands, three products; before: bf16 x 3, six products
        mult = B3_ISSUE[name] / 2.0 if h2 else B3_ISSUE[name]
        issued = tf * mult
        r = {"bound": "mfma", "achieved": round(issued, 3), "peak": F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(issued / F16_PEAK_TFLOPS, 4),
             "fp32_equivalent_TFLOPs": round(tf, 3), "hbm_GBps": round(gb, 1), "hbm_frac": round(gb / HBM_PEAK_GBS, 4),
             "note": (f"fp32 operands as fp16 hi + lo, three fp16 MFMAs per product, fp32 accumulate: achieved = issued fp16 flops ({mult:.2f} x algorithmic) against the dense fp16 peak"
                      if h2 else
                      f"fp32 operands as three bf16 terms, six bf16 MFMAs per product, fp32 accumulate: achieved = issued bf16 flops ({mult:.2f} x "
                      "algorithmic) against the dense bf16 peak; with random operands the pipe sustains 1830 TFLOP/s (power-limited clock, profiles/r03/mfma_rate.txt)")}
    elif name in MFMA_FP16 and wtype.startswith("q"):
        # round 6: from 65 rows on the decoder GEMMs are qgemm_tile_kernel — one v_mfma_i32_32x32x32_i8 per 32-wide quantisation block, every block's 16 i32
        # results per lane scaled by d_w * d_a in ggml's order on the vector pipe (~48 VALU instructions per MFMA: the measured bound, profiles/r06/valu_rate.txt)
        r = {"bound": "mfma", "achieved": round(tf, 3), "peak": I8_PEAK_TOPS, "unit": "TFLOP/s", "frac": round(tf / I8_PEAK_TOPS, 4),
             "hbm_GBps": round(gb, 1), "hbm_frac": round(gb / HBM_PEAK_GBS, 4),
             "note": "int8 ops against the dense int8 MFMA peak (TOP/s); the kernel is bound by the per-block fp32 scaling on the vector pipe, not by the matrix pipe"}
        name = f"qgemm_tile_kernel (decoder GEMMs with {wtype} matrices: Q8_0-quantised rows x int8 weight codes on v_mfma_i32_32x32x32_i8, block scales on the vector pipe)"
    elif name in MFMA_FP32 or name in MFMA_FP16 or name in B3_ISSUE:
        peak = F16_PEAK_TFLOPS if (name in MFMA_FP16 or dac_wtype == "f16") else F32_PEAK_TFLOPS
        r = {"bound": "mfma", "achieved": round(tf, 3), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4),
             "hbm_GBps": round(gb, 1), "hbm_frac": round(gb / HBM_PEAK_GBS, 4)}
        if peak == F32_PEAK_TFLOPS:
            r["note"] = "fp32 conv: peak = fp32 vector / fp32-input-MFMA peak (exact-fp32 numerics)"
    else:
        r = {"bound": "hbm", "achieved": round(gb, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gb / HBM_PEAK_GBS, 4)}
    r.update({"kernel": name, "timing_source": src, "avg_launch_us": round(per_launch_ms * 1e3, 3), "launches": st["launches"],
              "share_of_kernel_time": round(st["ms_total"] / tot, 3),
              "algorithmic_bytes_per_launch": round(st["bytes_total"] / max(st["launches"], 1), 1),
              "algorithmic_flops_per_launch": round(st["flops_total"] / max(st["launches"], 1), 1)})
    return r


def long_sentences(rn, n, lo, hi, seed):
    """`n` pseudo-sentences whose id counts (the runner's tokenizer, + EOS) spread evenly over lo..hi"""
    rng = np.random.default_rng(seed)
    letters = "abcdefghijklmnopqrstuvwxyz"
    out = []
    for i in range(n):
        target = lo + (hi - lo) * i // max(n - 1, 1)
        words = []
        while True:
            words.append("".join(rng.choice(list(letters), size=int(rng.integers(2, 7)))))
            if len(words) % 16 == 0 or len(words) * 2 >= target:
                if len(rn.tokenize(" ".join(words))) >= target:
                    break
        text = " ".join(words)
        while len(rn.tokenize(text)) > target and len(text) > 1:
            text = text[:-1].rstrip()
        out.append(text)
    return out


class TimeBudget:
    """Wall-clock budget of one bench.py run.  The contract's part — warm-up, the K timed steps, the roofline pass, the CPU baseline — always runs;
    every EXTRA section (batch-1 sweep, end-to-end harness, secondary configs, the long_utterances parts) runs only while the time already spent plus
    the section's cost (seconds on an MI355X box, measured: `time_budget.sections` of the last full line under profiles/r05/) stays inside the budget.
    Round 4's table over-priced the short sections by 2-3 x and warmed every long part up with a full run of itself (250 of 440 s), so the driver's
    `--steps 20 --warmup 5` line (25 steps of ~11 s = 280 s before any extra) dropped the round's own feature.  Round 5: the long parts warm up on a
    48-step run of the same row count, the request stream is 2 x the rows instead of 3 x, the costs are the measured ones, and the order inside
    long_utterances is uniform, ragged_stream, ragged, uniform_same_mix (most wanted first).  With 25 steps of 11.3 s the whole line takes 519-525 s (profiles/r05/bench_driver_flags.json).  `--time-budget-s 0` = no limit."""
    COST = {"decode_step_batch1": 5, "generate_batch1_end_to_end": 8, "secondary.kokoro": 15, "secondary.dia": 8, "secondary.orpheus": 5,
            "long_utterances.uniform": 60, "long_utterances.ragged_stream": 69, "long_utterances.ragged": 36, "long_utterances.uniform_same_mix": 30}
    RESERVED = 12   # the CPU baseline that still has to run after the extras of the main context

    def __init__(self, budget_s):
        self.budget, self.skipped, self.reserved, self.sections = float(budget_s), [], 0.0, {}

    def elapsed(self):
        return time.perf_counter() - _T_START

    def room(self, section):
        if self.budget <= 0 or self.elapsed() + self.reserved + self.COST[section] <= self.budget:
            return True
        self.skipped.append(section)
        return False

    def took(self, section, t0):
        """book what a section really cost (so that the next COST table can be read off a line)"""
        self.sections[section] = round(time.perf_counter() - t0, 1)

    def report(self):
        return {"budget_s": self.budget, "elapsed_s": round(self.elapsed(), 1), "skipped": self.skipped, "sections": self.sections,
                "note": "extra sections run while elapsed + their measured cost fits the budget (--time-budget-s 0: no limit); the timed steps, "
                        "the roofline and the CPU baseline always run; sections = seconds each extra took in this run"}


def long_utterances(args, mk, wt, local_rank, gen_cfg, torch, L, budget=None, model=None):
    """SURVEY §8(d)'s long workload: 1024 audio steps per utterance (11.7 s of audio; the reference's perf_battery sentences average
    10.8 s), the largest 3-runner lock-step batch whose fp32 KV cache fits (`uniform`); then the ragged mix (prompts of 16 .. 784 ids, so that
    the rows stop — reach max_generation — after 1024 down to 256 steps) as a STREAM of requests through continuous-batching sessions
    (`ragged_stream`), a lock-step batch in which every utterance carries the MEAN prompt of that mix (`uniform_same_mix`: the yardstick the two
    ragged numbers are quoted against — same ids prefilled, same steps generated, same cached positions on average, nothing ragged), and the mix as
    one lock-step batch (`ragged`)."""
    n_steps = args.long_steps
    if budget is not None and not budget.room("long_utterances.uniform"):
        budget.skipped += ["long_utterances.ragged_stream", "long_utterances.ragged", "long_utterances.uniform_same_mix"]
        return {"skipped": "time budget (--time-budget-s)"}
    t_sec = time.perf_counter()
    cfg = mk(weight_type=wt, dac_f16=args.dac_wtype == "f16", max_gen=args.prompt_len + n_steps)
    free_b, _ = torch.cuda.mem_get_info(local_rank)
    kv_per_seq = cfg.layers * 2 * (args.prompt_len + n_steps) * cfg.hidden * (2 if args.kv == "f16" else 4)
    frame_elems = max(cfg.latent, cfg.c0, max((cfg.c0 >> (i + 1)) * int(np.prod(cfg.strides[:i + 1])) for i in range(len(cfg.strides))))
    codec = 4 * 64 * (n_steps - cfg.n_out + 1) * frame_elems * 4        # the device's codec buffers: three activation buffers + planes of a 64-utterance pass
    mem_budget = 0.85 * free_b - codec - 6e9
    batch = int(min(args.batch, mem_budget // (args.streams * kv_per_seq))) // 32 * 32
    if batch < 32:
        return {"skipped": f"not enough free memory for {args.streams} x 32 sequences of {n_steps} steps ({free_b / 1e9:.0f} GB free)"}
    path = os.path.join(tempfile.gettempdir(), f"tts_bench_long_{os.getpid()}.gguf")
    if model is None or getattr(model, "shapes_only", False):
        model = synth.build(cfg)
    else:   # the headline's tensors under this section's max_generation (the only difference between the two files is that key)
        import copy
        model = copy.copy(model)
        model.cfg, model.kv = cfg, synth.build(cfg, shapes_only=True).kv   # (the vocabulary is seeded by cfg.seed: the same pieces in both files)
    model.write_gguf(path)
    first = runner.Runner(path, device=local_rank, max_seqs=batch, **gen_cfg)
    runners = [first] + [runner.Runner(path, device=local_rank, max_seqs=batch, share_with=first, **gen_cfg) for _ in range(1, args.streams)]
    os.unlink(path)
    out = {"audio_steps": n_steps, "lockstep_batch": batch, "contexts_per_gpu": args.streams, "kv_cache": args.kv,
           "kv_cache_GB": round(args.streams * batch * kv_per_seq / 1e9, 1)}
    frames = n_steps - cfg.n_out + 1

    def timed(texts, stream=False):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = run_all(runners, texts, stream=stream)
        torch.cuda.synchronize()
        return n, time.perf_counter() - t0

    try:
        # warm-up for every part: ONE utterance text of max_generation - 48 ids in every row — the same row count (the captured steps are keyed by
        # it), the codec and session buffers, 48 decode steps instead of a whole generation (round 4 ran every part twice: 250 s of warm-up)
        warm = long_sentences(first, 1, args.prompt_len + n_steps - 48, args.prompt_len + n_steps - 48, 4000)[0]
        run_all(runners, [[warm] * batch for _ in range(args.streams)])
        texts = [make_sentences(first, batch, args.prompt_len, 5000 + i) for i in range(args.streams)]
        n_samples, dt = timed(texts)
        out["uniform"] = {"audio_seconds_per_sec": round(n_samples / SAMPLE_RATE / dt, 2), "seconds": round(dt, 3),
                          "utterances": batch * args.streams, "audio_s_per_utterance": round(frames * cfg.hop / SAMPLE_RATE, 2)}
        # the attention family over the long cache: eager, event-timed pass of one runner
        L.tts_hip_profile(first.device_context(), 1)
        first.generate_batch_sizes(texts[0])
        st = profile_get(L, first).get("attn_self", {})
        L.tts_hip_profile(first.device_context(), 0)
        if st.get("launches"):
            out["uniform"]["attn_self"] = {"GBps": round(st["bytes_total"] / st["ms_total"] / 1e6, 1), "frac_of_hbm_peak": round(st["bytes_total"] / st["ms_total"] / 1e6 / HBM_PEAK_GBS, 4),
                                           "avg_launch_us": round(st["ms_total"] / st["launches"] * 1e3, 2), "mean_cached_positions": args.prompt_len + n_steps // 2}
        if budget is not None:
            budget.took("long_utterances.uniform", t_sec)
        lo, hi = args.prompt_len, args.prompt_len + (3 * n_steps) // 4     # 16 .. 784 ids: rows run 1024 .. 256 steps
        mid = (lo + hi) // 2
        mix = {"prompt_ids": [lo, hi], "audio_steps_per_utterance": [n_steps - (hi - lo), n_steps], "mean_prompt_ids": mid, "mean_audio_steps": n_steps - (mid - lo)}

        def part(name):
            if budget is not None and not budget.room("long_utterances." + name):
                out[name] = {"skipped": "time budget (--time-budget-s)"}
                return False
            return True

        # ---- the ragged mix as a STREAM of requests: twice the rows a runner holds, one continuous-batching session per runner
        # (tts_c_generate_stream): a row freed by an utterance that reached max_generation is refilled at the next 32-step look-in point
        if part("ragged_stream"):
            t_sec = time.perf_counter()
            n_req = 2 * (batch - 1)
            srag = [long_sentences(first, n_req, lo, hi, 9000 + i) for i in range(args.streams)]
            rngs = np.random.default_rng(17)
            for t in srag:
                rngs.shuffle(t)                                      # arrival order is not sorted by length
            run_all(runners, [sorted(t, key=len)[-8:] for t in srag], stream=True)   # warm-up: the session's buffers (eight of the shortest-running requests)
            n_samples, dt = timed(srag, stream=True)
            out["ragged_stream"] = dict(mix, audio_seconds_per_sec=round(n_samples / SAMPLE_RATE / dt, 2), seconds=round(dt, 3), requests=n_req * args.streams,
                                        rows_per_runner=batch - 1,
                                        note="continuous batching (tts_hip_parler_stream_*): 2 x the rows of requests with the ragged length mix per runner; "
                                             "finished utterances leave, waiting ones are prefilled as a side batch and enter the freed rows")
            budget and budget.took("long_utterances.ragged_stream", t_sec)
        # ---- the mix as ONE lock-step batch per runner: the loop runs as long as the longest row, finished rows leave every 32 steps
        if part("ragged"):
            t_sec = time.perf_counter()
            rag = [long_sentences(first, batch, lo, hi, 7000 + i) for i in range(args.streams)]
            n_samples, dt = timed(rag)
            out["ragged"] = dict(mix, audio_seconds_per_sec=round(n_samples / SAMPLE_RATE / dt, 2), seconds=round(dt, 3), utterances=batch * args.streams,
                                 note="lock-step: the loop runs as long as the longest row; every 32 steps the rows that reached max_generation leave the forward (row "
                                      "compaction, TTS_HIP_GEN_COMPACT=0: they idle instead — 196 against 268 audio-s/s, profiles/r03/compaction_call20.txt)")
            budget and budget.took("long_utterances.ragged", t_sec)
        # ---- the yardstick of the mix: every utterance with the mean prompt (400 ids) and hence the mean number of steps (640), lock-step
        if part("uniform_same_mix"):
            t_sec = time.perf_counter()
            same = [long_sentences(first, batch, mid, mid, 6000 + i) for i in range(args.streams)]   # (make_sentences wants exact counts: a minute of tokenizer calls at 400 ids)
            n_samples, dt = timed(same)
            out["uniform_same_mix"] = {"audio_seconds_per_sec": round(n_samples / SAMPLE_RATE / dt, 2), "seconds": round(dt, 3), "utterances": batch * args.streams,
                                       "prompt_ids": mid, "audio_steps": n_steps - (mid - lo),
                                       "note": "what the hardware does with the mix's mean utterance when nothing is ragged: the ragged numbers are quoted against this"}
            budget and budget.took("long_utterances.uniform_same_mix", t_sec)
        ref = out.get("uniform_same_mix", {}).get("audio_seconds_per_sec")
        for name in ("ragged_stream", "ragged"):
            v = out.get(name, {}).get("audio_seconds_per_sec")
            if v:
                out[name]["of_uniform"] = round(v / out["uniform"]["audio_seconds_per_sec"], 3)
                if ref:
                    out[name]["of_uniform_same_mix"] = round(v / ref, 3)
        if out.get("ragged_stream", {}).get("audio_seconds_per_sec") and out.get("ragged", {}).get("audio_seconds_per_sec"):
            out["ragged_stream"]["of_lockstep_ragged"] = round(out["ragged_stream"]["audio_seconds_per_sec"] / out["ragged"]["audio_seconds_per_sec"], 3)
    finally:
        for rn in reversed(runners):
            rn.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--join-steps", action="store_true", help="join all runners of the GPU after every batch (rounds 1-5); default: the runners make their K calls back to back")
    ap.add_argument("--batch", type=int, default=int(os.environ.get("TTS_BENCH_BATCH", "1024")),
                    help="utterances per context decoded in lock-step (1024: every decoder GEMM is whole rounds of tiles over the 256 CUs; 3 x 384 was the round-2 default)")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("TTS_BENCH_STREAMS", "3")),
                    help="independent runners (contexts, HIP streams) per GPU; each decodes --batch utterances per step")
    ap.add_argument("--audio-steps", type=int, default=256, help="AR audio steps per utterance (random weights never emit EOS: max_generation = prompt + this)")
    ap.add_argument("--long-steps", type=int, default=1024, help="audio steps per utterance of the long_utterances section")
    ap.add_argument("--prompt-len", type=int, default=16)
    ap.add_argument("--kv", choices=["f32", "f16"], default="f32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-step-sweep", action="store_true")
    ap.add_argument("--no-long", action="store_true", help="skip the long_utterances section (1024 audio steps, uniform + ragged)")
    ap.add_argument("--no-e2e", action="store_true", help="skip generate_batch1_end_to_end (the reference's perf_battery protocol, one utterance at a time)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short runs of BASELINE configs 2-4 (secondary)")
    ap.add_argument("--time-budget-s", type=float, default=550.0,
                    help="wall-clock budget of the whole run: extra sections are skipped (and named in time_budget.skipped) once they no longer fit; 0 = no limit")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--model", choices=["mini", "small", "tiny"], default="mini")
    ap.add_argument("--workload", choices=["parler", "dia", "orpheus", "kokoro"], default="parler",
                    help="parler (default) = BASELINE configs[1], the headline; dia / orpheus / kokoro = configs[3] / [4] / [2] at one GPU's share "
                         "(profiles/secondary_bench.py: same JSON contract, measured through the C ABI engines)")
    ap.add_argument("--sample", action="store_true",
                    help="sampler::sample on the device (top_k 50, temperature 1, top_p 1: the reference's defaults) instead of greedy")
    ap.add_argument("--dac-wtype", choices=["f32", "f16"], default="f32",
                    help="GGUF type of the codec tensors: f32 (quantize default) or f16 (--convert-dac-to-f16: fp16 im2col, fp16 MFMA)")
    ap.add_argument("--wtype", choices=["f16", "f32", "q8_0", "q5_0", "q4_0"], default="f16",
                    help="GGUF type of the decoder matrices (headline: f16; q*: integer path with Q8_0 activations)")
    args = ap.parse_args()
    budget = TimeBudget(args.time_budget_s)

    if args.workload in ("orpheus", "kokoro") and (int(os.environ.get("WORLD_SIZE", "1")) > 1 or args.gpus > 1):
        raise SystemExit("--workload orpheus / kokoro are single-utterance configurations of BASELINE.json (configs[4] / [2]: 1 x MI355X); run them with --gpus 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: the launcher started WORLD_SIZE={world} ranks but --gpus {args.gpus} was asked for")
    import torch

    if os.environ.get("TTS_BENCH_LAUNCH_ONLY"):   # CPU test hook: prove that the ranks start and meet, without touching a device
        import torch.distributed as dist
        tdist.init(os.environ.get("TTS_BENCH_DIST_BACKEND", "gloo"), rank, world)
        t = torch.tensor([float(rank)])
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"launched_ranks": world, "rank_sum": t.item(), "n_gpus": args.gpus}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    # test hooks: run the multi-rank flow on a single-GPU box (all ranks on one device, gloo instead of RCCL)
    if os.environ.get("TTS_BENCH_FORCE_DEVICE") is not None:
        local_rank = int(os.environ["TTS_BENCH_FORCE_DEVICE"])
    elif world > torch.cuda.device_count():
        raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPU(s) on this box")
    backend = os.environ.get("TTS_BENCH_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        tdist.init(backend, rank, world, device=torch.device("cuda", local_rank))

    if args.workload != "parler":
        # BASELINE configs[2] / [3] / [4] through the C ABI engines (profiles/secondary_bench.py: same JSON contract).  Dia is the one that shards:
        # configs[3] = batch 32 over 8 GPUs = 4 utterances x 2 guidance rows per rank, rank 0's arena broadcast like the headline's.
        sys.path.insert(0, os.path.join(ROOT, "profiles"))
        import secondary_bench
        ranks = None
        if world > 1:
            ranks = dict(rank=rank, world=world, local_rank=local_rank, backend=backend, dist=dist, torch=torch,
                         broadcast=lambda ctx: broadcast_arena_ctx(hip.load_lib(), ctx, rank, world, local_rank, backend, dist, torch),
                         reduce=lambda el, units: tdist.reduce_timing(el, units, device=f"cuda:{local_rank}"))
        line = secondary_bench.RUNNERS[args.workload](args, ranks) if args.workload == "dia" else secondary_bench.RUNNERS[args.workload](args)
        if rank == 0:
            print(json.dumps(line), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    mk = {"mini": synth.parler_mini, "small": synth.small, "tiny": synth.tiny}[args.model]
    wt = {"f16": gguf.F16, "f32": gguf.F32, "q8_0": gguf.Q8_0, "q5_0": gguf.Q5_0, "q4_0": gguf.Q4_0}[args.wtype]
    cfg_full = mk(weight_type=wt, dac_f16=args.dac_wtype == "f16")
    n_audio = min(args.audio_steps, cfg_full.max_gen - args.prompt_len, cfg_full.ctx - args.prompt_len)
    # the file the runners load: same tensors, max_generation = prompt + audio steps (check_stopping ends every utterance there,
    # model.cpp:720-722; random weights never emit EOS)
    cfg = mk(weight_type=wt, dac_f16=args.dac_wtype == "f16", max_gen=args.prompt_len + n_audio)
    WNAME = dict(f16="fp16", f32="fp32").get(args.wtype, args.wtype)
    DECODE = "top-k 50 sampling" if args.sample else "greedy decode"
    if args.kv == "f16":
        os.environ["TTS_HIP_KV_F16"] = "1"

    # ---- the model file: rank 0 mints the synthetic GGUF; the others only need its metadata (they load declare-only) ------------
    t_load = time.perf_counter()
    path = os.path.join(tempfile.gettempdir(), f"tts_bench_{args.model}_{args.wtype}_{args.dac_wtype}_{args.prompt_len + n_audio}_{os.environ.get('MASTER_PORT', '0')}.gguf")
    model = synth.build(cfg, shapes_only=(rank != 0 and args.no_cpu_baseline))
    if rank == 0:
        model.write_gguf(path)
    if dist is not None:
        dist.barrier()
    gen_cfg = dict(sample=1 if args.sample else 0, top_k=50, top_p=1.0, temperature=1.0, seed=1234 + rank)
    L = hip.load_lib()
    runners, bcast = load_runners(path, args, rank, world, local_rank, backend, dist, torch, gen_cfg, args.batch)
    arena_bytes = L.tts_hip_arena_bytes(runners[0].device_context())
    arith = int(L.tts_hip_dac_arith(runners[0].device_context()))
    log(f"[rank {rank}] {len(runners)} runner(s) ready in {time.perf_counter() - t_load:.1f}s, one arena of {arena_bytes / 1e6:.0f} MB per rank")
    if dist is not None:
        dist.barrier()

    all_texts = [make_sentences(runners[0], args.batch, args.prompt_len, 1000 + rank * 64 + i) for i in range(args.streams)]

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def profile(mode):
        for rn in runners:
            if L.tts_hip_profile(rn.device_context(), mode) != 0:
                raise RuntimeError(L.tts_hip_last_error().decode())

    for _ in range(args.warmup):
        run_all(runners, all_texts)
    barrier()
    profile(2)   # HIP-event pairs around the (never graph-captured) DAC launches, live in the timed region
    timings = []
    t0 = time.perf_counter()
    n_samples = 0
    if args.join_steps:
        for _ in range(args.steps):
            n_samples += run_all(runners, all_texts, timings)
    else:   # K steps = every runner's K batches, free-running between the two barriers (round 6; --join-steps: all runners joined after every batch)
        n_samples += run_all(runners, all_texts, timings, reps=args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    live = {}
    for rn in runners:
        for k, v in profile_get(L, rn).items():
            a = live.setdefault(k, dict(ms_total=0.0, launches=0, bytes_total=0.0, flops_total=0.0))
            for f in a:
                a[f] += v[f]
    profile(0)
    if dist is not None:
        elapsed, n_samples = tdist.reduce_timing(elapsed, n_samples, device=f"cuda:{local_rank}")

    audio_seconds = n_samples / SAMPLE_RATE
    value = audio_seconds / elapsed
    frames = n_audio - cfg.n_out + 1
    b3_on = args.dac_wtype == "f32" and (arith & 7) and not (arith & 24)
    detail = DTYPE_DETAIL[args.wtype]
    if args.dac_wtype == "f16":
        detail = detail.replace("DAC codec f32 (exact-f32 MFMA)", "DAC codec F16 tensors (fp16 im2col x fp16 kernels, fp16 MFMA, f32 accumulate)")
    elif b3_on:
        detail = detail.replace("DAC codec f32 (exact-f32 MFMA)", _H2 if (arith & 64) else _B3)
    out = {
        "metric": "audio-seconds/sec (Parler-TTS-Mini fp16, greedy decode + DAC to 44.1 kHz PCM)",
        "value": round(value, 3),
        "unit": "audio-seconds/sec",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": {"f16": "f16", "f32": "f32"}.get(args.wtype, "i8"),
        "dtype_detail": detail,
        "data": "synthetic (seeded random weights of the Parler-TTS-Mini + DAC-44k architecture written as a GGUF file; fixed-length greedy generation)",
        "config": {
            "workload": f"configs[1]: Parler-TTS-Mini {WNAME} on MI355X, {DECODE} + DAC codec ({args.dac_wtype} tensors), through the C++ runner "
                        f"(tts_c_generate_batch: tokenizer, device-resident AR loop, un-delay, DAC); {args.streams} runner(s) x {args.batch} utterances/GPU "
                        f"in lock-step, {args.prompt_len}-id prompts, {n_audio} audio steps (={frames} frames, {frames * cfg.hop / SAMPLE_RATE:.2f} s audio) per utterance",
            "utterances_per_gpu": args.batch * args.streams, "contexts_per_gpu": args.streams, "lockstep_batch": args.batch, "audio_steps": n_audio,
            "prompt_len": args.prompt_len, "kv_cache": args.kv, "codec_arithmetic_bits": arith,
            "parallelism": f"dp{world} (one process per GPU, one weight arena per GPU shared by its runners, RCCL weight broadcast, no per-step collective)",
        },
        "real_time_factor": round(elapsed / audio_seconds, 6),
        "x_real_time_per_gpu": round(value / world, 3),
        "ranks": world,
        # ranks whose weight arena arrived through RCCL (ncclBroadcast behind tts_hip_broadcast_weights_rank); 0 when the transport was anything else
        # (one rank, or the gloo test hook): the driver cross-checks this field against n_gpus
        "rccl_ranks": world if (world > 1 and bcast and "RCCL" in bcast.get("via", "")) else 0,
        "weight_broadcast": bcast,
        "ms_per_generate_batch": round(float(np.mean(timings)) * 1e3, 3),
        "runners_joined_per_step": bool(args.join_steps),
        "ms_per_generate_batch_note": "mean wall time of one runner's tts_c_generate_batch call; ms_per_step is elapsed / K where K = the calls every runner makes between the two barriers "
                                      "(runners_joined_per_step false: back to back, no join between a runner's calls; true: the wall time until ALL runners of the "
                                      "step are done); runners take turns at the per-device codec mutex: one 64-utterance codec pass at a time",
    }

    if rank == 0:
        if not args.no_roofline:
            # per-kernel-class HIP-event timing of the same call on runner 0 (eager launches, every launch bracketed by an event pair on
            # the context's stream); classes are summed per rocprof symbol family
            rn = runners[0]
            L.tts_hip_profile(rn.device_context(), 1)
            rn.generate_batch_sizes(all_texts[0])
            stats = profile_get(L, rn)
            L.tts_hip_profile(rn.device_context(), 0)
            fam = family_stats(stats)
            tot = sum(v["ms_total"] for v in fam.values()) or 1.0
            dom = max(fam, key=lambda k: fam[k]["ms_total"])
            src = "separate eager pass of the same tts_c_generate_batch call on runner 0 (decoder launches live inside hipGraphs in the timed region)"
            st = fam[dom]
            keys = FAMILIES[dom]
            if all(live.get(k, {}).get("launches") for k in keys if stats.get(k, {}).get("launches")):
                st = dict(ms_total=0.0, launches=0, bytes_total=0.0, flops_total=0.0)
                for k in keys:
                    for f in st:
                        st[f] += live.get(k, {}).get(f, 0)
                src = "HIP events around every launch of this kernel family in the timed region (all runners of rank 0)"
            roof = roof_of(dom, st, src, tot, arith, args.dac_wtype, args.wtype)
            roof["share_of_kernel_time"] = round(fam[dom]["ms_total"] / tot, 3)
            roof["traffic"] = pmc_traffic(dom, args, n_audio, arith)
            out["roofline"] = roof
            out["roofline_families"] = [dict(roof_of(n, v, "eager pass (launch-by-launch HIP events: short kernels read ~1-2 us long)", tot, arith, args.dac_wtype, args.wtype),
                                             traffic=pmc_traffic(n, args, n_audio, arith))
                                        for n, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms_total"]) if n != dom]
            out["kernel_classes"] = {
                k: {"ms": round(v["ms_total"], 3), "launches": v["launches"],
                    "GBps": round(v["bytes_total"] / max(v["ms_total"], 1e-9) / 1e6, 1),
                    "TFLOPs": round(v["flops_total"] / max(v["ms_total"], 1e-9) / 1e9, 3)}
                for k, v in stats.items() if v["launches"]}
        budget.reserved = 0.0 if args.no_cpu_baseline else TimeBudget.RESERVED
        if not args.no_step_sweep and args.wtype in ("f16", "f32") and budget.room("decode_step_batch1"):
            t_sec = time.perf_counter()
            full_model = synth.build(cfg_full, shapes_only=True)
            out["decode_step_batch1"] = decode_step_sweep(cfg_full, full_model, L.tts_hip_arena_ptr(runners[0].device_context()), local_rank)
            budget.took("decode_step_batch1", t_sec)
        if not args.no_e2e and args.model == "mini" and not args.sample and budget.room("generate_batch1_end_to_end"):
            t_sec = time.perf_counter()
            try:
                out["generate_batch1_end_to_end"] = generate_batch1_end_to_end(path, local_rank, both=budget.budget <= 0)
            except Exception as e:   # the headline must survive a failure of an extra section
                out["generate_batch1_end_to_end"] = {"error": str(e)[:300]}
            budget.took("generate_batch1_end_to_end", t_sec)
        if not args.no_cpu_baseline:
            t_sec = time.perf_counter()
            threads = args.cpu_threads or min(len(os.sched_getaffinity(0)), 32)
            prompt = runners[0].tokenize(all_texts[0][0])
            out["cpu_baseline"] = cpu_baseline(model, cfg, prompt, threads)
            budget.took("cpu_baseline", t_sec)
        budget.reserved = 0.0
    for rn in reversed(runners):
        rn.close()
    if rank == 0 and not os.environ.get("TTS_BENCH_KEEP_GGUF"):
        os.unlink(path)
    if rank == 0 and world == 1 and args.model == "mini":
        if not args.no_secondary:   # before the long section: cheaper per line of evidence when the budget is short
            sys.path.insert(0, os.path.join(ROOT, "profiles"))
            import secondary_bench
            sec = {}
            sargs = argparse.Namespace(steps=1, warmup=1, no_cpu_baseline=True, cpu_threads=0)
            for name in ("kokoro", "dia", "orpheus"):
                if not budget.room("secondary." + name):
                    sec[name] = {"skipped": "time budget (--time-budget-s)"}
                    continue
                t_sec = time.perf_counter()
                try:
                    sec[name] = secondary_bench.RUNNERS[name](sargs)
                except Exception as e:
                    sec[name] = {"error": str(e)[:300]}
                budget.took("secondary." + name, t_sec)
            out["secondary"] = sec
        if not args.no_long:
            try:
                out["long_utterances"] = long_utterances(args, mk, wt, local_rank, gen_cfg, torch, L, budget, model=model)
            except Exception as e:   # the headline must survive a failure of an extra section
                out["long_utterances"] = {"error": str(e)[:300]}
    if rank == 0:
        out["time_budget"] = budget.report()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

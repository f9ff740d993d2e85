#!/usr/bin/env python3
"""bench.py — audio-seconds/sec of the Parler-TTS-Mini hot path (AR decoder + DAC) on MI355X.

A "step" is one pass of the hot path over one batch of synthetic utterances:
  text-prompt prefill -> N greedy audio steps (device-resident delay-pattern loop) -> un-delay ->
  DAC decode to 44.1 kHz PCM, for `--batch` utterances per GPU decoded in lock-step.
Inputs (weights, prompts) are resident in HBM before the timed region; sampled ids and PCM come back
to the host inside it, exactly as tts_generation_runner::generate() returns them.

Multi-GPU (launched by torch.distributed.run): rank 0 mints the synthetic GGUF tensors and uploads them,
the other ranks only declare shapes; the finished weight arena (incl. precomputed cross K/V) is
broadcast once with RCCL; afterwards utterances are independent, no data-path collective
 -> "scaling": "weak".

Prints ONE JSON line on rank 0 (see the task contract): metric/value/unit + roofline + cpu_baseline.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import tts_cpp_amd  # noqa: E402,F401
from tts_cpp_amd import dist as tdist  # noqa: E402
from tts_cpp_amd import gguf, hip, synth  # noqa: E402
from tts_cpp_amd.pattern import undelay  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
F32_PEAK_TFLOPS = 157.3   # fp32 vector == fp32-input MFMA peak
F16_PEAK_TFLOPS = 2516.8  # dense fp16 MFMA = 16 x the fp32 matrix rate (MI355X_MICROARCH.md)
SAMPLE_RATE = 44100.0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def make_prompts(cfg, batch, prompt_len, rank):
    rng = np.random.default_rng(1000 + rank)
    return [np.concatenate([rng.integers(3, cfg.prompt_vocab, prompt_len - 1), [1]]).astype(np.uint32) for _ in range(batch)]


DAC_GROUP = int(os.environ.get("TTS_BENCH_DAC_GROUP", "32"))
import threading  # noqa: E402

# One DAC pass fills the chip (compute-bound MFMA convs); passes of different contexts are serialised so that they
# overlap the other context's latency-bound decoder loop instead of each other.
DAC_LOCK = threading.Lock()


SAMPLE_UNIFORMS = None  # [steps][batch][heads] U[0,1) draws when --sample


def run_utterance_batch(eng, cfg, prompts, n_audio, timings=None, dac_group=None):
    dac_group = dac_group or DAC_GROUP
    """one bench step; returns total PCM samples produced"""
    t0 = time.perf_counter()
    eng.prefill_batch(prompts)
    t1 = time.perf_counter()
    if SAMPLE_UNIFORMS is not None:   # --sample: sampler::sample on the device, the reference's default parameters
        u = SAMPLE_UNIFORMS[:n_audio, :len(prompts)]
        toks, _ = eng.generate_sampled([len(p) for p in prompts], n_audio, u, top_k=50, top_p=1.0, temperature=1.0)
    else:
        toks, _ = eng.generate_greedy([len(p) for p in prompts], n_audio)
    t2 = time.perf_counter()
    frames = [undelay(toks[:, s, :], cfg.audio_vocab) for s in range(len(prompts))]
    n_samples = 0
    group = max(1, dac_group)
    for g in range(0, len(frames), group):  # DAC for `group` utterances per pass (bounds the activation buffers)
        with DAC_LOCK:
            pcms = eng.dac_decode_batch(frames[g:g + group])
        for pcm in pcms:
            n_samples += pcm.size
    t3 = time.perf_counter()
    if timings is not None:
        timings.append((t1 - t0, t2 - t1, t3 - t2))
    return n_samples


def pmc_traffic(kclass, args, n_audio):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate runs of this same command, FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM);
    None when the committed measurement was taken on a different workload."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    t = json.load(open(path))
    w = t.get("workload", {})
    if kclass.startswith("dac_"):  # DAC launches depend only on the group size and the frame count
        if args.dac_wtype != "f32":
            return None
        if w.get("audio_steps") != n_audio or w.get("dac_group") != DAC_GROUP:
            return None
    elif w.get("batch") != args.batch or w.get("audio_steps") != n_audio:
        return None
    v = t.get("kernels", {}).get(kclass)
    return None if v is None else round(v["hbm_bytes_per_launch"], 1)


_TAIL = "f32 accumulate and residual stream; KV cache f32; DAC codec f32 (exact-f32 MFMA)"
DTYPE_DETAIL = {
    "f16": "decoder: f16 weights, f16 MFMA inputs, " + _TAIL,
    "f32": "decoder: f32 weights and activations (exact-f32 MFMA), " + _TAIL,
    "q8_0": "decoder: Q8_0 weights x Q8_0-quantised activations, int8 MFMA block dots, fp16 block scales, " + _TAIL,
    "q5_0": "decoder: Q5_0 weights x Q8_0-quantised activations, int8 MFMA block dots, fp16 block scales, " + _TAIL,
    "q4_0": "decoder: Q4_0 weights x Q8_0-quantised activations, int8 MFMA block dots, fp16 block scales, " + _TAIL,
}


def cpu_baseline(model, cfg, prompt, threads):
    """The oracle ("port": restated CPU path, ggml unavailable) on the host cores, bounded sample:
    prompt prefill + a few audio steps + a few DAC frames, extrapolated per frame."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["ORACLE_THREADS"] = str(threads)
    import oracle as orc

    orc.lib().orc_set_threads(threads)
    o = orc.ParlerOracle(model, act_mode=1, gelu_mode=1)
    o.decode(prompt, 0, audio=False, want_logits=False)
    ids = np.full(cfg.n_out, cfg.bos, dtype=np.uint32)
    n_steps = 12
    t0 = time.perf_counter()
    for s in range(n_steps):
        lg, _ = o.decode(ids, len(prompt) + s, audio=True)
        ids = lg[:, 0, :].argmax(-1).astype(np.uint32)
    t_step = (time.perf_counter() - t0) / n_steps
    d = orc.DacOracle(model)
    n_frames = 4
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("TTS_BENCH_BATCH", "128")), help="utterances per context decoded in lock-step")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("TTS_BENCH_STREAMS", "3")),
                    help="independent contexts (HIP streams) per GPU sharing one weight arena; each decodes --batch utterances")
    ap.add_argument("--audio-steps", type=int, default=256, help="AR audio steps per utterance (random weights never emit EOS)")
    ap.add_argument("--prompt-len", type=int, default=16)
    ap.add_argument("--kv", choices=["f32", "f16"], default="f32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--model", choices=["mini", "small", "tiny"], default="mini")
    ap.add_argument("--sample", action="store_true",
                    help="sampler::sample on the device (top_k 50, temperature 1, top_p 1: the reference's defaults) instead of greedy")
    ap.add_argument("--dac-wtype", choices=["f32", "f16"], default="f32",
                    help="GGUF type of the codec tensors: f32 (quantize default) or f16 (--convert-dac-to-f16: fp16 im2col, fp16 MFMA)")
    ap.add_argument("--wtype", choices=["f16", "f32", "q8_0", "q5_0", "q4_0"], default="f16",
                    help="GGUF type of the decoder matrices (headline: f16; q*: integer path with Q8_0 activations)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    import torch

    # test hooks: run the multi-rank flow on a single-GPU box (all ranks on one device, gloo instead of RCCL)
    if os.environ.get("TTS_BENCH_FORCE_DEVICE") is not None:
        local_rank = int(os.environ["TTS_BENCH_FORCE_DEVICE"])
    backend = os.environ.get("TTS_BENCH_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        tdist.init(backend, rank, world, device=torch.device("cuda", local_rank))

    cfg = {"mini": synth.parler_mini, "small": synth.small, "tiny": synth.tiny}[args.model](
        weight_type={"f16": gguf.F16, "f32": gguf.F32, "q8_0": gguf.Q8_0, "q5_0": gguf.Q5_0, "q4_0": gguf.Q4_0}[args.wtype],
        dac_f16=args.dac_wtype == "f16")
    WNAME = dict(f16="fp16", f32="fp32").get(args.wtype, args.wtype)
    DECODE = "top-k 50 sampling" if args.sample else "greedy decode"
    if args.sample:
        global SAMPLE_UNIFORMS
        SAMPLE_UNIFORMS = np.random.default_rng(1234 + rank).random((args.audio_steps, args.batch, cfg.n_out), dtype=np.float32)
    n_audio = min(args.audio_steps, cfg.max_gen - args.prompt_len, cfg.ctx - args.prompt_len)
    kv_type = gguf.F16 if args.kv == "f16" else gguf.F32

    # ---- weights: rank 0 generates + uploads, everyone else receives the arena over RCCL ------
    t_load = time.perf_counter()
    model = synth.build(cfg, shapes_only=(rank != 0))
    kv_pos = min(cfg.ctx, cfg.max_gen)  # generation stops at position max_generation (check_stopping, model.cpp:720-722)
    eng = hip.HipEngine(cfg, device=local_rank, max_seqs=args.batch, kv_type=kv_type, kv_positions=kv_pos)
    for t in model.tensors:
        eng.upload(t, declare_only=(rank != 0))
    arena = None
    if world > 1:
        arena = torch.empty(eng.arena_bytes(), dtype=torch.uint8, device=f"cuda:{local_rank}")
        eng.finalize(external_arena=arena.data_ptr())
        torch.cuda.synchronize()
        tdist.broadcast_arena(arena, src=0)   # RCCL over xGMI: the one collective of the path
        torch.cuda.synchronize()
        eng.arena_filled()
    else:
        eng.finalize()
    # extra contexts on the same GPU share the finished arena (weights + cross K/V): own stream, own KV cache
    engines = [eng]
    for _ in range(1, args.streams):
        e2 = hip.HipEngine(cfg, device=local_rank, max_seqs=args.batch, kv_type=kv_type, kv_positions=kv_pos)
        for t in model.tensors:
            e2.upload(t, declare_only=True)
        e2.finalize(external_arena=eng.arena_ptr())
        e2.arena_filled()
        engines.append(e2)
    log(f"[rank {rank}] weights ready in {time.perf_counter() - t_load:.1f}s, arena {eng.arena_bytes() / 1e6:.0f} MB, {len(engines)} context(s)")

    all_prompts = [make_prompts(cfg, args.batch, args.prompt_len, rank * 64 + i) for i in range(args.streams)]
    prompts = all_prompts[0]
    if args.streams > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=args.streams)

        def run_all(n_audio_, timings_=None):
            tl = [[] for _ in engines]
            futs = [pool.submit(run_utterance_batch, e, cfg, p, n_audio_, t) for e, p, t in zip(engines, all_prompts, tl)]
            tot = sum(f.result() for f in futs)
            if timings_ is not None:
                timings_.append(tuple(np.mean([t[0][i] for t in tl]) for i in range(3)))
            return tot
    else:
        def run_all(n_audio_, timings_=None):
            return run_utterance_batch(eng, cfg, prompts, n_audio_, timings_)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        run_all(n_audio)
    barrier()
    for e in engines:
        e.profile(2)   # HIP-event pairs around the (never graph-captured) DAC launches, live in the timed region
    timings = []
    t0 = time.perf_counter()
    n_samples = 0
    for _ in range(args.steps):
        n_samples += run_all(n_audio, timings)
    barrier()
    elapsed = time.perf_counter() - t0
    live = {}
    for e in engines:
        for k, v in e.profile_get().items():
            a = live.setdefault(k, dict(ms_total=0.0, launches=0, bytes_total=0.0, flops_total=0.0))
            for f in a:
                a[f] += v[f]
        e.profile(0)
    if dist is not None:
        elapsed, n_samples = tdist.reduce_timing(elapsed, n_samples, device=f"cuda:{local_rank}")

    audio_seconds = n_samples / SAMPLE_RATE
    value = audio_seconds / elapsed
    tm = np.array(timings)
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
        "dtype_detail": DTYPE_DETAIL[args.wtype] if args.dac_wtype == "f32" else
                        DTYPE_DETAIL[args.wtype].replace("DAC codec f32 (exact-f32 MFMA)", "DAC codec F16 tensors (fp16 im2col x fp16 kernels, fp16 MFMA, f32 accumulate)"),
        "data": "synthetic (seeded random weights of the Parler-TTS-Mini + DAC-44k architecture; fixed-length greedy generation)",
        "config": {
            "workload": f"configs[1]: Parler-TTS-Mini {WNAME} on MI355X, {DECODE} + DAC codec ({args.dac_wtype} tensors); {args.streams} context(s) x {args.batch} utterances/GPU in lock-step, "
                        f"{args.prompt_len}-id prompt, {n_audio} audio steps (={n_audio - cfg.n_out + 1} frames, "
                        f"{(n_audio - cfg.n_out + 1) * cfg.hop / SAMPLE_RATE:.2f} s audio) per utterance",
            "utterances_per_gpu": args.batch * args.streams, "contexts_per_gpu": args.streams, "lockstep_batch": args.batch, "audio_steps": n_audio, "prompt_len": args.prompt_len,
            "kv_cache": args.kv, "parallelism": f"dp{world} (one process per GPU, RCCL weight broadcast, no per-step collective)",
        },
        "real_time_factor": round(elapsed / audio_seconds, 6),
        "x_real_time_per_gpu": round(value / world, 3),
        "ms_per_decode_step": round(float(tm[:, 1].mean()) / n_audio * 1e3, 4),
        "phase_ms": {"prefill": round(float(tm[:, 0].mean()) * 1e3, 3), "ar_loop": round(float(tm[:, 1].mean()) * 1e3, 3),
                     "dac": round(float(tm[:, 2].mean()) * 1e3, 3)},
    }

    if rank == 0 and world == 1:
        if not args.no_roofline:
            # per-kernel-class HIP-event timing of the same workload on context 0 (eager launches, every launch
            # of every class bracketed by an event pair on the context's stream)
            prof_steps = n_audio
            eng.profile(True)
            run_utterance_batch(eng, cfg, prompts, prof_steps)
            stats = eng.profile_get()
            eng.profile(False)
            tot = sum(v["ms_total"] for v in stats.values()) or 1.0
            dom = max(stats, key=lambda k: stats[k]["ms_total"])
            st = stats[dom]
            src = "separate eager pass of the same workload on context 0 (decoder launches live inside hipGraphs in the timed region)"
            share = st["ms_total"] / tot
            if live.get(dom, {}).get("launches"):
                st = live[dom]   # the dominant kernel was bracketed live in the timed region
                src = "HIP events around every launch of this kernel in the timed region (all contexts)"
            per_launch_ms = st["ms_total"] / max(st["launches"], 1)
            if dom.startswith("dac_conv"):
                ach = st["flops_total"] / (st["ms_total"] * 1e-3) / 1e12
                peak = F16_PEAK_TFLOPS if args.dac_wtype == "f16" else F32_PEAK_TFLOPS
                roof = {"bound": "mfma", "achieved": round(ach, 3), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(ach / peak, 4), "traffic": None,
                        "note": "fp16 conv (F16 tensors, fp16 im2col): dense fp16 MFMA peak" if args.dac_wtype == "f16" else
                                "fp32 conv: peak = fp32 vector/fp32-input-MFMA peak (exact-fp32 numerics)"}
            else:
                ach = st["bytes_total"] / (st["ms_total"] * 1e-3) / 1e9
                roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None}
            roof["traffic"] = pmc_traffic(dom, args, n_audio)
            roof.update({"kernel": dom, "timing_source": src, "avg_launch_us": round(per_launch_ms * 1e3, 3), "launches": st["launches"],
                         "share_of_kernel_time": round(share, 3),
                         "algorithmic_bytes_per_launch": round(st["bytes_total"] / max(st["launches"], 1), 1)})
            out["roofline"] = roof
            out["kernel_classes"] = {
                k: {"ms": round(v["ms_total"], 3), "launches": v["launches"],
                    "GBps": round(v["bytes_total"] / max(v["ms_total"], 1e-9) / 1e6, 1),
                    "TFLOPs": round(v["flops_total"] / max(v["ms_total"], 1e-9) / 1e9, 3)}
                for k, v in stats.items() if v["launches"]}
        if not args.no_cpu_baseline:
            threads = args.cpu_threads or min(len(os.sched_getaffinity(0)), 32)
            out["cpu_baseline"] = cpu_baseline(model, cfg, prompts[0], threads)
    for e in engines[1:]:
        e.close()
    eng.close()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

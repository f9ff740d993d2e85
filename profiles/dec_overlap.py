#!/usr/bin/env python3
"""Do two decoder loops on one GPU overlap (one runner's HBM-bound self-attention under the other's L2-bound GEMMs)?  And does a decoder loop
make progress under a codec pass?  Parler-Mini fp16, fp32 KV, lock-step rows; everything through the C ABI, contexts on their own streams,
one Python thread per context (ctypes releases the GIL).  Prints ms per decoder step of 1024 rows for: one context, two contexts side by
side, and one context with a codec pass (64 x 248 frames, its own context) looping beside it.
usage: dec_overlap.py [rows=1024] [steps=96]"""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tts_cpp_amd  # noqa: F401
from tts_cpp_amd import gguf, hip, synth

R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 96
cfg = synth.parler_mini(weight_type=gguf.F16, max_gen=16 + 272)
model = synth.build(cfg)
rng = np.random.default_rng(3)


def mk(rows, share=None):
    e = hip.HipEngine(cfg, max_seqs=rows, kv_positions=16 + 272)
    if share is None:
        e.load(model)
    else:
        e.load(model, declare_only=True, external_arena=share.arena_ptr())
        e.arena_filled()
    return e


def loop(e, rows, out, key):
    prompts = [rng.integers(3, cfg.prompt_vocab, 16).astype(np.uint32) for _ in range(rows)]
    e.reset(); e.prefill_batch(prompts)
    e.generate_greedy([16] * rows, 8)          # capture
    e.reset(); e.prefill_batch(prompts)
    e.synchronize()
    out[key + ":ready"] = True
    while not out.get("go"):
        time.sleep(0.0005)
    t0 = time.perf_counter()
    e.generate_greedy([16] * rows, STEPS)
    out[key] = (time.perf_counter() - t0) * 1e3 / STEPS


def run(engines, rows, extra=None):
    out = {}
    ths = [threading.Thread(target=loop, args=(e, rows, out, f"d{i}")) for i, e in enumerate(engines)]
    for t in ths: t.start()
    while sum(1 for k in out if k.endswith(":ready")) < len(engines): time.sleep(0.001)
    stop = {}
    th2 = None
    if extra:
        th2 = threading.Thread(target=extra, args=(out, stop)); th2.start(); time.sleep(0.3)
    t0 = time.perf_counter()
    out["go"] = True
    for t in ths: t.join()
    wall = (time.perf_counter() - t0) * 1e3
    if th2:
        stop["stop"] = True; th2.join()
    return out, wall


a = mk(R)
o, wall = run([a], R)
base = o["d0"]
print(f"one context, {R} rows: {base:.3f} ms/step")
if os.environ.get("DEC_OVERLAP_ONE"):
    sys.exit(0)
b = mk(R, share=a)
o, wall = run([a, b], R)
print(f"two contexts x {R} rows side by side: {o['d0']:.3f} / {o['d1']:.3f} ms/step each, wall {wall / STEPS:.3f} ms per step pair = {wall / STEPS / 2:.3f} per {R} rows  ({2 * base / (wall / STEPS):.2f}x of serial)")
b.close()
if R >= 512:
    h1, h2 = mk(R // 2, share=a), mk(R // 2, share=a)
    o, wall = run([h1, h2], R // 2)
    print(f"two contexts x {R // 2} rows: wall {wall / STEPS:.3f} ms per step of {R} rows ({base / (wall / STEPS):.2f}x of one context)")
    h1.close(); h2.close()

# decoder loop under a looping codec pass
dcfg = synth.parler_mini(layers=1, prompt_vocab=64, ctx=64, max_gen=256)
dmodel = synth.build(dcfg)
d = hip.HipEngine(dcfg, flags=hip.FLAG_NO_PARLER)
d.load(dmodel)
utts = [rng.integers(0, dcfg.cb_size, (248, dcfg.n_out)).astype(np.uint32) for _ in range(64)]
d.dac_decode_batch(utts)
t0 = time.perf_counter(); n = 3
for _ in range(n): d.dac_decode_batch(utts)
codec_ms = (time.perf_counter() - t0) * 1e3 / n
print(f"codec pass alone (64 x 248 frames): {codec_ms:.1f} ms")


def codec_loop(out, stop):
    k = 0; t0 = time.perf_counter()
    while not stop.get("stop"):
        d.dac_decode_batch(utts); k += 1
    out["codec_passes"] = k; out["codec_wall"] = (time.perf_counter() - t0) * 1e3


STEPS = STEPS * 2
o, wall = run([a], R, extra=codec_loop)
passes, cw = o["codec_passes"], o["codec_wall"]
print(f"decoder under a looping codec: {o['d0']:.3f} ms/step (alone {base:.3f}); codec {cw / passes:.1f} ms per pass (alone {codec_ms:.1f}) over {passes} passes")
dec_work = STEPS * base
print(f"   work done in {wall:.0f} ms of wall: decoder {dec_work:.0f} ms-alone + codec ~{wall / (cw / passes) * codec_ms:.0f} ms-alone -> {(dec_work + wall / (cw / passes) * codec_ms) / wall:.2f}x of time-shared")

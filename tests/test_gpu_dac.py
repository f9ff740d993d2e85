"""GPU parity: the HIP DAC decoder against the oracle and the golden vectors.
Tolerance: PCM (tanh output, |x|<=1) 1e-4 absolute; intermediate activations 1e-4 relative to max|x|."""
import os

import numpy as np
import pytest

import oracle as orc
from tts_cpp_amd import gguf, hip, synth

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden", "tiny_f32.npz")


def relerr(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def dac_engine(cfg, model):
    eng = hip.HipEngine(cfg, flags=hip.FLAG_NO_PARLER)
    eng.load(model)
    return eng


def test_tiny_stages_and_golden():
    model = synth.build(synth.tiny(weight_type=gguf.F32))
    cfg = model.cfg
    g = np.load(GOLD)
    eng = dac_engine(cfg, model)
    eng.set_debug(True)
    pcm = eng.dac_decode(g["codes"])
    assert np.abs(pcm - g["pcm"]).max() < 1e-4, "golden PCM (float64 torch restatement)"
    d = orc.DacOracle(model)
    for st in range(2 + len(cfg.strides)):
        _, ref = d.decode(g["codes"], stage=st)
        act = eng.debug_read(f"dac:{st}", ref.size).reshape(ref.shape)
        assert relerr(act, ref) < 1e-4, f"stage {st}"
        assert relerr(act, g[f"dac_stage{st}"]) < 1e-4, f"golden stage {st}"
    eng.close()


@pytest.mark.parametrize("frames", [1, 2, 13, 70])
def test_small_dac_matches_oracle(frames):
    model = synth.build(synth.small(weight_type=gguf.F32))
    cfg = model.cfg
    eng = dac_engine(cfg, model)
    codes = np.random.default_rng(frames).integers(0, cfg.cb_size, (frames, cfg.n_out)).astype(np.uint32)
    pcm = eng.dac_decode(codes)
    ref = orc.DacOracle(model).decode(codes)
    assert pcm.shape == (frames * 512,)
    assert np.abs(pcm - ref).max() < 1e-4
    assert np.abs(pcm).max() <= 1.0
    eng.close()


@pytest.mark.parametrize("frames", [3, 37])
def test_mfma_conv_paths_match_oracle_and_valu(frames):
    """Channel counts that exercise every MFMA tile shape (conv 128/64/96-channel tiles, convT strides
    8/4/2) at a size the oracle finishes quickly; the scalar-FMA kernels must agree too."""
    cfg = synth.small(weight_type=gguf.F32, latent=64, c0=768, strides=(8, 4, 2), max_gen=64)
    model = synth.build(cfg)
    codes = np.random.default_rng(frames).integers(0, cfg.cb_size, (frames, cfg.n_out)).astype(np.uint32)
    ref = orc.DacOracle(model).decode(codes)
    outs = []
    for flags in (hip.FLAG_NO_PARLER, hip.FLAG_NO_PARLER | hip.FLAG_VALU_GEMM):
        eng = hip.HipEngine(cfg, flags=flags)
        eng.load(model)
        eng.set_debug(True)
        pcm = eng.dac_decode(codes)
        assert pcm.shape == ref.shape
        assert np.abs(pcm - ref).max() < 1e-4, flags
        if flags & hip.FLAG_VALU_GEMM == 0:
            d = orc.DacOracle(model)
            for st in range(2 + len(cfg.strides)):
                _, r = d.decode(codes, stage=st)
                act = eng.debug_read(f"dac:{st}", r.size).reshape(r.shape)
                assert relerr(act, r) < 1e-4, f"stage {st}"
        outs.append(pcm)
        eng.close()
    assert np.abs(outs[0] - outs[1]).max() < 1e-5


def test_batched_dac_equals_single_decodes():
    """utterances of different lengths in one pass (grid.z, per-utterance valid lengths) == one decode each"""
    cfg = synth.small(weight_type=gguf.F32, latent=64, c0=768, strides=(8, 4, 2), max_gen=64)
    model = synth.build(cfg)
    eng = dac_engine(cfg, model)
    rng = np.random.default_rng(5)
    codes = [rng.integers(0, cfg.cb_size, (f, cfg.n_out)).astype(np.uint32) for f in (7, 1, 19, 0, 12)]
    batch = eng.dac_decode_batch(codes)
    for c, pcm in zip(codes, batch):
        single = eng.dac_decode(c)
        assert pcm.shape == single.shape
        assert np.array_equal(pcm, single), "batched and single decode must agree bit for bit"
    ref = orc.DacOracle(model).decode(codes[2])
    assert np.abs(batch[2] - ref).max() < 1e-4
    eng.close()


# fp16-im2col noise floor.  Rounding a conv input to fp16 turns a relative difference eps between two fp32
# implementations (summation order) into flips of 1 fp16 ulp (2^-10) with probability eps*2^10 per element, i.e. an
# output difference of about 2^-10*sqrt(eps*2^10); iterating eps -> 0.03*sqrt(eps) over the 6 roundings per block
# converges to ~5e-4 whatever the starting point (measured: 2.5e-7 after the first conv, 3e-5 / 1.3e-4 / 4e-4 after
# blocks 1 / 2 / 4).  ggml run with another thread count differs from itself by the same amount.  So: the first
# convs are checked at rounding level (that is where fp16 vs fp32 im2col differ 1000x), the end at the floor.
F16_DAC_TOL = 1e-3


def test_f16_dac_tensors_and_empty_input():
    """--convert-dac-to-f16 GGUFs (quantize_impl.cpp:264-266): ggml's conv_1d / conv_transpose_1d round their input to
    fp16 when the kernel is F16 (fp16 im2col), accumulate in fp32."""
    model = synth.build(synth.tiny(weight_type=gguf.F32, dac_f16=True))
    cfg = model.cfg
    eng = dac_engine(cfg, model)
    codes = np.random.default_rng(1).integers(0, cfg.cb_size, (5, cfg.n_out)).astype(np.uint32)
    eng.set_debug(True)
    pcm = eng.dac_decode(codes)
    o16, o32 = orc.DacOracle(model), orc.DacOracle(model, f16_conv=0)  # default follows the GGUF types: fp16 im2col
    assert o16.m.f16_conv == 1
    for stage in range(0, 2 + len(cfg.strides)):
        _, st_ref = o16.decode(codes, stage=stage)
        _, st_32 = o32.decode(codes, stage=stage)
        st = eng.debug_read(f"dac:{stage}", st_ref.size).reshape(st_ref.shape)
        if stage <= 1:
            assert relerr(st, st_ref) < 1e-5, f"stage {stage}"          # quantizer + first conv: exact semantics
        else:
            assert relerr(st, st_ref) < F16_DAC_TOL, f"stage {stage}"
        if stage in (1, 2):
            assert relerr(st, st_ref) * 5 < relerr(st, st_32), f"stage {stage}: not the fp16-im2col result"
    assert np.abs(pcm - o16.decode(codes)).max() < F16_DAC_TOL
    # fp32 activations on the same F16 tensors (TTS_HIP_FLAG_DAC_F32)
    eng32 = hip.HipEngine(cfg, flags=hip.FLAG_NO_PARLER | hip.FLAG_DAC_F32)
    eng32.load(model)
    assert np.abs(eng32.dac_decode(codes) - o32.decode(codes)).max() < 1e-4
    eng32.close()
    assert eng.dac_decode(np.zeros((0, cfg.n_out), dtype=np.uint32)).size == 0
    with pytest.raises(hip.HipError):
        eng.dac_decode(np.full((2, cfg.n_out), cfg.cb_size, dtype=np.uint32))  # code outside the codebook
    eng.close()


@pytest.mark.parametrize("dac_f16", [False, True])
def test_full_size_dac_and_locality(dac_f16):
    """DAC 44 kHz dims (1536->96 channels, x512), F32 tensors (exact-fp32 MFMA) and F16 tensors (fp16 MFMA, fp16
    im2col).  Parity on 3 frames; at 40 frames, size-independent properties: determinism, and locality — the decoder
    is a finite-receptive-field convolution stack, so PCM far from an edited frame is unchanged bit for bit."""
    model = synth.build(synth.parler_mini(layers=1, prompt_vocab=64, ctx=64, dac_f16=dac_f16))  # decoder part irrelevant here
    cfg = model.cfg
    eng = dac_engine(cfg, model)
    rng = np.random.default_rng(2)
    codes = rng.integers(0, cfg.cb_size, (3, cfg.n_out)).astype(np.uint32)
    eng.set_debug(True)
    pcm = eng.dac_decode(codes)
    eng.set_debug(False)
    ref = orc.DacOracle(model).decode(codes)
    assert np.abs(pcm - ref).max() < (F16_DAC_TOL if dac_f16 else 2e-4)
    _, st_ref = orc.DacOracle(model).decode(codes, stage=1)       # every 128-channel k7 tile, K = 7 x 1024
    assert relerr(eng.debug_read("dac:1", st_ref.size).reshape(st_ref.shape), st_ref) < 1e-5
    big = rng.integers(0, cfg.cb_size, (40, cfg.n_out)).astype(np.uint32)
    a = eng.dac_decode(big)
    b = eng.dac_decode(big)
    assert np.array_equal(a, b)
    edited = big.copy()
    edited[35] = (edited[35] + 1) % cfg.cb_size
    c = eng.dac_decode(edited)
    # receptive field: initial conv 3 frames + per block (convT 1 + residual 3*(1+3+9)=39 samples at that rate) < 8 frames
    assert np.array_equal(a[: 20 * 512], c[: 20 * 512])
    assert not np.array_equal(a[34 * 512: 36 * 512], c[34 * 512: 36 * 512])
    eng.close()


def test_full_size_quantizer_embedding_tile_kernel():
    """9 codebooks x 8 dims -> 1024 channels (dac_embed_tile_kernel): stage 0 against the oracle, and utterances of
    0 / 1 / 70 / 129 frames in one pass (the kernel works on 64-frame tiles) equal to one decode each."""
    model = synth.build(synth.parler_mini(layers=1, prompt_vocab=64, ctx=64))
    cfg = model.cfg
    eng = dac_engine(cfg, model)
    rng = np.random.default_rng(9)
    codes = rng.integers(0, cfg.cb_size, (70, cfg.n_out)).astype(np.uint32)
    eng.set_debug(True)
    pcm = eng.dac_decode(codes)
    eng.set_debug(False)
    _, st0 = orc.DacOracle(model).decode(codes, stage=0)
    assert relerr(eng.debug_read("dac:0", st0.size).reshape(st0.shape), st0) < 1e-6
    ragged = [rng.integers(0, cfg.cb_size, (f, cfg.n_out)).astype(np.uint32) for f in (1, 129, 0)] + [codes]
    batch = eng.dac_decode_batch(ragged)
    assert np.array_equal(batch[3], pcm)
    for c, out in zip(ragged[:3], batch[:3]):
        assert np.array_equal(out, eng.dac_decode(c))
    eng.close()


ARITH_CASES = [
    # (id, environment, tts_hip_tune keys, tts_hip_dac_arith bits expected)
    ("default_fp16_hi_lo", {}, {}, 1 | 2 | 4 | 32 | 64),              # the product default since round 6: every conv as fp16 hi + lo split products (three per product)
    ("bf16x3", {}, {"dac_split": 0}, 1 | 2 | 4 | 32),                 # rounds 3 - 5: bf16 x 3 split products (six per product)
    ("bf16x3_by_env", {"TTS_HIP_DAC_SPLIT": "0"}, {}, 1 | 2 | 4 | 32),
    ("exact_fp32", {"TTS_HIP_DAC_BF16X3": "0"}, {}, 0),              # the exact-fp32 MFMA pipe the default is held against
    ("units_unfused", {}, {"dac_fuse": 0}, 1 | 4 | 32 | 64),          # residual units at 96 / 192 channels as two launches
    ("convt_fp32", {}, {"dac_convt_b3": 0}, 1 | 2 | 32 | 64),         # transposed convs on the exact-fp32 kernel
    ("no_planes", {}, {"dac_planes": 0}, 1 | 2 | 4 | 64),             # wide classes keep fp32 activations (conv1d_mfma_b3_kernel + fp32 k = 1)
    ("tap_pairs", {}, {"dac_tap7": 0}, 1 | 2 | 4 | 32 | 64),          # k = 7 convs with tap-pair k-steps (8 slots for 7 taps); the tap-pair unit kernel stays bf16 x 3
    ("convt_fp32_input", {}, {"dac_convt_planes": 0}, 1 | 2 | 4 | 32 | 64),  # the split transposed convs stage fp32 input themselves instead of the producer's planes
]


@pytest.mark.parametrize("case", ARITH_CASES, ids=[c[0] for c in ARITH_CASES])
def test_codec_arithmetic_default_and_every_fallback_match_oracle(case, monkeypatch):
    """The DAC-44k stage check under the product default (since round 6: fp16 hi + lo split products on v_mfma_f32_32x32x16_f16 — three per
    product, ~2^-22 relative per product —, fused residual units, split planes) and under every switch that survives as a fallback, the bf16 x 3
    products of rounds 3 - 5 among them — one bar for all: PCM 2e-4, stages 1e-5 relative (measured values are printed: run with -s).
    A fallback nobody runs is a second product nobody verifies: the switches this test does not name were deleted in round 4."""
    _, env, tune, arith = case
    worst = {}
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    model = synth.build(synth.parler_mini(layers=1, prompt_vocab=64, ctx=64))
    cfg = model.cfg
    eng = hip.HipEngine(cfg, flags=hip.FLAG_NO_PARLER, tune=tune)
    eng.load(model)
    assert eng.L.tts_hip_dac_arith(eng.ctx) == arith
    codes = np.random.default_rng(2).integers(0, cfg.cb_size, (3, cfg.n_out)).astype(np.uint32)
    eng.set_debug(True)
    pcm = eng.dac_decode(codes)
    eng.set_debug(False)
    o = orc.DacOracle(model)
    worst["pcm"] = float(np.abs(pcm - o.decode(codes)).max())
    assert worst["pcm"] < 2e-4
    for st in range(2 + len(cfg.strides)):
        _, ref = o.decode(codes, stage=st)
        worst[st] = relerr(eng.debug_read(f"dac:{st}", ref.size).reshape(ref.shape), ref)
        assert worst[st] < 1e-5, f"stage {st}"
    print(case[0], "PCM %.2e, stages " % worst["pcm"] + " ".join("%.1e" % worst[st] for st in range(2 + len(cfg.strides))))
    ragged = [np.random.default_rng(f).integers(0, cfg.cb_size, (f, cfg.n_out)).astype(np.uint32) for f in (1, 5)] + [codes]
    batch = eng.dac_decode_batch(ragged)
    assert np.array_equal(batch[2], pcm)
    eng.close()


def test_unknown_tuning_key_is_refused():
    model = synth.build(synth.tiny(weight_type=gguf.F32))
    eng = hip.HipEngine(model.cfg, flags=hip.FLAG_NO_PARLER)
    with pytest.raises(hip.HipError, match="unknown key"):
        eng.tune("dac_variant", 3)
    eng.close()


def test_measured_codec_shape_windows_match_oracle():
    """The shape bench.py measures: ONE codec pass of 64 utterances x 248 frames at the DAC-44k dims.  The decoder is a stack of finite
    receptive-field convolutions (3 frames for the first conv, 1 + 39 / 8 for the first block at 8 samples per frame, under 1 for the
    rest: about 10 frames on either side), so a 2-frame window of any utterance equals the same window of an oracle decode of the
    surrounding +-12 frames: three random windows per utterance for eight utterances spread
    over the pass (first, last, tile boundaries of grid.z) at the full-size tolerance, plus every utterance against its own single decode
    bit for bit for four of them."""
    model = synth.build(synth.parler_mini(layers=1, prompt_vocab=64, ctx=64))
    cfg = model.cfg
    eng = dac_engine(cfg, model)
    rng = np.random.default_rng(64248)
    U, F, HALO, hop = 64, 248, 12, 512
    utts = [rng.integers(0, cfg.cb_size, (F, cfg.n_out)).astype(np.uint32) for _ in range(U)]
    out = eng.dac_decode_batch(utts)
    assert len(out) == U and all(o.shape == (F * hop,) for o in out)
    o = orc.DacOracle(model)
    worst = 0.0
    for u in (0, 1, 15, 16, 31, 32, 62, 63):
        for f0 in [0, F - 2] + list(rng.integers(HALO, F - HALO - 2, 1)):
            lo, hi = max(0, int(f0) - HALO), min(F, int(f0) + 2 + HALO)
            ref = o.decode(utts[u][lo:hi])
            w = slice((int(f0) - lo) * hop, (int(f0) - lo + 2) * hop)
            err = float(np.abs(out[u][int(f0) * hop:(int(f0) + 2) * hop] - ref[w]).max())
            worst = max(worst, err)
            assert err < 2e-4, f"utterance {u}, frames {int(f0)}..{int(f0) + 1}: {err:.2e}"
    for u in (0, 17, 40, 63):
        assert np.array_equal(out[u], eng.dac_decode(utts[u])), f"utterance {u}: batched pass differs from its own decode"
    print(f"64 x 248 frames: worst window error {worst:.2e}")
    eng.close()

"""GPU parity: the Dia encoder + decoder step (tts_hip_dia_encode / tts_hip_dia_step) against the oracle (orc_dia_*, pinned
to a float64 torch golden by tests/test_oracle_cpu.py).  Tolerances as for the other decoders: 2e-4 of max|oracle| with
F32 weights, 2e-3 with F16, the Q8_0-activation flip bound for quantised matrices."""
import os

import numpy as np
import pytest

import oracle as orc
from tts_cpp_amd import gguf, hip, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "tiny_dia.npz")


def relerr(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize("wtype,tol", [(gguf.F32, 2e-4), (gguf.F16, 2e-3), (gguf.Q8_0, 3e-2)])
def test_dia_encoder_and_steps_match_oracle(wtype, tol):
    # quantised matrices take the integer path when their rows are multiples of 256: widen the tiny encoder for that case
    kw = dict(enc_hidden=256) if wtype == gguf.Q8_0 else {}
    model = synth.build_dia(synth.dia_tiny(weight_type=wtype, **kw))
    eng = hip.DiaEngine(model.cfg)
    eng.load(model)
    o = orc.DiaOracle(model, act_mode=1)
    g = np.load(GOLD)
    n = int(g["sentence_len"])
    enc = eng.encode(g["tokens"], n, want_states=True)
    ref_enc = o.encode(g["tokens"], n, want_states=True)
    assert relerr(enc, ref_enc) < tol
    if wtype == gguf.F32:
        assert relerr(enc, g["enc"]) < 2e-4
    for s_ in range(len(g["ids"])):
        lg, raw = eng.step(g["ids"][s_], s_, want_raw=True)
        ref, ref_raw = o.step(g["ids"][s_], s_, want_raw=True)
        assert relerr(raw, ref_raw) < tol, s_
        assert relerr(lg, ref) < 4 * tol, s_          # cond + 3 (cond - uncond) amplifies the raw error
        if wtype == gguf.F32:
            assert relerr(raw, g["raw"][s_]) < 2e-4
    eng.close()


def test_dia_second_sentence_and_errors():
    """A second encode on the same context replaces the cross K/V (keys beyond the new, shorter sentence are zero again,
    as in a fresh runner), and the self-attention cache restarts from position 0."""
    model = synth.build_dia(synth.dia_tiny())
    cfg = model.cfg
    eng = hip.DiaEngine(cfg)
    eng.load(model)
    with pytest.raises(hip.HipError):
        eng.step(np.full(cfg.n_out, cfg.bos, dtype=np.uint32), 0)      # no encode yet
    long_t, long_n = orc.dia_tokenize("[S1] a longer text.", cfg.max_ctx)
    short_t, short_n = orc.dia_tokenize("[S2] hi", cfg.max_ctx)
    ids = np.full(cfg.n_out, cfg.bos, dtype=np.uint32)
    eng.encode(long_t, long_n)
    eng.step(ids, 0)
    eng.step(ids, 1)
    eng.encode(short_t, short_n)
    got = eng.step(ids, 0)
    o = orc.DiaOracle(model, act_mode=1)
    o.encode(short_t, short_n)
    assert relerr(got, o.step(ids, 0)) < 8e-4
    with pytest.raises(hip.HipError):
        eng.step(ids, cfg.max_gen)
    with pytest.raises(hip.HipError):
        eng.step(np.full(cfg.n_out, cfg.out_vocab, dtype=np.uint32), 1)
    with pytest.raises(hip.HipError):
        eng.encode(short_t, 0)
    eng.close()


def test_dia_runner_generates_through_both_contexts(tmp_path):
    """runner_from_file on a Dia GGUF (dia.* + audio_encoder.*): byte tokenisation with speaker tags (model.cpp:661-705),
    greedy generate_from_batch with the end-of-sequence countdown (:767-833), un-delay (:787-808), DAC — equal to the oracle
    pipeline."""
    from tts_cpp_amd import runner
    model = synth.build_dia(synth.dia_tiny(), suppress_special=True)
    cfg = model.cfg
    path = model.write_gguf(str(tmp_path / "dia.gguf"))
    r = runner.Runner(path, sample=0)
    assert r.arch == "dia" and r.sampling_rate == 44100.0
    text = " Hi there [S2] ok"
    pcm = r.generate(text, sample=0)
    toks, n = orc.dia_tokenize(text, cfg.max_ctx)
    assert r.last_tokens(0).tolist() == toks.tolist()
    o = orc.DiaOracle(model, act_mode=1)
    outs, frames = o.generate(text)
    got = r.last_tokens(1).reshape(-1, cfg.n_out)
    assert got.shape == outs.shape == (cfg.max_gen - 1, cfg.n_out)
    assert np.array_equal(got, outs)
    ref = orc.DacOracle(model.dac).decode(frames)
    assert pcm.shape == ref.shape == (len(frames) * cfg.hop,)
    assert np.abs(pcm - ref).max() < 1e-4
    short = r.generate(text, sample=0, max_tokens=24)               # config.max_tokens bounds the loop (:812-818): the countdown starts at step 9
    outs_s, frames_s = o.generate(text, max_tokens=24)
    assert np.array_equal(r.last_tokens(1).reshape(-1, cfg.n_out), outs_s) and len(outs_s) == 23
    assert np.abs(short - orc.DacOracle(model.dac).decode(frames_s)).max() < 1e-4
    sampled = r.generate(text, sample=1, top_k=8, seed=5)
    assert sampled.size % cfg.hop == 0 and np.isfinite(sampled).all()
    with pytest.raises(runner.RunnerError):
        r.generate(text, sample=0, max_tokens=7)                    # GGML_ASSERT(max_tokens == 0 || max_tokens > max_delay)
    r.close()


@pytest.mark.parametrize("wtype,tol", [(gguf.F32, 2e-4), (gguf.F16, 2e-3)])
def test_dia_lockstep_batch_of_4_equals_4_single_oracles(wtype, tol):
    """BASELINE config 3's per-GPU unit: 4 utterances x 2 guidance rows in one step (tts_hip_dia_step_batch), each utterance with
    its own sentence (cross K/V slot), ids and position; every utterance must equal a stand-alone oracle of that utterance.
    Utterance 2 starts two steps late (its position lags), utterance 3 sits in a non-contiguous slot order."""
    model = synth.build_dia(synth.dia_tiny(weight_type=wtype))
    cfg = model.cfg
    eng = hip.DiaEngine(cfg, max_utterances=4)
    eng.load(model)
    texts = ["[S1] first one.", "[S2] the second is long.", "[S1] hi.", "[S1] a [S2] b [S1] c."]   # <= 24 characters (tiny max_ctx) once the tags are single bytes
    slots = [0, 1, 3, 2]
    oracles, rng = [], np.random.default_rng(11)
    for s, t in zip(slots, texts):
        toks, n = orc.dia_tokenize(t, cfg.max_ctx)
        eng.encode_slot(s, toks, n)
        o = orc.DiaOracle(model, act_mode=1)
        o.encode(toks, n)
        oracles.append(o)
    pos = np.zeros(4, dtype=np.uint32)
    ids = np.full((4, cfg.n_out), cfg.bos, dtype=np.uint32)
    for step in range(6):
        active = [u for u in range(4) if not (u == 2 and step < 2)]      # utterance 2 joins at step 2
        lg, raw = eng.step_batch(ids[active], pos[active], slots=[slots[u] for u in active], want_raw=True)
        for k, u in enumerate(active):
            ref, ref_raw = oracles[u].step(ids[u], int(pos[u]), want_raw=True)
            assert relerr(raw[k], ref_raw) < tol, (step, u)
            assert relerr(lg[k], ref) < 4 * tol, (step, u)
            ids[u] = rng.integers(0, cfg.audio_vocab, cfg.n_out)        # the same (arbitrary) ids on both sides
            pos[u] += 1
    with pytest.raises(hip.HipError):
        eng.step_batch(ids[:1], pos[:1], slots=[4])                     # slot outside max_utterances
    eng.close()
    one = hip.DiaEngine(cfg)                                            # a one-utterance context refuses a second slot
    one.load(model)
    with pytest.raises(hip.HipError):
        one.encode_slot(1, *orc.dia_tokenize(texts[0], cfg.max_ctx))
    one.close()


def test_dia_seven_utterances_take_the_16_row_streaming_gemm():
    """7 utterances = 14 rows: gemv_stream_kernel keeps 16 rows of the activation slice in LDS (8 up to 4 utterances), slabs folded by the
    consumers — every utterance against its own oracle, fp16 matrices"""
    model = synth.build_dia(synth.dia_tiny(weight_type=gguf.F16))
    cfg = model.cfg
    eng = hip.DiaEngine(cfg, max_utterances=7)
    eng.load(model)
    texts = ["[S1] one.", "[S2] two two.", "[S1] three.", "[S2] four.", "[S1] five five.", "[S2] six.", "[S1] seven."]
    oracles, rng = [], np.random.default_rng(7)
    for u, t in enumerate(texts):
        toks, n = orc.dia_tokenize(t, cfg.max_ctx)
        eng.encode_slot(u, toks, n)
        o = orc.DiaOracle(model, act_mode=1)
        o.encode(toks, n)
        oracles.append(o)
    ids = np.full((7, cfg.n_out), cfg.bos, dtype=np.uint32)
    for step in range(3):
        lg = eng.step_batch(ids, np.full(7, step, dtype=np.uint32))
        for u in range(7):
            assert relerr(lg[u], oracles[u].step(ids[u], step)) < 8e-3, (step, u)
            ids[u] = rng.integers(0, cfg.audio_vocab, cfg.n_out)
    eng.close()


def test_dia_runner_generate_batch_equals_separate_generates(tmp_path):
    """dia_runner::generate_batch (4 utterances in lock-step, per-utterance countdown and un-delay) == 4 generate() calls, greedy:
    token streams identical, audio equal"""
    from tts_cpp_amd import runner
    model = synth.build_dia(synth.dia_tiny(), suppress_special=True)
    cfg = model.cfg
    path = model.write_gguf(str(tmp_path / "dia.gguf"))
    texts = [" Hi there [S2] ok", "[S1] another one.", "[S2] short", "[S1] the fourth one."]
    r = runner.Runner(path, sample=0, max_seqs=4)
    singles, toks = [], []
    for t in texts:
        singles.append(r.generate(t, sample=0, max_tokens=30))
        toks.append(r.last_tokens(1).copy())
    batch = r.generate_batch(texts, sample=0, max_tokens=30)
    for b, s_ in zip(batch, singles):
        assert b.shape == s_.shape and b.size > 0 and np.abs(b - s_).max() < 1e-5
    with pytest.raises(runner.RunnerError):
        r.generate_batch(texts + ["one too many"], sample=0, max_tokens=30)
    r.close()


def test_dia_device_loop_equals_the_per_step_host_loop(tmp_path):
    """tts_hip_dia_generate (check_stopping, step, guidance, sampler::sample / sampler::max and the delay-pattern feedback replayed as one
    captured graph) == the reference-shaped host loop (TTS_HOST_LOOP=1: logits D2H + sampler::sample + check_stopping on the host every
    step): ids identical under a fixed seed, audio equal; one utterance and three in lock-step.  The model keeps its special-id head
    rows (no block of exactly equal logits: the order of equal keys is where index order and std::sort may differ)."""
    import os
    from tts_cpp_amd import runner
    model = synth.build_dia(synth.dia_tiny(), suppress_special=False)
    path = model.write_gguf(str(tmp_path / "dia.gguf"))
    text = " Hi there [S2] ok"
    r = runner.Runner(path, sample=0, max_seqs=3)

    def host(fn):
        os.environ["TTS_HOST_LOOP"] = "1"
        try:
            return fn()
        finally:
            del os.environ["TTS_HOST_LOOP"]
    for kw in (dict(sample=1, top_k=8, seed=5), dict(sample=1, top_k=0, top_p=0.9, temperature=1.2, seed=7), dict(sample=1, top_k=20, repetition_penalty=1.3, seed=9),
               dict(sample=1, top_k=12, top_p=0.8, temperature=0.7, seed=11), dict(sample=0)):
        dev = r.generate(text, max_tokens=40, **kw)
        dt = r.last_tokens(1).copy()
        hst = host(lambda: r.generate(text, max_tokens=40, **kw))
        assert len(dt) > 0 and np.array_equal(dt, r.last_tokens(1)), kw
        assert np.array_equal(dev, hst), kw
    texts = [" Hi there [S2] ok", "[S1] another one.", "[S2] short"]
    for kw in (dict(sample=1, top_k=10, seed=4), dict(sample=0)):
        dev = r.generate_batch(texts, max_tokens=36, **kw)
        hst = host(lambda: r.generate_batch(texts, max_tokens=36, **kw))
        assert len(dev) == len(hst) == 3
        for a, b in zip(dev, hst):
            assert a.shape == b.shape and np.array_equal(a, b), kw
    r.close()


def test_dia_generate_entry_point_checks():
    model = synth.build_dia(synth.dia_tiny())
    cfg = model.cfg
    eng = hip.DiaEngine(cfg, max_utterances=2)
    eng.load(model)
    args = dict(delay_pattern=[0, 8, 9, 10, 11, 12, 13, 14, 15], bos=cfg.bos, eos=cfg.eos, pad=cfg.pad, max_delay=cfg.max_delay)
    with pytest.raises(hip.HipError):
        eng.generate(1, 24, **args)                       # slot 0 not encoded
    toks, n = orc.dia_tokenize("[S1] hi", cfg.max_ctx)
    eng.encode_slot(0, toks, n)
    with pytest.raises(hip.HipError):
        eng.generate(2, 24, **args)                       # slot 1 not encoded
    with pytest.raises(hip.HipError):
        eng.generate(1, cfg.max_gen + 1, **args)          # beyond the self-attention cache
    with pytest.raises(hip.HipError):
        eng.generate(1, cfg.max_delay, **args)            # max_gen must exceed max_delay
    out = eng.generate(1, 24, **args)                     # greedy: the countdown starts at position 24 - 15 and ends the loop at 23 ids
    assert len(out) == 1 and out[0].shape == (23, cfg.n_out)
    o = orc.DiaOracle(model, act_mode=1)
    outs, _ = o.generate("[S1] hi", max_tokens=24)
    assert np.array_equal(out[0], outs)
    eng.close()


def test_dia_1_6b_layer_shapes():
    """one encoder and one decoder layer at nari-labs/Dia-1.6B's widths (encoder 1024 / ffn 4096 over 2 x 1024 positions, decoder
    2048 with 16 query heads on 4 k/v groups x 128, ffn 8192, 9 x 1028 logits), fp16 matrices: BASELINE config 3's shapes,
    two utterances in lock-step against per-utterance oracles"""
    model = synth.build_dia(synth.dia_1_6b(enc_layers=1, dec_layers=1, max_gen=32, weight_type=gguf.F16))
    cfg = model.cfg
    eng = hip.DiaEngine(cfg, max_utterances=2)
    eng.load(model)
    texts = ["[S1] The birch canoe slid on the smooth planks.", "[S2] Glue the sheet to the dark blue background."]
    oracles = []
    for u, t in enumerate(texts):
        toks, n = orc.dia_tokenize(t, cfg.max_ctx)
        eng.encode_slot(u, toks, n)
        o = orc.DiaOracle(model, act_mode=1)
        o.encode(toks, n)
        oracles.append(o)
    ids = np.full((2, cfg.n_out), cfg.bos, dtype=np.uint32)
    rng = np.random.default_rng(5)
    for step in range(3):
        lg, raw = eng.step_batch(ids, np.full(2, step, dtype=np.uint32), want_raw=True)
        for u in range(2):
            ref, ref_raw = oracles[u].step(ids[u], step, want_raw=True)
            assert relerr(raw[u], ref_raw) < 2e-3, (step, u)
            assert relerr(lg[u], ref) < 8e-3, (step, u)
            ids[u] = rng.integers(0, cfg.audio_vocab, cfg.n_out)
    eng.close()


def test_dia_cross_attention_in_rolling_passes_matches_the_split_kernel():
    """Round 5: the cross-attention of a Dia-1.6B step (16 heads x 2 rows x 8 slices of 128 text positions) runs as attn_gqa_wave_kernel<128, 3, EXT>:
    keys of a slice interleaved 16 by 16, a running softmax per 16-lane group, eight passes through three register slots, the query's slabs folded
    and rotated under the first rows.  Another association of the same softmax than attn_gqa_split_kernel (tune("attn_wave") = 0): guided logits
    of three steps agree inside the fp16-matrix bar (2e-3 of the largest logit), and both stay inside the oracle's bar (test_dia_1_6b_layer_shapes runs the default)."""
    model = synth.build_dia(synth.dia_1_6b(enc_layers=1, dec_layers=2, max_gen=32, weight_type=gguf.F16))
    cfg = model.cfg
    texts = ["[S1] The birch canoe slid on the smooth planks.", "[S2] Glue the sheet to the dark blue background."]
    res = {}
    for wave in (1, 0):
        eng = hip.DiaEngine(cfg, max_utterances=2)
        eng.tune("attn_wave", wave)
        eng.load(model)
        for u, t in enumerate(texts):
            eng.encode_slot(u, *orc.dia_tokenize(t, cfg.max_ctx))
        ids = np.full((2, cfg.n_out), cfg.bos, dtype=np.uint32)
        out = []
        for step in range(3):
            lg = eng.step_batch(ids, np.full(2, step, dtype=np.uint32))
            out.append(np.array(lg, copy=True))
            ids[:] = (np.arange(2 * cfg.n_out).reshape(2, -1) * 37 + step * 11) % cfg.audio_vocab
        res[wave] = np.stack(out)
        eng.close()
    assert not np.array_equal(res[1], res[0])          # the switch reaches the kernel under test
    # two layers of fp16 matrices: every projection rounds its rows to fp16 while it stages them, so a last-bit difference of the attention flips
    # roundings downstream — the two kernels are as far apart as either is from the oracle (measured 7.4e-4; the oracle bar of the raw logits: 2e-3)
    assert relerr(res[1], res[0]) < 2e-3


@pytest.mark.parametrize("shapes", ["tiny_f16", "1_6b_layer"])
def test_dia_captured_step_folds_slice_merge_and_silu_into_the_projections(shapes):
    """The captured decoder step with fp16 matrices leaves the self- (and, over >= 1024 text positions, cross-) attention's eight key slices
    unmerged and the gate | up slabs unmultiplied: the output / down projections do both while they stage their rows
    (gemv_stream_kernel<.., PRO_ATTN8 / PRO_SILU, ..>, three launches fewer per layer).  Same arithmetic as the separate launches: the ids of a
    greedy generation equal those of tune("attn_fold") = 0 exactly, for one utterance and for two in lock-step."""
    if shapes == "tiny_f16":
        model = synth.build_dia(synth.dia_tiny(weight_type=gguf.F16))
        texts, steps = ["[S1] hi there", "[S2] ok"], 40
    else:
        model = synth.build_dia(synth.dia_1_6b(enc_layers=1, dec_layers=2, max_gen=64, weight_type=gguf.F16))
        texts, steps = ["[S1] The birch canoe slid on the smooth planks.", "[S2] Glue the sheet."], 40
    cfg = model.cfg
    args = dict(delay_pattern=[0, 8, 9, 10, 11, 12, 13, 14, 15], bos=cfg.bos, eos=cfg.eos, pad=cfg.pad, max_delay=cfg.max_delay)
    outs = {}
    for fold in (1, 0):
        eng = hip.DiaEngine(cfg, max_utterances=2)
        eng.tune("attn_fold", fold)
        eng.load(model)
        for u, t in enumerate(texts):
            eng.encode_slot(u, *orc.dia_tokenize(t, cfg.max_ctx))
        outs[fold] = eng.generate(1, steps, **args) + eng.generate(2, steps, **args)
        eng.close()
    assert len(outs[1]) == 3 and all(len(o) > 0 for o in outs[1])
    for a, b in zip(outs[1], outs[0]):
        assert a.shape == b.shape and np.array_equal(a, b)

"""GPU parity: the HIP Parler decoder (through the C ABI) against the oracle, same seeded inputs.

Tolerances (relative to max|oracle| of the compared tensor):
  F32 weights                      : 2e-4   (fp32 MFMA / FMA chains vs the oracle's double accumulation)
  F16 weights (act rounded to f16) : 2e-3   (both sides round activations to fp16 before each matmul, so
                                             a 1-ulp fp32 difference can flip an fp16 rounding: 1e-3 class)
  F16 KV cache (extension)         : 1e-2
Greedy token ids must be identical wherever the oracle's top-2 logit margin exceeds the tolerance.
"""
import os

import numpy as np
import pytest

import oracle as orc
from tts_cpp_amd import gguf, hip, synth

pytestmark = pytest.mark.gpu

TOL = {gguf.F32: 2e-4, gguf.F16: 2e-3, gguf.Q8_0: 2e-4, gguf.Q4_0: 2e-4, gguf.Q5_0: 2e-4}


def relerr(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - b).max() / (np.abs(b).max() + 1e-30))


_models = {}


def get_model(kind, wtype):
    key = (kind, wtype)
    if key not in _models:
        cfg = {"tiny": synth.tiny, "small": synth.small, "mini": synth.parler_mini}[kind](weight_type=wtype)
        _models[key] = synth.build(cfg)
    return _models[key]


def check_tokens(lg_gpu, lg_ref, tol):
    """argmax must agree unless the oracle's own margin is inside the tolerance band"""
    a, b = lg_gpu.argmax(-1), lg_ref.argmax(-1)
    srt = np.sort(lg_ref, axis=-1)
    margin = srt[..., -1] - srt[..., -2]
    band = tol * np.abs(lg_ref).max() * 2
    bad = (a != b) & (margin > band)
    assert not bad.any(), f"greedy ids differ outside the tolerance band: gpu {a[bad]} ref {b[bad]} margin {margin[bad]}"


@pytest.mark.parametrize("kind,wtype,flags", [
    ("tiny", gguf.F32, 0),
    ("tiny", gguf.F32, hip.FLAG_VALU_GEMM),
    ("tiny", gguf.F16, 0),
    ("tiny", gguf.F16, hip.FLAG_VALU_GEMM),
    ("small", gguf.F16, 0),
    ("small", gguf.F32, hip.FLAG_NO_GRAPH),
])
def test_prefill_and_steps_match_oracle(kind, wtype, flags):
    model = get_model(kind, wtype)
    cfg = model.cfg
    tol = TOL[wtype]
    eng = hip.HipEngine(cfg, max_seqs=1, flags=flags)
    eng.load(model)
    o = orc.ParlerOracle(model, act_mode=1, gelu_mode=1)
    rng = np.random.default_rng(5)
    prompt = rng.integers(3, cfg.prompt_vocab, 9).astype(np.uint32)

    # cross K/V computed at finalize (prep_cross_key_values)
    ck = eng.debug_read("cross:0:0", cfg.enc_len * cfg.hidden).reshape(cfg.enc_len, cfg.hidden)
    ref_ck = orc.mul_mat(model.by_name["decoder.layers.0.encoder_attn.k_proj.weight"].type,
                         model.by_name["decoder.layers.0.encoder_attn.k_proj.weight"].raw(), cfg.hidden, cfg.hidden,
                         model.by_name["decoder.text_encoding"].to_f32(), act_mode=1)
    assert relerr(ck, ref_ck) < tol

    eng.prefill(0, prompt)
    _, h_ref = o.decode(prompt, 0, audio=False, want_logits=False, want_hidden=True)
    h = eng.debug_read("hidden", len(prompt) * cfg.hidden).reshape(len(prompt), cfg.hidden)
    assert relerr(h, h_ref) < tol, "prompt hidden states"

    ids = np.full(cfg.n_out, cfg.bos, dtype=np.uint32)
    for step in range(1, 7):
        pos = len(prompt) + step - 1
        lg = eng.step(ids[None], [pos])[0]
        ref, _ = o.decode(ids, pos, audio=True)
        ref = ref[:, 0, :]
        assert np.isfinite(lg).all()
        assert relerr(lg, ref) < tol, f"step {step}"
        check_tokens(lg, ref, tol)
        toks = ref.argmax(-1).astype(np.uint32)  # teacher forcing with the oracle's ids
        ids = np.array([toks[i] if step > i else cfg.bos for i in range(cfg.n_out)], dtype=np.uint32)

    n_pos = len(prompt) + 6
    for layer in (0, cfg.layers - 1):
        k_ref, v_ref = o.get_kv(layer, n_pos)
        k = eng.debug_read(f"k:{layer}:0", n_pos * cfg.hidden).reshape(n_pos, cfg.hidden)
        v = eng.debug_read(f"v:{layer}:0", n_pos * cfg.hidden).reshape(n_pos, cfg.hidden)
        assert relerr(k, k_ref) < tol and relerr(v, v_ref) < tol, f"cache layer {layer}"
    eng.close()


def test_lockstep_sequences_are_independent():
    """3 utterances with different prompt lengths decoded in one batch == 3 single-utterance oracles."""
    model = get_model("tiny", gguf.F16)
    cfg = model.cfg
    eng = hip.HipEngine(cfg, max_seqs=3)
    eng.load(model)
    rng = np.random.default_rng(8)
    prompts = [rng.integers(3, cfg.prompt_vocab, n).astype(np.uint32) for n in (4, 11, 7)]
    oracles = [orc.ParlerOracle(model, act_mode=1, gelu_mode=1) for _ in prompts]
    for s, (p, o) in enumerate(zip(prompts, oracles)):
        eng.prefill(s, p)
        o.decode(p, 0, audio=False, want_logits=False)
    ids = np.full((3, cfg.n_out), cfg.bos, dtype=np.uint32)
    for step in range(1, 5):
        pos = [len(p) + step - 1 for p in prompts]
        lg = eng.step(ids, pos)
        for s, o in enumerate(oracles):
            ref, _ = o.decode(ids[s], pos[s], audio=True)
            assert relerr(lg[s], ref[:, 0, :]) < TOL[gguf.F16], (step, s)
            toks = ref[:, 0, :].argmax(-1)
            ids[s] = [toks[i] if step > i else cfg.bos for i in range(cfg.n_out)]
    # a permuted slot mapping gives the same logits for the same sequence
    lg_a = eng.step(ids[[2, 0]], [len(prompts[2]) + 4, len(prompts[0]) + 4], seqs=[2, 0])
    lg_b = eng.step(ids[[0]], [len(prompts[0]) + 4], seqs=[0])
    assert relerr(lg_a[1], lg_b[0]) < 1e-5
    eng.close()


@pytest.mark.parametrize("wtype", [gguf.F16, gguf.F32])
def test_many_lockstep_sequences_split_layernorm_path(wtype):
    """more than 8 rows per forward: LayerNorm runs as its own kernel and the GEMMs read the normalised
    rows from memory (other template instantiations than the small-batch path)"""
    model = get_model("tiny", wtype)
    cfg = model.cfg
    n = 12
    eng = hip.HipEngine(cfg, max_seqs=n)
    eng.load(model)
    rng = np.random.default_rng(21)
    prompts = [rng.integers(3, cfg.prompt_vocab, 3 + (i % 5)).astype(np.uint32) for i in range(n)]
    oracles = [orc.ParlerOracle(model, act_mode=1, gelu_mode=1) for _ in prompts]
    for s, (p, o) in enumerate(zip(prompts, oracles)):
        eng.prefill(s, p)
        o.decode(p, 0, audio=False, want_logits=False)
    ids = np.full((n, cfg.n_out), cfg.bos, dtype=np.uint32)
    for step in range(1, 4):
        pos = [len(p) + step - 1 for p in prompts]
        lg = eng.step(ids, pos)
        for s, o in enumerate(oracles):
            ref, _ = o.decode(ids[s], pos[s], audio=True)
            assert relerr(lg[s], ref[:, 0, :]) < TOL[wtype], (step, s)
            toks = ref[:, 0, :].argmax(-1)
            ids[s] = [toks[i] if step > i else cfg.bos for i in range(cfg.n_out)]
    eng.close()


def test_large_lockstep_batch_and_kv_position_cap():
    """80 utterances in one forward: the GEMM walks the rows in groups of 64 with the weights held in
    registers; device-resident generation == per-sequence oracles; the cache holds kv_positions rows."""
    model = get_model("tiny", gguf.F16)
    cfg = model.cfg
    n, n_steps = 80, 6
    eng = hip.HipEngine(cfg, max_seqs=n, kv_positions=40)
    eng.load(model)
    rng = np.random.default_rng(77)
    prompts = [rng.integers(3, cfg.prompt_vocab, 2 + (i % 7)).astype(np.uint32) for i in range(n)]
    for s, p in enumerate(prompts):
        eng.prefill(s, p)
    toks, done = eng.generate_greedy([len(p) for p in prompts], n_steps)
    for s in (0, 15, 16, 63, 64, 79):
        o = orc.ParlerOracle(model, act_mode=1, gelu_mode=1)
        ref_toks, ref_logits = o.generate_greedy(prompts[s], n_steps)
        mism = np.argwhere(toks[:, s, :] != ref_toks)
        if len(mism):
            st, hd = mism[0]
            srt = np.sort(ref_logits[st, hd])
            assert srt[-1] - srt[-2] < TOL[gguf.F16] * 2 * np.abs(ref_logits[st]).max(), (s, st, hd)
    with pytest.raises(hip.HipError):
        eng.step(np.zeros((1, cfg.n_out)), [40])  # beyond the cached positions
    eng.close()


def test_batched_prefill_equals_sequential_prefill():
    model = get_model("tiny", gguf.F16)
    cfg = model.cfg
    rng = np.random.default_rng(31)
    prompts = [rng.integers(3, cfg.prompt_vocab, 1 + (i * 5) % 13).astype(np.uint32) for i in range(9)]
    outs = []
    for batched in (True, False):
        eng = hip.HipEngine(cfg, max_seqs=len(prompts))
        eng.load(model)
        if batched:
            eng.prefill_batch(prompts)
        else:
            for s, p in enumerate(prompts):
                eng.prefill(s, p)
        ids = np.full((len(prompts), cfg.n_out), cfg.bos, dtype=np.uint32)
        outs.append(eng.step(ids, [len(p) for p in prompts]))
        eng.close()
    # the batched prefill carries enough rows for the LDS-tiled GEMM (fc2: K = 128), the sequential one stays on the 16-feature
    # workgroups, and forwards of up to 4 rows slice the keys of a (row, head) over 8 workgroups whose partial softmax results are
    # merged by the last one to finish (attn_kernel): same products, different fp32 summation order (measured 1.7e-4)
    assert relerr(outs[0], outs[1]) < 4e-4


def test_graph_replay_equals_eager_and_greedy_equals_argmax():
    model = get_model("tiny", gguf.F16)
    cfg = model.cfg
    prompt = np.array([7, 8, 9, 10, 1], dtype=np.uint32)
    outs = []
    for flags in (0, hip.FLAG_NO_GRAPH):
        eng = hip.HipEngine(cfg, max_seqs=2, flags=flags)
        eng.load(model)
        eng.prefill(0, prompt)
        eng.prefill(1, prompt[:3])
        ids = np.full((2, cfg.n_out), cfg.bos, dtype=np.uint32)
        seq = []
        for step in range(4):
            lg = eng.step(ids, [5 + step, 3 + step])
            tk = eng.step_greedy(ids, [5 + step, 3 + step])  # same positions: rewrites identical K/V
            assert (tk == lg.argmax(-1)).all()
            seq.append(lg)
            ids = lg.argmax(-1).astype(np.uint32)
        outs.append(np.stack(seq))
        eng.close()
    assert np.array_equal(outs[0], outs[1]), "hipGraph replay must be bit-identical to eager launches"


@pytest.mark.parametrize("nsplit", [1, 4, 16])
def test_long_context_split_attention(nsplit):
    """fill most of the context, with the split-T attention path forced"""
    os.environ["TTS_HIP_ATTN_NSPLIT"] = str(nsplit)
    try:
        model = get_model("tiny", gguf.F32)
        cfg = model.cfg
        eng = hip.HipEngine(cfg, max_seqs=1)
        eng.load(model)
    finally:
        del os.environ["TTS_HIP_ATTN_NSPLIT"]
    o = orc.ParlerOracle(model, act_mode=1, gelu_mode=1)
    rng = np.random.default_rng(nsplit)
    prompt = rng.integers(3, cfg.prompt_vocab, 100).astype(np.uint32)  # > 64 rows: chunked prefill
    eng.prefill(0, prompt)
    o.decode(prompt, 0, audio=False, want_logits=False)
    ids = rng.integers(0, cfg.audio_vocab, cfg.n_out).astype(np.uint32)
    for pos in range(100, 110):
        lg = eng.step(ids[None], [pos])[0]
        ref, _ = o.decode(ids, pos, audio=True)
        assert relerr(lg, ref[:, 0, :]) < TOL[gguf.F32], pos
        ids = ref[:, 0, :].argmax(-1).astype(np.uint32)
    eng.close()


def test_device_resident_greedy_generation_matches_reference_loop():
    """tts_hip_parler_generate_greedy (delay pattern + EOS flags on the device, one host sync) against
    the reference loop restated in the oracle (model.cpp:762-792)."""
    model = get_model("tiny", gguf.F32)
    cfg = model.cfg
    n_steps = 24
    prompts = [np.array([5, 6, 7, 1], dtype=np.uint32), np.array([9, 3, 44, 12, 13, 1], dtype=np.uint32)]
    eng = hip.HipEngine(cfg, max_seqs=2)
    eng.load(model)
    for s, p in enumerate(prompts):
        eng.prefill(s, p)
    toks, done = eng.generate_greedy([len(p) for p in prompts], n_steps)
    assert toks.shape == (n_steps, 2, cfg.n_out)
    for s, p in enumerate(prompts):
        o = orc.ParlerOracle(model, act_mode=1, gelu_mode=1)
        ref_toks, ref_logits = o.generate_greedy(p, n_steps)
        mism = np.argwhere(toks[:, s, :] != ref_toks)
        if len(mism):
            st, hd = mism[0]
            srt = np.sort(ref_logits[st, hd])
            margin = srt[-1] - srt[-2]
            assert margin < TOL[gguf.F32] * 2 * np.abs(ref_logits[st]).max(), \
                f"seq {s}: first divergence at step {st} head {hd} with oracle margin {margin}"
        else:
            assert np.array_equal(toks[:, s, :], ref_toks)
    # running the host-driven loop over the same prompts gives the same ids as the device loop
    eng2 = hip.HipEngine(cfg, max_seqs=2)
    eng2.load(model)
    for s, p in enumerate(prompts):
        eng2.prefill(s, p)
    ids = np.full((2, cfg.n_out), cfg.bos, dtype=np.uint32)
    eos_seen = np.zeros((2, cfg.n_out), dtype=bool)
    for step in range(1, n_steps + 1):
        tk = eng2.step_greedy(ids, [len(p) + step - 1 for p in prompts])
        assert np.array_equal(tk, toks[step - 1]), step
        eos_seen |= tk == cfg.eos
        for s in range(2):
            ids[s] = [(cfg.eos if eos_seen[s, i] else tk[s, i]) if step > i else cfg.bos for i in range(cfg.n_out)]
    eng.close()
    eng2.close()


def replace_tensors(model, repl):
    import copy
    m2 = copy.copy(model)
    m2.tensors = list(model.tensors)
    m2.by_name = dict(model.by_name)
    names = [x.name for x in m2.tensors]
    for name, arr in repl.items():
        t = gguf.Tensor.from_array(name, arr, gguf.F32)
        m2.by_name[name] = t
        m2.tensors[names.index(name)] = t
    return m2


def test_eos_bookkeeping_on_device():
    """Force EOS: with a zero final-LayerNorm weight the final hidden state is the LN bias, so a head row
    aligned with it wins every arg-max.  All heads emit EOS at audio step 1 -> check_stopping
    (model.cpp:715-732) turns true before step 2 (steps_done == 1) and every later input id is EOS."""
    model = get_model("tiny", gguf.F32)
    cfg = model.cfg
    lnb = model.by_name["decoder.layer_norm.bias"].to_f32()
    repl = {"decoder.layer_norm.weight": np.zeros_like(lnb)}
    for i in range(cfg.n_out):
        name = f"decoder.lm_heads.{i}.weight.head"
        w = model.by_name[name].to_f32().copy()
        w[cfg.eos] = 10.0 * lnb / float((lnb * lnb).sum())
        repl[name] = w
    m2 = replace_tensors(model, repl)
    eng = hip.HipEngine(cfg, max_seqs=1)
    eng.load(m2)
    prompt = np.array([4, 5, 1], dtype=np.uint32)
    eng.prefill(0, prompt)
    toks, done = eng.generate_greedy([3], 6)
    assert (toks == cfg.eos).all()
    assert done[0] == 1
    o = orc.ParlerOracle(m2, act_mode=1, gelu_mode=1)
    ref_toks, _ = o.generate_greedy(prompt, 6)
    assert np.array_equal(toks[:, 0, :], ref_toks)
    eng.close()


def test_fp16_kv_cache_extension():
    model = get_model("tiny", gguf.F16)
    cfg = model.cfg
    eng = hip.HipEngine(cfg, max_seqs=1, kv_type=gguf.F16)
    eng.load(model)
    o = orc.ParlerOracle(model, act_mode=1, gelu_mode=1)
    prompt = np.array([11, 12, 13, 14, 15, 1], dtype=np.uint32)
    eng.prefill(0, prompt)
    o.decode(prompt, 0, audio=False, want_logits=False)
    ids = np.full(cfg.n_out, cfg.bos, dtype=np.uint32)
    for pos in range(6, 10):
        lg = eng.step(ids[None], [pos])[0]
        ref, _ = o.decode(ids, pos, audio=True)
        assert relerr(lg, ref[:, 0, :]) < 1e-2
        ids = ref[:, 0, :].argmax(-1).astype(np.uint32)
    eng.close()


# end-to-end bound of the integer path.  Q8_0 activation quantisation is discontinuous: where x*127/amax lies within
# an fp32 ulp of k+0.5, a last-bit difference in the summation order upstream (MFMA tree vs the oracle's serial sum;
# ggml itself differs between SIMD widths and thread counts in the same way) moves that activation by one whole step
# (amax/127).  Measured here: one such flip in ~30k quantised activations moves a K/V row by ~1e-2 of the row
# maximum, everything else agrees to 1e-7.  So: exact checks where inputs are bit-identical, a flip-sized bound after.
Q_FLIP_TOL = 3e-2
Q_EXACT_TOL = 2e-6


@pytest.mark.parametrize("wtype", [gguf.Q8_0, gguf.Q5_0, gguf.Q4_0])
@pytest.mark.parametrize("mode", ["integer", "dequant"])
def test_quantised_gguf_models(wtype, mode):
    """Q4_0/Q5_0/Q8_0 GGUF tensors.  integer: ggml's CPU semantics (activations quantised to Q8_0 blocks, exact
    integer block dots on the int8 MFMA, fp16 block scales) == oracle act_mode 1.  dequant: blocks decoded exactly
    to fp32 at upload with fp32 activations == oracle act_mode 0."""
    model = get_model("tiny", wtype)
    cfg = model.cfg
    integer = mode == "integer"
    eng = hip.HipEngine(cfg, max_seqs=2, flags=0 if integer else hip.FLAG_DEQUANT_Q)
    eng.load(model)
    act = 1 if integer else 0
    o = orc.ParlerOracle(model, act_mode=act, gelu_mode=1)
    H, E = cfg.hidden, cfg.enc_len

    # (1) the GEMM alone on bit-identical inputs: cross K/V = W_{k,v} x text_encoding for every layer
    enc = model.by_name["decoder.text_encoding"].to_f32()
    for layer in range(cfg.layers):
        for kv, nm in enumerate(("k_proj", "v_proj")):
            w = model.by_name[f"decoder.layers.{layer}.encoder_attn.{nm}.weight"]
            ref = orc.mul_mat(w.type, w.raw(), H, H, enc, act_mode=act)
            got = eng.debug_read(f"cross:{layer}:{kv}", E * H).reshape(E, H)
            assert relerr(got, ref) < (Q_EXACT_TOL if integer else TOL[wtype]), (layer, nm)

    prompt = np.array([21, 22, 23, 24, 25, 26, 27, 28, 29, 1], dtype=np.uint32)
    eng.prefill(0, prompt)
    eng.prefill(1, prompt[:3])
    o.decode(prompt, 0, audio=False, want_logits=False)

    # (2) layer-0 K/V of the prompt rows: embeddings -> LayerNorm -> quantise -> fused QKV GEMM.  Rows without a
    # flipped activation agree to rounding; allow one flipped row.
    k_ref, v_ref = o.get_kv(0, len(prompt))
    k = eng.debug_read("k:0:0", len(prompt) * H).reshape(len(prompt), H)
    v = eng.debug_read("v:0:0", len(prompt) * H).reshape(len(prompt), H)
    row_err = np.maximum(np.abs(k - k_ref).max(axis=1) / np.abs(k_ref).max(), np.abs(v - v_ref).max(axis=1) / np.abs(v_ref).max())
    if integer:
        assert (row_err < 1e-5).sum() >= len(prompt) - 1, row_err
        assert row_err.max() < Q_FLIP_TOL, row_err
    else:
        assert row_err.max() < TOL[wtype], row_err

    # (3) end to end
    tol = Q_FLIP_TOL if integer else TOL[wtype]
    ids = np.full((2, cfg.n_out), cfg.bos, dtype=np.uint32)
    for step in range(3):
        lg = eng.step(ids, [10 + step, 3 + step])
        ref, _ = o.decode(ids[0], 10 + step, audio=True)
        assert relerr(lg[0], ref[:, 0, :]) < tol, (mode, step)
        check_tokens(lg[0], ref[:, 0, :], tol)
        ids[0] = ref[:, 0, :].argmax(-1)
        ids[1] = lg[1].argmax(-1)
    eng.close()


def test_integer_path_is_closer_to_ggml_semantics_than_dequantised_weights():
    """the two quantised modes differ by the activation quantisation error (~1e-2); the integer path tracks the
    act_mode-1 oracle, the dequant path tracks act_mode 0 — on the small model (H=512: 2-wave blocks, F=1024)."""
    model = get_model("small", gguf.Q5_0)
    cfg = model.cfg
    prompt = np.array([5, 6, 7, 1], dtype=np.uint32)
    res = {}
    for mode, flags in (("integer", 0), ("dequant", hip.FLAG_DEQUANT_Q)):
        eng = hip.HipEngine(cfg, max_seqs=1, flags=flags)
        eng.load(model)
        eng.prefill(0, prompt)
        res[mode] = eng.step(np.full((1, cfg.n_out), cfg.bos, dtype=np.uint32), [len(prompt)])[0]
        eng.close()
    refs = {}
    for act in (0, 1):
        o = orc.ParlerOracle(model, act_mode=act, gelu_mode=1)
        o.decode(prompt, 0, audio=False, want_logits=False)
        refs[act] = o.decode(np.full(cfg.n_out, cfg.bos, dtype=np.uint32), len(prompt), audio=True)[0][:, 0, :]
    assert relerr(res["dequant"], refs[0]) < TOL[gguf.Q5_0]
    assert relerr(res["integer"], refs[1]) < Q_FLIP_TOL
    assert relerr(res["integer"], refs[1]) < relerr(res["integer"], refs[0]) or relerr(res["integer"], refs[1]) < 1e-5


@pytest.mark.parametrize("rows", [1, 3, 20, 40])
def test_integer_path_full_width_shapes(rows):
    """Parler-Mini's matrix shapes (H=1024, F=4096; one layer): split-K slabs of the residual GEMMs (K=1024 -> 4x256,
    K=4096 -> 4x1024 with 4 waves), folded by the next LayerNorm; rows <= 16 quantise inside the GEMM workgroups,
    more rows through quant_rows_q8_kernel with 2 / 4 row blocks per wave."""
    key = ("wide1", gguf.Q5_0)
    if key not in _models:
        _models[key] = synth.build(synth.tiny(hidden=1024, heads=16, ffn=4096, layers=1, weight_type=gguf.Q5_0))
    model = _models[key]
    cfg = model.cfg
    eng = hip.HipEngine(cfg, max_seqs=rows)
    eng.load(model)
    rng = np.random.default_rng(rows)
    prompts = [rng.integers(3, cfg.prompt_vocab, 2 + (i % 3)).astype(np.uint32) for i in range(rows)]
    eng.prefill_batch(prompts)
    ids = np.full((rows, cfg.n_out), cfg.bos, dtype=np.uint32)
    lg = eng.step(ids, [len(p) for p in prompts])
    exact = 0
    for s in range(min(rows, 6)):
        o = orc.ParlerOracle(model, act_mode=1, gelu_mode=1)
        o.decode(prompts[s], 0, audio=False, want_logits=False)
        ref, _ = o.decode(ids[s], len(prompts[s]), audio=True)
        e = relerr(lg[s], ref[:, 0, :])
        assert e < Q_FLIP_TOL, (s, e)
        exact += e < 1e-5
    assert exact >= min(rows, 6) - 2   # most utterances see no flipped activation at all
    eng.close()


QTILE_ROWS = {65: (0, 31, 32, 63, 64), 257: (0, 63, 64, 127, 128, 255, 256), 320: (0, 31, 32, 191, 192, 256, 319), 1024: (0, 63, 64, 511, 512, 959, 960, 1023)}


@pytest.mark.parametrize("wtype", [gguf.Q5_0, gguf.Q4_0, gguf.Q8_0])
@pytest.mark.parametrize("rows", [65, 257, 320, 1024])
def test_integer_tile_path_many_rows(rows, wtype):
    """GGUF-quantised matrices with more rows than the 16-feature kernel carries: qgemm_tile_kernel (one
    v_mfma_i32_32x32x32_i8 per quantisation block, per-block (float) sumi * (d_w * d_a) like ggml_vec_dot_q*_q8_0) at Parler-Mini's matrix
    shapes, ragged row counts (rows % 64 != 0), split-K slabs of the residual GEMMs, the GELU output and the attended cross rows leaving
    their GEMMs as Q8_0 blocks.  (1) layer-0 K / V rows of the prompt forward = the fused QKV GEMM on inputs that are bit-identical on both
    sides (embedding -> LayerNorm -> Q8_0): rows without a flipped activation at 1e-5, all inside the flip bound; (2) a step end to end
    against per-row oracles at rows on both sides of every tile boundary."""
    key = ("wide1", wtype)
    if key not in _models:
        _models[key] = synth.build(synth.tiny(hidden=1024, heads=16, ffn=4096, layers=1, weight_type=wtype))
    model = _models[key]
    cfg = model.cfg
    H = cfg.hidden
    eng = hip.HipEngine(cfg, max_seqs=rows, kv_positions=16)
    eng.load(model)
    rng = np.random.default_rng(rows)
    prompts = [rng.integers(3, cfg.prompt_vocab, 2 + (i % 3)).astype(np.uint32) for i in range(rows)]
    eng.prefill_batch(prompts)   # sum(len) = 3 * rows rows in one forward: beyond 1024 the host splits it
    sample = QTILE_ROWS[rows]
    oracles = {}
    exact = 0
    for r in sample:
        o = orc.ParlerOracle(model, act_mode=1, gelu_mode=1)
        o.decode(prompts[r], 0, audio=False, want_logits=False)
        oracles[r] = o
        n = len(prompts[r])
        k_ref, v_ref = o.get_kv(0, n)
        k = eng.debug_read(f"k:0:{r}", n * H).reshape(n, H)
        v = eng.debug_read(f"v:0:{r}", n * H).reshape(n, H)
        e = max(np.abs(k - k_ref).max() / np.abs(k_ref).max(), np.abs(v - v_ref).max() / np.abs(v_ref).max())
        assert e < Q_FLIP_TOL, (rows, r, e)
        exact += e < 1e-5
    assert exact >= len(sample) - 1, exact
    ids = np.full((rows, cfg.n_out), cfg.bos, dtype=np.uint32)
    worst, close = 0.0, 0
    for step in range(2):
        lg = eng.step(ids, [len(p) + step for p in prompts])
        for r in sample:
            ref, _ = oracles[r].decode(ids[r], len(prompts[r]) + step, audio=True)
            e = relerr(lg[r], ref[:, 0, :])
            worst = max(worst, e)
            close += e < 1e-4
            assert e < Q_FLIP_TOL, (rows, r, step, e)
            ids[r] = ref[:, 0, :].argmax(-1)
        for r in range(rows):
            if r not in sample:
                ids[r] = lg[r].argmax(-1)
    assert close >= len(sample), (close, worst)   # most (row, step) pairs see no flipped activation
    print(f"rows={rows} {wtype}: worst relative logit error {worst:.2e}, {close} of {2 * len(sample)} below 1e-4")
    eng.close()


@pytest.mark.parametrize("shape", [0, 1, 2, 3, 4])
def test_integer_tile_path_every_tile_shape(shape):
    """every tile shape of qgemm_tile_kernel forced onto every GEMM of a Parler-Mini-width layer, ragged rows, 1 / 2 / 4 K slices on the residual
    GEMMs; the tiled path against the 16-feature kernel (tune qtile_min_rows = 0) on the same rows: the same block terms in another order."""
    key = ("wide1", gguf.Q5_0)
    if key not in _models:
        _models[key] = synth.build(synth.tiny(hidden=1024, heads=16, ffn=4096, layers=1, weight_type=gguf.Q5_0))
    model = _models[key]
    cfg = model.cfg
    rows = 200
    rng = np.random.default_rng(shape)
    prompts = [rng.integers(3, cfg.prompt_vocab, 2 + (i % 3)).astype(np.uint32) for i in range(rows)]
    ids = np.full((rows, cfg.n_out), cfg.bos, dtype=np.uint32)
    res = {}
    for mode in ("tile", "wg16"):
        eng = hip.HipEngine(cfg, max_seqs=rows, kv_positions=16)
        eng.tune("qtile_min_rows", 65 if mode == "tile" else 0)
        if mode == "tile":
            eng.tune("qtile_shape", shape)
            eng.tune("qtile_ks", (1, 2, 4, 2, 1)[shape])
        eng.load(model)
        eng.prefill_batch(prompts)
        res[mode] = eng.step(ids, [len(p) for p in prompts])
        eng.close()
    err = np.abs(res["tile"] - res["wg16"]).reshape(rows, -1).max(axis=1) / np.abs(res["wg16"]).max()
    # a row differs only where a last-bit difference (another summation order) flips one of its ~12 000 Q8_0 activation codes of the layer: measured 14 of 200 rows
    assert (err < 1e-5).sum() >= rows * 0.8, (shape, np.sort(err)[-10:])
    assert err.max() < Q_FLIP_TOL
    for r in (0, 63, 64, 199):
        o = orc.ParlerOracle(model, act_mode=1, gelu_mode=1)
        o.decode(prompts[r], 0, audio=False, want_logits=False)
        ref, _ = o.decode(ids[r], len(prompts[r]), audio=True)
        assert relerr(res["tile"][r], ref[:, 0, :]) < Q_FLIP_TOL, (shape, r)


def test_update_conditional_prompt():
    model = get_model("tiny", gguf.F32)
    cfg = model.cfg
    eng = hip.HipEngine(cfg, max_seqs=1)
    eng.load(model)
    o = orc.ParlerOracle(model, act_mode=1, gelu_mode=1)
    enc = (np.random.default_rng(3).standard_normal((11, cfg.hidden)) * 0.5).astype(np.float32)
    eng.set_text_encoding(enc)
    o.set_text_encoding(enc)
    prompt = np.array([5, 6, 1], dtype=np.uint32)
    eng.prefill(0, prompt)
    o.decode(prompt, 0, audio=False, want_logits=False)
    ids = np.full(cfg.n_out, cfg.bos, dtype=np.uint32)
    lg = eng.step(ids[None], [3])[0]
    ref, _ = o.decode(ids, 3, audio=True)
    assert relerr(lg, ref[:, 0, :]) < TOL[gguf.F32]
    eng.close()


def test_no_cross_attention_mode():
    model = get_model("tiny", gguf.F32)
    cfg = model.cfg
    eng = hip.HipEngine(cfg, max_seqs=1, use_cross_attn=False)
    eng.load(model)
    o = orc.ParlerOracle(model, act_mode=1, gelu_mode=1, use_cross=False)
    prompt = np.array([5, 6, 7, 1], dtype=np.uint32)
    eng.prefill(0, prompt)
    o.decode(prompt, 0, audio=False, want_logits=False)
    ids = np.full(cfg.n_out, cfg.bos, dtype=np.uint32)
    lg = eng.step(ids[None], [4])[0]
    ref, _ = o.decode(ids, 4, audio=True)
    assert relerr(lg, ref[:, 0, :]) < TOL[gguf.F32]
    eng.close()


def test_argument_errors_are_reported():
    model = get_model("tiny", gguf.F32)
    cfg = model.cfg
    eng = hip.HipEngine(cfg, max_seqs=1)
    with pytest.raises(hip.HipError):
        eng.step(np.zeros((1, cfg.n_out)), [0])  # not finalized
    eng.load(model)
    with pytest.raises(hip.HipError):
        eng.step(np.full((1, cfg.n_out), cfg.out_vocab + 5), [0])  # id outside the embedding table
    with pytest.raises(hip.HipError):
        eng.step(np.zeros((1, cfg.n_out)), [cfg.ctx])  # position outside the context
    with pytest.raises(hip.HipError):
        eng.prefill(3, [1, 2])  # sequence slot out of range
    with pytest.raises(hip.HipError):
        eng.step(np.zeros((2, cfg.n_out)), [0, 0])  # more rows than max_seqs
    eng.close()


def test_parler_mini_full_size_step():
    """BASELINE config dims (H=1024, 24 layers, 9x1088 heads, fp16 weights): prefill + 2 steps."""
    model = get_model("mini", gguf.F16)
    cfg = model.cfg
    eng = hip.HipEngine(cfg, max_seqs=2)
    eng.load(model)
    o = orc.ParlerOracle(model, act_mode=1, gelu_mode=1)
    prompt = np.random.default_rng(0).integers(3, cfg.prompt_vocab, 6).astype(np.uint32)
    eng.prefill(0, prompt)
    eng.prefill(1, prompt[:4])
    o.decode(prompt, 0, audio=False, want_logits=False)
    ids = np.full((2, cfg.n_out), cfg.bos, dtype=np.uint32)
    for step in range(2):
        lg = eng.step(ids, [6 + step, 4 + step])
        ref, _ = o.decode(ids[0], 6 + step, audio=True)
        assert relerr(lg[0], ref[:, 0, :]) < TOL[gguf.F16]
        check_tokens(lg[0], ref[:, 0, :], TOL[gguf.F16])
        ids[0] = ref[:, 0, :].argmax(-1)
        ids[1] = lg[1].argmax(-1)
    eng.close()


def test_parler_mini_batch1_greedy_128_steps_against_oracle():
    """The batch-1 chain at the BASELINE dims (fc2 as four K-slice slabs folded by the next LayerNorm prologue / residual epilogue, the
    self-attention's key-split partials folded by out_proj's prologue) over a whole run of 128 steps: the device-resident greedy loop
    (hipGraph replays) against the reference loop restated in the oracle.  The streams must be identical, or the first divergence must sit
    at an oracle top-2 margin inside the fp16 tolerance band (reported) — round 3 saw two of its own chains part at step 109 and only
    ever checked 2 steps against the oracle.  After a divergence the two runs see different histories, so the check ends there; the logits
    of the 128 steps are also held to the 2e-3 bar step by step with the oracle's own ids fed to both (no history drift)."""
    model = get_model("mini", gguf.F16)
    cfg = model.cfg
    n_steps = 128
    prompt = np.random.default_rng(109).integers(3, cfg.prompt_vocab, 6).astype(np.uint32)
    eng = hip.HipEngine(cfg, max_seqs=1)
    eng.load(model)
    eng.prefill(0, prompt)
    toks, _ = eng.generate_greedy([len(prompt)], n_steps)
    o = orc.ParlerOracle(model, act_mode=1, gelu_mode=1)
    ref_toks, ref_logits = o.generate_greedy(prompt, n_steps)
    mism = np.argwhere(toks[:, 0, :] != ref_toks)
    if len(mism):
        st, hd = mism[0]
        srt = np.sort(ref_logits[st, hd])
        margin, band = srt[-1] - srt[-2], TOL[gguf.F16] * 2 * np.abs(ref_logits[st]).max()
        print(f"first divergence at step {st} head {hd}: oracle top-2 margin {margin:.3e}, band {band:.3e}")
        assert margin < band, f"greedy streams part at step {st} head {hd} with an oracle margin of {margin} (band {band})"
        assert np.array_equal(toks[:st, 0, :], ref_toks[:st])
    else:
        print(f"{n_steps} steps: device greedy stream identical to the oracle's")
    # same prompt, every step fed the ORACLE's ids: logits within the bar at every step
    eng.reset()
    eng.prefill(0, prompt)
    ids = np.full((1, cfg.n_out), cfg.bos, dtype=np.uint32)
    eos_seen = np.zeros(cfg.n_out, dtype=bool)
    worst = 0.0
    for step in range(1, n_steps + 1):
        lg = eng.step(ids, [len(prompt) + step - 1])[0]
        ref = ref_logits[step - 1]
        err = relerr(lg, ref)
        worst = max(worst, err)
        assert err < TOL[gguf.F16], f"step {step}: logits {err:.2e}"
        check_tokens(lg, ref, TOL[gguf.F16])
        tk = ref_toks[step - 1]
        eos_seen |= tk == cfg.eos
        ids[0] = [(cfg.eos if eos_seen[i] else tk[i]) if step > i else cfg.bos for i in range(cfg.n_out)]
    print(f"worst logits error over {n_steps} teacher-forced steps: {worst:.2e}")
    eng.close()


SAMPLE_ROWS = {128: (0, 15, 16, 63, 64, 127), 384: (0, 31, 32, 127, 128, 255, 256, 383), 1024: (0, 127, 128, 511, 512, 640, 895, 896, 1023),
               1152: (0, 127, 128, 511, 512, 640, 1023, 1024, 1151)}


@pytest.mark.parametrize("rows", [128, 384, 1024] + ([1152] if int(os.environ.get("TTS_HIP_MAX_ROWS", "0")) >= 1152 else []))
def test_parler_mini_full_size_many_rows(rows):
    """The measured configuration (bench.py: H=1024, 24 layers, fp16 weights, 128..1024 lock-step rows): the LDS-tiled GEMM
    path (gemm_tile_kernels.h: every tile shape the cost model picks at these sizes, split-K slabs folded by the next
    LayerNorm, KV append from the tile epilogue), 4 steps, a sample of rows at the tile / wave / row-group boundaries against
    per-row oracles.  Each row feeds its own arg-max back, like the device loop does."""
    model = get_model("mini", gguf.F16)
    cfg = model.cfg
    eng = hip.HipEngine(cfg, max_seqs=rows, kv_positions=32)
    eng.load(model)
    rng = np.random.default_rng(rows)
    prompts = [rng.integers(3, cfg.prompt_vocab, 3 + (i % 4)).astype(np.uint32) for i in range(rows)]
    eng.prefill_batch(prompts)
    sample = SAMPLE_ROWS[rows]
    oracles = {}
    for r in sample:
        o = orc.ParlerOracle(model, act_mode=1, gelu_mode=1)
        o.decode(prompts[r], 0, audio=False, want_logits=False)
        oracles[r] = o
    ids = np.full((rows, cfg.n_out), cfg.bos, dtype=np.uint32)
    worst = 0.0
    for step in range(4):
        lg = eng.step(ids, [len(p) + step for p in prompts])
        for r in sample:
            ref, _ = oracles[r].decode(ids[r], len(prompts[r]) + step, audio=True)
            e = relerr(lg[r], ref[:, 0, :])
            worst = max(worst, e)
            assert e < TOL[gguf.F16], (rows, r, step, e)
            check_tokens(lg[r], ref[:, 0, :], TOL[gguf.F16])
            ids[r] = ref[:, 0, :].argmax(-1)   # the oracle's token keeps both sides on one trajectory
        for r in range(rows):
            if r not in sample:
                ids[r] = lg[r].argmax(-1)
    print(f"rows={rows}: worst relative logit error over {len(sample)} rows x 4 steps = {worst:.2e}")
    eng.close()


@pytest.mark.parametrize("shape", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("rows", [40, 100, 200])
def test_tiled_gemm_every_tile_shape(rows, shape, monkeypatch):
    """every tile shape of gemm_tile_kernel forced onto every GEMM of a Parler-Mini-width layer (H=1024, F=4096, fp16), ragged
    row counts (rows % tile != 0), split-K 1/2/4/8 on the residual GEMMs."""
    monkeypatch.setenv("TTS_HIP_TILE_FORCE", str(shape))
    monkeypatch.setenv("TTS_HIP_TILE_KS", str((1, 2, 4, 8, 1, 2)[shape]))
    key = ("wide1", gguf.F16)
    if key not in _models:
        _models[key] = synth.build(synth.tiny(hidden=1024, heads=16, ffn=4096, layers=1, weight_type=gguf.F16))
    model = _models[key]
    cfg = model.cfg
    eng = hip.HipEngine(cfg, max_seqs=rows, kv_positions=16)
    eng.load(model)
    rng = np.random.default_rng(rows * 8 + shape)
    prompts = [rng.integers(3, cfg.prompt_vocab, 2 + (i % 3)).astype(np.uint32) for i in range(rows)]
    eng.prefill_batch(prompts)
    ids = np.full((rows, cfg.n_out), cfg.bos, dtype=np.uint32)
    sample = sorted({0, 15, 16, 31, 32, rows // 2, rows - 1})
    oracles = {}
    for r in sample:
        o = orc.ParlerOracle(model, act_mode=1, gelu_mode=1)
        o.decode(prompts[r], 0, audio=False, want_logits=False)
        oracles[r] = o
    for step in range(2):
        lg = eng.step(ids, [len(p) + step for p in prompts])
        for r in sample:
            ref, _ = oracles[r].decode(ids[r], len(prompts[r]) + step, audio=True)
            assert relerr(lg[r], ref[:, 0, :]) < TOL[gguf.F16], (rows, shape, r, step)
        ids = lg.argmax(-1).astype(np.uint32)
    eng.close()


@pytest.mark.parametrize("sampled", [False, True])
def test_row_compaction_keeps_every_utterance_token_for_token(sampled, monkeypatch):
    """A ragged lock-step batch: 200 utterances whose prompts leave 20 ... 76 steps before their position reaches max_generation
    (check_stopping, model.cpp:720-722).  generate_loop drops finished utterances from the forward every 32 steps (row compaction: live rows
    gathered to the front, the rest of the loop state indexed by utterance).  Every utterance must get exactly the tokens and the step count
    it gets when finished rows keep idling in the forward (TTS_HIP_GEN_COMPACT=0), greedy and with the device sampler (uniforms, repetition
    penalty state per utterance).  The tile shape of the GEMMs is pinned for the comparison: the cost model picks shapes and K splits by the
    row count, i.e. a forward of 64 rows sums in another order than one of 200 (last-bit differences that the fp16 rounding of the
    activations occasionally turns into another sampled token — the same reason a lock-step batch is compared with single-utterance runs
    through a tolerance band); with the shape fixed the two runs are bit-identical (profiles/dbg_compact.py shows both)."""
    monkeypatch.setenv("TTS_HIP_TILE_FORCE", "3")
    monkeypatch.setenv("TTS_HIP_TILE_KS", "1")
    monkeypatch.setenv("TTS_HIP_ATTN_NSPLIT", "1")
    monkeypatch.setenv("TTS_HIP_ATTN_ROWS", "0")   # a >= 1024-row forward (the reference run's prefill) would take the row-major attention kernel: other summation order
    cfg = synth.small(weight_type=gguf.F16, ctx=80, max_gen=80)
    model = synth.build(cfg)
    rng = np.random.default_rng(11)
    n, cap = 200, 80
    lens = rng.integers(4, 61, n)
    prompts = [rng.integers(3, cfg.prompt_vocab, int(l)).astype(np.uint32) for l in lens]
    n_steps = int(cap - lens.min())
    uni = rng.random((n_steps, n, cfg.n_out), dtype=np.float32)
    res = []
    for compact in ("1", "0"):
        monkeypatch.setenv("TTS_HIP_GEN_COMPACT", compact)
        eng = hip.HipEngine(cfg, max_seqs=n, kv_positions=cap, flags=hip.FLAG_NO_DAC)
        eng.load(model)
        eng.prefill_batch(prompts)
        if sampled:
            toks, done = eng.generate_sampled(lens, n_steps, uni, top_k=20, temperature=0.9, repetition_penalty=1.1)
        else:
            toks, done = eng.generate_greedy(lens, n_steps)
        res.append((toks, done))
        eng.close()
    (ta, da), (tb, db) = res
    assert np.array_equal(da, db)
    assert sorted(set(da.tolist()))[0] < 32 < da.max(), "some utterances must finish before and some after the first compaction point"
    for u in range(n):
        k = int(da[u]) if da[u] else n_steps
        assert k == cap - lens[u] or da[u] == 0 or k < cap - lens[u]
        assert np.array_equal(ta[:k, u], tb[:k, u]), f"utterance {u}"


@pytest.mark.parametrize("sampled", [False, True])
def test_utterance_admitted_mid_flight_gets_the_tokens_of_its_own_run(sampled, monkeypatch):
    """Continuous batching (tts_hip_parler_stream_*): 150 utterances through 70 rows.  The first 70 open the stream; whenever utterances
    finish (their position reaches max_generation after 20 ... 76 steps, check_stopping model.cpp:720-722) the rows are refilled from the
    waiting list at the next 32-step look-in point — newcomers are prefilled as a side batch while the others hold their place.  Every
    utterance must get exactly the tokens and the step count it gets from ONE lock-step generation of all 150 (greedy, and with the device
    sampler: its own uniforms and repetition-penalty state), whatever it shared the forward with and whenever it entered.  The GEMM tile
    shape is pinned as in the compaction test (a forward of 64 rows otherwise sums in another order than one of 150)."""
    monkeypatch.setenv("TTS_HIP_TILE_FORCE", "3")
    monkeypatch.setenv("TTS_HIP_TILE_KS", "1")
    monkeypatch.setenv("TTS_HIP_ATTN_NSPLIT", "1")
    monkeypatch.setenv("TTS_HIP_ATTN_ROWS", "0")   # a >= 1024-row forward (the reference run's prefill) would take the row-major attention kernel: other summation order
    cfg = synth.small(weight_type=gguf.F16, ctx=80, max_gen=80)
    model = synth.build(cfg)
    rng = np.random.default_rng(23)
    n, cap, slots = 150, 80, 70
    lens = rng.integers(4, 61, n)
    prompts = [rng.integers(3, cfg.prompt_vocab, int(l)).astype(np.uint32) for l in lens]
    n_steps = int(cap - lens.min())
    uni = rng.random((n_steps, n, cfg.n_out), dtype=np.float32)
    samp = (20, 1.0, 0.9, 1.1)   # top_k, top_p, temperature, repetition penalty
    ref = hip.HipEngine(cfg, max_seqs=n, kv_positions=cap, flags=hip.FLAG_NO_DAC)
    ref.load(model)
    ref.prefill_batch(prompts)
    if sampled:
        rt, rd = ref.generate_sampled(lens, n_steps, uni, top_k=samp[0], temperature=samp[2], repetition_penalty=samp[3])
    else:
        rt, rd = ref.generate_greedy(lens, n_steps)
    ref.close()

    eng = hip.HipEngine(cfg, max_seqs=slots + 1, kv_positions=cap, flags=hip.FLAG_NO_DAC)
    eng.load(model)
    max_steps = cap - 1
    eng.stream_begin(slots, max_steps, sampling=samp if sampled else None)

    def draws(us):
        u = np.zeros((len(us), max_steps, cfg.n_out), dtype=np.float32)
        for i, k in enumerate(us):
            u[i, :n_steps] = uni[:, k]
        return u

    waiting = list(range(n))
    slot_utt, free = {}, list(range(slots))
    got, joined_late, rounds = {}, 0, 0
    while waiting or slot_utt:
        take = waiting[:len(free)]
        if take:
            waiting = waiting[len(take):]
            sl = [free.pop(0) for _ in take]
            joined_late += len(take) if slot_utt else 0
            eng.stream_admit(sl, [prompts[k] for k in take], draws(take) if sampled else None)
            slot_utt.update(dict(zip(sl, take)))
        for slot, steps in eng.stream_run(32):
            k = slot_utt.pop(slot)
            got[k] = (steps, eng.stream_collect(slot, steps))
            free.append(slot)
        rounds += 1
        assert rounds < 100
    eng.stream_end()
    eng.close()
    assert len(got) == n and joined_late >= n - slots, "utterances must have entered while others were generating"
    for k in range(n):
        want = int(rd[k]) if rd[k] else n_steps
        steps, toks = got[k]
        assert steps == want == cap - lens[k], (k, steps, want)
        assert np.array_equal(toks, rt[:want, k]), f"utterance {k} (admitted with {lens[k]} prompt ids)"


def test_generation_graphs_follow_reallocated_outputs_and_sampling_parameters():
    """The captured generation step bakes the tokens_out pointer and the sampling parameters in; both may change between calls on one
    context (a longer request reallocates tokens_out, a request may sample differently).  Round 3 keyed the graphs in units of 8192 rows
    and dropped them in units of 1000: stale graphs wrote through a freed pointer (memory access fault in the 3-runner bench) and sampled
    with the first call's top_k (tests/test_gpu_runner.py::test_host_sampling_loop_modes)."""
    model = get_model("small", gguf.F16)
    cfg = model.cfg
    prompt = np.random.default_rng(21).integers(3, cfg.prompt_vocab, 12).astype(np.uint32)

    def fresh(n_steps, sampled=None):
        eng = hip.HipEngine(cfg, max_seqs=2)
        eng.load(model)
        eng.prefill_batch([prompt, prompt[:7]])
        out = run(eng, n_steps, sampled)
        eng.close()
        return out

    def run(eng, n_steps, sampled):
        start = [len(prompt), 7]
        if sampled is None:
            return eng.generate_greedy(start, n_steps)[0]
        u = np.random.default_rng(5).random((n_steps, 2, cfg.n_out), dtype=np.float32)
        return eng.generate_sampled(start, n_steps, u, top_k=sampled, temperature=1.0)[0]

    eng = hip.HipEngine(cfg, max_seqs=2)
    eng.load(model)
    for n_steps, sampled in [(6, None), (40, None), (40, 1), (40, 30), (64, 30), (64, None)]:   # growing outputs, changing sampler
        eng.reset(); eng.prefill_batch([prompt, prompt[:7]])
        got = run(eng, n_steps, sampled)
        assert np.array_equal(got, fresh(n_steps, sampled)), (n_steps, sampled)
    eng.close()
    assert not np.array_equal(fresh(40, 30), fresh(40, None))


# ---- the kernel the roofline is quoted on, at the history it is measured at ----------------------------------------------------
# bench.py runs attn_rows_kernel<8> at 1024 rows over T = 17 ... 281 cached positions (one slice per row) and `long_utterances` at 320 rows
# over T <= 1040 (four key slices + attn_combine_kernel).  The many-rows test above only reaches T <= 10.  (parler/model.cpp:543-572)
HISTORY_CASES = {
    # rows: [(row, prompt length)]: the compared steps see T = length + 1 and length + 2 keys
    1024: [(0, 6), (63, 7), (64, 15), (511, 16), (512, 136), (640, 280), (895, 99), (1022, 136), (1023, 280)],   # T = 7, 8, 9, 16, 17, 18, 100, 137, 281, 282
    320: [(0, 4), (1, 63), (63, 527), (64, 1039), (127, 1039), (128, 527), (255, 62), (319, 3)],                   # T = 4, 5, 63, 64, 65, 528, 529, 1040, 1041
    256: [(0, 4), (15, 31), (16, 32), (100, 263), (128, 529), (254, 2), (255, 264)],                               # slices of 1, 8, 9 keys; 66 + 66 + 66 + 67
}


@pytest.mark.parametrize("kv", [gguf.F32, gguf.F16])
@pytest.mark.parametrize("rows", [1024, 320, 256])
def test_row_major_self_attention_at_measured_history(rows, kv):
    """attn_rows_kernel (one workgroup per row, whole K / V rows per wave-instruction set; parler_kernels.h) at the cached lengths of the
    headline (1024 rows, one slice) and of long_utterances (320 rows, four slices + combine), fp32 and fp16 caches: Parler-Mini widths,
    two layers, a handful of rows with long prompts among short ones, logits of those rows against per-row oracles over two steps.  The
    prompts' prefill runs through the same kernel (every row at its own position), so the cached rows the steps read were produced by it too."""
    key = ("mini2", gguf.F16)
    if key not in _models:
        _models[key] = synth.build(synth.parler_mini(layers=2, weight_type=gguf.F16))
    model = _models[key]
    cfg = model.cfg
    tol = TOL[gguf.F16] if kv == gguf.F32 else 1e-2
    cases = dict(HISTORY_CASES[rows])
    cap = max(cases.values()) + 8
    eng = hip.HipEngine(cfg, max_seqs=rows, kv_type=kv, kv_positions=cap)
    eng.load(model)
    rng = np.random.default_rng(rows + kv)
    prompts = [rng.integers(3, cfg.prompt_vocab, cases.get(i, 3 + (i % 4))).astype(np.uint32) for i in range(rows)]
    eng.prefill_batch(prompts)
    oracles = {}
    for r in cases:
        o = orc.ParlerOracle(model, act_mode=1, gelu_mode=1)
        o.decode(prompts[r], 0, audio=False, want_logits=False)
        oracles[r] = o
    ids = np.full((rows, cfg.n_out), cfg.bos, dtype=np.uint32)
    worst = {}
    for step in range(2):
        lg = eng.step(ids, [len(p) + step for p in prompts])
        assert np.isfinite(lg).all()
        nxt = lg.argmax(-1).astype(np.uint32)
        for r in cases:
            ref, _ = oracles[r].decode(ids[r], len(prompts[r]) + step, audio=True)
            nxt[r] = ref[:, 0, :].argmax(-1)   # the oracle's token keeps both sides on one trajectory
            e = relerr(lg[r], ref[:, 0, :])
            worst[len(prompts[r]) + step + 1] = max(e, worst.get(len(prompts[r]) + step + 1, 0.0))
            assert e < tol, (rows, r, step, len(prompts[r]), e)
            if kv == gguf.F32:
                check_tokens(lg[r], ref[:, 0, :], tol)
        ids = nxt
    print(f"rows={rows} kv={'f16' if kv == gguf.F16 else 'f32'}: relative logit error by cached length " +
          ", ".join(f"T={t}: {e:.1e}" for t, e in sorted(worst.items())))
    eng.close()


def test_row_major_self_attention_is_the_kernel_under_test(monkeypatch):
    """The dispatch takes attn_rows_kernel from 256 rows on (run_attn, shim_decoder.hip): with it switched off the same forward sums its
    scores in another order, so the two logit sets agree to rounding but not bit for bit — if they were identical the test above would be
    exercising the one-(head, row) kernel without saying so."""
    key = ("mini2", gguf.F16)
    if key not in _models:
        _models[key] = synth.build(synth.parler_mini(layers=2, weight_type=gguf.F16))
    model = _models[key]
    cfg = model.cfg
    rows = 256
    rng = np.random.default_rng(77)
    prompts = [rng.integers(3, cfg.prompt_vocab, 40 + (i % 9)).astype(np.uint32) for i in range(rows)]
    out = []
    for sw in (None, "0"):
        if sw is None:
            monkeypatch.delenv("TTS_HIP_ATTN_ROWS", raising=False)
        else:
            monkeypatch.setenv("TTS_HIP_ATTN_ROWS", sw)
        eng = hip.HipEngine(cfg, max_seqs=rows, kv_positions=64)
        eng.load(model)
        eng.prefill_batch(prompts)
        out.append(eng.step(np.full((rows, cfg.n_out), cfg.bos, dtype=np.uint32), [len(p) for p in prompts]))
        eng.close()
    assert relerr(out[0], out[1]) < 1e-3
    assert not np.array_equal(out[0], out[1]), "TTS_HIP_ATTN_ROWS made no difference: which attention kernel ran?"


@pytest.mark.parametrize("sampled", [False, True])
def test_row_compaction_with_the_row_major_attention(sampled, monkeypatch):
    """The compaction test above pins the self-attention to the one-(head, row) kernel to demand bit-equal streams.  Here the dispatch is
    left alone: 400 utterances start in the row-major kernel (four key slices + combine), are compacted every 32 steps and drop below 256
    live rows, where the other kernel takes over — a run that keeps finished rows idling never switches.  The two kernels sum in different
    orders, so streams may part at a near-tie; a bookkeeping error (a row reading another utterance's cache after the gather) would part
    nearly all of them at once.  Band: every utterance stops at the same step, and at least 90 % of the streams are identical."""
    monkeypatch.setenv("TTS_HIP_TILE_FORCE", "3")
    monkeypatch.setenv("TTS_HIP_TILE_KS", "1")
    monkeypatch.delenv("TTS_HIP_ATTN_ROWS", raising=False)
    cfg = synth.small(weight_type=gguf.F16, ctx=80, max_gen=80)
    model = synth.build(cfg)
    rng = np.random.default_rng(12)
    n, cap = 400, 80
    lens = rng.integers(4, 61, n)
    prompts = [rng.integers(3, cfg.prompt_vocab, int(l)).astype(np.uint32) for l in lens]
    n_steps = int(cap - lens.min())
    uni = rng.random((n_steps, n, cfg.n_out), dtype=np.float32)
    res = []
    for compact in ("1", "0"):
        monkeypatch.setenv("TTS_HIP_GEN_COMPACT", compact)
        eng = hip.HipEngine(cfg, max_seqs=n, kv_positions=cap, flags=hip.FLAG_NO_DAC)
        eng.load(model)
        eng.prefill_batch(prompts)
        if sampled:
            res.append(eng.generate_sampled(lens, n_steps, uni, top_k=20, temperature=0.9, repetition_penalty=1.1))
        else:
            res.append(eng.generate_greedy(lens, n_steps))
        eng.close()
    (ta, da), (tb, db) = res
    assert np.array_equal(da, db)
    same = 0
    for u in range(n):
        k = int(da[u]) if da[u] else n_steps
        same += bool(np.array_equal(ta[:k, u], tb[:k, u]))
    print(f"{same} of {n} streams identical with the attention kernel changing under compaction")
    assert same >= int(0.9 * n), same


def test_utterance_admitted_mid_flight_with_the_row_major_attention(monkeypatch):
    """Continuous batching with the self-attention dispatch left alone: 600 utterances through 300 rows (row-major kernel, four slices)
    against one lock-step generation of all 600 (same kernel, other row count and neighbours).  Same band as above."""
    monkeypatch.setenv("TTS_HIP_TILE_FORCE", "3")
    monkeypatch.setenv("TTS_HIP_TILE_KS", "1")
    monkeypatch.delenv("TTS_HIP_ATTN_ROWS", raising=False)
    cfg = synth.small(weight_type=gguf.F16, ctx=80, max_gen=80)
    model = synth.build(cfg)
    rng = np.random.default_rng(24)
    n, cap, slots = 600, 80, 300
    lens = rng.integers(4, 61, n)
    prompts = [rng.integers(3, cfg.prompt_vocab, int(l)).astype(np.uint32) for l in lens]
    n_steps = int(cap - lens.min())
    ref = hip.HipEngine(cfg, max_seqs=n, kv_positions=cap, flags=hip.FLAG_NO_DAC)
    ref.load(model)
    ref.prefill_batch(prompts)
    rt, rd = ref.generate_greedy(lens, n_steps)
    ref.close()
    eng = hip.HipEngine(cfg, max_seqs=slots + 1, kv_positions=cap, flags=hip.FLAG_NO_DAC)
    eng.load(model)
    eng.stream_begin(slots, cap - 1)
    waiting, slot_utt, free, got, rounds = list(range(n)), {}, list(range(slots)), {}, 0
    while waiting or slot_utt:
        take = waiting[:len(free)]
        if take:
            waiting = waiting[len(take):]
            sl = [free.pop(0) for _ in take]
            eng.stream_admit(sl, [prompts[k] for k in take], None)
            slot_utt.update(dict(zip(sl, take)))
        for slot, steps in eng.stream_run(32):
            k = slot_utt.pop(slot)
            got[k] = (steps, eng.stream_collect(slot, steps))
            free.append(slot)
        rounds += 1
        assert rounds < 200
    eng.stream_end()
    eng.close()
    same = 0
    for k in range(n):
        want = int(rd[k]) if rd[k] else n_steps
        steps, toks = got[k]
        assert steps == want == cap - lens[k], (k, steps, want)
        same += bool(np.array_equal(toks, rt[:want, k]))
    print(f"{same} of {n} streams identical")
    assert same >= int(0.9 * n), same


@pytest.mark.parametrize("rows,enc", [(100, 8), (1024, 8), (200, 27)])
def test_cross_attention_inside_the_q_projection_tiles(rows, enc, monkeypatch):
    """Round 5: with the 64 x 64 GEMM tile a tile column of the cross-attention's q projection is one head, and the attention over the voice prompt
    (parler/model.cpp:576-593) runs in that tile's epilogue (gemm_tile_kernel<64, 64, .., EPI_CROSS>) instead of a launch of its own.  Against the
    oracle (ragged row counts: partial tiles; a 27-position prompt: scores beyond one 16-group), and against the same forward with the fold
    switched off (tune cross_fold = 0: attn_short_kernel) — same mathematics, another order of the 64-term score sums."""
    if rows != 1024:
        monkeypatch.setenv("TTS_HIP_TILE_FORCE", "3")   # smaller forwards would pick another tile shape (and keep the launch)
    key = ("wide1e%d" % enc, gguf.F16)
    if key not in _models:
        _models[key] = synth.build(synth.tiny(hidden=1024, heads=16, ffn=4096, layers=1, enc_len=enc, weight_type=gguf.F16))
    model = _models[key]
    cfg = model.cfg
    rng = np.random.default_rng(rows + enc)
    prompts = [rng.integers(3, cfg.prompt_vocab, 2 + (i % 3)).astype(np.uint32) for i in range(rows)]
    ids = np.full((rows, cfg.n_out), cfg.bos, dtype=np.uint32)
    out = {}
    for fold in (1, 0):
        eng = hip.HipEngine(cfg, max_seqs=rows, kv_positions=16, tune={"cross_fold": fold})
        eng.load(model)
        eng.prefill_batch(prompts)
        out[fold] = eng.step(ids, [len(p) for p in prompts])
        eng.close()
    assert relerr(out[1], out[0]) < TOL[gguf.F16]
    for r in sorted({0, 15, 16, 63, 64, rows // 2, rows - 1}):
        o = orc.ParlerOracle(model, act_mode=1, gelu_mode=1)
        o.decode(prompts[r], 0, audio=False, want_logits=False)
        ref, _ = o.decode(ids[r], len(prompts[r]), audio=True)
        assert relerr(out[1][r], ref[:, 0, :]) < TOL[gguf.F16], (rows, enc, r)


@pytest.mark.parametrize("rows,enc", [(1, 8), (3, 8), (4, 27)])
def test_cross_attention_inside_the_out_projection_prologue(rows, enc):
    """The one-sequence chain (<= 4 rows): the cross-attention over the voice prompt runs in the prologue of its own out projection
    (gemm16_kernel<.., PRO_CROSS, ..>: every workgroup recomputes it from the q rows and K_c / V_c) instead of attn_short_kernel + a launch
    boundary.  Against the oracle over three steps, and against the same chain with the fold switched off (tune cross_fold = 0)."""
    key = ("wide1e%d" % enc, gguf.F16)
    if key not in _models:
        _models[key] = synth.build(synth.tiny(hidden=1024, heads=16, ffn=4096, layers=1, enc_len=enc, weight_type=gguf.F16))
    model = _models[key]
    cfg = model.cfg
    rng = np.random.default_rng(rows * 7 + enc)
    prompts = [rng.integers(3, cfg.prompt_vocab, 3 + i).astype(np.uint32) for i in range(rows)]
    oracles = []
    for p in prompts:
        o = orc.ParlerOracle(model, act_mode=1, gelu_mode=1)
        o.decode(p, 0, audio=False, want_logits=False)
        oracles.append(o)
    engines = []
    for fold in (1, 0):
        eng = hip.HipEngine(cfg, max_seqs=rows, kv_positions=16, tune={"cross_fold": fold})
        eng.load(model)
        eng.prefill_batch(prompts)
        engines.append(eng)
    ids = np.full((rows, cfg.n_out), cfg.bos, dtype=np.uint32)
    for step in range(3):
        pos = [len(p) + step for p in prompts]
        a, b = engines[0].step(ids, pos), engines[1].step(ids, pos)
        assert relerr(a, b) < TOL[gguf.F16], step
        for r in range(rows):
            ref, _ = oracles[r].decode(ids[r], pos[r], audio=True)
            assert relerr(a[r], ref[:, 0, :]) < TOL[gguf.F16], (rows, enc, step, r)
            ids[r] = ref[:, 0, :].argmax(-1)
    for eng in engines:
        eng.close()

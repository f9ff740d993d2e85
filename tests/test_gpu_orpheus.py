"""GPU parity: the Orpheus decoder (Llama-3 blocks; tts_hip_orpheus_decode) against the oracle (orc_orpheus_decode, pinned
to a float64 torch golden by tests/test_oracle_cpu.py).  Tolerances as for the Parler decoder: 2e-4 of max|oracle| with F32
weights, 2e-3 with F16, and the Q8_0-activation flip bound for Q4_0 (the BASELINE config-4 type)."""
import os

import numpy as np
import pytest

import oracle as orc
from tts_cpp_amd import gguf, hip, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "tiny_orpheus.npz")


def relerr(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize("wtype,tol", [(gguf.F32, 2e-4), (gguf.F16, 2e-3), (gguf.Q4_0, 3e-2)])
def test_orpheus_prompt_and_steps_match_oracle(wtype, tol):
    model = synth.build_orpheus(synth.orpheus_tiny(weight_type=wtype))
    eng = hip.OrpheusEngine(model.cfg)
    eng.load(model)
    o = orc.OrpheusOracle(model, act_mode=1)
    g = np.load(GOLD)
    prompt = g["prompt"]
    lg, tok = eng.decode(prompt, 0)
    ref = o.decode(prompt, 0)
    assert relerr(lg, ref) < tol
    assert tok == int(lg.argmax())
    if wtype == gguf.F32:
        assert relerr(lg, g["logits"][0]) < 2e-4
    pos = len(prompt)
    for step in range(6):
        t = int(ref.argmax())   # teacher forcing with the oracle's id
        lg, tok = eng.decode([t], pos)
        ref = o.decode([t], pos)
        assert relerr(lg, ref) < tol, step
        pos += 1
    eng.close()


def test_orpheus_greedy_generation_and_chunked_prompt():
    model = synth.build_orpheus(synth.orpheus_tiny())
    cfg = model.cfg
    eng = hip.OrpheusEngine(cfg)
    eng.load(model)
    g = np.load(GOLD)
    toks = eng.generate_greedy(g["prompt"], 6, stop_id=cfg.vocab + 5)
    assert toks.tolist() == g["tokens"].tolist()          # the float64 torch golden's greedy ids
    # stop id ends the loop with the stop token recorded last (generate_from_batch :379)
    assert eng.generate_greedy(g["prompt"], 6, stop_id=int(g["tokens"][2])).tolist() == g["tokens"][:3].tolist()
    # a prompt fed in two decode() calls == one call (KV cache positions)
    a, _ = eng.decode(g["prompt"], 0)
    eng.decode(g["prompt"][:4], 0)
    b, _ = eng.decode(g["prompt"][4:], 4)
    assert relerr(b, a) < 1e-5
    with pytest.raises(hip.HipError):
        eng.decode([cfg.vocab], 0)
    with pytest.raises(hip.HipError):
        eng.decode([1], cfg.ctx)
    eng.close()


@pytest.mark.parametrize("wtype,tol", [(gguf.F16, 2e-3), (gguf.Q4_0, 3e-2)])
def test_orpheus_3b_layer_shapes(wtype, tol):
    """one layer of canopylabs/orpheus-3b's shapes: hidden 3072 (12 K-slices), 24 q heads / 8 kv heads x 128, ffn 8192
    (down_proj in two 4096-column slabs folded by the next rms norm), a vocabulary that is not a multiple of 16 — in F16 and in
    Q4_0, BASELINE config 4's type (Q8_0-quantised activations x 4-bit codes: the 20-token prompt takes the int8 MFMA workgroups,
    the single token the streaming kernels on the Q4_0 codes; tolerance = the activation-flip bound of the integer path)."""
    model = synth.build_orpheus(synth.orpheus_3b(layers=1, vocab=5001, ctx=64, weight_type=wtype))
    eng = hip.OrpheusEngine(model.cfg)
    eng.load(model)
    o = orc.OrpheusOracle(model, act_mode=1)
    ids = np.random.default_rng(1).integers(0, 5001, 20).astype(np.uint32)
    lg, tok = eng.decode(ids, 0)
    ref = o.decode(ids, 0)
    assert lg.shape == (5001,) and relerr(lg, ref) < tol
    lg2, _ = eng.decode([int(ref.argmax())], 20)
    ref2 = o.decode([int(ref.argmax())], 20)
    assert relerr(lg2, ref2) < tol
    # a second single-token step reads the cache rows the first one appended (Q4_0: projection + rope + append in one launch,
    # gemv_q4_qkv_rope_kernel)
    lg3, _ = eng.decode([int(ref2.argmax())], 21)
    assert relerr(lg3, o.decode([int(ref2.argmax())], 21)) < tol
    eng.close()


def test_orpheus_3b_shapes_few_row_calls_take_the_fused_projections():
    """decode() calls of 2-4 tokens at the 3B widths (Q4_0): every row has its own position in the fused launches — rms norm inside the
    staging, q/k/v + rope + cache append, gate|up + silu, down with on-the-fly Q8_0 — and the cache rows they append are read by the next
    call; against the oracle fed the same pieces"""
    model = synth.build_orpheus(synth.orpheus_3b(layers=2, vocab=5001, ctx=64, weight_type=gguf.Q4_0))
    eng = hip.OrpheusEngine(model.cfg)
    eng.load(model)
    o = orc.OrpheusOracle(model, act_mode=1)
    ids = np.random.default_rng(4).integers(0, 5001, 9).astype(np.uint32)
    pos = 0
    for piece in (ids[:3], ids[3:7], ids[7:9]):
        lg, _ = eng.decode(piece, pos)
        ref = o.decode(piece, pos)
        assert relerr(lg, ref) < 3e-2, pos
        pos += len(piece)
    eng.close()


@pytest.mark.parametrize("wtype", [gguf.Q4_0, gguf.Q8_0])
def test_orpheus_3b_shapes_5_to_64_rows_take_the_weight_streaming_integer_gemm(wtype):
    """5 .. 64 rows on GGUF-quantised matrices at the 3B widths (round 6: qgemv_stream_kernel — every projection and, for lock-step utterances, the
    LM head with its 5001 = 312 x 16 + 9 features): K slices of 512 / 768 / 1536 columns as slabs folded by rope / silu * up / the next rms norm
    (3072-wide rows: eight slabs per round trip), 8-, 16-, 32- and 64-row LDS images.  decode() pieces of 5, 8, 9, 16, 23 and 40 rows against the oracle
    fed the same pieces, then lock-step steps of 6, 13 and 21 utterances: every row's logits against the oracle's for that utterance's history; with
    tune("q_stream") = 0 (qgemm16_kernel) the same calls stay inside the same bound and the arg-max tokens agree wherever the oracle's margin allows."""
    model = synth.build_orpheus(synth.orpheus_3b(layers=2, vocab=5001, ctx=128, weight_type=wtype))
    cfg = model.cfg
    o = orc.OrpheusOracle(model, act_mode=1)
    ids = np.random.default_rng(11).integers(0, 5001, 101).astype(np.uint32)
    pieces = (ids[:5], ids[5:13], ids[13:22], ids[22:38], ids[38:61], ids[61:101])   # 23 and 40 rows: two and four row tiles of 16 per workgroup
    refs, p = [], 0
    for piece in pieces:
        refs.append(o.decode(piece, p)); p += len(piece)
    for qs in (31, 0):
        eng = hip.OrpheusEngine(cfg)
        eng.tune("q_stream", qs)
        eng.load(model)
        p = 0
        for piece, ref in zip(pieces, refs):
            lg, tok = eng.decode(piece, p)
            assert relerr(lg, ref) < 3e-2, (qs, p)
            assert ref[tok] >= ref.max() - 2 * 3e-2 * np.abs(ref).max(), (qs, p)
            p += len(piece)
        eng.close()
    # lock-step utterances: B rows of one step, every row with its own cache slot and position
    for B in (6, 13, 21):
        rng = np.random.default_rng(B)
        prompts = [rng.integers(0, 5001, 2 + (u % 5)).astype(np.uint32) for u in range(B)]
        eng = hip.OrpheusEngine(cfg, max_seqs=B)
        eng.load(model)
        first = eng.generate_batch(prompts, 1, stop_id=cfg.vocab + 5)          # fills the slots' caches, returns every utterance's first id
        lg, tok = eng.step_batch(list(range(B)), [int(f[0]) for f in first], [len(q) for q in prompts])
        assert lg.shape == (B, 5001)
        for u in (0, B // 2, B - 1):
            ob = orc.OrpheusOracle(model, act_mode=1)
            ob.decode(prompts[u], 0)
            ref = ob.decode([int(first[u][0])], len(prompts[u]))
            assert relerr(lg[u], ref) < 3e-2, (B, u)
        assert tok.tolist() == [int(l.argmax()) for l in lg]
        eng.close()


def test_orpheus_3b_shapes_captured_step_with_split_attention():
    """The captured greedy step at the 3B widths (Q4_0) cuts the keys of every (row, head) into eight slices (attn_gqa_split_kernel; at these
    positions most of them are empty) and attn_gqa_combine_kernel merges them with every slice requested at once (round 4 kept two more places
    for the merge as switches — measured equal, removed in round 5).  A second generation on the same context repeats the first; four slices
    instead of eight give the same ids wherever the oracle's margin allows; every id must be the oracle's arg-max up to the Q8_0
    activation-flip bound when the oracle is fed the same history."""
    model = synth.build_orpheus(synth.orpheus_3b(layers=2, vocab=5001, ctx=256, weight_type=gguf.Q4_0))
    ids = np.random.default_rng(7).integers(0, 5001, 20).astype(np.uint32)
    toks = {}
    for split in (8, 4):
        eng = hip.OrpheusEngine(model.cfg)
        eng.tune("attn_split", split)
        eng.load(model)
        toks[split] = eng.generate_greedy(ids, 40, stop_id=model.cfg.vocab + 5).tolist()
        if split == 8:
            assert eng.generate_greedy(ids, 40, stop_id=model.cfg.vocab + 5).tolist() == toks[8]
        eng.close()
    assert len(toks[8]) == 40
    o = orc.OrpheusOracle(model, act_mode=1)
    for split in (8, 4):
        ref = o.decode(ids, 0)
        for s, t in enumerate(toks[split]):
            assert ref[t] >= ref.max() - 2 * 3e-2 * np.abs(ref).max(), (split, s, t, int(ref.argmax()))
            ref = o.decode([t], 20 + s)


def test_orpheus_runner_generates_through_both_contexts(tmp_path):
    """runner_from_file on an Orpheus GGUF (orpheus.* + snac.* + byte-pair vocabulary): prompt framing (model.cpp:341-356),
    greedy generate_from_batch (:378-392), 7 ids -> 3 SNAC levels (:358-376), SNAC decode — equal to the oracle pipeline."""
    import os
    import tokenizer_oracle
    from tts_cpp_amd import runner
    full = synth.SynthOrpheusFull(max_gen=28)
    path = full.write_gguf(str(tmp_path / "orpheus.gguf"))
    os.environ["TTS_SNAC_NO_NOISE"] = "1"     # the reference's noise comes from a process-wide engine; parity runs without it
    try:
        r = runner.Runner(path, sample=0)
        assert r.arch == "orpheus" and r.sampling_rate == 24000.0
        pcm = r.generate("hello the zebra", voice=b"zoe", sample=0)
    finally:
        del os.environ["TTS_SNAC_NO_NOISE"]
    tok = tokenizer_oracle.BpeOracle(full.vocab_tokens, full.merges)
    prompt = full.specials["pre"] + tok.tokenize("zoe: hello the zebra") + full.specials["app"]
    assert r.last_tokens(0).tolist() == prompt
    o = orc.OrpheusOracle(full.orpheus)
    lg = o.decode(prompt, 0)
    toks, pos = [], len(prompt)
    while len(toks) < full.max_gen:
        toks.append(int(lg.argmax()))
        if toks[-1] == full.specials["stop"] or len(toks) >= full.max_gen:
            break
        lg = o.decode([toks[-1]], pos)
        pos += 1
    got = r.last_tokens(1).tolist()
    assert got == toks
    heads = [0, 1, 2, 2, 1, 2, 2]
    levels = [[], [], []]
    for i in range(len(toks) // 7):
        for ii in range(7):
            levels[heads[ii]].append(toks[i * 7 + ii] - full.audio_offset)
    T = len(levels[2])
    ref = orc.SnacOracle(full.snac).decode(np.array(levels[0] + levels[1] + levels[2], dtype=np.uint32), T, None)
    assert pcm.shape == ref.shape == (T * full.scfg.hop,)
    assert np.abs(pcm - ref).max() < 1e-4
    # with the noise block active the audio differs and stays a tanh output
    noisy = r.generate("hello the zebra", voice=b"zoe", sample=0)
    assert noisy.shape == pcm.shape and np.abs(noisy).max() <= 1.0 and not np.array_equal(noisy, pcm)
    # seeded sampling: the device sampler loop (tts_hip_orpheus_generate_sampled; top_k 1..64, any top_p) == the per-step host loop
    # (logits D2H + sampler::sample on the host), token for token; other configurations take the host loop
    os.environ["TTS_SNAC_NO_NOISE"] = "1"
    try:
        def run(**kw):   # random weights may sample a text id where an audio id belongs: the SNAC context refuses it, the ids are still there
            try:
                pcm_ = r.generate("hello the zebra", voice=b"zoe", sample=1, **kw)
            except runner.RunnerError as e:
                assert "codebook size" in str(e)
                pcm_ = None
            return pcm_, r.last_tokens(1).copy()
        # top_k stays below the number of distinct logits: this synthetic model has a block of exactly equal logits (untrained text rows),
        # and the order of equal keys is where the device (index order) and std::sort (unspecified, sampler.cpp:167) may differ
        for kw in (dict(top_k=8, seed=3), dict(top_k=16, temperature=0.8, seed=11), dict(top_k=20, temperature=1.3, repetition_penalty=1.3, seed=5),
                   dict(top_k=12, top_p=0.7, temperature=0.9, seed=7)):   # top_p < 1 on the device since round 5
            sampled, dev_toks = run(**kw)
            os.environ["TTS_HOST_LOOP"] = "1"
            try:
                hst, hst_toks = run(**kw)
            finally:
                del os.environ["TTS_HOST_LOOP"]
            assert len(dev_toks) > 0 and np.array_equal(dev_toks, hst_toks), kw
            assert (sampled is None) == (hst is None)
            if sampled is not None:
                assert np.array_equal(sampled, hst) and sampled.size % full.scfg.hop == 0 and np.isfinite(sampled).all()
        assert not np.array_equal(dev_toks, np.array(toks[:len(dev_toks)]))     # it does sample
        nucleus, ntoks = run(top_k=0, top_p=0.9, seed=3)   # host loop
        assert len(ntoks) > 0 and (nucleus is None or (nucleus.size % full.scfg.hop == 0 and np.isfinite(nucleus).all()))
    finally:
        del os.environ["TTS_SNAC_NO_NOISE"]
    with pytest.raises(runner.RunnerError):
        r.generate("hello", voice=b"nobody", sample=0)
    r.close()


def test_orpheus_greedy_through_the_captured_step():
    """TTS_HIP_LLAMA_GRAPH=1: the greedy step (forward + arg-max + feedback) as one hipGraph replayed per position — same ids as the eager loop"""
    model = synth.build_orpheus(synth.orpheus_tiny())
    g = np.load(GOLD)
    os.environ["TTS_HIP_LLAMA_GRAPH"] = "1"
    try:
        eng = hip.OrpheusEngine(model.cfg)
    finally:
        del os.environ["TTS_HIP_LLAMA_GRAPH"]
    eng.load(model)
    assert eng.generate_greedy(g["prompt"], 6, stop_id=model.cfg.vocab + 5).tolist() == g["tokens"].tolist()
    assert eng.generate_greedy(g["prompt"], 40, stop_id=model.cfg.vocab + 5).tolist()[:6] == g["tokens"].tolist()   # several chunks of 8 replays
    assert eng.generate_greedy(g["prompt"], 6, stop_id=int(g["tokens"][2])).tolist() == g["tokens"][:3].tolist()
    eng.close()


def test_orpheus_runner_with_the_noise_block(tmp_path):
    """The SNAC noise block draws from a never-reseeded std::default_random_engine through std::normal_distribution<float>
    (util.cpp:73-79); oracle/rng_oracle.py restates that stream, so the first generate of a fresh runner is comparable with the
    noise active (the other runner test switches it off)."""
    import tokenizer_oracle
    from rng_oracle import minstd0_normal
    from tts_cpp_amd import runner
    full = synth.SynthOrpheusFull(max_gen=28)
    path = full.write_gguf(str(tmp_path / "orpheus.gguf"))
    r = runner.Runner(path, sample=0)
    pcm = r.generate("hello the zebra", voice=b"zoe", sample=0)
    toks = r.last_tokens(1).tolist()
    heads = [0, 1, 2, 2, 1, 2, 2]
    levels = [[], [], []]
    for i in range(len(toks) // 7):
        for ii in range(7):
            levels[heads[ii]].append(toks[i * 7 + ii] - full.audio_offset)
    T = len(levels[2])
    so = orc.SnacOracle(full.snac)
    noise, state, saved = minstd0_normal(so.noise_len(T))
    ref = so.decode(np.array(levels[0] + levels[1] + levels[2], dtype=np.uint32), T, noise)
    assert pcm.shape == ref.shape and np.abs(pcm - ref).max() < 1e-4
    # the second generate continues the stream where the first stopped
    pcm2 = r.generate("hello the zebra", voice=b"zoe", sample=0)
    noise2, _, _ = minstd0_normal(so.noise_len(T), state, saved)
    assert np.abs(pcm2 - so.decode(np.array(levels[0] + levels[1] + levels[2], dtype=np.uint32), T, noise2)).max() < 1e-4
    r.close()


def _oracle_sampler(v, top_k, temp, rep, top_p=1.0):
    import ctypes as C
    smp = orc.Sampler()
    orc.lib().orc_sampler_init(C.byref(smp), 1, v)
    smp.top_k, smp.temperature, smp.top_p, smp.repetition_penalty, smp.do_sample = top_k, temp, top_p, rep, 1
    orc.lib().orc_sampler_reset(C.byref(smp))
    return smp


@pytest.mark.parametrize("top_k,temp,rep,top_p", [(50, 1.0, 1.0, 1.0), (50, 0.6, 1.1, 1.0), (64, 1.4, 1.0, 1.0), (1, 1.0, 1.0, 1.0), (7, 0.9, 1.5, 1.0),
                                                   (50, 1.0, 1.0, 0.9), (20, 0.7, 1.2, 0.5), (64, 1.3, 1.0, 0.95), (50, 1.0, 1.0, 0.05)])
def test_orpheus_device_sampler_over_the_full_vocabulary(top_k, temp, rep, top_p):
    """sampler::sample over 156 940 logits (orpheus/model.cpp:389-398): topk_parts_kernel + topk_sample_kernel against the oracle sampler
    (pinned to the compiled src/sampler.cpp by tests/test_sampler.py), same logits, same uniform draw, same repetition state, three calls
    deep.  top_p < 1 (round 5: softmax_total_kernel — the softmax over the whole vocabulary before the top k, its total accumulated in index
    order like sampler.cpp:82-116; then topp, :118-150): the logits are sharpened so that the nucleus is a handful of candidates and the trim
    point matters.  Bar: identical ids; a draw within a few ulp of a CDF boundary may land on the neighbouring candidate (device expf vs libm)."""
    import ctypes as C
    cfg = synth.orpheus_tiny(vocab=156940)
    model = synth.build_orpheus(cfg)
    eng = hip.OrpheusEngine(cfg)
    eng.load(model)
    V = cfg.vocab
    rng = np.random.default_rng(top_k * 31 + int(temp * 10) + int(top_p * 1000))
    bad = 0
    for case in range(6):
        last, cnt = (int(rng.integers(0, V)), int(rng.integers(1, 5))) if rep != 1.0 else (-1, 0)
        smp = _oracle_sampler(V, top_k, temp, rep, top_p)
        if rep != 1.0:
            smp.last_token_ids[0], smp.repetition_counts[0] = last, cnt
        for call in range(3):
            lg = (rng.standard_normal(V) * (3.0 if top_p >= 1.0 else 6.0)).astype(np.float32)
            if rep != 1.0 and call == 1 and last >= 0:
                lg[last] = 20.0     # the token sampled last is the arg-max: its penalised value decides the softmax maximum
            u = float(np.float32([0.0, 0.99999994][case]) if case < 2 and call == 0 else rng.random(dtype=np.float32))
            tok, last, cnt = eng.sample_logits(lg, u, top_k=top_k, temperature=temp, repetition_penalty=rep, top_p=top_p, last_id=last, rep_count=cnt)
            ref = np.zeros(1, dtype=np.uint32)
            orc.lib().orc_sampler_sample(C.byref(smp), orc.f32p(lg.copy()), orc.f32p(np.float32([u])), orc.u32p(ref))
            if tok != int(ref[0]):
                bad += 1
                break   # the states have diverged
            if rep != 1.0:
                assert (smp.last_token_ids[0], smp.repetition_counts[0]) == (last, cnt)
            if top_k == 1:
                pen = lg.copy()
                assert tok == int(pen.argmax()) or rep != 1.0
    assert bad <= 1, bad
    with pytest.raises(hip.HipError):
        eng.sample_logits(np.zeros(V, dtype=np.float32), 0.5, top_k=65)       # beyond the device sampler: host loop
    with pytest.raises(hip.HipError):
        eng.sample_logits(np.zeros(V, dtype=np.float32), 0.5, top_k=50, top_p=0.0)
    eng.close()


@pytest.mark.parametrize("graph", ["0", "1"])
def test_orpheus_sampled_generation_equals_the_per_step_host_loop(graph):
    """tts_hip_orpheus_generate_sampled (eager steps and the captured step) == decode + logits D2H + the oracle sampler per step with the same
    uniform draws: the ids, the stop on the stopping token, several chunks of 8 replays"""
    import ctypes as C
    model = synth.build_orpheus(synth.orpheus_tiny())
    cfg = model.cfg
    os.environ["TTS_HIP_LLAMA_GRAPH"] = graph
    try:
        eng = hip.OrpheusEngine(cfg)
    finally:
        del os.environ["TTS_HIP_LLAMA_GRAPH"]
    eng.load(model)
    g = np.load(GOLD)
    prompt = g["prompt"]
    for top_k, temp, rep, seed, top_p in ((8, 1.0, 1.0, 1, 1.0), (50, 0.8, 1.2, 2, 1.0), (3, 1.5, 1.0, 3, 1.0), (20, 0.9, 1.1, 4, 0.6)):
        n = 30
        u = np.random.default_rng(seed).random(n, dtype=np.float32)
        got = eng.generate_sampled(prompt, n, stop_id=cfg.vocab + 5, uniforms=u, top_k=top_k, temperature=temp, repetition_penalty=rep, top_p=top_p)
        smp = _oracle_sampler(cfg.vocab, top_k, temp, rep, top_p)
        ref, pos = [], len(prompt)
        lg, _ = eng.decode(prompt, 0)
        for s in range(n):
            t = np.zeros(1, dtype=np.uint32)
            orc.lib().orc_sampler_sample(C.byref(smp), orc.f32p(lg.copy()), orc.f32p(u[s:s + 1].copy()), orc.u32p(t))
            ref.append(int(t[0]))
            if s + 1 < n:
                lg, _ = eng.decode([ref[-1]], pos)
                pos += 1
        assert got.tolist() == ref, (top_k, temp, rep)
        stop = ref[4]
        first = ref.index(stop)
        assert eng.generate_sampled(prompt, n, stop_id=stop, uniforms=u, top_k=top_k, temperature=temp, repetition_penalty=rep, top_p=top_p).tolist() == ref[:first + 1]
    assert len(set(ref)) > 1
    eng.close()


@pytest.mark.parametrize("wtype", [gguf.F16, gguf.Q4_0])
@pytest.mark.parametrize("n_utt", [3, 7])
def test_orpheus_lockstep_batch_equals_single_sequence_generations(n_utt, wtype):
    """VERDICT r5 item 5 (B utterances batched in lock-step inside a GPU; the reference's only concurrency is N independent workers,
    server.cpp:225-321): tts_hip_orpheus_generate_batch — prompts of different lengths in their own cache slots, one row per live utterance, finished
    utterances leaving the step — gives every utterance the ids its own one-sequence generation gives, token for token, greedy and with the device sampler
    (own uniforms and repetition state per utterance); the first utterance's ids are also the oracle's.  3 rows take the streaming 1-4 row kernels,
    7 rows the weight-streaming integer GEMM (qgemv_stream_kernel, round 6; the 5- and 7-row prompts of either engine go through it as well).
    The reference is the one-sequence EAGER loop (TTS_HIP_LLAMA_GRAPH=0): its rows meet the same kernels as a lock-step row, so the ids are equal by
    construction.  The captured one-sequence step attends with attn_gqa_wave_kernel, another association of the same softmax ("agrees to rounding"):
    with Q4_0 matrices one ulp there can flip a Q8_0 code of the next activation row, so against the captured step only the greedy ids are compared
    (equal on these seeds; captured == eager is test_orpheus_greedy_through_the_captured_step's subject)."""
    model = synth.build_orpheus(synth.orpheus_tiny(weight_type=wtype))
    cfg = model.cfg
    rng = np.random.default_rng(10 * n_utt + wtype)
    prompts = [rng.integers(0, cfg.vocab, 3 + 2 * (u % 4)).astype(np.uint32) for u in range(n_utt)]
    max_new = 14
    captured = hip.OrpheusEngine(cfg)
    captured.load(model)
    cap_greedy = [captured.generate_greedy(p, max_new, stop_id=cfg.vocab + 5) for p in prompts]
    captured.close()
    os.environ["TTS_HIP_LLAMA_GRAPH"] = "0"
    try:
        single = hip.OrpheusEngine(cfg)
    finally:
        del os.environ["TTS_HIP_LLAMA_GRAPH"]
    single.load(model)
    ref_greedy = [single.generate_greedy(p, max_new, stop_id=cfg.vocab + 5) for p in prompts]
    stop = int(ref_greedy[1][5])                                  # a stopping token that some utterances meet early and others never
    ref_stop = [single.generate_greedy(p, max_new, stop_id=stop) for p in prompts]
    uni = rng.random((n_utt, max_new), dtype=np.float32)
    ref_smp = [single.generate_sampled(p, max_new, stop_id=cfg.vocab + 5, uniforms=uni[u], top_k=12, temperature=0.9, repetition_penalty=1.2) for u, p in enumerate(prompts)]
    single.close()
    eng = hip.OrpheusEngine(cfg, max_seqs=n_utt)
    eng.load(model)
    got = eng.generate_batch(prompts, max_new, stop_id=cfg.vocab + 5)
    assert [g.tolist() for g in got] == [r.tolist() for r in ref_greedy]
    assert [g.tolist() for g in got] == [r.tolist() for r in cap_greedy]
    got = eng.generate_batch(prompts, max_new, stop_id=stop)
    assert [g.tolist() for g in got] == [r.tolist() for r in ref_stop]
    assert len({len(g) for g in got}) > 1                          # the utterances really finish at different steps
    got = eng.generate_batch(prompts, max_new, stop_id=cfg.vocab + 5, uniforms=uni, top_k=12, temperature=0.9, repetition_penalty=1.2)
    assert [g.tolist() for g in got] == [r.tolist() for r in ref_smp]
    # the one-sequence entry points still work on a multi-slot context (slot 0) ...
    assert eng.generate_greedy(prompts[0], max_new, stop_id=cfg.vocab + 5).tolist() == ref_greedy[0].tolist()
    # ... and a step's logits are the oracle's
    o = orc.OrpheusOracle(model, act_mode=1)
    ref = o.decode(prompts[0], 0)
    eng.generate_batch(prompts, 1, stop_id=cfg.vocab + 5)          # fills the slots' caches with the prompts
    t0 = int(ref.argmax())
    lg, tok = eng.step_batch([0, 1], [t0, int(ref_greedy[1][0])], [len(prompts[0]), len(prompts[1])])
    ref1 = o.decode([t0], len(prompts[0]))
    assert relerr(lg[0], ref1) < (2e-3 if wtype == gguf.F16 else 3e-2)
    assert tok.tolist() == [int(l.argmax()) for l in lg]
    with pytest.raises(hip.HipError, match="max_seqs"):
        eng.generate_batch(prompts + [prompts[0]], 2, stop_id=0)
    eng.close()


def test_orpheus_runner_generate_batch_equals_single_calls(tmp_path):
    """orpheus_runner::generate_batch (host/orpheus_runner.cpp): runner_from_file with max_seqs slots, three sentences in lock-step through
    tts_hip_orpheus_generate_batch, SNAC per utterance — the audio of every utterance is bit for bit the audio of a generate() call of its own,
    greedy and with the seeded device sampler."""
    import os
    from tts_cpp_amd import runner
    full = synth.SynthOrpheusFull(max_gen=28)
    path = full.write_gguf(str(tmp_path / "orpheus.gguf"))
    texts = ["hello the zebra", "a zebra", "the quick hello of the zebra there"]
    os.environ["TTS_SNAC_NO_NOISE"] = "1"
    try:
        one = runner.Runner(path, sample=0)
        many = runner.Runner(path, sample=0, max_seqs=4)
        for kw in (dict(sample=0), dict(sample=1, top_k=8, seed=3)):
            singles = []
            for t in texts:
                try:
                    singles.append(one.generate(t, voice=b"zoe", **kw))
                except runner.RunnerError as e:     # random weights may sample a text id where an audio id belongs
                    assert "codebook size" in str(e)
                    singles.append(None)
            if any(s is None for s in singles):
                continue
            batch = many.generate_batch(texts, voice=b"zoe", **kw)
            assert len(batch) == 3
            for s, b in zip(singles, batch):
                assert np.array_equal(s, b)
        one.close(); many.close()
    finally:
        del os.environ["TTS_SNAC_NO_NOISE"]

"""GPU parity: the first device version of the Kokoro graphs (tts_hip_kokoro_*, csrc/kokoro_kernels.h) against the oracle
(oracle/kokoro_oracle.c, matched to a float64 torch golden by tests/test_oracle_cpu.py)."""
import os

import numpy as np
import pytest

import oracle as orc
from tts_cpp_amd import hip, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "tiny_kokoro.npz")


def relerr(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def test_kokoro_durations_and_audio_match_oracle():
    g = np.load(GOLD)
    model = synth.build_kokoro(synth.kokoro_tiny())
    eng = hip.KokoroEngine(model)
    o = orc.KokoroOracle(model)
    toks = g["tokens"]
    lens, hid = eng.durations(toks, "af_test")
    ref_lens, ref_hid = o.durations(toks, "af_test")
    assert np.array_equal(lens, ref_lens) and np.array_equal(lens, g["lens"])
    assert relerr(hid, ref_hid) < 2e-4 and relerr(hid, g["hidden"]) < 2e-4
    total = int(ref_lens.sum())
    noise = np.random.default_rng(int(g["noise_seed"])).random(9 * 50 * model.cfg.up_sampling_factor, dtype=np.float32)[:o.noise_len(total)]
    # the STFT conditioning as complex numbers (its phase channels wrap at +-pi), then the audio from a shared conditioning
    ref_pcm, _, _, ref_hs = o.generate(toks, ref_lens, ref_hid, "af_test", noise, want_curves=True)
    pcm_own, hs = eng.generate(toks, ref_lens, ref_hid, "af_test", noise, want_hsrc=True)
    nb = model.cfg.n_fft // 2 + 1
    za, zb = hs[:nb] * np.exp(1j * hs[nb:]), ref_hs[:nb] * np.exp(1j * ref_hs[nb:])
    assert np.abs(za - zb).max() < 1e-2 * np.abs(zb).max()
    pcm = eng.generate(toks, ref_lens, ref_hid, "af_test", noise, hsrc_in=ref_hs)
    assert pcm.shape == ref_pcm.shape and relerr(pcm, ref_pcm) < 2e-4
    assert relerr(eng.generate(toks, g["lens"], g["hidden"], "af_test", noise, hsrc_in=g["hsrc"]), g["pcm"]) < 2e-4
    assert pcm_own.shape == pcm.shape and np.isfinite(pcm_own).all()
    # ... and the audio from the device's OWN conditioning.  The conditioning's phase channels are atan2 values that feed convolutions
    # directly: a bin sitting at +-pi flips by 2 pi under rounding-level differences of the fp32 harmonic source, so the comparison is
    # phase-aware: (1) the conditioning agrees as complex numbers (above); (2) bins whose phase value differs by more than pi are the
    # wrapped ones and must be rare; (3) with the wrapped bins' phase taken from the oracle, the audio agrees sample for sample;
    # (4) the unpatched audio differs from the oracle's only by what those few bins explain (small in the mean).
    dphi = np.abs(hs[nb:] - ref_hs[nb:])
    small = np.abs(zb) < 1e-3 * np.abs(zb).max()            # bins with (almost) no energy have no defined phase
    wrapped = (dphi > np.pi) & ~small
    patched = hs.copy()
    for sel in (wrapped, small):
        patched[nb:][sel] = ref_hs[nb:][sel]
    patched[:nb][small] = ref_hs[:nb][small]
    pcm_patched = eng.generate(toks, ref_lens, ref_hid, "af_test", noise, hsrc_in=patched)
    print(f"kokoro own conditioning: {int(wrapped.sum())} wrapped phase values among {int((~small).sum())} bins with energy, {int(small.sum())} empty bins, "
          f"patched-vs-oracle {relerr(pcm_patched, ref_pcm):.2e}, own-vs-oracle mean abs {np.abs(pcm_own - ref_pcm).mean():.2e} (max |oracle| {np.abs(ref_pcm).max():.2e})")
    assert wrapped.sum() < 0.05 * max(1, (~small).sum()), f"{wrapped.sum()} of {(~small).sum()} phase values wrapped"
    # measured 9e-3 (single-workgroup LSTM) and 4.4e-2 (split LSTM: another summation order of the recurrent dot products): the phases of the
    # bins that did not wrap still move at the level the complex comparison above allows (1e-2 of the largest bin), and they feed convolutions
    assert relerr(pcm_patched, ref_pcm) < 1e-1
    assert np.abs(pcm_own - ref_pcm).mean() < 5e-2 * np.abs(ref_pcm).max()
    # forced durations (BASELINE's Kokoro configuration) and the other voice
    lens2 = np.array([1, 4, 2, 1, 3, 2, 5, 1], dtype=np.float32)
    n2 = np.random.default_rng(3).random(o.noise_len(int(lens2.sum())), dtype=np.float32)
    r2, _, _, h2 = o.generate(toks, lens2, ref_hid, "bm_test", n2, want_curves=True)
    assert relerr(eng.generate(toks, lens2, ref_hid, "bm_test", n2, hsrc_in=h2), r2) < 2e-4
    with pytest.raises(hip.HipError):
        eng.durations(toks, "nobody")
    with pytest.raises(hip.HipError):
        eng.generate(toks, lens2 + 0.5, ref_hid, "af_test", n2)
    eng.close()


def test_kokoro_runner_from_file(tmp_path):
    """runner_from_file on a Kokoro GGUF: phoneme string -> clause chunks -> durations -> source noise -> audio (kokoro/model.cpp:1277-1446).
    The reference phonemizes first; the runner is handed the phonemes (host/kokoro_runner.h)."""
    import tokenizer_oracle
    from rng_oracle import minstd0_uniform
    from tts_cpp_amd import gguf, runner
    model = synth.build_kokoro(synth.kokoro_tiny())
    cfg = model.cfg
    path = model.write_gguf(str(tmp_path / "kokoro.gguf"))
    vocab = gguf.Reader(path).kv["tokenizer.ggml.tokens"]
    r = runner.Runner(path, voice=b"af_test")
    assert r.arch == "kokoro" and r.sampling_rate == 24000.0
    text = "abc de. fgh"
    pcm = r.generate(text, voice=b"af_test")
    tok = tokenizer_oracle.SinglePassOracle(vocab)
    chunks = tokenizer_oracle.kokoro_chunks(tok, text, cfg.max_ctx)
    assert r.last_tokens(0).tolist() == [t for ch in chunks for t in ch]
    o = orc.KokoroOracle(model)
    eng = hip.KokoroEngine(model)
    state, want = 1, []
    for ch in chunks:
        lens, hid = eng.durations(ch, "af_test")                           # the runner's own two device calls, same noise stream: bit-equal audio
        assert np.array_equal(lens, o.durations(ch, "af_test")[0])
        noise, state = minstd0_uniform(o.noise_len(int(lens.sum())), state)
        want.append(eng.generate(ch, lens, hid, "af_test", noise))
    want = np.concatenate(want)
    assert pcm.shape == want.shape and np.array_equal(pcm, want)
    with pytest.raises(runner.RunnerError):
        r.generate(text, voice=b"nobody")
    r.close()


def test_kokoro_device_pool_workers_share_one_weight_arena(tmp_path):
    """device_pool with two workers on device 0 and a Kokoro file (the reference: every worker loads the whole model, server.cpp:316-321): the file is parsed
    and uploaded once, the second worker's runner is loaded with share_with (kokoro_runner::device_context, declare-only tensors, finalize on the first
    worker's arena).  Every request comes back with the number of samples its own generate() gives (the durations are deterministic; the source noise of a
    worker depends on what that worker synthesised before, as with the reference's process-wide engine)."""
    from tts_cpp_amd import runner
    model = synth.build_kokoro(synth.kokoro_tiny())
    path = model.write_gguf(str(tmp_path / "kokoro.gguf"))
    texts = ["abc de. fgh", "hgf ed cba", "a", "cab. bac! abc? cba", "de de de de", "fgh abc. de"]
    r = runner.Runner(path, voice=b"af_test")
    expect = [r.generate(t, voice=b"af_test").size for t in texts]
    r.close()
    pool = runner.Pool(path, n_workers=2, devices=[0], max_batch=2, voice=b"af_test")
    assert pool.load_stats() == {"weight_broadcasts": 0, "shared_arena_loads": 1}
    ids = [pool.submit(t, voice=b"af_test") for t in texts]
    for i, n in zip(ids, expect):
        audio, bs, wk, err = pool.wait(i, timeout_ms=60000)
        assert err == "" and audio.size == n and np.isfinite(audio).all() and wk in (0, 1)
    pool.close()


def test_kokoro_runner_generate_batch_equals_generate_calls_in_a_row(tmp_path):
    """kokoro_runner::generate_batch (VERDICT r5 item 3 for Kokoro: utterances batched inside one GPU; the reference's only concurrency is N workers with a
    model each, server.cpp:225-321): the clauses of n utterances run through `max_seqs` device contexts on their own streams, all reading ONE weight arena
    (lanes declared with the runner's tensors and finalized on its arena).  The audio of every utterance is BIT FOR BIT that of the same utterances given
    to generate() one after the other — including the source noise, one minstd stream in the reference: clause i's stretch starts where clause i - 1's ends
    (engine jumped ahead once the earlier durations are known) — and the runner's engine is left where the sequential calls would leave it.  A second runner
    loaded with share_with (the device pool's path for further workers of a device) reads the same arena and gives the same audio."""
    from tts_cpp_amd import runner
    model = synth.build_kokoro(synth.kokoro_tiny())
    path = model.write_gguf(str(tmp_path / "kokoro.gguf"))
    texts = ["abc de. fgh", "hgf ed cba", "a", "cab. bac! abc? cba", "de de de de", "fgh abc. de", "b"]
    seq = runner.Runner(path, voice=b"af_test")
    want = [seq.generate(t, voice=b"af_test") for t in texts]
    after = seq.generate("abc", voice=b"af_test")
    assert len({w.size for w in want}) > 2 and all(w.size for w in want)
    bat = runner.Runner(path, voice=b"af_test", max_seqs=3)
    got = bat.generate_batch(texts, voice=b"af_test")
    for u, (g, w) in enumerate(zip(got, want)):
        assert g.shape == w.shape and np.array_equal(g, w), u
    assert np.array_equal(bat.generate("abc", voice=b"af_test"), after)       # the noise engine continues as after the sequential calls
    # a second batch on the same lanes (already created), fewer utterances than lanes
    seq2 = runner.Runner(path, voice=b"af_test")
    want2 = [seq2.generate(t, voice=b"af_test") for t in texts[:2]]
    bat2 = runner.Runner(path, voice=b"af_test", max_seqs=4, share_with=bat)   # no weights of its own: bat's arena
    got2 = bat2.generate_batch(texts[:2], voice=b"af_test")
    assert all(np.array_equal(g, w) for g, w in zip(got2, want2))
    with pytest.raises(runner.RunnerError):
        bat.generate_batch(texts, voice=b"nobody")
    bat2.close()
    bat.close()
    assert seq.generate(texts[2], voice=b"af_test").size == want[2].size      # closing the sharers left the other runners' arenas alone
    seq.close()
    seq2.close()


@pytest.mark.parametrize("tune", [{}, {"kokoro_split": 0}, {"kokoro_b3": 0}], ids=["fp16_hi_lo_convs", "bf16x3_convs", "exact_fp32_convs"])
def test_kokoro_82m_shapes_match_oracle(tune):
    """(default since round 6: the k = 3 / 5 / 7 / 11 same-convolutions as fp16 hi + lo split products, three MFMAs per product,
    conv1d_mfma_b3_kernel<.., KT, SplitH2>; tune("kokoro_split") = 0: three bf16 planes, six products; tune("kokoro_b3") = 0: the exact-fp32 MFMA kernel)
    BASELINE config 2's dimensions (hexgrad/Kokoro-82M: ALBERT 768 x 12 recurrences, predictor / text encoder 512, decoder 1024,
    generator 512 -> 256 -> 128, (10, 6) upsampling, n_fft 20 / hop 5) with seeded weights, 10 phoneme ids: durations identical,
    duration states, and the audio from the oracle's conditioning sample for sample"""
    model = synth.build_kokoro(synth.kokoro_82m())
    cfg = model.cfg
    eng = hip.KokoroEngine(model, tune=tune)
    o = orc.KokoroOracle(model)
    rng = np.random.default_rng(82)
    toks = np.concatenate([[0], rng.integers(1, cfg.vocab, 10), [0]]).astype(np.uint32)
    lens, hid = eng.durations(toks, cfg.voices[0])
    ref_lens, ref_hid = o.durations(toks, cfg.voices[0])
    # ALBERT's GELU goes through ggml's fp16 table on both sides (kokoro/model.cpp:1000): an x within fp32 rounding distance of an fp16
    # boundary lands in the neighbouring entry (1e-3 of that element) -> measured 3.4e-4 over the states (9e-7 with an fp32 GELU)
    assert np.array_equal(lens, ref_lens) and relerr(hid, ref_hid) < 2e-3
    forced = np.full(toks.size, 2.0, dtype=np.float32)       # forced durations: shape determinism, and both sides see the same alignment
    noise = rng.random(o.noise_len(int(forced.sum())), dtype=np.float32)
    ref_pcm, _, _, ref_hs = o.generate(toks, forced, ref_hid, cfg.voices[0], noise, want_curves=True)
    pcm = eng.generate(toks, forced, ref_hid, cfg.voices[0], noise, hsrc_in=ref_hs)
    print(f"kokoro-82m: duration states {relerr(hid, ref_hid):.2e}, predicted lengths equal: {np.array_equal(lens, ref_lens)}, audio {relerr(pcm, ref_pcm):.2e}")
    assert pcm.shape == ref_pcm.shape and relerr(pcm, ref_pcm) < 2e-4          # measured 2.8e-6
    eng.close()


def test_kokoro_albert_attention_through_lds_equals_the_row_walking_kernel():
    """kk_albert_attn64_kernel (round 6: a chunk of 64 keys staged through LDS, four rows per workgroup) keeps kk_albert_attn_kernel's arithmetic term for
    term: the duration states of 75 phoneme ids (two key chunks, a ragged last workgroup) are bit-identical with tune("kokoro_attn_lds") = 0 and 1, and
    the predicted lengths are the oracle's."""
    model = synth.build_kokoro(synth.kokoro_82m())
    cfg = model.cfg
    rng = np.random.default_rng(75)
    toks = np.concatenate([[0], rng.integers(1, cfg.vocab, 75), [0]]).astype(np.uint32)
    res = {}
    for v in (0, 1):
        eng = hip.KokoroEngine(model, tune={"kokoro_attn_lds": v})
        res[v] = eng.durations(toks, cfg.voices[0])
        eng.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    ref_lens, ref_hid = orc.KokoroOracle(model).durations(toks, cfg.voices[0])
    assert np.array_equal(res[1][0], ref_lens) and relerr(res[1][1], ref_hid) < 2e-3


def test_kokoro_82m_projections_over_a_long_sequence():
    """40 phoneme ids at the 82M widths: from 32 rows on, ALBERT's and the predictor's projections run on kk_linear_mfma_kernel (64 x 64
    tiles of the exact-fp32 MFMA) instead of one wave per output — durations identical, duration states against the oracle"""
    model = synth.build_kokoro(synth.kokoro_82m())
    cfg = model.cfg
    eng = hip.KokoroEngine(model)
    o = orc.KokoroOracle(model)
    rng = np.random.default_rng(40)
    toks = np.concatenate([[0], rng.integers(1, cfg.vocab, 40), [0]]).astype(np.uint32)
    lens, hid = eng.durations(toks, cfg.voices[0])
    ref_lens, ref_hid = o.durations(toks, cfg.voices[0])
    print(f"kokoro-82m, 42 rows: duration states {relerr(hid, ref_hid):.2e}")
    assert np.array_equal(lens, ref_lens) and relerr(hid, ref_hid) < 2e-3     # fp16-table GELU on both sides (see the 82M test above)
    eng.close()

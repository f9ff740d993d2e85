"""CPU tests of the C++ host layer (libtts.so): GGUF reader vs the Python writer/reader, unigram tokenizer
vs the Python restatement, sampler vs the REAL reference sampler, the test:dummy plumbing backend, and the
exported C ABI (include/tts_c.h)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle as orc
import tokenizer_oracle
from tts_cpp_amd import gguf, runner, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tiny_gguf(tmp_path_factory):
    m = synth.build(synth.tiny(weight_type=gguf.Q8_0))
    p = str(tmp_path_factory.mktemp("gguf") / "tiny.gguf")
    m.write_gguf(p)
    return m, p


def test_tts_c_exports_match_header():
    hdr = open(os.path.join(ROOT, "include", "tts_c.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(tts_c_[a-z_0-9]+)\s*\(", hdr)))
    assert sorted(runner.EXPORTS) == declared
    L = C.CDLL(runner.lib_path())
    for name in declared:
        assert hasattr(L, name), name


def test_cpp_gguf_reader_matches_python(tiny_gguf):
    m, p = tiny_gguf
    L = runner.load_lib()
    nt, nkv, off = C.c_uint64(), C.c_uint64(), C.c_uint64()
    arch = C.create_string_buffer(64)
    assert L.tts_c_gguf_summary(p.encode(), C.byref(nt), C.byref(nkv), C.byref(off), arch, 64) == 0
    r = gguf.Reader(p)
    assert nt.value == len(m.tensors) and nkv.value == len(m.kv) and off.value == r.data_offset
    assert arch.value == b"parler-tts"
    for idx in (0, 7, len(m.tensors) // 2, len(m.tensors) - 1):
        name = C.create_string_buffer(256)
        ttype, ne, cs = C.c_int(), (C.c_int64 * 4)(), C.c_uint64()
        assert L.tts_c_gguf_tensor(p.encode(), idx, name, 256, C.byref(ttype), ne, C.byref(cs)) == 0
        t = m.tensors[idx]
        assert name.value.decode() == t.name and ttype.value == t.type
        assert list(ne)[:len(t.ne)] == t.ne
        h = 1469598103934665603
        for b in bytes(t.raw()):
            h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        assert cs.value == h, t.name


TEXTS = ["hello world", "The birch canoe slid on the smooth planks.", "  two  spaces\tand\n\nnewlines  ", "", "a",
         "zzzzqqqq xkcd", "naïve café ☕ 你好", "It's easy to tell the depth of a well.", "ALL CAPS and MiXeD", "x" * 200]


@pytest.mark.parametrize("text", TEXTS)
def test_tokenizer_matches_restatement(tiny_gguf, text):
    m, p = tiny_gguf
    o = tokenizer_oracle.UnigramOracle(m.vocab, m.scores, 2, 1)
    exp = o.tokenize(text) + [1]
    got = runner.tokenize(p, text)
    assert got.tolist() == exp
    assert all(t < m.cfg.prompt_vocab for t in got)


def test_tokenizer_prefers_high_scoring_segmentation():
    vocab = ["<pad>", "</s>", "<unk>", " ", "a", "b", "ab", " a", " ab", "abab"]
    scores = [0, 0, -10, -1, -2, -2, -3.5, -2.5, -4.2, -9]
    o = tokenizer_oracle.UnigramOracle(vocab, scores, 2)
    assert o.tokenize("abab") == [8, 6]        # " ab" + "ab" (-7.7) beats " a","b","ab" (-8) and " "+"abab" (-10)
    assert o.tokenize("a?b") == [7, 2, 5]      # unknown code point -> <unk>
    assert o.tokenize("a??b") == [7, 2, 5]     # consecutive unknowns are joined


def _ref():
    r = orc.ref_sampler_lib()
    if r is None:
        pytest.skip("oracle/_ref not built")
    return r


@pytest.mark.parametrize("top_k,top_p,temp,rep", [(50, 1.0, 1.0, 1.0), (50, 0.9, 0.8, 1.0), (0, 0.7, 1.0, 1.0), (30, 0.95, 1.1, 1.3), (1, 1.0, 1.0, 1.0)])
def test_host_cpp_sampler_matches_reference(top_k, top_p, temp, rep):
    """the product's sampler (host/sampler.cpp) against the real reference sampler's distribution"""
    R, L = _ref(), runner.load_lib()
    NH, V = 9, 1088
    for seed in range(3):
        lg = (np.random.default_rng(50 + seed).standard_normal((NH, V)) * 3).astype(np.float32)
        rc = orc.RefSamplerCfg(NH, V, top_k, temp, top_p, rep, 1)
        last = lg.argmax(axis=1).astype(np.int32)
        last[1::2] = 3
        counts = np.arange(1, NH + 1, dtype=np.uint32)
        ref_l, picks, n_picks, mhp = lg.copy(), np.zeros((NH, V), dtype=np.uint32), np.zeros(NH, dtype=np.uint32), np.zeros(NH, dtype=np.float32)
        R.ref_sampler_distribution(C.byref(rc), last.ctypes.data_as(C.POINTER(C.c_int32)), orc.u32p(counts), orc.f32p(ref_l),
                                   orc.u32p(picks), orc.u32p(n_picks), orc.f32p(mhp))
        u = np.random.default_rng(seed).random(NH).astype(np.float32)
        exp = []
        for i in range(NH):
            a = np.float32(u[i] * mhp[i]) if top_p < 1.0 else u[i]
            cum, n = np.float32(0), int(n_picks[i])
            for j in range(n):
                cum = np.float32(cum + ref_l[i, picks[i, j]])
                if a <= cum or j >= n - 1:
                    exp.append(int(picks[i, j]))
                    break
        sc = runner.SamplerCfg(NH, V, top_k, temp, top_p, rep, 1, 0)
        out = np.zeros(NH, dtype=np.uint32)
        mine = lg.copy()
        L.tts_c_sampler_sample(C.byref(sc), last.ctypes.data_as(C.POINTER(C.c_int32)), orc.u32p(counts), orc.f32p(mine), orc.f32p(u), orc.u32p(out))
        assert out.tolist() == exp


def test_host_sampler_greedy_and_seed():
    R, L = _ref(), runner.load_lib()
    NH, V = 9, 1088
    lg = (np.random.default_rng(9).standard_normal((NH, V)) * 2).astype(np.float32)
    lg[:, 500] = lg.max() + 1
    lg[:, 20] = lg[:, 500]  # tie: the first maximum (20) wins
    rc = orc.RefSamplerCfg(NH, V, 50, 1.0, 1.0, 1.0, 0)
    ref_out = np.zeros(NH, dtype=np.uint32)
    R.ref_sampler_sample_greedy(C.byref(rc), orc.f32p(lg.copy()), orc.u32p(ref_out))
    sc = runner.SamplerCfg(NH, V, 50, 1.0, 1.0, 1.0, 0, 0)
    out = np.zeros(NH, dtype=np.uint32)
    L.tts_c_sampler_sample(C.byref(sc), None, None, orc.f32p(lg.copy()), None, orc.u32p(out))
    assert (out == ref_out).all() and (out == 20).all()
    # seeded sampling is reproducible (extension), different seeds differ
    outs = []
    for seed in (7, 7, 8):
        sc = runner.SamplerCfg(NH, V, 50, 1.0, 1.0, 1.0, 1, seed)
        o = np.zeros(NH, dtype=np.uint32)
        L.tts_c_sampler_sample(C.byref(sc), None, None, orc.f32p(lg.copy()), None, orc.u32p(o))
        outs.append(o.copy())
    assert (outs[0] == outs[1]).all() and not (outs[0] == outs[2]).all()


def test_dummy_backend_plumbing():
    """config 1 of BASELINE.json ("plumbing, no GPU"): runner_from_file("test:dummy") -> generate -> PCM"""
    r = runner.Runner("test:dummy")
    pcm = r.generate("ab")
    assert r.arch == "dummy" and r.sampling_rate == 44100.0
    assert pcm.shape == (2 * 44100,)
    j = np.arange(44100, dtype=np.float32)
    for i, ch in enumerate("ab"):
        wl = np.float32(44100 / np.pi / 2) / np.float32(200 + ord(ch))
        exp = np.sin(j * np.float32(np.pi / 44100)) * np.sin(j / wl)
        assert np.abs(pcm[i * 44100:(i + 1) * 44100] - exp).max() < 2e-3
    r.close()
    with pytest.raises(runner.RunnerError):
        runner.Runner("test:nope")
    with pytest.raises(runner.RunnerError):
        runner.Runner("/nonexistent/model.gguf")


def test_runner_from_file_fails_loudly_without_gpu(tiny_gguf, have_gpu):
    if have_gpu:
        pytest.skip("a GPU is present")
    with pytest.raises(runner.RunnerError) as e:
        runner.Runner(tiny_gguf[1])
    assert "tts_hip_create failed" in str(e.value)


def test_parler_metadata_that_cannot_describe_a_model_is_refused_at_load(tiny_gguf, tmp_path):
    """read_hparams: zero output heads (a division further down the runner), a hidden size the heads do not divide — refused with the metadata
    message before anything touches the device (so the check runs without a GPU too)."""
    m, _ = tiny_gguf
    for key, bad in (("parler-tts.decoder.output_heads", 0), ("parler-tts.decoder.attention.head_count", 7), ("parler-tts.decoder.num_hidden_layers", 0)):
        kv = [(k, t, bad if k == key else v) for k, t, v in m.kv]
        path = str(tmp_path / "bad.gguf")
        gguf.write(path, kv, m.tensors)
        with pytest.raises(runner.RunnerError) as e:
            runner.Runner(path)
        assert "metadata out of range" in str(e.value), (key, str(e.value))


def test_malformed_gguf_files_never_crash_the_reader(have_gpu):
    """tests/tools/fuzz_gguf.py in a subprocess (a crash must not take the suite down): truncated and bit-flipped model files through
    tts_c_gguf_summary / tts_c_gguf_tensor / tts_c_runner_from_file — error codes, never a signal."""
    if have_gpu:
        pytest.skip("the loader would go on to create device contexts for the files that still parse")
    import subprocess, sys
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "fuzz_gguf.py"), runner.lib_path(), "7", "160"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "no crash" in p.stdout, (p.returncode, p.stdout[-300:], p.stderr[-600:])


def test_hostile_text_never_crashes_the_tokenizers():
    """tests/tools/fuzz_tokenizers.py in a subprocess: arbitrary bytes, invalid / truncated UTF-8, long inputs and short output buffers through the
    unigram, Dia, single-pass and Kokoro chunking entry points of include/tts_c.h."""
    import subprocess, sys
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "fuzz_tokenizers.py"), runner.lib_path(), "5", "120"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "no crash" in p.stdout, (p.returncode, p.stdout[-300:], p.stderr[-600:])


def test_device_pool_queue_batching_and_responses():
    """device_pool (host/device_pool.h ~ examples/server/server.cpp:126-330): tasks pushed from several threads,
    pulled by 2 workers, compatible queued tasks decoded together, responses fetched by id.  Runs on the weightless
    test:dummy backend (its generate_batch is the base-class loop), so this is the queue/batch/response logic alone."""
    import threading

    ref = runner.Runner("test:dummy")
    pool = runner.Pool("test:dummy", n_workers=2, max_batch=4, batch_window_ms=30)
    texts = [f"utterance number {i}" for i in range(13)]
    ids = {}
    lock = threading.Lock()

    def producer(lo, hi):
        for i in range(lo, hi):
            t = pool.submit(texts[i])
            with lock:
                ids[i] = t

    th = [threading.Thread(target=producer, args=(a, b)) for a, b in ((0, 5), (5, 9), (9, 13))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert len(set(ids.values())) == 13
    workers, sizes = set(), []
    for i in range(13):
        audio, bs, wk, err = pool.wait(ids[i], timeout_ms=20000)
        assert err == "" and np.array_equal(audio, ref.generate(texts[i])), i
        assert 1 <= bs <= 4
        workers.add(wk)
        sizes.append(bs)
    st = pool.stats()
    assert st["tasks"] == 13 and st["timed_out"] == 0
    assert st["batches"] < 13 and 2 <= st["largest_batch"] <= 4, st   # requests did share passes
    assert workers <= {0, 1}
    # a different generation_configuration is not batched with the others
    a = pool.submit("same text", top_k=5)
    b = pool.submit("same text", top_k=7)
    ra, rb = pool.wait(a, 20000), pool.wait(b, 20000)
    assert ra[1] == 1 and rb[1] == 1 and np.array_equal(ra[0], rb[0])
    # an empty prompt gives an empty response: finished, success == false (server.cpp:258)
    e = pool.submit("")
    audio, bs, wk, err = pool.wait(e, 20000)
    assert audio.size == 0 and err != ""
    # unknown id: wait times out
    with pytest.raises(runner.RunnerError):
        pool.wait(10 ** 6, timeout_ms=50)
    pool.close()
    ref.close()


def test_device_pool_continuous_batching_admits_arrivals_during_a_generation():
    """pool_options::continuous on the weightless backend (four rows; an utterance of n characters generates for n look-in intervals of
    10 ms): two long requests open a session, five short ones arrive while those are generating — they must enter the rows that are free
    (and the rows that free up) of the SAME session instead of waiting for the long ones to finish, and everybody gets the audio of a
    generate() call of their own.  generate_stream (any number of sentences through one session) is checked against generate()."""
    import time
    pool = runner.Pool("test:dummy", n_workers=1, max_batch=4, continuous=True)
    t0 = time.perf_counter()
    long_ids = [pool.submit("l" * 80) for _ in range(2)]             # 0.8 s of generation: a loaded CI box may stall the submitter for a while
    time.sleep(0.08)
    short_ids = [pool.submit("ab") for _ in range(5)]
    short = [pool.wait(i) for i in short_ids]
    t_short = time.perf_counter() - t0
    long_ = [pool.wait(i) for i in long_ids]
    t_long = time.perf_counter() - t0
    for audio, bs, wk, err in short:
        assert err == "" and audio.size == 2 * 44100 and wk == 0
    for audio, bs, wk, err in long_:
        assert err == "" and audio.size == 80 * 44100
    st = pool.stats()
    assert st["tasks"] == 7 and st["batches"] == 1, st          # one session served all seven
    # the worker may open the session on the first long request before the second is queued: then that one is an in-flight admission too
    assert st["admitted_in_flight"] in (5, 6) and st["largest_batch"] == 4, st
    assert t_short < t_long - 0.1, (t_short, t_long)               # the short requests did not wait for the long ones
    pool.close()
    r = runner.Runner("test:dummy")
    texts = ["abc", "d", "efgh", "ij", "k", "lmnop", "q"]          # seven utterances through four rows
    outs = r.generate_stream(texts)
    for t, o in zip(texts, outs):
        assert np.array_equal(o, r.generate(t))
    r.close()


def test_device_pool_continuous_session_that_cannot_open_answers_every_request():
    """stream_begin() throws in the worker (a real runner refuses a cross-attention mismatch, a host-only sampler or a failed device
    allocation there).  The first batch was in none of the lists the handler failed, so wait(id) never returned.  Every request of the batch
    must come back with the message, and the pool must keep serving."""
    pool = runner.Pool("test:dummy", n_workers=1, max_batch=4, continuous=True)
    bad = [pool.submit("abc", voice=b"test:stream_begin-fails") for _ in range(3)]
    for i in bad:
        audio, bs, wk, err = pool.wait(i, 20000)
        assert audio.size == 0 and "test:stream_begin-fails" in err, err
    good = pool.submit("de")
    audio, bs, wk, err = pool.wait(good, 20000)
    assert err == "" and audio.size == 2 * 44100
    assert pool.stats()["tasks"] == 4
    pool.close()


def test_device_pool_continuous_session_yields_to_an_older_incompatible_request():
    """a continuous session admitted compatible requests from anywhere in the queue for as long as they kept coming; a request with
    other sampling parameters (or another model) waited without bound.  With continuous_yield_ms the session stops admitting once such a request
    has waited that long at the head of the queue, drains, and the next session serves it: it must come back before the compatible requests
    that were submitted after the bound had passed."""
    import time
    pool = runner.Pool("test:dummy", n_workers=1, max_batch=4, continuous=True, continuous_yield_ms=100)
    first = [pool.submit("a" * 20) for _ in range(2)]          # 0.2 s of generation each
    time.sleep(0.03)
    other = pool.submit("zz", top_k=7)                          # incompatible with the running session
    later = []
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.6:                       # a steady stream of compatible requests
        later.append((time.perf_counter() - t0, pool.submit("b" * 10)))
        time.sleep(0.02)
    audio, bs, wk, err = pool.wait(other, 20000)
    t_other = time.perf_counter() - t0
    assert err == "" and audio.size == 2 * 44100
    done_late = [pool.wait(i, 20000) for _, i in later]
    t_all = time.perf_counter() - t0
    assert all(e == "" for _, _, _, e in done_late)
    assert t_other < t_all - 0.1, (t_other, t_all)              # served before the stream behind it had drained
    assert pool.stats()["batches"] >= 3                         # the session ended for it, and another one followed
    for i in first:
        pool.wait(i, 20000)
    pool.close()


def test_device_pool_under_concurrent_submitters():
    """host/pool_stress.c: four threads submit 24 requests each to three workers and wait for them while reading the statistics, once as lock-step
    batches and once as continuous sessions.  Every request gets its own audio, and the statistics are complete the moment the last wait returns (a
    continuous session used to book its counters only when it ended: a waiter could read 57 of 96 tasks).  `make -C tts.cpp_amd/host sanitize` runs the
    same driver with the host library under ThreadSanitizer and AddressSanitizer (clean at the end of round 4; about a minute to build, not part of
    this suite)."""
    import subprocess
    host = os.path.join(ROOT, "tts.cpp_amd", "host")
    subprocess.run(["make", "-C", host, "pool_stress"], check=True, capture_output=True)
    p = subprocess.run([os.path.join(host, "pool_stress")], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "lock-step: tasks 96" in p.stdout and "continuous: tasks 96" in p.stdout, p.stdout


def test_device_pool_conditional_prompt_task():
    """CONDITIONAL_PROMPT (server.cpp:263-271): without --text-encoder-path the task is answered with the reference's
    message; with one, every worker applies it (here the dummy backend refuses, as any non-Parler architecture does)."""
    pool = runner.Pool("test:dummy", n_workers=2, max_batch=2)
    audio, bs, wk, err = pool.wait(pool.conditional_prompt("a calm voice"), 20000)
    assert audio.size == 0 and "text encoder path must be specified" in err
    pool.close()
    pool = runner.Pool("test:dummy", n_workers=2, max_batch=2, text_encoder_path="/nonexistent/t5.gguf")
    audio, bs, wk, err = pool.wait(pool.conditional_prompt("a calm voice"), 20000)
    assert audio.size == 0 and "does not support update_conditional_prompt" in err
    t = pool.submit("still serving")          # TTS tasks keep flowing after a failed control task
    assert pool.wait(t, 20000)[0].size > 0
    pool.close()


def test_device_pool_reports_load_failure():
    with pytest.raises(runner.RunnerError):
        runner.Pool("/nonexistent/model.gguf", n_workers=2, max_batch=2)


def test_bpe_tokenizer_matches_python_restatement(tmp_path):
    """host/tokenizer.cpp bpe_tokenizer (Orpheus; src/tokenizer.cpp:209-296) against oracle/tokenizer_oracle.BpeOracle on a
    synthetic merge table: whole-piece hits, rank order, equal-rank position ties, the sticky "Ġ" prefix, UTF-8 symbols,
    unknown symbols -> id 0."""
    import ctypes as C
    import tokenizer_oracle
    from tts_cpp_amd import gguf as gg
    vocab = ["<unk>", "a", "b", "c", "d", "e", "Ġ", "ab", "abc", "bc", "cd", "Ġa", "Ġab", "é", "éa", "hello", "Ġhello", "de", "cde", "dd", "ddd"]
    merges = ["a b", "ab c", "b c", "c d", "Ġ a", "Ġa b", "é a", "d e", "c de", "d d", "dd d"]
    path = str(tmp_path / "bpe.gguf")
    gg.write(path, [("general.architecture", gg.T_STR, "orpheus"), ("tokenizer.ggml.tokens", gg.T_ARR, (gg.T_STR, vocab)),
                    ("tokenizer.ggml.merges", gg.T_ARR, (gg.T_STR, merges)), ("tokenizer.ggml.bos_token_id", gg.T_U32, 0),
                    ("tokenizer.ggml.eos_token_id", gg.T_U32, 1)], [])
    o = tokenizer_oracle.BpeOracle(vocab, merges)
    L = runner.load_lib()
    for text in ["abc", "ab cd", "hello hello", "abcd abcd", "éab x", "a  b", " a", "dddd cde", "bcde", "", "zzz", "dédé abc"]:
        out = (C.c_uint32 * 64)()
        n = L.tts_c_tokenize(path.encode(), text.encode("utf-8"), out, 64)
        assert n >= 0, L.tts_c_last_error()
        assert list(out[:n]) == o.tokenize(text), text
    assert o.tokenize("abc") == [8]                 # whole-piece hit
    assert o.tokenize("hello hello") == [15, 16]    # "Ġ" prefix after the first space


# ---- Dia host logic (host/dia_runner.cpp) against the oracle's restatement of src/models/dia/model.cpp:661-808 --------------
def test_dia_host_tokenizer_matches_oracle():
    import oracle as orc
    for text in ["hello world", "  [S2] spaced out.  ", "[S1] a [S2] b [S1] c", "no tag, with comma", "x", "[S1]", "café naïve"]:
        want, n = orc.dia_tokenize(text, 48)
        got, m = runner.dia_tokenize(text, 48)
        assert m == n and np.array_equal(got, want), text
    assert runner.dia_tokenize("café", 16)[0].max() < 256          # UTF-8 bytes stay byte values (see DESIGN: the reference sign-extends)
    with pytest.raises(runner.RunnerError):
        runner.dia_tokenize("a" * 60, 48)                                # longer than the encoder context (model.cpp:689-691)


def test_dia_host_stopping_and_undelay_match_oracle():
    import oracle as orc
    from tts_cpp_amd import synth
    cfg = synth.dia_tiny()
    o = orc.DiaOracle(synth.build_dia(cfg))
    rng = np.random.default_rng(4)
    # every (position, countdown state) the loop can be in, with and without an EOS on head 0
    for max_gen in (40, 100):
        for pos in range(0, max_gen):
            for d in (-1, 15, 9, 1):
                ids = rng.integers(0, cfg.audio_vocab, cfg.n_out).astype(np.uint32)
                if pos % 7 == 3:
                    ids[0] = cfg.eos
                want = o.check_stopping(ids, pos, max_gen, d)
                got = runner.dia_check_stopping(ids, cfg.eos, cfg.pad, cfg.max_delay, pos, max_gen, d)
                assert got[0] == want[0] and np.array_equal(got[1], want[1]) and got[2] == want[2], (max_gen, pos, d)
    for steps in (10, 16, 17, 40):
        toks = rng.integers(0, cfg.audio_vocab + 3, (steps, cfg.n_out)).astype(np.uint32)   # a few specials: frames get dropped
        assert np.array_equal(runner.dia_adjust_output_tokens(toks, cfg.audio_vocab, cfg.max_delay), o.adjust_output_tokens(toks)), steps


# ---- Kokoro host logic (host/kokoro_runner.cpp) against the restatement in oracle/tokenizer_oracle.py ---------------------------
KOKORO_VOCAB = ["", "a", "b", "ab", "ʃ", "ə", "ˈ", "t", "h", "e", "l", "o", "w", "r", "d", "k", " ", ",", "ɪ", "n", "ŋ"]   # id 16 is the space (model.h:185)


def test_kokoro_single_pass_tokenizer_matches_oracle():
    tok = tokenizer_oracle.SinglePassOracle(KOKORO_VOCAB)
    for text in ["hello world", "ab abba", "ˈʃə təl", "xyz", "", "a?b", "həˈloʊ wɜːld", "ŋŋ nn ɪ"]:
        assert runner.single_pass_tokenize(KOKORO_VOCAB, text).tolist() == tok.tokenize(text), text
    # the shortest matching entry wins: "ab" is never produced while "a" is in the vocabulary (tokenizer.cpp:163-170)
    assert runner.single_pass_tokenize(KOKORO_VOCAB, "ab").tolist() == [1, 2]
    # a multi-byte symbol outside the vocabulary costs one unknown id per byte
    assert runner.single_pass_tokenize(KOKORO_VOCAB, "é").tolist() == [0, 0]


def test_kokoro_clause_chunking_matches_oracle():
    tok = tokenizer_oracle.SinglePassOracle(KOKORO_VOCAB)
    rng = np.random.default_rng(8)
    words = ["hello", "world", "talk", "ˈʃə", "ten", "doll", "loaned", "a"]
    for max_ctx, n_words in ((64, 5), (24, 12), (24, 40), (16, 30), (12, 25)):
        for trial in range(6):
            parts = []
            for i in range(n_words):
                parts.append(words[int(rng.integers(len(words)))])
                parts.append(rng.choice([" ", " ", " ", ". ", "! ", "? ", "\n"]))
            text = "".join(parts)
            want = tokenizer_oracle.kokoro_chunks(tok, text, max_ctx)
            got = runner.kokoro_chunks(KOKORO_VOCAB, text, max_ctx)
            assert got == want, (max_ctx, text)
            assert all(c[0] == 0 and c[-1] == 0 for c in got)
    assert runner.kokoro_chunks(KOKORO_VOCAB, " . ! ", 64) == []
    # a clause longer than the context without any space is cut mid-word
    long_word = "hello" * 10
    got = runner.kokoro_chunks(KOKORO_VOCAB, long_word, 16)
    assert got == tokenizer_oracle.kokoro_chunks(tok, long_word, 16) and len(got) > 1


def test_a_caller_written_against_the_reference_api_builds_and_runs(tmp_path):
    """The reference's applications and its one test (tests/aPaleBlueDot/main.cpp) only use `generation_configuration`'s positional
    constructor, `runner_from_file(path, n_threads, config, cpu_only)`, `runner->generate(text, response, config)` and the
    `tts_response` / `sampling_rate` fields.  A translation unit written against exactly that surface compiles against host/common.h
    and runs on the weightless backend (BASELINE config 1: plumbing, no GPU)."""
    import subprocess
    host = os.path.join(ROOT, "tts.cpp_amd", "host")
    src = tmp_path / "caller.cpp"
    src.write_text(r'''
#include <cstdio>
#include <memory>
#include "common.h"
int main() {
    generation_configuration config("", 30, 1.0f, 1.1f, false, "", 256, 0.95f, true);   // voice, top_k, temperature, repetition_penalty,
                                                                                         // use_cross_attn, espeak_voice_id, max_tokens, top_p, sample
    std::unique_ptr<tts_generation_runner> runner{runner_from_file("test:dummy", 4, config, true)};
    tts_response response{};
    runner->generate("Hello", response, config);
    std::printf("%zu %.0f %s %d\n", response.n_outputs, runner->sampling_rate, runner->loader.get().arch, (int) runner->supports_voices);
    return response.data && response.n_outputs == 5 * 44100 ? 0 : 1;
}
''')
    exe = tmp_path / "caller"
    cc = subprocess.run(["g++", "-std=c++20", "-O1", "-I", host, str(src), "-o", str(exe), "-L", host, "-ltts", "-L", os.path.join(ROOT, "tts.cpp_amd"), "-ltts_hip",
                         f"-Wl,-rpath,{host}", f"-Wl,-rpath,{os.path.join(ROOT, 'tts.cpp_amd')}", "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True, timeout=300)
    assert cc.returncode == 0, cc.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    assert run.stdout.split() == ["220500", "44100", "dummy", "0"]


REFERENCE = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "examples", "cli")), reason="needs the reference checkout (absent on the GPU box)")
def test_reference_cli_and_perf_battery_build_unchanged_against_the_overlay(tmp_path):
    """The reference's OWN examples/cli/cli.cpp (+ playback / vad / write_file / src/args.cpp) and examples/perf_battery/perf_battery.cpp,
    compiled from where they lie with compat/'s three headers in place of include/common.h, src/models/loaders.h and ggml.h
    (compat/make_overlay.py), link against libtts.so and run on the weightless backend (BASELINE configs[0])."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "compat"))
    from make_overlay import make_overlay
    ovl = make_overlay(REFERENCE, str(tmp_path / "ovl"))
    # the application sources are links into the checkout, not copies
    assert os.path.realpath(os.path.join(ovl, "examples/cli/cli.cpp")) == os.path.join(REFERENCE, "examples/cli/cli.cpp")
    host = os.path.join(ROOT, "tts.cpp_amd", "host")
    link = ["-L", host, "-ltts", "-L", os.path.join(ROOT, "tts.cpp_amd"), "-ltts_hip", f"-Wl,-rpath,{host}",
            f"-Wl,-rpath,{os.path.join(ROOT, 'tts.cpp_amd')}", "-Wl,-rpath,/opt/rocm/lib"]
    builds = {
        "tts-cli": ["examples/cli/cli.cpp", "examples/cli/playback.cpp", "examples/cli/vad.cpp", "examples/cli/write_file.cpp", "src/args.cpp"],
        "perf_battery": ["examples/perf_battery/perf_battery.cpp", "src/args.cpp"],
    }
    for exe, srcs in builds.items():
        cc = subprocess.run(["g++", "-std=c++20", "-O1", "-I", "include", *srcs, "-o", exe, *link], cwd=ovl, capture_output=True, text=True, timeout=600)
        assert cc.returncode == 0, cc.stderr[-4000:]
    wav = tmp_path / "out.wav"
    run = subprocess.run(["./tts-cli", "--model-path", "test:dummy", "--prompt", "Hello", "--save-path", str(wav)], cwd=ovl, capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "total time" in run.stdout                       # cli.cpp:10-20 through compat/include/ggml.h
    assert wav.stat().st_size == 44 + 2 * 5 * 44100          # "Hello" -> 5 s of 16-bit mono at 44.1 kHz (test runner: 1 s per character)
    run = subprocess.run(["./perf_battery", "--model-path", "test:dummy"], cwd=ovl, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "Mean Stats for arch dummy" in run.stdout and "Generation Real Time Factor" in run.stdout
    # no ggml anywhere in what was linked
    ldd = subprocess.run(["ldd", os.path.join(ovl, "tts-cli")], capture_output=True, text=True).stdout
    assert "libtts.so" in ldd and "ggml" not in ldd


def test_reference_server_builds_unchanged_against_the_overlay_and_serves_a_wav(tmp_path):
    """north_star: "drops in under the existing CLI/server".  The reference's OWN examples/server/server.cpp (with its vendored httplib.h / json.hpp),
    compiled from where it lies against compat/ (common.h, loaders.h, ggml.h = timers + GGML_ASSERT, util.h = the string helpers server.cpp:809-821
    uses) and the page header its CMake generates from public/index.html (make_overlay.py writes it the way cmake/xxd.cmake does), linked to libtts.so,
    started on the weightless backend; POST /v1/audio/speech must come back as a RIFF / WAVE body."""
    import json
    import socket
    import subprocess
    import sys
    import time
    import urllib.request
    sys.path.insert(0, os.path.join(ROOT, "compat"))
    from make_overlay import make_overlay
    ovl = make_overlay(REFERENCE, str(tmp_path / "ovl"))
    assert os.path.realpath(os.path.join(ovl, "examples/server/server.cpp")) == os.path.join(REFERENCE, "examples/server/server.cpp")
    host = os.path.join(ROOT, "tts.cpp_amd", "host")
    link = ["-L", host, "-ltts", "-L", os.path.join(ROOT, "tts.cpp_amd"), "-ltts_hip", f"-Wl,-rpath,{host}",
            f"-Wl,-rpath,{os.path.join(ROOT, 'tts.cpp_amd')}", "-Wl,-rpath,/opt/rocm/lib", "-lpthread"]
    # server.cpp:5 includes the C++20 header <format>; g++ 11 has none: compat/shims/format stands in ONLY then (a toolchain with <format> never sees it)
    probe = subprocess.run(["g++", "-std=c++20", "-x", "c++", "-fsyntax-only", "-"], input="#include <format>\nint main(){}\n", capture_output=True, text=True)
    shim = [] if probe.returncode == 0 else ["-I", os.path.join(ROOT, "compat", "shims")]
    cc = subprocess.run(["g++", "-std=c++20", "-O1", "-I", "include", "-I", "examples/server", *shim, "examples/server/server.cpp", "src/args.cpp", "-o", "tts-server", *link],
                        cwd=ovl, capture_output=True, text=True, timeout=900)
    assert cc.returncode == 0, cc.stderr[-4000:]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    srv = subprocess.Popen(["./tts-server", "--model-path", "test:dummy", "--port", str(port)], cwd=ovl, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        base = f"http://127.0.0.1:{port}"
        for _ in range(100):
            try:
                if json.loads(urllib.request.urlopen(base + "/health", timeout=2).read())["status"] == "ok":
                    break
            except OSError:
                time.sleep(0.1)
        else:
            raise AssertionError("the server never answered /health")
        req = urllib.request.Request(base + "/v1/audio/speech", data=json.dumps({"input": "Hi"}).encode(), headers={"Content-Type": "application/json"})
        r = urllib.request.urlopen(req, timeout=60)
        body = r.read()
        assert r.status == 200 and r.headers.get("Content-Type") == "audio/wav"
        assert body[:4] == b"RIFF" and body[8:12] == b"WAVE" and len(body) > 44 + 2 * 44100   # "Hi": 2 s of audio from the weightless runner
        models = json.loads(urllib.request.urlopen(base + "/v1/models", timeout=5).read())
        assert models["data"] and models["data"][0]["object"] == "model"
        page = urllib.request.urlopen(base + "/", timeout=5).read()
        assert page == open(os.path.join(REFERENCE, "examples/server/public/index.html"), "rb").read()   # the generated page header round-trips
    finally:
        srv.terminate()
        try:
            srv.wait(timeout=10)
        except subprocess.TimeoutExpired:
            srv.kill()
    ldd = subprocess.run(["ldd", os.path.join(ovl, "tts-server")], capture_output=True, text=True).stdout
    assert "libtts.so" in ldd and "ggml" not in ldd


def test_gguf_reader_survives_damaged_files(tiny_gguf, tmp_path):
    """Truncations at every structural boundary region and random byte flips in the metadata: the mmap reader (host/gguf.cpp) must
    answer with an error or a consistent parse, never read outside the mapping (run in a child process so that a crash is a failure,
    not the end of the test session)."""
    import subprocess
    import sys
    path = tiny_gguf[1] if isinstance(tiny_gguf, tuple) else tiny_gguf
    data = open(path, "rb").read()
    r = gguf.Reader(path)
    meta_end = r.data_offset
    rng = np.random.default_rng(12)
    cases = []
    for cut in sorted(set([0, 3, 4, 8, 16, 23, 24, 40] + rng.integers(24, meta_end, 60).tolist() + [meta_end - 1, meta_end, meta_end + 1, len(data) - 1])):
        cases.append(data[:cut])
    for _ in range(120):
        b = bytearray(data[:meta_end + 4096])
        for pos in rng.integers(0, meta_end, int(rng.integers(1, 6))):
            b[int(pos)] = int(rng.integers(0, 256))
        cases.append(bytes(b) + data[meta_end + 4096:])
    paths = []
    for i, c in enumerate(cases):
        p = tmp_path / f"d{i}.gguf"
        p.write_bytes(c)
        paths.append(str(p))
    code = r'''
import ctypes as C, sys
sys.path.insert(0, %r)
import tts_cpp_amd
from tts_cpp_amd import runner
L = runner.load_lib()
ok = bad = 0
for p in sys.argv[1:]:
    nt, nkv, off = C.c_uint64(), C.c_uint64(), C.c_uint64()
    arch = C.create_string_buffer(64)
    rc = L.tts_c_gguf_summary(p.encode(), C.byref(nt), C.byref(nkv), C.byref(off), arch, 64)
    if rc == 0:
        ok += 1
        name = C.create_string_buffer(256); tt = C.c_int(); ne = (C.c_int64 * 4)(); cs = C.c_uint64()
        for idx in range(min(int(nt.value), 4)):
            L.tts_c_gguf_tensor(p.encode(), idx, name, 256, C.byref(tt), ne, C.byref(cs))
    else:
        bad += 1
print(ok, bad)
''' % ROOT
    out = subprocess.run([sys.executable, "-c", code] + paths, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    ok, bad = map(int, out.stdout.split())
    assert ok + bad == len(cases) and bad >= 40        # every truncation inside the metadata is refused


def test_gguf_reader_refuses_hostile_lengths(tmp_path):
    """length fields chosen to wrap pointer / size arithmetic: string length, array count, tensor dimensions, data offset"""
    import struct
    L = runner.load_lib()

    def summary(b):
        p = tmp_path / "x.gguf"
        p.write_bytes(b)
        nt, nkv, off = C.c_uint64(), C.c_uint64(), C.c_uint64()
        return L.tts_c_gguf_summary(str(p).encode(), C.byref(nt), C.byref(nkv), C.byref(off), C.create_string_buffer(64), 64), L.tts_c_last_error().decode()

    hdr = b"GGUF" + struct.pack("<IQQ", 3, 1, 1)
    big = 0xFFFFFFFFFFFFFFF0
    kv = struct.pack("<Q", 1) + b"k" + struct.pack("<II", 4, 7)
    tensor = lambda dims, off: struct.pack("<Q", 1) + b"t" + struct.pack("<I", len(dims)) + b"".join(struct.pack("<Q", d) for d in dims) + struct.pack("<IQ", 0, off)
    assert summary(hdr + struct.pack("<Q", big) + b"abc") == (-1, "truncated or corrupt GGUF metadata")
    assert summary(hdr + struct.pack("<Q", 1) + b"k" + struct.pack("<II", 9, 4) + struct.pack("<Q", big) + b"\0" * 64)[0] == -1
    assert "impossible dimensions" in summary(hdr + kv + tensor([1 << 62, 1 << 62], 0) + b"\0" * 64)[1]
    assert "impossible dimensions" in summary(hdr + kv + tensor([0, 4], 0) + b"\0" * 64)[1]
    assert "past the end" in summary(hdr + kv + tensor([8], big) + b"\0" * 96)[1]
    assert summary(hdr + kv + tensor([8], 0) + b"\0" * 96)[0] == 0           # the well-formed twin loads


def test_noise_stream_restatement_matches_the_local_standard_library(tmp_path):
    """The Kokoro runner draws its source noise like the reference (std::default_random_engine + uniform_real_distribution<float>,
    util.cpp:65-71); oracle/rng_oracle.py restates that stream for the runner parity test.  Checked against a program built here."""
    import subprocess
    from rng_oracle import minstd0_uniform
    src = tmp_path / "rng.cpp"
    src.write_text('#include <cstdio>\n#include <random>\nint main() { std::default_random_engine e; std::uniform_real_distribution<float> d(0.0f, 1.0f);\n'
                   '  for (int i = 0; i < 3000; i++) std::printf("%.9g\\n", d(e)); }\n')
    exe = tmp_path / "rng"
    assert subprocess.run(["g++", "-O2", str(src), "-o", str(exe)], capture_output=True, text=True, timeout=120).returncode == 0
    want = np.array([float(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, timeout=60).stdout.split()], dtype=np.float32)
    got, state = minstd0_uniform(3000)
    assert np.array_equal(got, want)
    more, _ = minstd0_uniform(10, state)                         # the state carries over between clauses
    assert np.array_equal(np.concatenate([got, more])[-10:], more) and 0.0 <= got.min() and got.max() < 1.0
    # the SNAC noise block's stream (random_normal_gen, util.cpp:73-79)
    from rng_oracle import minstd0_normal
    src.write_text('#include <cstdio>\n#include <random>\nint main() { std::default_random_engine e; std::normal_distribution<float> d(0.0f, 1.0f);\n'
                   '  for (int i = 0; i < 2001; i++) std::printf("%.9g\\n", d(e)); }\n')
    assert subprocess.run(["g++", "-O2", str(src), "-o", str(exe)], capture_output=True, text=True, timeout=120).returncode == 0
    want = np.array([float(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, timeout=60).stdout.split()], dtype=np.float32)
    a, st, sv = minstd0_normal(1001)                             # an odd count leaves the pair's second value saved
    b, _, _ = minstd0_normal(1000, st, sv)
    assert np.array_equal(np.concatenate([a, b]), want)


def test_noise_stream_jump_ahead_equals_drawing():
    """kokoro_runner::generate_batch (host/kokoro_runner.cpp) gives every clause its stretch of the ONE minstd stream the reference draws source noise from
    (util.cpp:65-71) by jumping the engine ahead: x -> 16807^k x mod 2^31 - 1.  The jump equals k draws, composes, and k = 0 is the identity."""
    from rng_oracle import minstd0_uniform
    L = runner.load_lib()
    L.tts_c_minstd0_jump.restype = C.c_uint32
    L.tts_c_minstd0_jump.argtypes = [C.c_uint32, C.c_uint64]
    for start in (1, 48271, 2147483646):
        assert L.tts_c_minstd0_jump(start, 0) == start
        for k in (1, 2, 999, 5400):
            _, st = minstd0_uniform(k, start)
            assert L.tts_c_minstd0_jump(start, k) == st
    big = 7 * 600 * 9 * 1200 + 13                                  # a long clause's noise: frames x 600 samples x 9 harmonics
    x = 1
    for _ in range(3):
        x = L.tts_c_minstd0_jump(x, big)
    assert x == L.tts_c_minstd0_jump(1, 3 * big) == pow(16807, 3 * big, 2147483647)
    assert L.tts_c_minstd0_jump(1, 2147483646) == 1               # the multiplicative group's order
    # a clause's noise drawn as parallel stretches (the standard library's own engine and distribution from jumped states) is the sequential stream
    L.tts_c_minstd0_uniform.restype = C.c_uint32
    L.tts_c_minstd0_uniform.argtypes = [C.c_uint32, C.c_uint64, C.POINTER(C.c_float), C.c_uint32]
    n = 3 * 65536 + 4321
    want, st = minstd0_uniform(5000, 12345)
    for threads in (1, 2, 7):
        out = np.empty(n, dtype=np.float32)
        end = L.tts_c_minstd0_uniform(12345, n, out.ctypes.data_as(C.POINTER(C.c_float)), threads)
        assert np.array_equal(out[:5000], want) and end == L.tts_c_minstd0_jump(12345, n)
        if threads == 1:
            ref = out
        else:
            assert np.array_equal(out, ref), threads


@pytest.mark.parametrize("kind", ["dia", "kokoro", "orpheus"])
def test_cpp_gguf_reader_on_the_other_architectures(tmp_path, kind):
    """the C++ reader sees every tensor (name, type, shape, bytes) and key of the Dia / Kokoro / Orpheus files the generators mint"""
    model = {"dia": lambda: synth.build_dia(synth.dia_tiny()), "kokoro": lambda: synth.build_kokoro(synth.kokoro_tiny()),
             "orpheus": lambda: synth.SynthOrpheusFull(max_gen=28)}[kind]()
    p = model.write_gguf(str(tmp_path / f"{kind}.gguf"))
    L = runner.load_lib()
    nt, nkv, off = C.c_uint64(), C.c_uint64(), C.c_uint64()
    arch = C.create_string_buffer(64)
    assert L.tts_c_gguf_summary(p.encode(), C.byref(nt), C.byref(nkv), C.byref(off), arch, 64) == 0
    assert arch.value.decode() == kind and nt.value == len(model.tensors) and nkv.value == len(model.kv)
    r = gguf.Reader(p)
    assert off.value == r.data_offset
    name = C.create_string_buffer(256)
    ttype, ne, cs = C.c_int(), (C.c_int64 * 4)(), C.c_uint64()
    for idx in sorted(set(np.linspace(0, len(model.tensors) - 1, 25).astype(int).tolist())):
        assert L.tts_c_gguf_tensor(p.encode(), idx, name, 256, C.byref(ttype), ne, C.byref(cs)) == 0
        t = model.tensors[idx]
        assert name.value.decode() == t.name and ttype.value == t.type and list(ne)[:len(t.ne)] == t.ne
        h = 1469598103934665603
        for b in bytes(t.raw()):
            h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        assert cs.value == h, t.name

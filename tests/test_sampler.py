"""The sampler restatement (oracle/tts_oracle.c) against the REAL reference sampler
(/root/reference/src/sampler.cpp compiled in place into oracle/_ref/libref_sampler.so)."""
import ctypes as C

import numpy as np
import pytest

import oracle as orc

NH, V = 9, 1088


def ref():
    r = orc.ref_sampler_lib()
    if r is None:
        pytest.skip("oracle/_ref/libref_sampler.so not built (reference sources absent)")
    return r


def mk_cfg(top_k=0, top_p=1.0, temp=1.0, rep=1.0, do_sample=1, nh=NH, v=V):
    c = orc.RefSamplerCfg()
    c.n_output_heads, c.vocab_size, c.top_k = nh, v, top_k
    c.temperature, c.top_p, c.repetition_penalty, c.do_sample = temp, top_p, rep, do_sample
    return c


def mk_orc(cfg):
    s = orc.Sampler()
    orc.lib().orc_sampler_init(C.byref(s), cfg.n_output_heads, cfg.vocab_size)
    s.top_k, s.temperature, s.top_p, s.repetition_penalty, s.do_sample = cfg.top_k, cfg.temperature, cfg.top_p, cfg.repetition_penalty, cfg.do_sample
    orc.lib().orc_sampler_reset(C.byref(s))
    return s


def logits(seed, scale=3.0):
    return (np.random.default_rng(seed).standard_normal((NH, V)) * scale).astype(np.float32)


@pytest.mark.parametrize("seed", range(5))
def test_greedy_matches_reference(seed):
    R = ref()
    lg = logits(seed)
    if seed == 4:  # exact ties: first maximum must win (sampler.cpp:197 `v > max`)
        lg[:, 100] = lg.max() + 1
        lg[:, 900] = lg[:, 100]
    cfg = mk_cfg(do_sample=0)
    out_ref = np.zeros(NH, dtype=np.uint32)
    R.ref_sampler_sample_greedy(C.byref(cfg), orc.f32p(lg.copy()), orc.u32p(out_ref))
    s = mk_orc(cfg)
    out = np.zeros(NH, dtype=np.uint32)
    orc.lib().orc_sampler_sample(C.byref(s), orc.f32p(lg.copy()), None, orc.u32p(out))
    assert (out == out_ref).all()
    assert (out == lg.argmax(axis=1)).all()


@pytest.mark.parametrize("top_k,top_p,temp,rep", [
    (50, 1.0, 1.0, 1.0),   # CLI default (cli.cpp:28)
    (50, 1.0, 0.7, 1.0),
    (50, 0.9, 1.0, 1.0),
    (0, 0.8, 1.3, 1.0),
    (20, 0.95, 0.9, 1.2),
    (1, 1.0, 1.0, 1.0),    # --topk 1 == greedy
])
def test_distribution_matches_reference(top_k, top_p, temp, rep):
    """max -> softmax/topk/topp exactly as sampler::sample runs them, then the inverse-CDF draw with
    the SAME uniform numbers: picks, probabilities and chosen ids must coincide."""
    R = ref()
    for seed in range(3):
        lg = logits(100 + seed)
        cfg = mk_cfg(top_k=top_k, top_p=top_p, temp=temp, rep=rep)
        last = np.array([int(x) for x in lg.argmax(axis=1)], dtype=np.int32)
        last[::2] = 7
        counts = np.arange(1, NH + 1, dtype=np.uint32)
        ref_l = lg.copy()
        picks = np.zeros((NH, V), dtype=np.uint32)
        n_picks = np.zeros(NH, dtype=np.uint32)
        mhp = np.zeros(NH, dtype=np.float32)
        R.ref_sampler_distribution(C.byref(cfg), last.ctypes.data_as(C.POINTER(C.c_int32)), orc.u32p(counts),
                                   orc.f32p(ref_l), orc.u32p(picks), orc.u32p(n_picks), orc.f32p(mhp))
        # expected draw from the reference's own distribution (sampler.cpp:49-68)
        u = np.random.default_rng(seed).random(NH).astype(np.float32)
        expect = []
        for i in range(NH):
            a = np.float32(u[i] * mhp[i]) if top_p < 1.0 else u[i]
            cum = np.float32(0)
            n = int(n_picks[i])
            assert n > 0
            for j in range(n):
                ii = int(picks[i, j])
                cum = np.float32(cum + ref_l[i, ii])
                if a <= cum or j >= n - 1:
                    expect.append(ii)
                    break
        s = mk_orc(cfg)
        if rep != 1.0:
            for i in range(NH):
                s.last_token_ids[i] = int(last[i])
                s.repetition_counts[i] = int(counts[i])
        mine_l = lg.copy()
        out = np.zeros(NH, dtype=np.uint32)
        orc.lib().orc_sampler_sample(C.byref(s), orc.f32p(mine_l), orc.f32p(u), orc.u32p(out))
        assert out.tolist() == expect
        # the in-place probability mass the reference leaves in `logits` must match on the picks
        for i in range(NH):
            idx = picks[i, :int(n_picks[i])]
            assert np.allclose(mine_l[i, idx], ref_l[i, idx], rtol=1e-6, atol=1e-9)


def test_max_with_repetition_state_matches_reference():
    R = ref()
    lg = logits(7)
    cfg = mk_cfg(rep=1.5)
    last = lg.argmax(axis=1).astype(np.int32)
    counts = np.full(NH, 3, dtype=np.uint32)
    out_ref = np.zeros(NH, dtype=np.uint32)
    R.ref_sampler_max(C.byref(cfg), last.ctypes.data_as(C.POINTER(C.c_int32)), orc.u32p(counts), orc.f32p(lg.copy()), orc.u32p(out_ref))
    s = mk_orc(cfg)
    for i in range(NH):
        s.last_token_ids[i], s.repetition_counts[i] = int(last[i]), 3
    out = np.zeros(NH, dtype=np.uint32)
    orc.lib().orc_sampler_max(C.byref(s), orc.f32p(lg), orc.u32p(out))
    assert (out == out_ref).all()

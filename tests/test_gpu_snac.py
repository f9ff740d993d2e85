"""GPU parity: the SNAC codec (tts_hip_snac_decode; Orpheus' audio decoder, src/decoder/snac_model.cpp) against the
oracle (orc_snac_decode, pinned to a float64 torch restatement in tests/golden/tiny_snac.npz by tests/test_oracle_cpu.py).
PCM is a tanh output: 1e-4 absolute, as for DAC."""
import os

import numpy as np
import pytest

import oracle as orc
from tts_cpp_amd import hip, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "tiny_snac.npz")


def rand_codes(cfg, T, seed):
    rng = np.random.default_rng(seed)
    return np.concatenate([rng.integers(0, cfg.cb_size, T // r) for r in cfg.repeats]).astype(np.uint32)


@pytest.mark.parametrize("flags", [0, hip.FLAG_VALU_GEMM])
def test_tiny_snac_matches_oracle_and_golden(flags):
    model = synth.build_snac(synth.snac_tiny())
    eng = hip.SnacEngine(model.cfg, flags=flags)
    eng.load(model)
    o = orc.SnacOracle(model)
    g = np.load(GOLD)
    T = int(g["T"])
    pcm_n = eng.decode(g["codes"], T, g["noise"])
    pcm_c = eng.decode(g["codes"], T, None)
    assert np.abs(pcm_n - g["pcm_noise"]).max() < 1e-4 and np.abs(pcm_c - g["pcm_clean"]).max() < 1e-4
    assert np.abs(pcm_n - pcm_c).max() > 1e-3          # the noise block does something
    for T2 in (4, 8, 40):
        codes = rand_codes(model.cfg, T2, T2)
        noise = np.random.default_rng(T2).standard_normal(o.noise_len(T2)).astype(np.float32)
        assert np.abs(eng.decode(codes, T2, noise) - o.decode(codes, T2, noise)).max() < 1e-4, T2
    assert eng.decode(np.zeros(0, dtype=np.uint32), 0).size == 0
    with pytest.raises(hip.HipError):
        eng.decode(rand_codes(model.cfg, 8, 1)[:-1].tolist() + [model.cfg.cb_size], 8)   # id outside the codebook
    with pytest.raises(hip.HipError):
        eng.decode(rand_codes(model.cfg, 8, 1), 6)                                          # T not a multiple of 4
    eng.close()


def test_snac_24khz_shapes():
    """hubertsiuzdak/snac_24khz dims (768 -> 1024 -> 512 -> 256 -> 128 -> 64 channels, strides 8,8,4,2, codebooks 4096 x 8):
    every pointwise conv and transposed conv on its MFMA tile."""
    model = synth.build_snac(synth.snac_24khz(max_frames=16))
    eng = hip.SnacEngine(model.cfg)
    eng.load(model)
    o = orc.SnacOracle(model)
    T = 8
    codes = rand_codes(model.cfg, T, 3)
    noise = np.random.default_rng(3).standard_normal(o.noise_len(T)).astype(np.float32)
    pcm = eng.decode(codes, T, noise)
    assert pcm.shape == (T * 512,)
    assert np.abs(pcm - o.decode(codes, T, noise)).max() < 2e-4
    assert np.array_equal(pcm, eng.decode(codes, T, noise))
    eng.close()

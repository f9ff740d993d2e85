"""GPU parity: the T5 voice-prompt encoder (tts_hip_t5_encode) against the oracle (oracle/tts_oracle.c orc_t5_encode,
pinned to a float64 torch restatement in tests/golden/tiny_t5.npz by tests/test_oracle_cpu.py).
Tolerances as for the decoder: 2e-4 of max|oracle| with F32 weights, 2e-3 with F16; the integer path is bounded by the
Q8_0 activation-flip size (see tests/test_gpu_parler.py)."""
import os

import numpy as np
import pytest

import oracle as orc
from tts_cpp_amd import gguf, hip, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "tiny_t5.npz")


def relerr(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize("wtype,tol", [(gguf.F32, 2e-4), (gguf.F16, 2e-3), (gguf.Q8_0, 3e-2), (gguf.Q5_0, 3e-2)])
def test_t5_encoder_matches_oracle(wtype, tol):
    model = synth.build_t5(synth.t5_tiny(weight_type=wtype))
    eng = hip.T5Engine(model.cfg)
    eng.load(model)
    o = orc.T5Oracle(model, act_mode=1, gelu_mode=1)
    g = np.load(GOLD)
    for ids in (g["ids"], g["ids"][:1], g["ids"][:17], np.arange(3, 3 + model.cfg.ctx, dtype=np.uint32) % model.cfg.vocab):
        out = eng.encode(ids)
        ref = o.encode(ids)
        assert out.shape == ref.shape == (len(ids), model.cfg.output_size)
        assert relerr(out, ref) < tol, len(ids)
    if wtype == gguf.F32:
        assert relerr(eng.encode(g["ids"]), g["out"]) < 5e-4   # golden is tanh-GELU in fp64; the engine uses ggml's fp16 GELU table
    with pytest.raises(hip.HipError):
        eng.encode(np.zeros(model.cfg.ctx + 1, dtype=np.uint32))      # beyond t5encoder.context_length
    with pytest.raises(hip.HipError):
        eng.encode(np.array([model.cfg.vocab], dtype=np.uint32))      # id outside the vocabulary
    eng.close()


def test_t5_down_projection_and_flan_large_shapes():
    """down_proj + bias (present when the encoder's width differs from the decoder's, py-gguf t5 encoder :72-74), and
    one layer of the real flan-t5-large shapes (d_model 1024, d_ff 2816 = 11 K-slices of 256, 16 heads)."""
    g = np.load(GOLD)
    m1 = synth.build_t5(synth.t5_tiny(output_size=192, seed=0x76))
    e1 = hip.T5Engine(m1.cfg)
    e1.load(m1)
    out = e1.encode(g["ids"][:23])
    assert out.shape == (23, 192)
    assert relerr(out, orc.T5Oracle(m1).encode(g["ids"][:23])) < 2e-4
    assert relerr(out, g["out_proj"]) < 5e-4
    e1.close()
    m2 = synth.build_t5(synth.t5_flan_large(layers=1, vocab=512, ctx=300, weight_type=gguf.F16))
    e2 = hip.T5Engine(m2.cfg)
    e2.load(m2)
    ids = np.random.default_rng(5).integers(3, 512, 270).astype(np.uint32)   # > 256 rows: two GEMM row chunks
    assert relerr(e2.encode(ids), orc.T5Oracle(m2).encode(ids)) < 2e-3
    e2.close()

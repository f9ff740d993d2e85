"""GPU parity: the streaming 1..4-row GEMV kernels (csrc/gemv_kernels.h, TTS_HIP_GEMV_ROWS) against the oracle through the Orpheus
and Dia steps.  Measured (profiles/r02/first_call_*.log): Orpheus-3B Q4_0 3.84 -> 2.98 ms/step (2.80 with the Q4_0 codes read
natively and the step captured), so they are the default for Orpheus contexts; Dia 3.71 vs 3.75 (no gain: stays off there).  The
other Orpheus tests run on the defaults; `test_orpheus_steps_through_the_lockstep_workgroups` here keeps the knobs-off path covered."""
import os

import numpy as np
import pytest

import oracle as orc
from tts_cpp_amd import gguf, hip, synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(__file__)


def relerr(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.fixture(autouse=True)
def _knob():
    os.environ["TTS_HIP_GEMV_ROWS"] = "1"     # read when a context is created
    os.environ["TTS_HIP_Q4_NATIVE"] = "0"     # the int8 streaming kernels; the 4-bit test below switches it on
    yield
    del os.environ["TTS_HIP_GEMV_ROWS"]
    del os.environ["TTS_HIP_Q4_NATIVE"]


@pytest.mark.parametrize("wtype,tol", [(gguf.F32, 2e-4), (gguf.F16, 2e-3), (gguf.Q4_0, 3e-2), (gguf.Q8_0, 3e-2)])
def test_orpheus_steps_through_the_row_kernels(wtype, tol):
    model = synth.build_orpheus(synth.orpheus_tiny(weight_type=wtype))
    eng = hip.OrpheusEngine(model.cfg)
    eng.load(model)
    o = orc.OrpheusOracle(model, act_mode=1)
    g = np.load(os.path.join(HERE, "golden", "tiny_orpheus.npz"))
    prompt = g["prompt"]
    ref = o.decode(prompt[:3], 0)                    # 3 rows: the row kernels; the 9-token prompt would take the MFMA path
    lg, _ = eng.decode(prompt[:3], 0)
    assert relerr(lg, ref) < tol
    pos = 3
    for step in range(6):
        t = int(ref.argmax())
        lg, _ = eng.decode([t], pos)
        ref = o.decode([t], pos)
        assert relerr(lg, ref) < tol, step
        pos += 1
    eng.close()


@pytest.mark.parametrize("wtype,tol", [(gguf.F32, 2e-4), (gguf.F16, 2e-3), (gguf.Q8_0, 3e-2)])
def test_dia_steps_through_the_row_kernels(wtype, tol):
    kw = dict(enc_hidden=256) if wtype == gguf.Q8_0 else {}
    model = synth.build_dia(synth.dia_tiny(weight_type=wtype, **kw))
    eng = hip.DiaEngine(model.cfg)
    eng.load(model)
    o = orc.DiaOracle(model, act_mode=1)
    g = np.load(os.path.join(HERE, "golden", "tiny_dia.npz"))
    n = int(g["sentence_len"])
    eng.encode(g["tokens"], n)
    o.encode(g["tokens"], n)
    for s_ in range(len(g["ids"])):
        lg, raw = eng.step(g["ids"][s_], s_, want_raw=True)
        ref, ref_raw = o.step(g["ids"][s_], s_, want_raw=True)
        assert relerr(raw, ref_raw) < tol, s_
        assert relerr(lg, ref) < 4 * tol, s_
    eng.close()


def test_wide_feed_forward_without_split_k():
    """ffn 8192: the MFMA path splits K in two slabs folded by the next rms norm; the row kernels walk all of K — same numbers"""
    cfg = synth.orpheus_tiny(ffn=8192, layers=1)
    model = synth.build_orpheus(cfg)
    o = orc.OrpheusOracle(model, act_mode=1)
    eng = hip.OrpheusEngine(cfg)
    eng.load(model)
    lg, _ = eng.decode([5, 7], 0)
    assert relerr(lg, o.decode([5, 7], 0)) < 2e-4
    eng.close()


def test_orpheus_q4_0_matrices_read_as_4_bit_codes():
    """TTS_HIP_Q4_NATIVE=1: the streaming kernel reads the Q4_0 codes (half the bytes of the int8 expansion); sum (n - 8) x is the same
    integer, so the logits equal the int8 streaming path bit for bit"""
    model = synth.build_orpheus(synth.orpheus_tiny(weight_type=gguf.Q4_0))
    g = np.load(os.path.join(HERE, "golden", "tiny_orpheus.npz"))
    base = hip.OrpheusEngine(model.cfg)            # TTS_HIP_GEMV_ROWS=1 from the fixture: int8 streaming kernels
    base.load(model)
    os.environ["TTS_HIP_Q4_NATIVE"] = "1"
    try:
        eng = hip.OrpheusEngine(model.cfg)
    finally:
        os.environ["TTS_HIP_Q4_NATIVE"] = "0"
    eng.load(model)
    o = orc.OrpheusOracle(model, act_mode=1)
    ref = o.decode(g["prompt"][:3], 0)
    a, _ = base.decode(g["prompt"][:3], 0)
    b, _ = eng.decode(g["prompt"][:3], 0)
    assert np.array_equal(a, b) and relerr(b, ref) < 3e-2
    for pos in range(3, 8):
        t = int(ref.argmax())
        a, _ = base.decode([t], pos)
        b, _ = eng.decode([t], pos)
        ref = o.decode([t], pos)
        assert np.array_equal(a, b) and relerr(b, ref) < 3e-2
    base.close(); eng.close()


@pytest.mark.parametrize("wtype,tol", [(gguf.F16, 2e-3), (gguf.Q4_0, 3e-2)])
def test_orpheus_steps_through_the_lockstep_workgroups(wtype, tol):
    """TTS_HIP_GEMV_ROWS=0 / TTS_HIP_LLAMA_GRAPH=0: single rows through the 16-feature MFMA workgroups (round 1's only path)"""
    os.environ["TTS_HIP_GEMV_ROWS"] = "0"
    os.environ["TTS_HIP_LLAMA_GRAPH"] = "0"
    try:
        model = synth.build_orpheus(synth.orpheus_tiny(weight_type=wtype))
        eng = hip.OrpheusEngine(model.cfg)
    finally:
        os.environ["TTS_HIP_GEMV_ROWS"] = "1"
        del os.environ["TTS_HIP_LLAMA_GRAPH"]
    eng.load(model)
    o = orc.OrpheusOracle(model, act_mode=1)
    g = np.load(os.path.join(HERE, "golden", "tiny_orpheus.npz"))
    ref = o.decode(g["prompt"][:3], 0)
    lg, _ = eng.decode(g["prompt"][:3], 0)
    assert relerr(lg, ref) < tol
    for pos in range(3, 6):
        t = int(ref.argmax())
        lg, _ = eng.decode([t], pos)
        ref = o.decode([t], pos)
        assert relerr(lg, ref) < tol
    eng.close()

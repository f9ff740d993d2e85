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


def test_orpheus_3b_layer_shapes():
    """one layer of canopylabs/orpheus-3b's shapes: hidden 3072 (12 K-slices), 24 q heads / 8 kv heads x 128, ffn 8192
    (down_proj in two 4096-column slabs folded by the next rms norm), a vocabulary that is not a multiple of 16."""
    model = synth.build_orpheus(synth.orpheus_3b(layers=1, vocab=5001, ctx=64, weight_type=gguf.F16))
    eng = hip.OrpheusEngine(model.cfg)
    eng.load(model)
    o = orc.OrpheusOracle(model)
    ids = np.random.default_rng(1).integers(0, 5001, 20).astype(np.uint32)
    lg, tok = eng.decode(ids, 0)
    ref = o.decode(ids, 0)
    assert lg.shape == (5001,) and relerr(lg, ref) < 2e-3
    lg2, _ = eng.decode([int(ref.argmax())], 20)
    assert relerr(lg2, o.decode([int(ref.argmax())], 20)) < 2e-3
    eng.close()

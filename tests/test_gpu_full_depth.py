"""GPU parity at FULL DEPTH ("Dia/Orpheus are oracle-checked at 1-2 layers of the real widths, never at full depth").

canopylabs/orpheus-3b (28 layers, hidden 3072, 24:8 heads x 128, ffn 8192, 156 940 logits; every matrix Q4_0 = BASELINE config 4) and
nari-labs/Dia-1.6B (encoder 12 x 1024, decoder 18 x 2048, 9 x 1028 logits; fp16 matrices = BASELINE config 3) through every layer against the oracle
(src/models/orpheus/model.cpp:186-325, src/models/dia/model.cpp:320-660 restated in oracle/tts_oracle.c).  The models are `pooled` synthetic ones
(synth._Pool: matrices cut out of one buffer of normals, quantised matrices minted as random blocks) so that minting them costs seconds instead of minutes.

What depth adds over the one-layer cases: rounding differences of one layer are the next layer's input, and with Q4_0 matrices every row is re-quantised to
Q8_0 blocks 7 times per layer — a value on a rounding boundary flips a code by one step (1/127 of its block's largest value).  The bars below are the
one-layer bars (2e-3 of the largest logit for fp16 matrices, 3e-2 for the integer path); the measured distances are printed (-s) and recorded in DESIGN.md 3."""
import numpy as np
import pytest

import oracle as orc
from tts_cpp_amd import gguf, hip, synth

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def test_orpheus_3b_q4_0_all_28_layers():
    """a 12-token prompt (the tiled int8-MFMA GEMM path), two one-token steps (the streaming Q4_0 kernels: fused rms norm, q/k/v + rope + append, gate|up + silu,
    down) reading the cache rows the prompt left, then a lock-step step of 8 utterances (qgemv_stream_kernel incl. the 156 940-row head): logits of every call
    against the oracle's for the same history, and the arg-max inside the oracle's margin."""
    model = synth.build_orpheus(synth.orpheus_3b(ctx=64, weight_type=gguf.Q4_0), pooled=True)
    cfg = model.cfg
    assert cfg.layers == 28 and cfg.vocab == 156940
    o = orc.OrpheusOracle(model, act_mode=1)
    rng = np.random.default_rng(1)
    ids = rng.integers(0, cfg.vocab, 12).astype(np.uint32)
    eng = hip.OrpheusEngine(cfg, max_seqs=8)
    eng.load(model)
    errs = []
    lg, tok = eng.decode(ids, 0)
    ref = o.decode(ids, 0)
    errs.append(relerr(lg, ref))
    assert ref[tok] >= ref.max() - 2 * 3e-2 * np.abs(ref).max()
    pos = len(ids)
    for _ in range(2):
        t = int(ref.argmax())
        lg, tok = eng.decode([t], pos)
        ref = o.decode([t], pos)
        errs.append(relerr(lg, ref))
        assert ref[tok] >= ref.max() - 2 * 3e-2 * np.abs(ref).max()
        pos += 1
    # lock-step: 8 utterances with prompts of 2..6 tokens in their own cache slots, one step of 8 rows
    prompts = [rng.integers(0, cfg.vocab, 2 + (u % 5)).astype(np.uint32) for u in range(8)]
    first = eng.generate_batch(prompts, 1, stop_id=cfg.vocab + 5)
    lgb, tokb = eng.step_batch(list(range(8)), [int(f[0]) for f in first], [len(q) for q in prompts])
    assert tokb.tolist() == [int(l.argmax()) for l in lgb]
    for u in (0, 3, 7):
        o.reset()
        o.decode(prompts[u], 0)
        refu = o.decode([int(first[u][0])], len(prompts[u]))
        errs.append(relerr(lgb[u], refu))
    eng.close()
    print("orpheus-3b Q4_0, 28 layers: logits rel. distance to the oracle (prompt, step, step, lock-step rows 0 / 3 / 7):", " ".join(f"{e:.2e}" for e in errs))
    assert max(errs) < 3e-2, errs


def test_dia_1_6b_f16_all_layers():
    """encoder (12 layers over 2 x 256 text positions: conditioned + unconditioned stream), cross K/V of 18 decoder layers, three guided decoder steps of two
    utterances in lock-step through all 18 layers: raw and guided logits of every step against per-utterance oracles."""
    model = synth.build_dia(synth.dia_1_6b(max_gen=32, max_ctx=256, weight_type=gguf.F16), pooled=True)
    cfg = model.cfg
    assert cfg.enc_layers == 12 and cfg.dec_layers == 18
    eng = hip.DiaEngine(cfg, max_utterances=2)
    eng.load(model)
    texts = ["[S1] The birch canoe slid on the smooth planks.", "[S2] Glue the sheet to the dark blue background."]
    oracles = []
    for u, t in enumerate(texts):
        toks, n = orc.dia_tokenize(t, cfg.max_ctx)
        eng.encode_slot(u, toks, n)
        ob = orc.DiaOracle(model, act_mode=1)
        ob.encode(toks, n)
        oracles.append(ob)
    ids = np.full((2, cfg.n_out), cfg.bos, dtype=np.uint32)
    rng = np.random.default_rng(5)
    raw_errs, errs = [], []
    for step in range(3):
        lg, raw = eng.step_batch(ids, np.full(2, step, dtype=np.uint32), want_raw=True)
        for u in range(2):
            ref, ref_raw = oracles[u].step(ids[u], step, want_raw=True)
            raw_errs.append(relerr(raw[u], ref_raw))
            errs.append(relerr(lg[u], ref))
            ids[u] = rng.integers(0, cfg.audio_vocab, cfg.n_out)
    eng.close()
    print("dia-1.6b fp16, 12 + 18 layers: raw logits", " ".join(f"{e:.2e}" for e in raw_errs), "| guided", " ".join(f"{e:.2e}" for e in errs))
    assert max(raw_errs) < 2e-3, raw_errs
    assert max(errs) < 8e-3, errs

#!/usr/bin/env python3
"""One-shot check of the bf16x3 conv experiment (TTS_HIP_DAC_BF16X3=1) against the oracle at the DAC-44k dims:
per-stage relative error with the experiment on and off, then the PCM error."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as orc
import tts_cpp_amd  # noqa: F401
from tts_cpp_amd import gguf, hip, synth

def relerr(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))

model = synth.build(synth.parler_mini(layers=1, prompt_vocab=64, ctx=64))
cfg = model.cfg
codes = np.random.default_rng(2).integers(0, cfg.cb_size, (3, cfg.n_out)).astype(np.uint32)
o = orc.DacOracle(model)
ref = o.decode(codes)
stages = [o.decode(codes, stage=s)[1] for s in range(2 + len(cfg.strides))]
for knob in os.environ.get("B3_KNOBS", "1 0").split():   # B3_KNOBS="2" checks the 96-channel tile as well
    os.environ["TTS_HIP_DAC_BF16X3"] = knob
    eng = hip.HipEngine(cfg, flags=hip.FLAG_NO_PARLER)
    eng.load(model)
    eng.set_debug(True)
    pcm = eng.dac_decode(codes)
    errs = [relerr(eng.debug_read(f"dac:{s}", st.size).reshape(st.shape), st) for s, st in enumerate(stages)]
    print(f"BF16X3={knob} pcm max abs err {np.abs(pcm - ref).max():.3e}  stage relerr " + " ".join(f"{e:.2e}" for e in errs), flush=True)
    eng.close()

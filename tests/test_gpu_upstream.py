"""GPU: the HIP path, through the C ABI engines, straight against the upstream goldens (tests/golden/upstream_*.npz: Hugging Face models in
float64, weights exported under the reference converters' naming rules — tests/golden/make_upstream_golden.py).  tests/test_upstream_golden.py
pins the oracle to the same fixtures on the CPU; here no oracle stands between the device and the upstream implementation.
Tolerances: the suite's fp32 bars (F32 weights: 2e-4 of max|ref| on logits / hidden states, PCM 1e-4 absolute, codec stages 1e-4)."""
import os
import types

import numpy as np
import pytest

from tts_cpp_amd import gguf, hip, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, name))
    tensors = [gguf.Tensor.from_array(k[2:], np.ascontiguousarray(z[k], dtype=np.float32), gguf.F32) for k in z.files if k.startswith("t:")]
    return z, tensors


def model_of(cfg, tensors):
    return types.SimpleNamespace(cfg=cfg, tensors=tensors, by_name={t.name: t for t in tensors})


def rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize("fixture", ["upstream_dac.npz", "upstream_dac_b3.npz"])
def test_dac_decoder_against_transformers_dac(fixture):
    """upstream_dac_b3: 192 -> 96 -> 48 channels, 300 frames: the 96-channel residual units run as resunit_t7_kernel (bf16 x 3, one launch per
    unit, three position tiles) and the stride-2 transposed conv as convt_b3_kernel — compared with transformers' DacModel, no oracle between."""
    z, tensors = load(fixture)
    latent, cb_dim, cb_size, c0, n_cb = (int(x) for x in z["cfg"])
    strides = tuple(int(s) for s in z["strides"])
    cfg = synth.tiny(latent=latent, cb_dim=cb_dim, cb_size=cb_size, c0=c0, n_out=n_cb, strides=strides, weight_type=gguf.F32, max_gen=max(96, len(z["codes"]) + 8))
    eng = hip.HipEngine(cfg, flags=hip.FLAG_NO_PARLER)
    eng.load(model_of(cfg, [t for t in tensors if t.name.startswith("audio_encoder.")]))
    eng.set_debug(True)
    pcm = eng.dac_decode(z["codes"])
    assert np.abs(pcm - z["pcm"]).max() < 1e-4
    for st in range(2 + len(strides)):
        ref = z[f"stage{st}"]
        act = eng.debug_read(f"dac:{st}", ref.size).reshape(ref.shape)
        assert rel(act, ref) < 1e-4, f"stage {st}"
    eng.close()


def test_parler_decoder_against_transformers_musicgen():
    z, tensors = load("upstream_parler.npz")
    H, L, HEADS, F, V, NCB, ENC, PV = (int(x) for x in z["cfg"])
    ctx = next(t for t in tensors if t.name == "decoder.positional_embed").ne[1]
    cfg = synth.tiny(hidden=H, layers=L, heads=HEADS, ffn=F, out_vocab=V, audio_vocab=64, n_out=NCB, ctx=ctx, enc_len=ENC, prompt_vocab=PV, weight_type=gguf.F32)
    eng = hip.HipEngine(cfg, flags=hip.FLAG_NO_DAC, gelu_mode=0)   # exact tanh-GELU (the default, 1, goes through ggml's fp16-indexed table)
    eng.load(model_of(cfg, [t for t in tensors if t.name.startswith("decoder.")]))
    prompt, audio = z["prompt"], z["audio"]
    eng.prefill(0, prompt)
    n = prompt.size
    for t in range(audio.shape[0]):
        lg = eng.step(audio[t][None], [n + t])[0]
        assert rel(lg, z["logits"][:, n + t, :]) < 2e-4, f"step {t}"
    eng.close()


def test_orpheus_decoder_against_transformers_llama():
    z, tensors = load("upstream_orpheus.npz")
    H, L, NH, NKV, HD, F, V = (int(x) for x in z["cfg"])
    ids = z["ids"]
    cfg = synth.OrpheusConfig(hidden=H, layers=L, heads=NH, kv_heads=NKV, head_dim=HD, ffn=F, vocab=V, ctx=ids.size + 8, weight_type=gguf.F32)
    eng = hip.OrpheusEngine(cfg)
    eng.load(model_of(cfg, tensors))
    lg, tok = eng.decode(ids, 0)                    # 2304 positions: the llama3 rope factors are visible in the logits
    assert rel(lg, z["logits_last"]) < 2e-4
    assert tok == int(np.argmax(z["logits_last"]))
    eng.close()
    eng = hip.OrpheusEngine(cfg)
    eng.load(model_of(cfg, tensors))
    eng.decode(ids[:40], 0)
    lg, _ = eng.decode(ids[40:41], 40)              # the cache path
    assert rel(lg, z["logits_at_40"]) < 2e-4
    eng.close()


def test_t5_encoder_against_transformers_t5():
    z, tensors = load("upstream_t5.npz")
    V, H, DKV, F, L, NH, OUT = (int(x) for x in z["cfg"])
    cfg = synth.T5Config(hidden=H, layers=L, heads=NH, ffn=F, vocab=V, ctx=64, buckets=32, output_size=OUT, weight_type=gguf.F32)
    eng = hip.T5Engine(cfg, gelu_mode=0)
    eng.load(model_of(cfg, tensors))
    assert rel(eng.encode(z["ids7"]), z["out7"]) < 2e-4
    assert rel(eng.encode(z["ids24"]), z["out24_refbuckets"]) < 2e-4   # the reference's bucket rule at 24 tokens
    assert rel(eng.encode(z["ids24"]), z["out24_hf"]) > 1e-2
    eng.close()

"""GGUF container round trip, synthetic model structure, and the C-ABI export check (no compute)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from tts_cpp_amd import gguf, hip, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gguf_roundtrip(tmp_path):
    m = synth.build(synth.tiny(weight_type=gguf.Q5_0))
    p = m.write_gguf(str(tmp_path / "tiny.gguf"))
    r = gguf.Reader(p)
    assert r.kv["general.architecture"] == "parler-tts"
    assert r.kv["parler-tts.decoder.encode_length"] == m.cfg.enc_len
    assert r.kv["parler-tts.decoder.attention.head_count"] == m.cfg.heads
    assert r.kv["dac.dac_layer_stride_0"] == m.cfg.strides[0]
    assert list(r.kv["tokenizer.ggml.tokens"]) == m.vocab
    assert np.array_equal(r.kv["tokenizer.ggml.scores"], m.scores)
    assert r.order == [t.name for t in m.tensors]
    assert r.data_offset % 32 == 0
    for t in m.tensors:
        rt = r.tensors[t.name]
        assert rt.type == t.type and rt.ne == t.ne
        assert bytes(rt.raw()) == bytes(t.raw())
    # quantisation allow-list (examples/quantize/quantize_impl.cpp:51-67)
    assert r.tensors["decoder.layers.0.fc1.weight"].type == gguf.Q5_0
    assert r.tensors["decoder.layers.0.final_layer_norm.weight"].type == gguf.F32
    assert r.tensors["decoder.positional_embed"].type == gguf.F32
    assert r.tensors["decoder.text_encoding"].type == gguf.F32
    assert r.tensors["audio_encoder.initial.weight"].type == gguf.F32


def test_synth_shapes_match_reference_layout():
    cfg = synth.parler_mini()
    assert (cfg.hidden, cfg.layers, cfg.heads, cfg.out_vocab, cfg.n_out, cfg.ctx, cfg.max_gen) == (1024, 24, 16, 1088, 9, 4096, 2580)
    assert cfg.hop == 512 and cfg.paddings == (4, 4, 2, 1)
    m = synth.build(synth.tiny())
    t = m.by_name
    c = m.cfg
    assert t["decoder.layers.1.fc1.weight"].ne == [c.hidden, c.ffn]              # ggml ne=[in,out]
    assert t["decoder.embed_tokens.0.weight"].ne == [c.hidden, c.out_vocab + 1]  # V+1 rows
    assert t["decoder.lm_heads.0.weight.head"].ne == [c.hidden, c.out_vocab]
    assert t["audio_encoder.decoder_block.1.final.weight"].ne == [2 * c.strides[0], c.c0 // 2, c.c0]  # [K,Cout,Cin]
    assert t["audio_encoder.decoder_block.1.residual_unit.2.res.initial.weight"].ne == [7, c.c0 // 2, c.c0 // 2]
    assert t["audio_encoder.quantizers.0.codebook.weight"].ne == [c.cb_dim, c.cb_size]
    assert t["audio_encoder.decoder_block.2.final.alpha"].ne == [1, c.c0 // 2, 1]


@pytest.mark.parametrize("header", ["tts_hip.h", "tts_c.h"])
@pytest.mark.parametrize("compiler,lang", [("gcc", "c"), ("g++", "c++")])
def test_public_headers_compile_as_c_and_cpp(header, compiler, lang):
    """The boundary headers are a C ABI: both must parse as plain C (and as C++) on their own, warnings as errors."""
    import subprocess
    r = subprocess.run([compiler, "-fsyntax-only", "-x", lang, "-Wall", "-Werror", os.path.join(ROOT, "include", header)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_c_abi_exports_every_declared_symbol():
    """include/tts_hip.h is the boundary: the built library must export each function it declares."""
    hdr = open(os.path.join(ROOT, "include", "tts_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(tts_hip_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 20
    assert sorted(hip.EXPORTS) == declared, "hip.py binding list and header out of sync"
    assert os.path.exists(hip.lib_path()), "libtts_hip.so not built (run __graft_entry__.build())"
    L = C.CDLL(hip.lib_path())
    for name in declared:
        assert hasattr(L, name), f"{name} not exported"
    assert hip.load_lib().tts_hip_version().startswith(b"tts_hip")
    assert C.sizeof(hip.Desc) == 4 * (1 + 8 + 1 + 16 + 1 + 4 + 1)


def test_product_path_fails_loudly_without_gpu(have_gpu):
    if have_gpu:
        pytest.skip("a GPU is present")
    with pytest.raises(hip.HipError) as e:
        hip.HipEngine(synth.tiny())
    assert "no HIP device" in str(e.value) or "no CPU fallback" in str(e.value)


def test_product_package_does_not_touch_the_oracle():
    pkg = os.path.join(ROOT, "tts.cpp_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp", ".c")) or f == "Makefile":
                src = open(os.path.join(base, f), errors="replace").read()
                assert "liboracle" not in src and "tts_oracle.h" not in src and "import oracle" not in src, os.path.join(base, f)


def test_undelay_matches_oracle_adjust_output_tokens():
    import oracle as orc
    from tts_cpp_amd.pattern import undelay
    rng = np.random.default_rng(0)
    for trial in range(30):
        steps, n_out = int(rng.integers(1, 40)), int(rng.integers(1, 10))
        t = rng.integers(0, 70, (steps, n_out)).astype(np.uint32)
        flat = t.reshape(-1).copy()
        out = np.empty_like(flat)
        n = orc.lib().orc_parler_adjust_output_tokens(orc.u32p(flat), flat.size, n_out, 64, 64, orc.u32p(out))
        assert np.array_equal(out[:n].reshape(-1, n_out), undelay(t, 64)), trial

"""The quantize tool (host/quantize.cpp behind tts_c_quantize_*; reference: examples/quantize/quantize_impl.cpp).

Bit-exact checks, all on the CPU: the C++ row quantisers against the two independent restatements the GPU parity tests
already rely on (oracle/orc_quantize in C, synth.quantize in numpy); the fp32->fp16 conversion against numpy over every
half value and every rounding midpoint; whole-file rewrites against models minted directly in the quantised type (the
very files the -m gpu tests decode), for each allow-list flag."""
import os
import subprocess

import numpy as np
import pytest

import oracle as orc
from tts_cpp_amd import gguf, runner, synth

QTYPES = [gguf.Q4_0, gguf.Q5_0, gguf.Q8_0]


def _rows(seed, n_rows=37, n=256):
    rng = np.random.default_rng(seed)
    a = (rng.standard_normal((n_rows, n)) * rng.choice([1e-3, 0.02, 1.0, 300.0], size=(n_rows, 1))).astype(np.float32)
    a[3, :32] = 0.0                      # an all-zero block: d == 0, id == 0
    a[4, 32:64] = np.float32(-0.5)       # constant block, the maximum is negative
    a[5, 64:96] = np.float32(0.25)       # constant block, the maximum is positive (clamped top code)
    a[6, :32] = np.linspace(-1, 1, 32)   # codes land on .5 boundaries
    return a


@pytest.mark.parametrize("qtype", QTYPES)
def test_row_quantisers_bit_exact(qtype):
    a = _rows(qtype)
    got = runner.quantize_rows(qtype, a)
    assert np.array_equal(got, orc.quantize(qtype, a))
    assert np.array_equal(got, synth.quantize(a, qtype))
    assert np.array_equal(got, runner.quantize_rows(qtype, a, n_threads=5))   # the row split cannot change the bytes


def test_f16_conversion_every_half_value_and_midpoint():
    bits = np.arange(0x7C00, dtype=np.uint16)                  # every finite non-negative half
    h = bits.view(np.float16).astype(np.float64)
    mid = (h[:-1] + h[1:]) / 2                                 # exact in fp32 (12 significant bits)
    near = np.concatenate([h, mid, np.nextafter(mid.astype(np.float32), np.float32(0)), np.nextafter(mid.astype(np.float32), np.float32(1e9)),
                           [65504.0, 65519.99, 65520.0, 65536.0, 1e9, 2.0 ** -25, np.nextafter(np.float32(2.0 ** -25), np.float32(1)), 1e-10, np.inf]])
    x = np.concatenate([near, -near]).astype(np.float32)
    x = np.concatenate([x, np.zeros((-x.size) % 32, dtype=np.float32)])
    got = runner.quantize_rows(gguf.F16, x.reshape(1, -1)).view(np.uint16)
    with np.errstate(over="ignore"):
        want = x.astype(np.float16).view(np.uint16)
    assert np.array_equal(got, want)
    rng = np.random.default_rng(5)
    r = (rng.standard_normal(1 << 16) * np.exp(rng.uniform(-20, 12, 1 << 16))).astype(np.float32)
    with np.errstate(over="ignore"):
        assert np.array_equal(runner.quantize_rows(gguf.F16, r.reshape(64, -1)).view(np.uint16).reshape(-1), r.astype(np.float16).view(np.uint16))


def _parler_rule(name, heads, text, kv):
    """parler_is_quanitizable restated from the tool's documentation of its flags (examples/quantize/README.md:22-31)."""
    if name.startswith("audio_encoder") or name.endswith(("norm.weight", "norm.bias", "text_encoding", "positional_embed")):
        return False
    if not heads and name.endswith("weight.head"):
        return False
    if not text and name.endswith("embed_prompts"):
        return False
    if not kv and name.endswith(("encoder_attn.k_proj.weight", "encoder_attn.v_proj.weight")):
        return False
    return True


@pytest.fixture(scope="module")
def f32_model(tmp_path_factory):
    m = synth.build(synth.tiny(weight_type=gguf.F32))
    return m, m.write_gguf(str(tmp_path_factory.mktemp("q") / "tiny_f32.gguf"))


@pytest.mark.parametrize("flags", [dict(), dict(output_heads=True), dict(text_embeddings=True), dict(cross_attn_kv=True),
                                   dict(output_heads=True, text_embeddings=True, cross_attn_kv=True, dac_f16=True)])
def test_parler_file_allow_list_and_bytes(f32_model, tmp_path, flags):
    m, src = f32_model
    dst = str(tmp_path / "q.gguf")
    runner.quantize_gguf(src, dst, gguf.Q5_0, n_threads=3, **flags)
    a, b = gguf.Reader(src), gguf.Reader(dst)
    assert b.order == a.order and b.version == 3 and b.data_offset % 32 == 0
    assert list(b.kv)[:len(a.kv)] == list(a.kv)                # records kept in order, the two new keys appended
    assert list(b.kv)[len(a.kv):] == ["general.quantization_version", "general.quantization_type"]
    assert b.kv["general.quantization_version"] == 2 and b.kv["general.quantization_type"] == gguf.Q5_0
    for k in a.kv:
        assert np.array_equal(np.asarray(a.kv[k]), np.asarray(b.kv[k])) and a.kv_types[k] == b.kv_types[k], k
    n_q = 0
    for name in a.order:
        s, d = a.tensors[name], b.tensors[name]
        assert d.ne[:len(s.ne)] == s.ne[:len(d.ne)] and int(np.prod(d.ne)) == int(np.prod(s.ne))
        if _parler_rule(name, flags.get("output_heads"), flags.get("text_embeddings"), flags.get("cross_attn_kv")):
            assert d.type == gguf.Q5_0, name
            assert bytes(d.raw()) == synth.quantize(s.to_f32(), gguf.Q5_0).tobytes(), name
            n_q += 1
        elif flags.get("dac_f16") and name.startswith("audio_encoder") and not name.endswith("alpha"):
            assert d.type == gguf.F16 and bytes(d.raw()) == s.to_f32().astype("<f2").tobytes(), name
        else:
            assert d.type == s.type and bytes(d.raw()) == bytes(s.raw()), name
    assert n_q >= 6 * m.cfg.layers


def test_tool_output_is_the_model_the_gpu_tests_decode(f32_model, tmp_path):
    """All flags on == the synthetic generator minting the same weights directly in the quantised type: those files are
    what tests/test_gpu_parler.py and test_gpu_runner.py run through the HIP path."""
    _, src = f32_model
    for qtype in QTYPES:
        dst = str(tmp_path / f"q{qtype}.gguf")
        runner.quantize_gguf(src, dst, qtype, output_heads=True, text_embeddings=True, cross_attn_kv=True, dac_f16=True)
        want = synth.build(synth.tiny(weight_type=qtype, dac_f16=True))
        got = gguf.Reader(dst)
        for t in want.tensors:
            g = got.tensors[t.name]
            assert g.type == t.type and bytes(g.raw()) == bytes(t.raw()), t.name


def test_quantised_input_is_refused_and_keys_are_replaced_in_place(f32_model, tmp_path):
    _, src = f32_model
    once, twice = str(tmp_path / "a.gguf"), str(tmp_path / "b.gguf")
    runner.quantize_gguf(src, once, gguf.Q8_0)
    with pytest.raises(runner.RunnerError, match="32bit floats"):      # quantize_impl.cpp:248-253
        runner.quantize_gguf(once, twice, gguf.Q4_0)
    # an F32 file that already carries the two keys: values replaced where they stand, no duplicates
    r = gguf.Reader(src)
    kv = [(k, r.kv_types[k], r.kv[k] if r.kv_types[k] != gguf.T_ARR else None) for k in r.kv]
    kv = [e for e in kv if e[2] is not None]
    kv.insert(1, ("general.quantization_type", gguf.T_U32, 99))
    kv.insert(3, ("general.quantization_version", gguf.T_U32, 1))
    arch = [("general.architecture", gguf.T_STR, "parler-tts")] if "general.architecture" not in r.kv else []
    small = [gguf.Tensor.from_array("decoder.layers.0.fc1.weight", np.ones((4, 64), dtype=np.float32)),
             gguf.Tensor.from_array("decoder.layer_norm.weight", np.ones(64, dtype=np.float32))]
    p = str(tmp_path / "keys.gguf")
    gguf.write(p, arch + kv, small)
    runner.quantize_gguf(p, twice, gguf.Q4_0)
    out = gguf.Reader(twice)
    assert list(out.kv) == [e[0] for e in arch + kv]
    assert out.kv["general.quantization_type"] == gguf.Q4_0 and out.kv["general.quantization_version"] == 2
    assert out.tensors["decoder.layers.0.fc1.weight"].type == gguf.Q4_0 and out.tensors["decoder.layer_norm.weight"].type == gguf.F32


def test_file_without_architecture_key_is_parler(tmp_path):
    """quantize_impl.cpp:188-192"""
    p, q = str(tmp_path / "noarch.gguf"), str(tmp_path / "noarch_q.gguf")
    gguf.write(p, [("x.y", gguf.T_U32, 1)], [gguf.Tensor.from_array("decoder.layers.0.fc2.weight", np.arange(128, dtype=np.float32).reshape(2, 64)),
                                             gguf.Tensor.from_array("audio_encoder.initial.weight", np.ones((2, 2, 7), dtype=np.float32))])
    runner.quantize_gguf(p, q, gguf.Q8_0)
    out = gguf.Reader(q)
    assert out.tensors["decoder.layers.0.fc2.weight"].type == gguf.Q8_0 and out.tensors["audio_encoder.initial.weight"].type == gguf.F32


def test_rows_that_do_not_split_into_blocks_are_refused(tmp_path):
    p, q = str(tmp_path / "odd.gguf"), str(tmp_path / "odd_q.gguf")
    gguf.write(p, [("general.architecture", gguf.T_STR, "parler-tts")], [gguf.Tensor.from_array("decoder.layers.0.fc2.weight", np.ones((2, 48), dtype=np.float32))])
    with pytest.raises(runner.RunnerError, match="block size"):
        runner.quantize_gguf(p, q, gguf.Q4_0)
    runner.quantize_gguf(p, q, gguf.F16)                        # fp16 has no block constraint
    assert gguf.Reader(q).tensors["decoder.layers.0.fc2.weight"].type == gguf.F16


def test_unknown_architecture_is_refused(tmp_path):
    p = str(tmp_path / "x.gguf")
    gguf.write(p, [("general.architecture", gguf.T_STR, "t5encoder")], [gguf.Tensor.from_array("w", np.ones((2, 32), dtype=np.float32))])
    with pytest.raises(runner.RunnerError, match="not supported"):
        runner.quantize_gguf(p, str(tmp_path / "y.gguf"), gguf.Q4_0)


DIA = [  # (name, quantised without -qh, with -qh)   quantize_impl.cpp:42-49
    ("dia.decoder.layers.0.self_attn.q_proj", 1, 1), ("dia.encoder.embedding", 1, 1), ("dia.decoder.heads.3", 0, 1),
    ("dia.decoder.layers.0.pre_sa_norm", 0, 0), ("dia.encoder.norm", 0, 0), ("audio_encoder.initial.weight", 0, 0)]
KOKORO = [  # quantize_impl.cpp:14-40
    ("kokoro.albert.layer.0.ffn", 1), ("kokoro.albert.layer.0.ffn_bias", 0), ("kokoro.albert.token_embd", 0), ("kokoro.albert.attn_norm", 0),
    ("kokoro.text_encoder.lstm.0.weight", 1), ("kokoro.text_encoder.embedding", 0), ("kokoro.duration_predictor.shared_lstm.0.w", 1),
    ("kokoro.duration_predictor.duration_proj", 1), ("kokoro.duration_predictor.layers.1.gamma_weight", 0), ("kokoro.duration_predictor.f0_proj", 0),
    ("kokoro.decoder.generator.m_source_weight", 0), ("kokoro.voice_tensors.af_sky", 0)]


def test_dia_and_kokoro_allow_lists():
    for name, plain, heads in DIA:
        assert runner.quantize_decision("dia", name, 2, gguf.Q4_0) == plain, name
        assert runner.quantize_decision("dia", name, 2, gguf.Q4_0, output_heads=True) == heads, name
    assert runner.quantize_decision("dia", "audio_encoder.initial.weight", 3, gguf.Q4_0, dac_f16=True) == 2
    assert runner.quantize_decision("dia", "audio_encoder.final.alpha", 3, gguf.Q4_0, dac_f16=True) == 0
    for name, q in KOKORO:
        assert runner.quantize_decision("kokoro", name, 2, gguf.Q4_0) == q, name
    # --convert-non-quantized-to-f16: everything "f16 compatible" that is not quantised (quantize_impl.cpp:264)
    assert runner.quantize_decision("kokoro", "kokoro.decoder.generator.m_source_weight", 2, gguf.Q4_0, non_quantizable_f16=True) == 2
    assert runner.quantize_decision("kokoro", "kokoro.albert.layer.0.ffn_bias", 1, gguf.Q4_0, non_quantizable_f16=True) == 0
    assert runner.quantize_decision("kokoro", "kokoro.voice_tensors.af_sky", 3, gguf.Q4_0, non_quantizable_f16=True) == 0


def test_orpheus_file(tmp_path):
    """Extension (the reference's tool aborts on Orpheus): Llama matrices quantised, norms / rope factors / SNAC kept; the
    result is the file the synthetic generator mints directly in Q4_0 — what tests/test_gpu_orpheus.py decodes."""
    full = synth.SynthOrpheusFull(max_gen=28)
    src, dst = full.write_gguf(str(tmp_path / "o.gguf")), str(tmp_path / "o_q.gguf")
    runner.quantize_gguf(src, dst, gguf.Q4_0, output_heads=True, n_threads=2)
    a, b = gguf.Reader(src), gguf.Reader(dst)
    assert b.order == a.order
    for name in a.order:
        s, d = a.tensors[name], b.tensors[name]
        matrix = name.startswith("orpheus.") and len(s.ne) == 2 and not name.endswith(("norm", "rope_frequencies"))
        if matrix:
            assert d.type == gguf.Q4_0 and bytes(d.raw()) == synth.quantize(s.to_f32(), gguf.Q4_0).tobytes(), name
        else:
            assert d.type == s.type and bytes(d.raw()) == bytes(s.raw()), name
    assert b.tensors["orpheus.lm_head"].type == gguf.Q4_0 and b.tensors["orpheus.norm"].type == gguf.F32
    runner.quantize_gguf(src, dst, gguf.Q4_0)
    assert gguf.Reader(dst).tensors["orpheus.lm_head"].type == gguf.F32          # heads need --quantize-output-heads


def test_command_line_matches_the_api(f32_model, tmp_path):
    _, src = f32_model
    exe = os.path.join(os.path.dirname(runner.lib_path()), "quantize")
    assert os.path.exists(exe), "host/quantize not built: run __graft_entry__.build()"
    a, b = str(tmp_path / "cli.gguf"), str(tmp_path / "api.gguf")
    r = subprocess.run([exe, "-mp", src, "-qp", a, "-qt", "Q8", "-qh", "-df", "-nt", "2"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "At tensor: 'decoder.layers.0.fc1.weight' with new size:" in r.stdout    # quantize_impl.cpp:287
    runner.quantize_gguf(src, b, gguf.Q8_0, output_heads=True, dac_f16=True)
    assert open(a, "rb").read() == open(b, "rb").read()
    assert subprocess.run([exe, "-mp", src, "-qp", a, "-qt", "Q3"], capture_output=True, text=True).returncode == 1
    assert "--quantized-type (-qt)" in subprocess.run([exe, "--help"], capture_output=True, text=True).stdout

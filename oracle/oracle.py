"""ctypes wrapper of oracle/liboracle.so and oracle/_ref/libref_sampler.so.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg — never by the product path (tts.cpp_amd/).  See tts_oracle.h for the parity status (sampler pinned against the real
reference, graph arithmetic against the upstream models of tests/golden/upstream_*.npz, ggml's kernel-level rounding unpinned).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
MAX_LAYERS, MAX_HEADS = 64, 16
F32, F16, Q4_0, Q5_0, Q8_0 = 0, 1, 2, 6, 8


class W(C.Structure):
    _fields_ = [("type", C.c_int), ("data", C.c_void_p)]


fp = C.POINTER(C.c_float)


class Layer(C.Structure):
    _fields_ = [("q", W), ("k", W), ("v", W), ("o", W), ("sa_ln_w", fp), ("sa_ln_b", fp),
                ("cq", W), ("ck", W), ("cv", W), ("co", W), ("ca_ln_w", fp), ("ca_ln_b", fp),
                ("fc1", W), ("fc2", W), ("f_ln_w", fp), ("f_ln_b", fp)]


class ParlerModel(C.Structure):
    _fields_ = [("H", C.c_int32), ("L", C.c_int32), ("n_heads", C.c_int32), ("F", C.c_int32), ("V", C.c_int32),
                ("n_out", C.c_int32), ("n_ctx", C.c_int32), ("E", C.c_int32), ("use_cross", C.c_int32),
                ("act_mode", C.c_int32), ("gelu_mode", C.c_int32),
                ("embed_prompts", W), ("embed_tokens", W * MAX_HEADS), ("lm_heads", W * MAX_HEADS),
                ("pos_embed", fp), ("text_encoding", fp), ("ln_w", fp), ("ln_b", fp),
                ("layers", Layer * MAX_LAYERS)]


class Sampler(C.Structure):
    _fields_ = [("n_output_heads", C.c_uint32), ("vocab_size", C.c_uint32), ("top_k", C.c_uint32),
                ("temperature", C.c_float), ("top_p", C.c_float), ("repetition_penalty", C.c_float),
                ("do_sample", C.c_int), ("last_token_ids", C.c_int32 * MAX_HEADS),
                ("repetition_counts", C.c_uint32 * MAX_HEADS), ("rep_initialised", C.c_int)]


class DacRes(C.Structure):
    _fields_ = [("in_alpha", fp), ("in_w", fp), ("in_b", fp), ("out_alpha", fp), ("out_w", fp), ("out_b", fp)]


class DacBlock(C.Structure):
    _fields_ = [("stride", C.c_int32), ("padding", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32),
                ("alpha", fp), ("w", fp), ("b", fp), ("res", DacRes * 3)]


class DacModel(C.Structure):
    _fields_ = [("n_codebooks", C.c_int32), ("codebook_dim", C.c_int32), ("codebook_size", C.c_int32), ("latent", C.c_int32),
                ("codebook", fp * MAX_HEADS), ("out_proj_w", fp * MAX_HEADS), ("out_proj_b", fp * MAX_HEADS),
                ("c0", C.c_int32), ("init_w", fp), ("init_b", fp), ("n_blocks", C.c_int32), ("blocks", DacBlock * 8),
                ("final_alpha", fp), ("final_w", fp), ("final_b", fp), ("f16_conv", C.c_int32)]


class T5Layer(C.Structure):
    _fields_ = [("q", W), ("k", W), ("v", W), ("o", W), ("attn_norm", fp), ("wi_0", W), ("wi_1", W), ("wo", W), ("mlp_norm", fp)]


class T5Model(C.Structure):
    _fields_ = [("H", C.c_int32), ("L", C.c_int32), ("n_heads", C.c_int32), ("F", C.c_int32), ("n_buckets", C.c_int32),
                ("out_size", C.c_int32), ("act_mode", C.c_int32), ("gelu_mode", C.c_int32), ("embd", W), ("rel_bias", fp),
                ("out_norm", fp), ("down_proj", W), ("down_proj_bias", fp), ("layers", T5Layer * MAX_LAYERS)]


class OrpheusLayer(C.Structure):
    _fields_ = [("q", W), ("k", W), ("v", W), ("o", W), ("gate", W), ("up", W), ("down", W), ("input_norm", fp), ("post_norm", fp)]


class OrpheusModel(C.Structure):
    _fields_ = [("H", C.c_int32), ("L", C.c_int32), ("n_heads", C.c_int32), ("n_kv_heads", C.c_int32), ("head_dim", C.c_int32),
                ("F", C.c_int32), ("V", C.c_int32), ("n_ctx", C.c_int32), ("act_mode", C.c_int32), ("embd", W), ("head", W),
                ("out_norm", fp), ("rope_freqs", fp), ("layers", OrpheusLayer * MAX_LAYERS)]


class DiaEncLayer(C.Structure):
    _fields_ = [(n, W) for n in ("q", "k", "v", "o", "gate", "up", "out")] + [("sa_norm", fp), ("mlp_norm", fp)]


class DiaDecLayer(C.Structure):
    _fields_ = [(n, W) for n in ("sq", "sk", "sv", "so", "cq", "ck", "cv", "co", "gate", "up", "out")] + [("sa_norm", fp), ("ca_norm", fp), ("mlp_norm", fp)]


class DiaModel(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("enc_H", "enc_L", "enc_heads", "enc_F", "dec_H", "dec_L", "dec_heads", "dec_kv_heads", "dec_F",
                                          "head_dim", "n_out", "V", "max_ctx", "max_gen", "act_mode")] + [
        ("cfg_scale", C.c_float), ("enc_embd", W), ("enc_norm", fp), ("dec_norm", fp), ("dec_embd", W * MAX_HEADS), ("heads", W * MAX_HEADS),
        ("enc", DiaEncLayer * MAX_LAYERS), ("dec", DiaDecLayer * MAX_LAYERS), ("no_cross_rope", C.c_int32)]


class SnacRes(C.Structure):
    _fields_ = [("in_alpha", fp), ("in_w", fp), ("in_b", fp), ("out_alpha", fp), ("out_w", fp), ("out_b", fp)]


class SnacBlock(C.Structure):
    _fields_ = [("stride", C.c_int32), ("padding", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32),
                ("alpha", fp), ("w", fp), ("b", fp), ("noise_w", fp), ("res", SnacRes * 3)]


class SnacModel(C.Structure):
    _fields_ = [("n_codebooks", C.c_int32), ("codebook_dim", C.c_int32), ("codebook_size", C.c_int32), ("latent", C.c_int32),
                ("repeats", C.c_int32 * 4), ("codebook", fp * 4), ("out_proj_w", fp * 4), ("out_proj_b", fp * 4),
                ("in_w", fp), ("in_b", fp), ("c0", C.c_int32), ("up_w", fp), ("up_b", fp), ("n_blocks", C.c_int32),
                ("blocks", SnacBlock * 8), ("final_alpha", fp), ("final_w", fp), ("final_b", fp)]


class RefSamplerCfg(C.Structure):
    _fields_ = [("n_output_heads", C.c_uint32), ("vocab_size", C.c_uint32), ("top_k", C.c_uint32),
                ("temperature", C.c_float), ("top_p", C.c_float), ("repetition_penalty", C.c_float), ("do_sample", C.c_int)]


_lib = None
_ref = None


def build():
    subprocess.check_call(["make", "-C", HERE], stdout=subprocess.DEVNULL)


def default_threads():
    """OpenMP threads for the oracle: bounded, because the GPU box reports hundreds of cores while the
    container may be CPU-limited (oversubscribed spinning OpenMP teams are catastrophically slow)."""
    if "ORACLE_THREADS" in os.environ:
        return max(1, int(os.environ["ORACLE_THREADS"]))
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def lib():
    global _lib
    if _lib is None:
        os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
        os.environ.setdefault("OMP_NUM_THREADS", str(default_threads()))
        p = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(p):
            build()
        L = C.CDLL(p)
        L.orc_h2f.restype = C.c_float
        L.orc_h2f.argtypes = [C.c_uint16]
        L.orc_f2h.restype = C.c_uint16
        L.orc_f2h.argtypes = [C.c_float]
        L.orc_row_bytes.restype = C.c_size_t
        L.orc_row_bytes.argtypes = [C.c_int, C.c_int64]
        L.orc_dequantize.argtypes = [C.c_int, C.c_void_p, fp, C.c_int64]
        L.orc_quantize.argtypes = [C.c_int, fp, C.c_void_p, C.c_int64]
        L.orc_mul_mat.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_int64, fp, C.c_int64, fp, C.c_int]
        L.orc_mul_mat.restype = None
        L.orc_parler_state_new.restype = C.c_void_p
        L.orc_parler_state_new.argtypes = [C.POINTER(ParlerModel)]
        L.orc_parler_state_free.argtypes = [C.c_void_p]
        L.orc_parler_state_free.restype = None
        L.orc_parler_prep_cross.argtypes = [C.POINTER(ParlerModel), C.c_void_p]
        L.orc_parler_prep_cross.restype = None
        L.orc_parler_decode.argtypes = [C.POINTER(ParlerModel), C.c_void_p, C.c_int, C.POINTER(C.c_uint32), C.c_int,
                                        C.c_uint32, fp, fp]
        L.orc_parler_decode.restype = None
        L.orc_parler_get_kv.argtypes = [C.c_void_p, C.c_int, C.c_int, fp, fp]
        L.orc_parler_get_kv.restype = None
        L.orc_parler_next_ids.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint8), C.c_uint32,
                                          C.c_uint32, C.POINTER(C.c_uint32)]
        L.orc_parler_next_ids.restype = None
        L.orc_parler_adjust_output_tokens.argtypes = [C.POINTER(C.c_uint32), C.c_size_t, C.c_int, C.c_uint32, C.c_uint32,
                                                      C.POINTER(C.c_uint32)]
        L.orc_parler_adjust_output_tokens.restype = C.c_size_t
        L.orc_sampler_init.argtypes = [C.POINTER(Sampler), C.c_uint32, C.c_uint32]
        L.orc_sampler_reset.argtypes = [C.POINTER(Sampler)]
        L.orc_sampler_max.argtypes = [C.POINTER(Sampler), fp, C.POINTER(C.c_uint32)]
        L.orc_sampler_sample.argtypes = [C.POINTER(Sampler), fp, fp, C.POINTER(C.c_uint32)]
        L.orc_dac_decode.argtypes = [C.POINTER(DacModel), C.POINTER(C.c_uint32), C.c_int, fp, C.c_int, fp]
        L.orc_dac_decode.restype = C.c_int64
        L.orc_conv1d.argtypes = [fp, C.c_int, C.c_int64, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, fp]
        L.orc_conv1d.restype = None
        L.orc_conv_transpose1d.argtypes = [fp, C.c_int, C.c_int64, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, fp]
        L.orc_conv_transpose1d.restype = None
        L.orc_snake.argtypes = [fp, C.c_int, C.c_int64, fp]
        L.orc_snake.restype = None
        L.orc_layer_norm.argtypes = [fp, C.c_int, fp, fp, fp]
        L.orc_layer_norm.restype = None
        L.orc_gelu.argtypes = [C.c_float, C.c_int]
        L.orc_gelu.restype = C.c_float
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_set_threads.restype = C.c_int
        L.orc_set_threads(default_threads())
        _lib = L
    return _lib


def ref_sampler_lib():
    """The REAL reference sampler (oracle/_ref/libref_sampler.so); None if it was never built."""
    global _ref
    if _ref is None:
        p = os.path.join(HERE, "_ref", "libref_sampler.so")
        if not os.path.exists(p):
            if os.path.exists("/root/reference/src/sampler.cpp"):
                build()
            if not os.path.exists(p):
                return None
        R = C.CDLL(p)
        u32p = C.POINTER(C.c_uint32)
        R.ref_sampler_max.argtypes = [C.POINTER(RefSamplerCfg), C.POINTER(C.c_int32), u32p, fp, u32p]
        R.ref_sampler_max.restype = None
        R.ref_sampler_sample_greedy.argtypes = [C.POINTER(RefSamplerCfg), fp, u32p]
        R.ref_sampler_sample_greedy.restype = None
        R.ref_sampler_distribution.argtypes = [C.POINTER(RefSamplerCfg), C.POINTER(C.c_int32), u32p, fp, u32p, u32p, fp]
        R.ref_sampler_distribution.restype = C.c_int
        _ref = R
    return _ref


def f32p(a):
    return a.ctypes.data_as(fp)


def u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def dequantize(ttype, raw, n):
    raw = np.frombuffer(bytes(raw), dtype=np.uint8)
    out = np.empty(n, dtype=np.float32)
    assert lib().orc_dequantize(ttype, raw.ctypes.data_as(C.c_void_p), f32p(out), n) == 0
    return out


def quantize(ttype, arr):
    arr = np.ascontiguousarray(arr, dtype=np.float32).reshape(-1)
    out = np.zeros(lib().orc_row_bytes(ttype, arr.size), dtype=np.uint8)
    assert lib().orc_quantize(ttype, f32p(arr), out.ctypes.data_as(C.c_void_p), arr.size) == 0
    return out


def mul_mat(ttype, raw, K, N, x, act_mode=0):
    raw = np.frombuffer(bytes(raw), dtype=np.uint8)
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, K)
    y = np.empty((x.shape[0], N), dtype=np.float32)
    lib().orc_mul_mat(ttype, raw.ctypes.data_as(C.c_void_p), K, N, f32p(x), x.shape[0], f32p(y), act_mode)
    return y


class ParlerOracle:
    """Oracle twin of a tts_cpp_amd.synth.SynthModel (any object with .cfg and .by_name of GGUF tensors)."""

    def __init__(self, model, act_mode=1, gelu_mode=1, use_cross=True):
        self.L = lib()
        cfg = model.cfg
        self.cfg = cfg
        self.keep = []
        m = ParlerModel()
        m.H, m.L, m.n_heads, m.F, m.V = cfg.hidden, cfg.layers, cfg.heads, cfg.ffn, cfg.out_vocab
        m.n_out, m.n_ctx, m.E, m.use_cross = cfg.n_out, cfg.ctx, cfg.enc_len, 1 if use_cross else 0
        m.act_mode, m.gelu_mode = act_mode, gelu_mode
        t = model.by_name

        def w(name):
            raw = np.frombuffer(bytes(t[name].raw()), dtype=np.uint8)
            self.keep.append(raw)
            return W(t[name].type, raw.ctypes.data_as(C.c_void_p).value)

        def f(name):
            a = np.ascontiguousarray(t[name].to_f32().reshape(-1))
            self.keep.append(a)
            return f32p(a)

        m.embed_prompts = w("decoder.embed_prompts")
        m.pos_embed = f("decoder.positional_embed")
        m.text_encoding = f("decoder.text_encoding")
        m.ln_w, m.ln_b = f("decoder.layer_norm.weight"), f("decoder.layer_norm.bias")
        for i in range(cfg.n_out):
            m.embed_tokens[i] = w(f"decoder.embed_tokens.{i}.weight")
            m.lm_heads[i] = w(f"decoder.lm_heads.{i}.weight.head")
        for l in range(cfg.layers):
            p = f"decoder.layers.{l}."
            y = m.layers[l]
            y.q, y.k, y.v, y.o = (w(p + f"self_attn.{n}_proj.weight") for n in ("q", "k", "v", "out"))
            y.sa_ln_w, y.sa_ln_b = f(p + "self_attn_layer_norm.weight"), f(p + "self_attn_layer_norm.bias")
            y.cq, y.ck, y.cv, y.co = (w(p + f"encoder_attn.{n}_proj.weight") for n in ("q", "k", "v", "out"))
            y.ca_ln_w, y.ca_ln_b = f(p + "encoder_attn_layer_norm.weight"), f(p + "encoder_attn_layer_norm.bias")
            y.fc1, y.fc2 = w(p + "fc1.weight"), w(p + "fc2.weight")
            y.f_ln_w, y.f_ln_b = f(p + "final_layer_norm.weight"), f(p + "final_layer_norm.bias")
        self.m = m
        self.state = None
        self.reset()

    def reset(self):
        if self.state:
            self.L.orc_parler_state_free(self.state)
        self.state = self.L.orc_parler_state_new(C.byref(self.m))
        self.L.orc_parler_prep_cross(C.byref(self.m), self.state)

    def set_text_encoding(self, enc):
        enc = np.ascontiguousarray(enc, dtype=np.float32)
        self.keep.append(enc)
        self.m.text_encoding = f32p(enc)
        self.m.E = enc.shape[0]
        self.reset()

    def __del__(self):
        try:
            if self.state:
                self.L.orc_parler_state_free(self.state)
        except Exception:
            pass

    def decode(self, tokens, pos0, audio, want_logits=True, want_hidden=False):
        tokens = np.ascontiguousarray(tokens, dtype=np.uint32)
        S = 1 if audio else tokens.size
        logits = np.empty((self.cfg.n_out, S, self.cfg.out_vocab), dtype=np.float32) if want_logits else None
        hidden = np.empty((S, self.cfg.hidden), dtype=np.float32) if want_hidden else None
        self.L.orc_parler_decode(C.byref(self.m), self.state, 1 if audio else 0, u32p(tokens), S, pos0,
                                 f32p(logits) if want_logits else None, f32p(hidden) if want_hidden else None)
        return logits, hidden

    def get_kv(self, layer, n_pos):
        k = np.empty((n_pos, self.cfg.hidden), dtype=np.float32)
        v = np.empty((n_pos, self.cfg.hidden), dtype=np.float32)
        self.L.orc_parler_get_kv(self.state, layer, n_pos, f32p(k), f32p(v))
        return k, v

    def generate_greedy(self, prompt_ids, n_steps):
        """The reference's generate_from_batch loop (model.cpp:762-792) under greedy sampling, for a
        fixed number of audio steps.  Returns (tokens [n_steps][n_out], logits [n_steps][n_out][V])."""
        cfg = self.cfg
        self.reset()
        self.decode(prompt_ids, 0, audio=False, want_logits=False)
        pos = len(prompt_ids)
        ids = np.full(cfg.n_out, cfg.bos, dtype=np.uint32)
        eos_seen = np.zeros(cfg.n_out, dtype=np.uint8)
        smp = Sampler()
        self.L.orc_sampler_init(C.byref(smp), cfg.n_out, cfg.out_vocab)
        smp.do_sample = 0
        toks_all, logits_all = [], []
        for step in range(1, n_steps + 1):
            logits, _ = self.decode(ids, pos, audio=True)
            lg = np.ascontiguousarray(logits[:, 0, :])
            toks = np.empty(cfg.n_out, dtype=np.uint32)
            self.L.orc_sampler_sample(C.byref(smp), f32p(lg.copy()), None, u32p(toks))
            toks_all.append(toks.copy())
            logits_all.append(lg)
            pos += 1
            eos_seen |= (toks == cfg.eos).astype(np.uint8)  # what the next check_stopping() records
            nxt = np.empty(cfg.n_out, dtype=np.uint32)
            # the reference evaluates eos_seen from the *previous* check_stopping; feeding eos when the
            # token just sampled IS eos gives the same id (model.cpp:781)
            self.L.orc_parler_next_ids(cfg.n_out, step, u32p(toks), eos_seen.ctypes.data_as(C.POINTER(C.c_uint8)),
                                       cfg.bos, cfg.eos, u32p(nxt))
            ids = nxt
        return np.stack(toks_all), np.stack(logits_all)


class T5Oracle:
    """Oracle twin of a tts_cpp_amd.synth.SynthT5 (src/models/parler/t5/model.cpp restated in tts_oracle.c)."""

    def __init__(self, model, act_mode=1, gelu_mode=1):
        self.L = lib()
        self.L.orc_t5_encode.argtypes = [C.POINTER(T5Model), C.POINTER(C.c_uint32), C.c_int, fp]
        self.L.orc_t5_encode.restype = None
        self.L.orc_t5_bucket.argtypes = [C.c_int, C.c_int, C.c_int]
        self.L.orc_t5_bucket.restype = C.c_uint32
        cfg = model.cfg
        self.cfg = cfg
        self.keep = []
        t = model.by_name

        def w(name):
            raw = np.frombuffer(bytes(t[name].raw()), dtype=np.uint8)
            self.keep.append(raw)
            return W(t[name].type, raw.ctypes.data_as(C.c_void_p).value)

        def f(name):
            a = np.ascontiguousarray(t[name].to_f32().reshape(-1))
            self.keep.append(a)
            return f32p(a)

        m = T5Model()
        m.H, m.L, m.n_heads, m.F, m.n_buckets, m.out_size = cfg.hidden, cfg.layers, cfg.heads, cfg.ffn, cfg.buckets, cfg.output_size
        m.act_mode, m.gelu_mode = act_mode, gelu_mode
        m.embd = w("t5encoder.token_embd")
        m.rel_bias = f("t5encoder.enc.blk.0.attn_rel_b")
        m.out_norm = f("t5encoder.enc.final_layer_norm")
        if "t5encoder.down_proj" in t:
            m.down_proj = w("t5encoder.down_proj")
            m.down_proj_bias = f("t5encoder.down_proj_bias")
        for l in range(cfg.layers):
            p = f"t5encoder.enc.blk.{l}."
            y = m.layers[l]
            y.q, y.k, y.v, y.o = w(p + "attn_q"), w(p + "attn_k"), w(p + "attn_v"), w(p + "attn_o")
            y.attn_norm, y.mlp_norm = f(p + "attn_norm"), f(p + "ffn_norm")
            y.wi_0, y.wi_1, y.wo = w(p + "ffn_up"), w(p + "ffn_gate"), w(p + "ffn_down")
        self.m = m

    def encode(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        out = np.empty((ids.size, self.cfg.output_size), dtype=np.float32)
        self.L.orc_t5_encode(C.byref(self.m), u32p(ids), ids.size, f32p(out))
        return out

    def bucket(self, key, query):
        return int(self.L.orc_t5_bucket(key, query, self.cfg.buckets))


class DacOracle:
    def __init__(self, model, f16_conv=None):
        """f16_conv: None = follow the GGUF (F16 conv kernels -> ggml's fp16 im2col semantics), or force 0 / 1; 2 = an F32 model read the way upstream
        ggml_conv_1d would (F16 im2col: the inputs and the kernels of the plain convs rounded to fp16, transposed convs exact fp32)"""
        self.L = lib()
        cfg = model.cfg
        self.cfg = cfg
        self.keep = []
        t = model.by_name

        def f(name, round16=False):
            a = np.ascontiguousarray(t[name].to_f32().reshape(-1))
            if round16 and f16_conv == 2:
                a = np.ascontiguousarray(a.astype(np.float16).astype(np.float32))   # mul_mat converts src1 (the kernel) to the F16 im2col's type
            self.keep.append(a)
            return f32p(a)

        m = DacModel()
        m.n_codebooks, m.codebook_dim, m.codebook_size, m.latent = cfg.n_out, cfg.cb_dim, cfg.cb_size, cfg.latent
        for i in range(cfg.n_out):
            p = f"audio_encoder.quantizers.{i}."
            m.codebook[i], m.out_proj_w[i], m.out_proj_b[i] = f(p + "codebook.weight"), f(p + "out_proj.weight", True), f(p + "out_proj.bias")
        m.c0 = cfg.c0
        m.init_w, m.init_b = f("audio_encoder.initial.weight", True), f("audio_encoder.initial.bias")
        m.n_blocks = len(cfg.strides)
        c = cfg.c0
        for bi, (s, pd) in enumerate(zip(cfg.strides, cfg.paddings)):
            p = f"audio_encoder.decoder_block.{bi + 1}."
            b = m.blocks[bi]
            b.stride, b.padding, b.cin, b.cout = s, pd, c, c // 2
            b.alpha, b.w, b.b = f(p + "final.alpha"), f(p + "final.weight"), f(p + "final.bias")
            for r in range(3):
                q = p + f"residual_unit.{r}.res."
                rr = b.res[r]
                rr.in_alpha, rr.in_w, rr.in_b = f(q + "initial.alpha"), f(q + "initial.weight", True), f(q + "initial.bias")
                rr.out_alpha, rr.out_w, rr.out_b = f(q + "final.alpha"), f(q + "final.weight", True), f(q + "final.bias")
            c //= 2
        m.final_alpha, m.final_w, m.final_b = f("audio_encoder.final.alpha"), f("audio_encoder.final.weight", True), f("audio_encoder.final.bias")
        if f16_conv is None:
            f16_conv = all(x.type == 1 for n, x in t.items() if n.startswith("audio_encoder.") and n.endswith(".weight") and ".in_proj" not in n and len(x.ne) == 3)
        m.f16_conv = int(f16_conv) if f16_conv in (0, 1, 2) else (1 if f16_conv else 0)
        self.m = m

    def stage_shape(self, stage, frames):
        cfg = self.cfg
        if stage == 0:
            return (cfg.latent, frames)
        if stage == 1:
            return (cfg.c0, frames)
        c, L = cfg.c0, frames
        for s in cfg.strides[:stage - 1]:
            c //= 2
            L *= s
        return (c, L)

    def decode(self, codes, stage=-1):
        codes = np.ascontiguousarray(codes, dtype=np.uint32).reshape(-1, self.cfg.n_out)
        frames = codes.shape[0]
        pcm = np.empty(frames * self.cfg.hop, dtype=np.float32)
        st = None
        if stage >= 0:
            st = np.empty(self.stage_shape(stage, frames), dtype=np.float32)
        n = self.L.orc_dac_decode(C.byref(self.m), u32p(codes), frames, f32p(pcm), stage, f32p(st) if st is not None else None)
        assert n == pcm.size
        return (pcm, st) if stage >= 0 else pcm


class SnacOracle:
    """Oracle twin of a tts_cpp_amd.synth.SynthSnac (src/decoder/snac_model.cpp restated in tts_oracle.c)."""

    def __init__(self, model):
        self.L = lib()
        self.L.orc_snac_decode.argtypes = [C.POINTER(SnacModel), C.POINTER(C.c_uint32), C.c_int, fp, fp]
        self.L.orc_snac_decode.restype = C.c_int64
        cfg = model.cfg
        self.cfg = cfg
        self.keep = []
        t = model.by_name

        def f(name, round16=False):   # (round16: DacOracle's fp16-im2col reading; SNAC's lines share the call shape)
            a = np.ascontiguousarray(t["snac." + name].to_f32().reshape(-1))
            self.keep.append(a)
            return f32p(a)

        m = SnacModel()
        m.n_codebooks, m.codebook_dim, m.codebook_size, m.latent = len(cfg.repeats), cfg.cb_dim, cfg.cb_size, cfg.latent
        for i, r in enumerate(cfg.repeats):
            m.repeats[i] = r
            p = f"quantizers.{i}."
            m.codebook[i], m.out_proj_w[i], m.out_proj_b[i] = f(p + "codebook.weight"), f(p + "out_proj.weight", True), f(p + "out_proj.bias")
        m.in_w, m.in_b, m.c0, m.up_w, m.up_b = f("in.weight"), f("in.bias"), cfg.c0, f("up.weight"), f("up.bias")
        m.n_blocks = len(cfg.strides)
        c = cfg.c0
        for bi, (s, pd) in enumerate(zip(cfg.strides, cfg.paddings)):
            p = f"layers.{bi}."
            b = m.blocks[bi]
            b.stride, b.padding, b.cin, b.cout = s, pd, c, c // 2
            b.alpha, b.w, b.b, b.noise_w = f(p + "alpha"), f(p + "weight"), f(p + "bias"), f(p + "noise_weight")
            for r in range(3):
                q = p + f"residual_unit.{r}.res."
                rr = b.res[r]
                rr.in_alpha, rr.in_w, rr.in_b = f(q + "initial.alpha"), f(q + "initial.weight", True), f(q + "initial.bias")
                rr.out_alpha, rr.out_w, rr.out_b = f(q + "final.alpha"), f(q + "final.weight", True), f(q + "final.bias")
            c //= 2
        m.final_alpha, m.final_w, m.final_b = f("alpha_out"), f("final.weight"), f("final.bias")
        self.m = m

    def noise_len(self, T):
        n, L = 0, T
        for s in self.cfg.strides:
            L *= s
            n += L
        return n

    def decode(self, codes, T, noise=None):
        """codes: level-major flat ids (T/4 + T/2 + T for repeats 4,2,1); noise: noise_len(T) floats or None"""
        codes = np.ascontiguousarray(codes, dtype=np.uint32)
        assert codes.size == sum(T // r for r in self.cfg.repeats)
        pcm = np.empty(T * self.cfg.hop, dtype=np.float32)
        nz = None
        if noise is not None:
            nz = np.ascontiguousarray(noise, dtype=np.float32)
            assert nz.size == self.noise_len(T)
        n = self.L.orc_snac_decode(C.byref(self.m), u32p(codes), T, f32p(nz) if nz is not None else None, f32p(pcm))
        assert n == pcm.size
        return pcm


class OrpheusOracle:
    """Oracle twin of a tts_cpp_amd.synth.SynthOrpheus (src/models/orpheus/model.cpp:186-325 restated in tts_oracle.c)."""

    def __init__(self, model, act_mode=1):
        self.L = lib()
        self.L.orc_orpheus_state_new.restype = C.c_void_p
        self.L.orc_orpheus_state_new.argtypes = [C.POINTER(OrpheusModel)]
        self.L.orc_orpheus_state_free.argtypes = [C.c_void_p]
        self.L.orc_orpheus_decode.argtypes = [C.POINTER(OrpheusModel), C.c_void_p, C.POINTER(C.c_uint32), C.c_int, C.c_uint32, fp, fp]
        self.L.orc_orpheus_decode.restype = None
        cfg = model.cfg
        self.cfg = cfg
        self.keep = []
        t = model.by_name

        def w(name):
            raw = np.frombuffer(bytes(t["orpheus." + name].raw()), dtype=np.uint8)
            self.keep.append(raw)
            return W(t["orpheus." + name].type, raw.ctypes.data_as(C.c_void_p).value)

        def f(name):
            a = np.ascontiguousarray(t["orpheus." + name].to_f32().reshape(-1))
            self.keep.append(a)
            return f32p(a)

        m = OrpheusModel()
        m.H, m.L, m.n_heads, m.n_kv_heads, m.head_dim = cfg.hidden, cfg.layers, cfg.heads, cfg.kv_heads, cfg.head_dim
        m.F, m.V, m.n_ctx, m.act_mode = cfg.ffn, cfg.vocab, cfg.ctx, act_mode
        m.embd, m.head, m.out_norm, m.rope_freqs = w("embed_tokens"), w("lm_head"), f("norm"), f("rope_frequencies")
        for l in range(cfg.layers):
            p = f"layers.{l}."
            y = m.layers[l]
            y.q, y.k, y.v, y.o = w(p + "self_attn.q_proj"), w(p + "self_attn.k_proj"), w(p + "self_attn.v_proj"), w(p + "self_attn.o_proj")
            y.gate, y.up, y.down = w(p + "mlp.gate_proj"), w(p + "mlp.up_proj"), w(p + "mlp.down_proj")
            y.input_norm, y.post_norm = f(p + "input_layernorm"), f(p + "post_attention_layernorm")
        self.m = m
        self.state = self.L.orc_orpheus_state_new(C.byref(m))

    def reset(self):
        self.L.orc_orpheus_state_free(self.state)
        self.state = self.L.orc_orpheus_state_new(C.byref(self.m))

    def decode(self, tokens, pos0, want_hidden=False):
        tokens = np.ascontiguousarray(tokens, dtype=np.uint32)
        logits = np.empty(self.cfg.vocab, dtype=np.float32)
        hid = np.empty((tokens.size, self.cfg.hidden), dtype=np.float32) if want_hidden else None
        self.L.orc_orpheus_decode(C.byref(self.m), self.state, u32p(tokens), tokens.size, pos0, f32p(logits), f32p(hid) if want_hidden else None)
        return (logits, hid) if want_hidden else logits

    def __del__(self):
        try:
            self.L.orc_orpheus_state_free(self.state)
        except Exception:
            pass


DIA_DELAY_PATTERN = (0, 8, 9, 10, 11, 12, 13, 14, 15)   # src/models/dia/model.h:84


def dia_tokenize(sentence, max_ctx):
    """dia_runner::tokenize_sentence (src/models/dia/model.cpp:661-705): byte tokens, [S1]/[S2] -> 1/2, zero padding.
    Returns (tokens [max_ctx], sentence_length)."""
    s = sentence.strip(" ")        # strip() in src/util.cpp trims spaces
    if s[:4] not in ("[S1]", "[S2]"):
        s = "[S1] " + s
    if s[-1] != ".":
        s += "."
    b = s.encode("utf-8").replace(b"[S1]", b"\x01").replace(b"[S2]", b"\x02")
    assert len(b) <= max_ctx
    toks = np.zeros(max_ctx, dtype=np.uint32)
    toks[:len(b)] = np.frombuffer(b, dtype=np.uint8)
    return toks, len(b)


class DiaOracle:
    """Oracle twin of a tts_cpp_amd.synth.SynthDia (src/models/dia/model.cpp restated in tts_oracle.c)."""

    def __init__(self, model, act_mode=1, cfg_scale=3.0, cross_rope=True):
        """cross_rope=False: no rope in cross-attention (Hugging Face's DiaModel; the reference ropes the cross query and keys, tts_oracle.h)"""
        self.L = lib()
        L = self.L
        L.orc_dia_state_new.restype = C.c_void_p
        L.orc_dia_state_new.argtypes = [C.POINTER(DiaModel)]
        L.orc_dia_state_free.argtypes = [C.c_void_p]
        L.orc_dia_encode.argtypes = [C.POINTER(DiaModel), C.c_void_p, C.POINTER(C.c_uint32), C.c_int, fp]
        L.orc_dia_encode.restype = None
        L.orc_dia_step.argtypes = [C.POINTER(DiaModel), C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32, fp, fp]
        L.orc_dia_step.restype = None
        L.orc_dia_check_stopping.argtypes = [C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                             C.POINTER(C.c_int)]
        L.orc_dia_adjust_output_tokens.restype = C.c_size_t
        L.orc_dia_adjust_output_tokens.argtypes = [C.POINTER(C.c_uint32), C.c_size_t, C.c_int, C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
        cfg = model.cfg
        self.cfg, self.model = cfg, model
        self.keep = []
        t = model.by_name

        def w(name):
            raw = np.frombuffer(bytes(t["dia." + name].raw()), dtype=np.uint8)
            self.keep.append(raw)
            return W(t["dia." + name].type, raw.ctypes.data_as(C.c_void_p).value)

        def f(name):
            a = np.ascontiguousarray(t["dia." + name].to_f32().reshape(-1))
            self.keep.append(a)
            return f32p(a)

        m = DiaModel()
        m.enc_H, m.enc_L, m.enc_heads, m.enc_F = cfg.enc_hidden, cfg.enc_layers, cfg.enc_heads, cfg.enc_ffn
        m.dec_H, m.dec_L, m.dec_heads, m.dec_kv_heads, m.dec_F = cfg.dec_hidden, cfg.dec_layers, cfg.dec_heads, cfg.dec_kv_heads, cfg.dec_ffn
        m.head_dim, m.n_out, m.V, m.max_ctx, m.max_gen, m.act_mode, m.cfg_scale = cfg.head_dim, cfg.n_out, cfg.out_vocab, cfg.max_ctx, cfg.max_gen, act_mode, cfg_scale
        m.enc_embd, m.enc_norm, m.dec_norm = w("encoder.embedding"), f("encoder.norm"), f("decoder.norm")
        for i in range(cfg.n_out):
            m.dec_embd[i], m.heads[i] = w(f"decoder.embeddings.{i}"), w(f"decoder.heads.{i}")
        for l in range(cfg.enc_layers):
            p, y = f"encoder.layers.{l}.", m.enc[l]
            y.q, y.k, y.v, y.o = w(p + "q_proj"), w(p + "k_proj"), w(p + "v_proj"), w(p + "o_proj")
            y.gate, y.up, y.out, y.sa_norm, y.mlp_norm = w(p + "gate"), w(p + "up"), w(p + "wo"), f(p + "pre_sa_norm"), f(p + "post_sa_norm")
        for l in range(cfg.dec_layers):
            p, y = f"decoder.layers.{l}.", m.dec[l]
            y.sq, y.sk, y.sv, y.so = w(p + "self_q_proj"), w(p + "self_k_proj"), w(p + "self_v_proj"), w(p + "self_o_proj")
            y.cq, y.ck, y.cv, y.co = w(p + "cross_q_proj"), w(p + "cross_k_proj"), w(p + "cross_v_proj"), w(p + "cross_o_proj")
            y.gate, y.up, y.out = w(p + "gate"), w(p + "up"), w(p + "wo")
            y.sa_norm, y.ca_norm, y.mlp_norm = f(p + "pre_sa_norm"), f(p + "pre_ca_norm"), f(p + "pre_mlp_norm")
        m.no_cross_rope = 0 if cross_rope else 1
        self.m = m
        self.state = L.orc_dia_state_new(C.byref(m))
        self.delay = np.array(DIA_DELAY_PATTERN[:cfg.n_out], dtype=np.uint32)

    def encode(self, tokens, sentence_len, want_states=False):
        tokens = np.ascontiguousarray(tokens, dtype=np.uint32)
        assert tokens.size == self.cfg.max_ctx
        out = np.empty((2, self.cfg.max_ctx, self.cfg.enc_hidden), dtype=np.float32) if want_states else None
        self.L.orc_dia_encode(C.byref(self.m), self.state, u32p(tokens), sentence_len, f32p(out) if want_states else None)
        return out

    def step(self, ids, pos, want_raw=False):
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        lg = np.empty((self.cfg.n_out, self.cfg.out_vocab), dtype=np.float32)
        raw = np.empty((2, self.cfg.n_out, self.cfg.out_vocab), dtype=np.float32) if want_raw else None
        self.L.orc_dia_step(C.byref(self.m), self.state, u32p(ids), pos, f32p(lg), f32p(raw) if want_raw else None)
        return (lg, raw) if want_raw else lg

    def check_stopping(self, ids, position, max_generation_size, delay_steps):
        """-> (stop, ids', delay_steps')"""
        ids = np.ascontiguousarray(ids, dtype=np.uint32).copy()
        d = C.c_int(delay_steps)
        stop = self.L.orc_dia_check_stopping(u32p(ids), self.cfg.n_out, u32p(self.delay), self.cfg.max_delay, self.cfg.eos, self.cfg.pad, position,
                                             max_generation_size, C.byref(d))
        return bool(stop), ids, d.value

    def adjust_output_tokens(self, tokens):
        tokens = np.ascontiguousarray(tokens, dtype=np.uint32).reshape(-1)
        out = np.empty(tokens.size, dtype=np.uint32)
        n = self.L.orc_dia_adjust_output_tokens(u32p(tokens), tokens.size, self.cfg.n_out, u32p(self.delay), self.cfg.max_delay, self.cfg.audio_vocab, u32p(out))
        return out[:n].reshape(-1, self.cfg.n_out)

    def generate(self, sentence, max_tokens=0, pick=None):
        """dia_runner::generate + generate_from_batch (:810-858) with greedy picks (or pick(logits) -> ids).
        Returns (output_tokens [steps][n_out], frames [frames][n_out])."""
        cfg = self.cfg
        max_gen = max_tokens if max_tokens > cfg.max_delay else cfg.max_gen
        toks, n = dia_tokenize(sentence, cfg.max_ctx)
        self.L.orc_dia_state_free(self.state)
        self.state = self.L.orc_dia_state_new(C.byref(self.m))
        self.encode(toks, n)
        ids = np.full(cfg.n_out, cfg.bos, dtype=np.uint32)
        pos, delay_steps, outs = 0, -1, []
        while True:
            stop, ids, delay_steps = self.check_stopping(ids, pos, max_gen, delay_steps)
            if stop:
                break
            lg = self.step(ids, pos)
            new = (pick(lg) if pick else lg.argmax(-1)).astype(np.uint32)
            outs.append(new)
            pos += 1
            ids = np.array([new[i] if pos > i else cfg.bos for i in range(cfg.n_out)], dtype=np.uint32)
        outs = np.array(outs, dtype=np.uint32).reshape(-1, cfg.n_out)
        return outs, self.adjust_output_tokens(outs)

    def __del__(self):
        try:
            self.L.orc_dia_state_free(self.state)
        except Exception:
            pass


class KokoroModelC(C.Structure):
    _fields_ = [("n_tensors", C.c_int32), ("names", C.POINTER(C.c_char_p)), ("data", C.POINTER(fp)), ("ne", C.POINTER(C.c_int64))] + [
        (n, C.c_int32) for n in ("n_heads", "n_recurrence", "n_dp_layers", "f0_n_blocks", "n_conv_layers", "n_decoder_blocks", "n_upsamples", "n_kernels",
                                 "n_fft", "hop", "harmonic_num", "up_sampling_factor", "out_conv_padding")] + [
        (n, C.c_float) for n in ("attn_scale", "upsample_scale", "sample_rate", "sin_amp", "noise_std", "voice_threshold")] + [
        ("up_stride", C.c_int32 * 4), ("up_padding", C.c_int32 * 4), ("noise_stride", C.c_int32 * 4), ("noise_padding", C.c_int32 * 4),
        ("res_padding", (C.c_int32 * 3) * 16), ("res_dilation", (C.c_int32 * 3) * 16), ("noise_res_padding", (C.c_int32 * 3) * 4),
        ("noise_res_dilation", (C.c_int32 * 3) * 4), ("gelu_mode", C.c_int32)]


class KokoroOracle:
    """Oracle twin of a tts_cpp_amd.synth.SynthKokoro (src/models/kokoro/model.cpp restated in kokoro_oracle.c; PARITY UNPINNED beyond the ALBERT stage,
    see that file's header).  The phonemizer is outside: inputs are phoneme ids with the bos / eos ids around them."""

    def __init__(self, model, attn_scale=0.125, gelu_mode=1):
        """gelu_mode 1: ALBERT's GELU through ggml's fp16 table like the reference's CPU path; 0: fp32 (the torch fixture)"""
        self.L = lib()
        L = self.L
        L.orc_kokoro_durations.argtypes = [C.POINTER(KokoroModelC), C.POINTER(C.c_uint32), C.c_int, fp, fp, fp]
        L.orc_kokoro_durations.restype = None
        L.orc_kokoro_generate.argtypes = [C.POINTER(KokoroModelC), C.POINTER(C.c_uint32), C.c_int, fp, fp, fp, fp, fp, fp, fp, fp, fp]
        L.orc_kokoro_generate.restype = C.c_int64
        cfg = model.cfg
        self.cfg, self.model = cfg, model
        ts = model.tensors
        self.arrs = [np.ascontiguousarray(t.to_f32().reshape(-1)) for t in ts]
        self.names = (C.c_char_p * len(ts))(*[t.name.encode() for t in ts])
        self.ptrs = (fp * len(ts))(*[f32p(a) for a in self.arrs])
        self.ne = np.array([t.ne + [1] * (4 - len(t.ne)) for t in ts], dtype=np.int64)
        m = KokoroModelC()
        m.n_tensors, m.names, m.data, m.ne = len(ts), self.names, self.ptrs, self.ne.ctypes.data_as(C.POINTER(C.c_int64))
        m.n_heads, m.n_recurrence, m.n_dp_layers, m.f0_n_blocks, m.n_conv_layers = cfg.heads, cfg.recurrence, cfg.dp_layers, cfg.f0_blocks, cfg.conv_layers
        m.n_decoder_blocks, m.n_upsamples, m.n_kernels = cfg.decoder_blocks, len(cfg.up_rates), len(cfg.res_kernels)
        m.n_fft, m.hop, m.harmonic_num, m.up_sampling_factor, m.out_conv_padding = cfg.n_fft, cfg.hop, cfg.harmonic_num, cfg.up_sampling_factor, 3
        # model.h:196,195,219-222 defaults
        m.gelu_mode = int(gelu_mode)
        m.attn_scale, m.upsample_scale, m.sample_rate, m.sin_amp, m.noise_std, m.voice_threshold = attn_scale, float(np.prod(cfg.up_rates) * cfg.hop), 24000.0, 0.1, 0.003, 10.0
        g = model.geometry
        for i, (st, pd) in enumerate(g["up"]):
            m.up_stride[i], m.up_padding[i] = st, pd
        for i, (st, pd) in enumerate(g["noise"]):
            m.noise_stride[i], m.noise_padding[i] = st, pd
        for i, blk in enumerate(g["res"]):
            for ii, (pd, dl) in enumerate(blk):
                m.res_padding[i][ii], m.res_dilation[i][ii] = pd, dl
        for i, blk in enumerate(g["noise_res"]):
            for ii, (pd, dl) in enumerate(blk):
                m.noise_res_padding[i][ii], m.noise_res_dilation[i][ii] = pd, dl
        self.m = m

    def voice(self, name):
        return np.ascontiguousarray(self.model.by_name["kokoro.voice_tensors." + name].to_f32())

    def durations(self, tokens, voice):
        """-> (lengths [n], hidden states [n][D + S])"""
        tokens = np.ascontiguousarray(tokens, dtype=np.uint32)
        v = self.voice(voice)
        lens = np.empty(tokens.size, dtype=np.float32)
        hid = np.empty((tokens.size, self.cfg.dp_hidden + self.cfg.style_half), dtype=np.float32)
        self.L.orc_kokoro_durations(C.byref(self.m), u32p(tokens), tokens.size, f32p(v), f32p(lens), f32p(hid))
        return lens, hid

    def noise_len(self, total):
        return (self.cfg.harmonic_num + 1) * total * self.cfg.up_sampling_factor

    def stft_shape(self, total):
        return (2 * (self.cfg.n_fft // 2 + 1), 2 * total * int(np.prod(self.cfg.up_rates)) + 1)

    def generate(self, tokens, lens, hidden, voice, noise, want_curves=False, hsrc_in=None):
        """want_curves: also return the F0 / N curves and the STFT conditioning; hsrc_in: use this conditioning instead (see kokoro_oracle.c)"""
        tokens = np.ascontiguousarray(tokens, dtype=np.uint32)
        lens = np.ascontiguousarray(lens, dtype=np.float32)
        hidden = np.ascontiguousarray(hidden, dtype=np.float32)
        total = int(lens.sum())
        noise = np.ascontiguousarray(noise, dtype=np.float32)
        assert noise.size == self.noise_len(total)
        v = self.voice(voice)
        pcm = np.empty(total * self.cfg.up_sampling_factor, dtype=np.float32)
        f0 = np.empty(2 * total, dtype=np.float32)
        nn = np.empty(2 * total, dtype=np.float32)
        hs = np.empty(self.stft_shape(total), dtype=np.float32)
        hin = None
        if hsrc_in is not None:
            hin = np.ascontiguousarray(hsrc_in, dtype=np.float32)
            assert hin.shape == hs.shape
        n = self.L.orc_kokoro_generate(C.byref(self.m), u32p(tokens), tokens.size, f32p(lens), f32p(hidden), f32p(v), f32p(noise), f32p(pcm), f32p(f0), f32p(nn),
                                       f32p(hs), f32p(hin) if hin is not None else None)
        assert n == pcm.size
        return (pcm, f0, nn, hs) if want_curves else pcm

"""ctypes binding of include/tts_c.h (host/libtts.so): the C++ runner API the reference's applications use."""
import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))


class RunnerError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("voice", C.c_char_p), ("top_k", C.c_int), ("temperature", C.c_float), ("repetition_penalty", C.c_float),
                ("use_cross_attn", C.c_int), ("max_tokens", C.c_int), ("top_p", C.c_float), ("sample", C.c_int), ("seed", C.c_uint64)]


class QuantizationParams(C.Structure):
    _fields_ = [("n_threads", C.c_uint32), ("quantize_type", C.c_int), ("quantize_output_heads", C.c_int), ("quantize_text_embeddings", C.c_int),
                ("quantize_cross_attn_kv", C.c_int), ("convert_dac_to_f16", C.c_int), ("convert_non_quantizable_to_f16", C.c_int)]


class SamplerCfg(C.Structure):
    _fields_ = [("n_output_heads", C.c_uint32), ("vocab_size", C.c_uint32), ("top_k", C.c_uint32), ("temperature", C.c_float),
                ("top_p", C.c_float), ("repetition_penalty", C.c_float), ("do_sample", C.c_int), ("seed", C.c_uint64)]


EXPORTS = ["tts_c_default_config", "tts_c_runner_from_file", "tts_c_generate", "tts_c_generate_batch", "tts_c_generate_stream", "tts_c_sampling_rate", "tts_c_arch", "tts_c_free",
           "tts_c_last_error", "tts_c_update_conditional_prompt", "tts_c_last_tokens", "tts_c_tokenize", "tts_c_sampler_sample", "tts_c_gguf_summary", "tts_c_gguf_tensor",
           "tts_c_pool_create", "tts_c_pool_set_text_encoder", "tts_c_pool_set_continuous", "tts_c_pool_set_continuous_yield_ms", "tts_c_pool_admitted_in_flight", "tts_c_pool_conditional_prompt", "tts_c_pool_submit", "tts_c_pool_wait", "tts_c_pool_release", "tts_c_pool_stats", "tts_c_pool_load_stats", "tts_c_pool_free", "tts_c_set_load_options", "tts_c_set_load_options_ex", "tts_c_runner_device_context", "tts_c_runner_tokenize",
           "tts_c_quantize_gguf", "tts_c_quantize_decision", "tts_c_quantize_rows",
           "tts_c_dia_tokenize", "tts_c_dia_check_stopping", "tts_c_dia_adjust_output_tokens", "tts_c_single_pass_tokenize", "tts_c_kokoro_chunks", "tts_c_minstd0_jump", "tts_c_minstd0_uniform"]

_lib = None


def lib_path():
    return os.path.join(PKG_DIR, "host", "libtts.so")


def load_lib():
    global _lib
    if _lib is None:
        p = lib_path()
        if not os.path.exists(p):
            raise RunnerError(f"{p} not built: run __graft_entry__.build()")
        L = C.CDLL(p)
        L.tts_c_default_config.argtypes = [C.POINTER(Config)]
        L.tts_c_runner_from_file.restype = C.c_void_p
        L.tts_c_runner_from_file.argtypes = [C.c_char_p, C.c_int, C.POINTER(Config), C.c_int]
        L.tts_c_generate.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(Config), C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_size_t)]
        L.tts_c_generate_batch.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_int, C.POINTER(Config), C.POINTER(C.POINTER(C.c_float)),
                                           C.POINTER(C.c_size_t)]
        L.tts_c_generate_stream.argtypes = L.tts_c_generate_batch.argtypes
        L.tts_c_sampling_rate.restype = C.c_float
        L.tts_c_sampling_rate.argtypes = [C.c_void_p]
        L.tts_c_arch.restype = C.c_char_p
        L.tts_c_arch.argtypes = [C.c_void_p]
        L.tts_c_free.argtypes = [C.c_void_p]
        L.tts_c_free.restype = None
        L.tts_c_last_error.restype = C.c_char_p
        L.tts_c_last_tokens.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint32), C.c_int]
        L.tts_c_tokenize.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_uint32), C.c_int]
        L.tts_c_sampler_sample.argtypes = [C.POINTER(SamplerCfg), C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_float),
                                           C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
        L.tts_c_gguf_summary.argtypes = [C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_char_p, C.c_int]
        L.tts_c_gguf_tensor.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_uint64)]
        L.tts_c_update_conditional_prompt.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.tts_c_single_pass_tokenize.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_char_p, C.POINTER(C.c_uint32), C.c_int]
        L.tts_c_kokoro_chunks.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.c_int]
        L.tts_c_dia_tokenize.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.tts_c_dia_check_stopping.argtypes = [C.POINTER(C.c_uint32)] + [C.c_uint32] * 5 + [C.POINTER(C.c_int)]
        L.tts_c_dia_adjust_output_tokens.restype = C.c_int64
        L.tts_c_dia_adjust_output_tokens.argtypes = [C.POINTER(C.c_uint32), C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
        L.tts_c_quantize_gguf.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(QuantizationParams)]
        L.tts_c_quantize_decision.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(QuantizationParams)]
        L.tts_c_quantize_rows.restype = C.c_int64
        L.tts_c_quantize_rows.argtypes = [C.c_int, C.POINTER(C.c_float), C.c_void_p, C.c_int64, C.c_int64, C.c_uint32]
        L.tts_c_pool_create.restype = C.c_void_p
        L.tts_c_pool_create.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.POINTER(Config)]
        L.tts_c_pool_submit.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(Config)]
        L.tts_c_pool_set_text_encoder.argtypes = [C.c_char_p]
        L.tts_c_pool_set_text_encoder.restype = None
        L.tts_c_pool_set_continuous.argtypes = [C.c_int]
        L.tts_c_pool_set_continuous.restype = None
        L.tts_c_pool_set_continuous_yield_ms.argtypes = [C.c_int]
        L.tts_c_pool_set_continuous_yield_ms.restype = None
        L.tts_c_pool_admitted_in_flight.argtypes = [C.c_void_p]
        L.tts_c_pool_admitted_in_flight.restype = C.c_uint64
        L.tts_c_pool_conditional_prompt.argtypes = [C.c_void_p, C.c_char_p]
        L.tts_c_pool_wait.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.tts_c_pool_release.argtypes = [C.c_void_p, C.c_int]
        L.tts_c_pool_release.restype = None
        L.tts_c_pool_stats.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint64)] * 4
        L.tts_c_pool_stats.restype = None
        L.tts_c_pool_load_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.tts_c_pool_load_stats.restype = None
        L.tts_c_pool_free.argtypes = [C.c_void_p]
        L.tts_c_set_load_options.argtypes = [C.c_int, C.c_int, C.c_int]
        L.tts_c_set_load_options.restype = None
        L.tts_c_set_load_options_ex.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.tts_c_set_load_options_ex.restype = None
        L.tts_c_runner_device_context.argtypes = [C.c_void_p]
        L.tts_c_runner_device_context.restype = C.c_void_p
        L.tts_c_runner_tokenize.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint32), C.c_int]
        L.tts_c_pool_free.restype = None
        _lib = L
    return _lib


def make_config(**kw):
    c = Config()
    load_lib().tts_c_default_config(C.byref(c))
    for k, v in kw.items():
        setattr(c, k, v)
    return c


class Runner:
    """runner_from_file() + generate(), as examples/cli/cli.cpp:79-95 uses them."""

    def __init__(self, path, n_threads=1, cpu_only=True, device=-1, max_seqs=0, declare_only=False, share_with=None, **cfg):
        self.L = load_lib()
        self.cfg = make_config(**cfg)
        if share_with is not None:
            self.L.tts_c_set_load_options_ex(device, max_seqs, 1 if declare_only else 0, share_with.h)   # use that runner's weight arena
        elif device >= 0 or max_seqs > 0 or declare_only:
            self.L.tts_c_set_load_options(device, max_seqs, 1 if declare_only else 0)   # tts_load_options of this thread, for this load
        try:
            self.h = self.L.tts_c_runner_from_file(path.encode(), n_threads, C.byref(self.cfg), 1 if cpu_only else 0)
        finally:
            self.L.tts_c_set_load_options(-1, 0, 0)
        if not self.h:
            raise RunnerError(self.L.tts_c_last_error().decode("utf-8", "replace"))

    @property
    def sampling_rate(self):
        return float(self.L.tts_c_sampling_rate(self.h))

    @property
    def arch(self):
        return self.L.tts_c_arch(self.h).decode()

    def generate(self, text, **cfg):
        c = make_config(**cfg) if cfg else self.cfg
        data = C.POINTER(C.c_float)()
        n = C.c_size_t()
        if self.L.tts_c_generate(self.h, text.encode("utf-8"), C.byref(c), C.byref(data), C.byref(n)) != 0:
            raise RunnerError(self.L.tts_c_last_error().decode("utf-8", "replace"))
        return np.ctypeslib.as_array(data, shape=(n.value,)).copy() if n.value else np.zeros(0, dtype=np.float32)

    def generate_batch(self, texts, **cfg):
        c = make_config(**cfg) if cfg else self.cfg
        n = len(texts)
        arr = (C.c_char_p * n)(*[t.encode("utf-8") for t in texts])
        data = (C.POINTER(C.c_float) * n)()
        ns = (C.c_size_t * n)()
        if self.L.tts_c_generate_batch(self.h, arr, n, C.byref(c), data, ns) != 0:
            raise RunnerError(self.L.tts_c_last_error().decode("utf-8", "replace"))
        return [np.ctypeslib.as_array(data[i], shape=(ns[i],)).copy() if ns[i] else np.zeros(0, dtype=np.float32) for i in range(n)]

    def generate_batch_sizes(self, texts, **cfg):
        """tts_c_generate_batch without copying the audio out of the runner-owned buffer: samples per utterance"""
        c = make_config(**cfg) if cfg else self.cfg
        n = len(texts)
        arr = (C.c_char_p * n)(*[t.encode("utf-8") for t in texts])
        data = (C.POINTER(C.c_float) * n)()
        ns = (C.c_size_t * n)()
        if self.L.tts_c_generate_batch(self.h, arr, n, C.byref(c), data, ns) != 0:
            raise RunnerError(self.L.tts_c_last_error().decode("utf-8", "replace"))
        return [int(ns[i]) for i in range(n)]

    def generate_stream(self, texts, sizes_only=False, **cfg):
        """tts_c_generate_stream: any number of utterances through one continuous-batching session (rows freed by finished utterances are refilled)"""
        c = make_config(**cfg) if cfg else self.cfg
        n = len(texts)
        arr = (C.c_char_p * n)(*[t.encode("utf-8") for t in texts])
        data = (C.POINTER(C.c_float) * n)()
        ns = (C.c_size_t * n)()
        if self.L.tts_c_generate_stream(self.h, arr, n, C.byref(c), data, ns) != 0:
            raise RunnerError(self.L.tts_c_last_error().decode("utf-8", "replace"))
        if sizes_only:
            return [int(ns[i]) for i in range(n)]
        return [np.ctypeslib.as_array(data[i], shape=(ns[i],)).copy() if ns[i] else np.zeros(0, dtype=np.float32) for i in range(n)]

    def tokenize(self, text):
        n = self.L.tts_c_runner_tokenize(self.h, text.encode("utf-8"), None, 0)
        if n < 0:
            raise RunnerError(self.L.tts_c_last_error().decode("utf-8", "replace"))
        out = np.zeros(max(n, 1), dtype=np.uint32)
        self.L.tts_c_runner_tokenize(self.h, text.encode("utf-8"), out.ctypes.data_as(C.POINTER(C.c_uint32)), n)
        return out[:n]

    def device_context(self):
        return self.L.tts_c_runner_device_context(self.h)

    def update_conditional_prompt(self, text_encoder_path, prompt):
        if self.L.tts_c_update_conditional_prompt(self.h, text_encoder_path.encode(), prompt.encode("utf-8")) != 0:
            raise RunnerError(self.L.tts_c_last_error().decode("utf-8", "replace"))

    def last_tokens(self, which):
        n = self.L.tts_c_last_tokens(self.h, which, None, 0)
        out = np.zeros(max(n, 1), dtype=np.uint32)
        self.L.tts_c_last_tokens(self.h, which, out.ctypes.data_as(C.POINTER(C.c_uint32)), n)
        return out[:n]

    def close(self):
        if self.h:
            self.L.tts_c_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def tokenize(gguf_path, text):
    L = load_lib()
    out = np.zeros(4096, dtype=np.uint32)
    n = L.tts_c_tokenize(gguf_path.encode(), text.encode("utf-8"), out.ctypes.data_as(C.POINTER(C.c_uint32)), out.size)
    if n < 0:
        raise RunnerError(L.tts_c_last_error().decode())
    return out[:n].copy()


def _vocab_array(vocab):
    return (C.c_char_p * len(vocab))(*[v.encode("utf-8") if isinstance(v, str) else v for v in vocab])


def single_pass_tokenize(vocab, text):
    L = load_lib()
    va = _vocab_array(vocab)
    t = text.encode("utf-8") if isinstance(text, str) else text
    n = L.tts_c_single_pass_tokenize(va, len(vocab), t, None, 0)
    out = np.zeros(max(n, 1), dtype=np.uint32)
    L.tts_c_single_pass_tokenize(va, len(vocab), t, out.ctypes.data_as(C.POINTER(C.c_uint32)), n)
    return out[:n]


def kokoro_chunks(vocab, phonemes, max_ctx, space_token_id=16):
    L = load_lib()
    va = _vocab_array(vocab)
    t = phonemes.encode("utf-8") if isinstance(phonemes, str) else phonemes
    n = L.tts_c_kokoro_chunks(va, len(vocab), t, max_ctx, space_token_id, None, 0)
    flat = np.zeros(max(n, 1), dtype=np.uint32)
    L.tts_c_kokoro_chunks(va, len(vocab), t, max_ctx, space_token_id, flat.ctypes.data_as(C.POINTER(C.c_uint32)), n)
    out, i = [], 0
    while i < n:
        k = int(flat[i])
        out.append(flat[i + 1:i + 1 + k].tolist())
        i += 1 + k
    return out


def dia_tokenize(sentence, max_ctx):
    L = load_lib()
    out = np.zeros(max_ctx, dtype=np.uint32)
    n = L.tts_c_dia_tokenize(sentence.encode("utf-8") if isinstance(sentence, str) else sentence, max_ctx, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    if n < 0:
        raise RunnerError(L.tts_c_last_error().decode("utf-8", "replace"))
    return out, n


def dia_check_stopping(ids, eos, pad, max_delay, position, max_generation_size, delay_steps):
    a = np.ascontiguousarray(ids, dtype=np.uint32).copy()
    d = C.c_int(delay_steps)
    stop = load_lib().tts_c_dia_check_stopping(a.ctypes.data_as(C.POINTER(C.c_uint32)), eos, pad, max_delay, position, max_generation_size, C.byref(d))
    return bool(stop), a, d.value


def dia_adjust_output_tokens(tokens, audio_vocab, max_delay):
    a = np.ascontiguousarray(tokens, dtype=np.uint32).reshape(-1)
    out = np.zeros(a.size, dtype=np.uint32)
    n = load_lib().tts_c_dia_adjust_output_tokens(a.ctypes.data_as(C.POINTER(C.c_uint32)), a.size, audio_vocab, max_delay, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out[:n].reshape(-1, 9)


def _qparams(qtype, n_threads=1, output_heads=False, text_embeddings=False, cross_attn_kv=False, dac_f16=False, non_quantizable_f16=False):
    return QuantizationParams(n_threads, qtype, int(output_heads), int(text_embeddings), int(cross_attn_kv), int(dac_f16), int(non_quantizable_f16))


def quantize_gguf(ifile, ofile, qtype, **flags):
    """The quantize tool (examples/quantize): rewrites `ifile` with the allow-listed F32 tensors converted to `qtype`."""
    L = load_lib()
    p = _qparams(qtype, **flags)
    if L.tts_c_quantize_gguf(str(ifile).encode(), str(ofile).encode(), C.byref(p)) != 0:
        raise RunnerError(L.tts_c_last_error().decode("utf-8", "replace"))


def quantize_decision(arch, name, n_dims, qtype, **flags):
    L = load_lib()
    p = _qparams(qtype, **flags)
    r = L.tts_c_quantize_decision(arch.encode(), name.encode(), n_dims, C.byref(p))
    if r < 0:
        raise RunnerError(L.tts_c_last_error().decode("utf-8", "replace"))
    return r


def quantize_rows(qtype, arr, n_threads=1):
    from . import gguf
    L = load_lib()
    a = np.ascontiguousarray(arr, dtype=np.float32)
    a2 = a.reshape(-1, a.shape[-1])
    out = np.zeros(gguf.nbytes(qtype, [a2.shape[1], a2.shape[0]]), dtype=np.uint8)
    n = L.tts_c_quantize_rows(qtype, a2.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.c_void_p), a2.shape[1], a2.shape[0], n_threads)
    if n < 0:
        raise RunnerError(L.tts_c_last_error().decode("utf-8", "replace"))
    assert n == out.size
    return out


class Pool:
    """device_pool (host/device_pool.h): the reference server's worker pool with one worker per device and dynamic
    lock-step batching.  submit() -> id; wait(id) -> (audio, batch_size, worker)."""

    def __init__(self, path, n_workers=1, devices=None, max_batch=1, batch_window_ms=0, text_encoder_path=None, continuous=False, continuous_yield_ms=2000, **cfg):
        self.L = load_lib()
        self.cfg = make_config(**cfg)
        self.L.tts_c_pool_set_text_encoder(text_encoder_path.encode() if text_encoder_path else None)
        self.L.tts_c_pool_set_continuous(1 if continuous else 0)   # pool_options::continuous: requests join a running generation
        self.L.tts_c_pool_set_continuous_yield_ms(int(continuous_yield_ms))
        dev = (C.c_int * len(devices))(*devices) if devices else None
        self.h = self.L.tts_c_pool_create(path.encode(), n_workers, dev, len(devices) if devices else 0, max_batch, batch_window_ms, C.byref(self.cfg))
        if not self.h:
            raise RunnerError(self.L.tts_c_last_error().decode("utf-8", "replace"))

    def submit(self, text, **cfg):
        c = make_config(**cfg) if cfg else self.cfg
        i = self.L.tts_c_pool_submit(self.h, text.encode("utf-8"), C.byref(c))
        if i < 0:
            raise RunnerError(self.L.tts_c_last_error().decode("utf-8", "replace"))
        return i

    def conditional_prompt(self, prompt):
        i = self.L.tts_c_pool_conditional_prompt(self.h, prompt.encode("utf-8"))
        if i < 0:
            raise RunnerError(self.L.tts_c_last_error().decode("utf-8", "replace"))
        return i

    def wait(self, task_id, timeout_ms=-1):
        data = C.POINTER(C.c_float)()
        n, bs, wk = C.c_size_t(), C.c_int(), C.c_int()
        rc = self.L.tts_c_pool_wait(self.h, task_id, timeout_ms, C.byref(data), C.byref(n), C.byref(bs), C.byref(wk))
        if rc < 0:
            raise RunnerError(self.L.tts_c_last_error().decode("utf-8", "replace"))
        audio = np.ctypeslib.as_array(data, shape=(n.value,)).copy() if n.value else np.zeros(0, dtype=np.float32)
        err = self.L.tts_c_last_error().decode("utf-8", "replace") if rc == 1 else ""
        self.L.tts_c_pool_release(self.h, task_id)
        return audio, bs.value, wk.value, err

    def load_stats(self):
        a, b = C.c_int(), C.c_int()
        self.L.tts_c_pool_load_stats(self.h, C.byref(a), C.byref(b))
        return {"weight_broadcasts": a.value, "shared_arena_loads": b.value}

    def stats(self):
        v = [C.c_uint64() for _ in range(4)]
        self.L.tts_c_pool_stats(self.h, *[C.byref(x) for x in v])
        d = dict(zip(("tasks", "batches", "largest_batch", "timed_out"), [x.value for x in v]))
        d["admitted_in_flight"] = int(self.L.tts_c_pool_admitted_in_flight(self.h))
        return d

    def close(self):
        if self.h:
            self.L.tts_c_pool_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

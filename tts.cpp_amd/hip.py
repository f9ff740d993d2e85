"""ctypes binding of the C ABI in include/tts_hip.h (libtts_hip.so).

Fails loudly when the library is missing or no MI355X is visible — there is no CPU fallback
in the product path.
"""
import ctypes as C
import os

import numpy as np

from . import gguf

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
MAX_DAC_BLOCKS = 8

FLAG_NO_GRAPH, FLAG_VALU_GEMM, FLAG_NO_DAC, FLAG_NO_PARLER, FLAG_DEQUANT_Q, FLAG_DAC_F32 = 1, 2, 4, 8, 16, 32
KCLASSES = ["embed", "ln", "gemm_qkv", "attn_self", "gemm_attn_out", "gemm_cross_q", "attn_cross", "gemm_cross_out", "gemm_fc1",
            "gemm_fc2", "gemm_heads", "sample", "gemm_other", "dac_embed", "dac_conv7", "dac_conv1", "dac_convt", "dac_final", "dac_resunit", "kokoro_conv_mfma"]


class HipError(RuntimeError):
    pass


class Desc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("hidden_size", C.c_uint32), ("n_layers", C.c_uint32), ("n_attn_heads", C.c_uint32),
        ("n_output_heads", C.c_uint32), ("output_vocab_size", C.c_uint32), ("max_ctx_length", C.c_uint32),
        ("n_encode_length", C.c_uint32), ("use_cross_attn", C.c_uint32),
        ("dac_n_blocks", C.c_uint32),
        ("dac_stride", C.c_uint32 * MAX_DAC_BLOCKS), ("dac_padding", C.c_uint32 * MAX_DAC_BLOCKS),
        ("dac_max_frames", C.c_uint32),
        ("max_seqs", C.c_uint32), ("kv_type", C.c_uint32), ("gelu_mode", C.c_uint32), ("flags", C.c_uint32),
        ("kv_positions", C.c_uint32),
    ]


class T5Desc(C.Structure):
    """tts_hip_t5_desc (include/tts_hip.h)"""
    _fields_ = [("struct_size", C.c_uint32), ("hidden_size", C.c_uint32), ("n_layers", C.c_uint32), ("n_attn_heads", C.c_uint32),
                ("max_ctx_length", C.c_uint32), ("n_buckets", C.c_uint32), ("output_size", C.c_uint32), ("gelu_mode", C.c_uint32),
                ("flags", C.c_uint32)]


class SnacDesc(C.Structure):
    """tts_hip_snac_desc (include/tts_hip.h)"""
    _fields_ = [("struct_size", C.c_uint32), ("n_blocks", C.c_uint32), ("stride", C.c_uint32 * MAX_DAC_BLOCKS),
                ("padding", C.c_uint32 * MAX_DAC_BLOCKS), ("groups", C.c_uint32 * MAX_DAC_BLOCKS), ("n_codebooks", C.c_uint32),
                ("repeats", C.c_uint32 * 4), ("max_frames", C.c_uint32), ("flags", C.c_uint32)]


class OrpheusDesc(C.Structure):
    """tts_hip_orpheus_desc (include/tts_hip.h)"""
    _fields_ = [("struct_size", C.c_uint32), ("hidden_size", C.c_uint32), ("n_layers", C.c_uint32), ("n_attn_heads", C.c_uint32),
                ("n_kv_heads", C.c_uint32), ("head_dim", C.c_uint32), ("vocab_size", C.c_uint32), ("n_ctx", C.c_uint32),
                ("rope_base", C.c_float), ("flags", C.c_uint32), ("max_seqs", C.c_uint32)]


class DiaDesc(C.Structure):
    """tts_hip_dia_desc (include/tts_hip.h)"""
    _fields_ = [(n, C.c_uint32) for n in ("struct_size", "enc_hidden_size", "enc_layers", "enc_attn_heads", "dec_hidden_size", "dec_layers", "dec_attn_heads",
                                          "dec_kv_heads", "head_dim", "n_output_heads", "output_vocab_size", "max_ctx", "max_gen")] + [
        ("cfg_scale", C.c_float), ("flags", C.c_uint32), ("max_utterances", C.c_uint32)]


class KokoroDesc(C.Structure):
    """tts_hip_kokoro_desc (include/tts_hip.h)"""
    _fields_ = [(n, C.c_uint32) for n in ("struct_size", "n_attn_heads", "n_recurrence", "n_dp_layers", "f0_n_blocks", "n_conv_layers", "n_decoder_blocks", "n_upsamples",
                                          "n_kernels", "n_fft", "hop", "harmonic_num", "up_sampling_factor", "out_conv_padding", "max_ctx")] + [
        (n, C.c_float) for n in ("attn_scale", "upsample_scale", "sample_rate", "sin_amp", "noise_std", "voice_threshold")] + [
        ("up_stride", C.c_uint32 * 4), ("up_padding", C.c_uint32 * 4), ("noise_stride", C.c_uint32 * 4), ("noise_padding", C.c_uint32 * 4),
        ("res_padding", (C.c_uint32 * 3) * 16), ("res_dilation", (C.c_uint32 * 3) * 16), ("noise_res_padding", (C.c_uint32 * 3) * 4),
        ("noise_res_dilation", (C.c_uint32 * 3) * 4), ("flags", C.c_uint32)]


class KStat(C.Structure):
    _fields_ = [("ms_total", C.c_double), ("launches", C.c_uint64), ("bytes_total", C.c_double), ("flops_total", C.c_double)]


EXPORTS = [
    "tts_hip_device_count", "tts_hip_create", "tts_hip_destroy", "tts_hip_last_error", "tts_hip_version",
    "tts_hip_upload", "tts_hip_arena_bytes", "tts_hip_finalize", "tts_hip_arena_ptr", "tts_hip_arena_filled",
    "tts_hip_parler_set_text_encoding", "tts_hip_parler_reset", "tts_hip_parler_prefill", "tts_hip_parler_prefill_batch", "tts_hip_parler_step",
    "tts_hip_parler_step_greedy", "tts_hip_parler_generate_greedy", "tts_hip_parler_generate_sampled", "tts_hip_sample_logits",
    "tts_hip_t5_create", "tts_hip_t5_encode", "tts_hip_t5_output_size", "tts_hip_snac_create", "tts_hip_snac_decode", "tts_hip_orpheus_create", "tts_hip_orpheus_decode", "tts_hip_orpheus_step_batch", "tts_hip_orpheus_generate_batch", "tts_hip_orpheus_generate_greedy", "tts_hip_orpheus_generate_sampled", "tts_hip_orpheus_sample_logits", "tts_hip_dia_create", "tts_hip_dia_encode", "tts_hip_dia_step", "tts_hip_dia_encode_slot", "tts_hip_dia_step_batch", "tts_hip_dia_generate", "tts_hip_kokoro_create", "tts_hip_kokoro_durations", "tts_hip_kokoro_generate", "tts_hip_dac_decode", "tts_hip_dac_decode_batch", "tts_hip_debug_read",
    "tts_hip_set_debug", "tts_hip_profile", "tts_hip_profile_get", "tts_hip_kclass_name", "tts_hip_stream",
    "tts_hip_synchronize", "tts_hip_dac_arith", "tts_hip_broadcast_weights", "tts_hip_comm_unique_id", "tts_hip_broadcast_weights_rank", "tts_hip_tune",
    "tts_hip_parler_stream_begin", "tts_hip_parler_stream_admit", "tts_hip_parler_stream_run", "tts_hip_parler_stream_collect", "tts_hip_parler_stream_end",
]

class Sampling(C.Structure):
    """tts_hip_sampling (include/tts_hip.h)"""
    _fields_ = [("top_k", C.c_uint32), ("top_p", C.c_float), ("temperature", C.c_float), ("repetition_penalty", C.c_float)]


_lib = None


def lib_path():
    return os.path.join(PKG_DIR, "libtts_hip.so")


def load_lib():
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise HipError(f"{p} not built: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950)")
    L = C.CDLL(p)
    u32p, f32p, vp = C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.c_void_p
    L.tts_hip_device_count.restype = C.c_int
    L.tts_hip_create.restype = vp
    L.tts_hip_create.argtypes = [C.c_int, C.POINTER(Desc)]
    L.tts_hip_destroy.argtypes = [vp]
    L.tts_hip_destroy.restype = None
    L.tts_hip_last_error.restype = C.c_char_p
    L.tts_hip_version.restype = C.c_char_p
    L.tts_hip_upload.argtypes = [vp, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_int64), vp]
    L.tts_hip_arena_bytes.argtypes = [vp]
    L.tts_hip_arena_bytes.restype = C.c_size_t
    L.tts_hip_finalize.argtypes = [vp, vp]
    L.tts_hip_arena_ptr.argtypes = [vp]
    L.tts_hip_arena_ptr.restype = vp
    L.tts_hip_arena_filled.argtypes = [vp]
    L.tts_hip_broadcast_weights.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int]
    L.tts_hip_dac_arith.argtypes = [vp]
    L.tts_hip_comm_unique_id.argtypes = [vp]
    L.tts_hip_broadcast_weights_rank.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int]
    L.tts_hip_parler_set_text_encoding.argtypes = [vp, f32p, C.c_uint32]
    L.tts_hip_parler_reset.argtypes = [vp]
    L.tts_hip_parler_prefill.argtypes = [vp, C.c_uint32, u32p, C.c_uint32, C.c_uint32]
    L.tts_hip_parler_prefill_batch.argtypes = [vp, C.c_uint32, u32p, u32p, u32p, u32p]
    L.tts_hip_parler_step.argtypes = [vp, C.c_uint32, u32p, u32p, u32p, f32p]
    L.tts_hip_parler_step_greedy.argtypes = [vp, C.c_uint32, u32p, u32p, u32p, u32p]
    L.tts_hip_parler_generate_greedy.argtypes = [vp, C.c_uint32, u32p, C.c_uint32, C.c_uint32, C.c_uint32, u32p, u32p]
    L.tts_hip_parler_generate_sampled.argtypes = [vp, C.c_uint32, u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(Sampling), f32p, u32p, u32p]
    L.tts_hip_sample_logits.argtypes = [vp, C.c_uint32, f32p, C.POINTER(Sampling), f32p, C.POINTER(C.c_int32), u32p, u32p]
    L.tts_hip_orpheus_create.restype = vp
    L.tts_hip_orpheus_create.argtypes = [C.c_int, C.POINTER(OrpheusDesc)]
    L.tts_hip_orpheus_decode.argtypes = [vp, u32p, C.c_uint32, C.c_uint32, f32p, u32p]
    L.tts_hip_orpheus_step_batch.argtypes = [vp, C.c_uint32, u32p, u32p, u32p, f32p, u32p]
    L.tts_hip_orpheus_generate_batch.argtypes = [vp, C.c_uint32, u32p, u32p, C.c_uint32, C.c_uint32, C.POINTER(Sampling), C.POINTER(C.c_float), u32p, u32p]
    L.tts_hip_orpheus_generate_greedy.argtypes = [vp, u32p, C.c_uint32, C.c_uint32, C.c_uint32, u32p, u32p]
    L.tts_hip_orpheus_generate_sampled.argtypes = [vp, u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(Sampling), C.POINTER(C.c_float), u32p, u32p]
    L.tts_hip_orpheus_sample_logits.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(Sampling), C.c_float, C.POINTER(C.c_int32), u32p, u32p]
    L.tts_hip_kokoro_create.restype = vp
    L.tts_hip_kokoro_create.argtypes = [C.c_int, C.POINTER(KokoroDesc)]
    L.tts_hip_kokoro_durations.argtypes = [vp, u32p, C.c_uint32, C.c_char_p, f32p, f32p]
    L.tts_hip_kokoro_generate.argtypes = [vp, u32p, C.c_uint32, f32p, f32p, C.c_char_p, f32p, f32p, f32p, f32p]
    L.tts_hip_dia_create.restype = vp
    L.tts_hip_dia_create.argtypes = [C.c_int, C.POINTER(DiaDesc)]
    L.tts_hip_dia_encode.argtypes = [vp, u32p, C.c_uint32, f32p]
    L.tts_hip_dia_step.argtypes = [vp, u32p, C.c_uint32, f32p, f32p]
    L.tts_hip_dia_encode_slot.argtypes = [vp, C.c_uint32, u32p, C.c_uint32, f32p]
    L.tts_hip_dia_step_batch.argtypes = [vp, C.c_uint32, u32p, u32p, u32p, f32p, f32p]
    L.tts_hip_dia_generate.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(DiaCodes), C.POINTER(Sampling), f32p, u32p, u32p]
    L.tts_hip_snac_create.restype = vp
    L.tts_hip_snac_create.argtypes = [C.c_int, C.POINTER(SnacDesc)]
    L.tts_hip_snac_decode.argtypes = [vp, u32p, C.c_uint32, f32p, f32p]
    L.tts_hip_t5_create.restype = vp
    L.tts_hip_t5_create.argtypes = [C.c_int, C.POINTER(T5Desc)]
    L.tts_hip_t5_encode.argtypes = [vp, u32p, C.c_uint32, f32p]
    L.tts_hip_t5_output_size.argtypes = [vp]
    L.tts_hip_dac_decode.argtypes = [vp, u32p, C.c_uint32, f32p]
    L.tts_hip_dac_decode_batch.argtypes = [vp, u32p, u32p, C.c_uint32, f32p]
    L.tts_hip_debug_read.argtypes = [vp, C.c_char_p, f32p, C.c_size_t]
    L.tts_hip_debug_read.restype = C.c_int64
    L.tts_hip_set_debug.argtypes = [vp, C.c_int]
    L.tts_hip_profile.argtypes = [vp, C.c_int]
    L.tts_hip_profile_get.argtypes = [vp, C.c_int, C.POINTER(KStat)]
    L.tts_hip_kclass_name.argtypes = [C.c_int]
    L.tts_hip_kclass_name.restype = C.c_char_p
    L.tts_hip_stream.argtypes = [vp]
    L.tts_hip_stream.restype = vp
    L.tts_hip_synchronize.argtypes = [vp]
    L.tts_hip_tune.argtypes = [vp, C.c_char_p, C.c_int]
    L.tts_hip_parler_stream_begin.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(Sampling)]
    L.tts_hip_parler_stream_admit.argtypes = [vp, C.c_uint32, u32p, u32p, u32p, f32p]
    L.tts_hip_parler_stream_run.argtypes = [vp, C.c_uint32, u32p, u32p, u32p]
    L.tts_hip_parler_stream_collect.argtypes = [vp, C.c_uint32, C.c_uint32, u32p]
    L.tts_hip_parler_stream_end.argtypes = [vp]
    _lib = L
    return L


def _u32(a):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint32))


class HipEngine:
    """One device context holding a Parler decoder and/or a DAC codec."""

    def __init__(self, cfg, device=0, max_seqs=1, kv_type=gguf.F32, gelu_mode=1, flags=0, use_cross_attn=True,
                 n_encode_length=None, kv_positions=0, tune=None):
        self.L = load_lib()
        self.cfg = cfg
        d = Desc()
        d.struct_size = C.sizeof(Desc)
        d.hidden_size, d.n_layers, d.n_attn_heads = cfg.hidden, cfg.layers, cfg.heads
        d.n_output_heads, d.output_vocab_size, d.max_ctx_length = cfg.n_out, cfg.out_vocab, cfg.ctx
        d.n_encode_length = cfg.enc_len if n_encode_length is None else n_encode_length
        d.use_cross_attn = 1 if use_cross_attn else 0
        d.dac_n_blocks = len(cfg.strides)
        for i, (s, p) in enumerate(zip(cfg.strides, cfg.paddings)):
            d.dac_stride[i], d.dac_padding[i] = s, p
        d.dac_max_frames = cfg.max_gen
        d.max_seqs, d.kv_type, d.gelu_mode, d.flags = max_seqs, kv_type, gelu_mode, flags
        d.kv_positions = kv_positions
        self.desc = d
        self.max_seqs = max_seqs
        self.ctx = self.L.tts_hip_create(device, C.byref(d))
        if not self.ctx:
            raise HipError(self.err())
        self.finalized = False
        for k, v in (tune or {}).items():
            self.tune(k, v)

    def err(self):
        return self.L.tts_hip_last_error().decode("utf-8", "replace")

    def _chk(self, rc):
        if rc != 0:
            raise HipError(self.err())

    def tune(self, key, value):
        """tts_hip_tune: a named tuning / fallback switch (before the first launch)"""
        self._chk(self.L.tts_hip_tune(self.ctx, key.encode(), int(value)))

    def close(self):
        if self.ctx:
            self.L.tts_hip_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights ------------------------------------------------------------------------------
    def upload(self, t: gguf.Tensor, declare_only=False):
        ne = (C.c_int64 * 4)(*(t.ne + [1] * (4 - len(t.ne))))
        raw = None if declare_only else np.frombuffer(bytes(t.raw()), dtype=np.uint8)
        ptr = None if declare_only else raw.ctypes.data_as(C.c_void_p)
        self._chk(self.L.tts_hip_upload(self.ctx, t.name.encode(), t.type, len(t.ne), ne, ptr))

    def load(self, model, declare_only=False, external_arena=None):
        for t in model.tensors:
            self.upload(t, declare_only)
        self.finalize(external_arena)

    def arena_bytes(self):
        n = self.L.tts_hip_arena_bytes(self.ctx)
        if n == 0:
            raise HipError(self.err())
        return n

    def finalize(self, external_arena=None):
        self._chk(self.L.tts_hip_finalize(self.ctx, external_arena))
        self.finalized = True

    def arena_ptr(self):
        return self.L.tts_hip_arena_ptr(self.ctx)

    def arena_filled(self):
        self._chk(self.L.tts_hip_arena_filled(self.ctx))

    @staticmethod
    def broadcast_weights(engines, root=0):
        """tts_hip_broadcast_weights: RCCL broadcast of engines[root]'s arena to the other engines (one per distinct device, one process)"""
        L = load_lib()
        arr = (C.c_void_p * len(engines))(*[e.ctx for e in engines])
        if L.tts_hip_broadcast_weights(arr, len(engines), root) != 0:
            raise HipError(L.tts_hip_last_error().decode())

    def set_text_encoding(self, enc):
        enc = np.ascontiguousarray(enc, dtype=np.float32)
        self._chk(self.L.tts_hip_parler_set_text_encoding(self.ctx, enc.ctypes.data_as(C.POINTER(C.c_float)), enc.shape[0]))

    # ---- parler -------------------------------------------------------------------------------
    def reset(self):
        self._chk(self.L.tts_hip_parler_reset(self.ctx))

    def prefill(self, seq, ids, pos0=0):
        a, p = _u32(ids)
        self._chk(self.L.tts_hip_parler_prefill(self.ctx, seq, p, len(a), pos0))

    def prefill_batch(self, prompts):
        """prompts: list of id arrays, sequence i -> cache slot i, positions from 0"""
        lens = np.array([len(p) for p in prompts], dtype=np.uint32)
        cat, cp = _u32(np.concatenate([np.asarray(p, dtype=np.uint32) for p in prompts]))
        l, lp = _u32(lens)
        self._chk(self.L.tts_hip_parler_prefill_batch(self.ctx, len(prompts), None, cp, lp, None))

    def step(self, ids, pos, seqs=None):
        """ids [n][n_out], pos [n] -> logits [n][n_out][V]"""
        a, ap = _u32(ids)
        b, bp = _u32(pos)
        n = len(b)
        sp = None
        if seqs is not None:
            s, sp = _u32(seqs)
        out = np.empty((n, self.cfg.n_out, self.cfg.out_vocab), dtype=np.float32)
        self._chk(self.L.tts_hip_parler_step(self.ctx, n, ap, bp, sp, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def step_greedy(self, ids, pos, seqs=None):
        a, ap = _u32(ids)
        b, bp = _u32(pos)
        n = len(b)
        sp = None
        if seqs is not None:
            s, sp = _u32(seqs)
        out = np.empty((n, self.cfg.n_out), dtype=np.uint32)
        self._chk(self.L.tts_hip_parler_step_greedy(self.ctx, n, ap, bp, sp, out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out

    def generate_greedy(self, start_pos, n_steps, bos=None, eos=None):
        b, bp = _u32(start_pos)
        n = len(b)
        out = np.empty((n_steps, n, self.cfg.n_out), dtype=np.uint32)
        done = np.zeros(n, dtype=np.uint32)
        self._chk(self.L.tts_hip_parler_generate_greedy(
            self.ctx, n, bp, n_steps, self.cfg.bos if bos is None else bos, self.cfg.eos if eos is None else eos,
            out.ctypes.data_as(C.POINTER(C.c_uint32)), done.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out, done

    # ---- continuous batching (tts_hip_parler_stream_*) ----
    def stream_begin(self, n_slots, max_steps, sampling=None, bos=None, eos=None):
        sp = None
        if sampling is not None:
            sp = Sampling(*sampling)
        self._chk(self.L.tts_hip_parler_stream_begin(self.ctx, n_slots, max_steps, self.cfg.bos if bos is None else bos, self.cfg.eos if eos is None else eos,
                                                     C.byref(sp) if sp is not None else None))
        self._stream_slots = n_slots

    def stream_admit(self, slots, prompts, uniforms=None):
        s, sp = _u32(slots)
        lens, lp = _u32([len(p) for p in prompts])
        cat, cp = _u32(np.concatenate([np.asarray(p, dtype=np.uint32) for p in prompts]))
        up = None
        if uniforms is not None:
            u = np.ascontiguousarray(uniforms, dtype=np.float32)
            up = u.ctypes.data_as(C.POINTER(C.c_float))
        self._chk(self.L.tts_hip_parler_stream_admit(self.ctx, len(s), sp, cp, lp, up))

    def stream_run(self, n_steps):
        """-> [(slot, steps)] of the utterances that finished inside these steps"""
        n = C.c_uint32()
        fs = np.zeros(self._stream_slots, dtype=np.uint32)
        fn = np.zeros(self._stream_slots, dtype=np.uint32)
        self._chk(self.L.tts_hip_parler_stream_run(self.ctx, n_steps, C.byref(n), fs.ctypes.data_as(C.POINTER(C.c_uint32)), fn.ctypes.data_as(C.POINTER(C.c_uint32))))
        return [(int(fs[i]), int(fn[i])) for i in range(n.value)]

    def stream_collect(self, slot, steps):
        out = np.zeros((steps, self.cfg.n_out), dtype=np.uint32)
        self._chk(self.L.tts_hip_parler_stream_collect(self.ctx, slot, steps, out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out

    def stream_end(self):
        self._chk(self.L.tts_hip_parler_stream_end(self.ctx))

    def generate_sampled(self, start_pos, n_steps, uniforms, top_k=50, top_p=1.0, temperature=1.0, bos=None, eos=None,
                         repetition_penalty=1.0):
        """uniforms [n_steps][n][n_out] -> (tokens [n_steps][n][n_out], steps_done [n])"""
        b, bp = _u32(start_pos)
        n = len(b)
        u = np.ascontiguousarray(uniforms, dtype=np.float32).reshape(n_steps, n, self.cfg.n_out)
        out = np.empty((n_steps, n, self.cfg.n_out), dtype=np.uint32)
        done = np.zeros(n, dtype=np.uint32)
        sp = Sampling(top_k, top_p, temperature, repetition_penalty)
        self._chk(self.L.tts_hip_parler_generate_sampled(
            self.ctx, n, bp, n_steps, self.cfg.bos if bos is None else bos, self.cfg.eos if eos is None else eos, C.byref(sp),
            u.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_uint32)), done.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out, done

    def sample_logits(self, logits, uniforms, top_k=50, top_p=1.0, temperature=1.0, repetition_penalty=1.0, last_ids=None, rep_counts=None):
        """the device sampler alone: logits [n][n_out][V], uniforms [n][n_out] -> ids [n][n_out];
        last_ids (int32) / rep_counts (uint32) [n][n_out] are updated in place when repetition_penalty != 1"""
        lg = np.ascontiguousarray(logits, dtype=np.float32).reshape(-1, self.cfg.n_out, self.cfg.out_vocab)
        u = np.ascontiguousarray(uniforms, dtype=np.float32).reshape(lg.shape[0], self.cfg.n_out)
        out = np.empty((lg.shape[0], self.cfg.n_out), dtype=np.uint32)
        sp = Sampling(top_k, top_p, temperature, repetition_penalty)
        lp = last_ids.ctypes.data_as(C.POINTER(C.c_int32)) if last_ids is not None else None
        cp = rep_counts.ctypes.data_as(C.POINTER(C.c_uint32)) if rep_counts is not None else None
        self._chk(self.L.tts_hip_sample_logits(self.ctx, lg.shape[0], lg.ctypes.data_as(C.POINTER(C.c_float)), C.byref(sp),
                                                u.ctypes.data_as(C.POINTER(C.c_float)), lp, cp, out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out

    # ---- dac ----------------------------------------------------------------------------------
    def dac_decode(self, codes):
        a, ap = _u32(codes)
        frames = a.size // self.cfg.n_out
        out = np.empty(frames * self.cfg.hop, dtype=np.float32)
        self._chk(self.L.tts_hip_dac_decode(self.ctx, ap, frames, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def dac_decode_batch(self, codes_list):
        """codes_list: list of [frames_i][n_out] arrays -> list of PCM arrays"""
        frames = np.array([len(np.asarray(c).reshape(-1, self.cfg.n_out)) for c in codes_list], dtype=np.uint32)
        if frames.sum() == 0:
            return [np.zeros(0, dtype=np.float32) for _ in codes_list]
        cat = np.concatenate([np.asarray(c, dtype=np.uint32).reshape(-1, self.cfg.n_out) for c in codes_list])
        a, ap = _u32(cat)
        f, fp = _u32(frames)
        out = np.empty(int(frames.sum()) * self.cfg.hop, dtype=np.float32)
        self._chk(self.L.tts_hip_dac_decode_batch(self.ctx, ap, fp, len(frames), out.ctypes.data_as(C.POINTER(C.c_float))))
        offs = np.concatenate([[0], np.cumsum(frames.astype(np.int64) * self.cfg.hop)])
        return [out[offs[i]:offs[i + 1]] for i in range(len(frames))]

    # ---- introspection ------------------------------------------------------------------------
    def set_debug(self, on=True):
        self._chk(self.L.tts_hip_set_debug(self.ctx, 1 if on else 0))

    def debug_read(self, what, max_floats):
        out = np.empty(max_floats, dtype=np.float32)
        n = self.L.tts_hip_debug_read(self.ctx, what.encode(), out.ctypes.data_as(C.POINTER(C.c_float)), max_floats)
        if n < 0:
            raise HipError(self.err())
        return out[:n].copy()

    def profile(self, enable):
        """True/1: every launch (eager forwards); 2: DAC launches only (usable inside a timed region); False/0: off"""
        self._chk(self.L.tts_hip_profile(self.ctx, int(enable)))

    def profile_get(self):
        res = {}
        for k, name in enumerate(KCLASSES):
            st = KStat()
            self._chk(self.L.tts_hip_profile_get(self.ctx, k, C.byref(st)))
            res[name] = dict(ms_total=st.ms_total, launches=int(st.launches), bytes_total=st.bytes_total, flops_total=st.flops_total)
        return res

    def synchronize(self):
        self._chk(self.L.tts_hip_synchronize(self.ctx))


class T5Engine:
    """A T5 voice-prompt encoder context (tts_hip_t5_create): load a synth.SynthT5 / GGUF tensor list, encode ids."""

    def __init__(self, cfg, device=0, gelu_mode=1, flags=0):
        self.L = load_lib()
        self.cfg = cfg
        d = T5Desc()
        d.struct_size = C.sizeof(T5Desc)
        d.hidden_size, d.n_layers, d.n_attn_heads, d.max_ctx_length = cfg.hidden, cfg.layers, cfg.heads, cfg.ctx
        d.n_buckets, d.output_size, d.gelu_mode, d.flags = cfg.buckets, cfg.output_size, gelu_mode, flags
        self.ctx = self.L.tts_hip_t5_create(device, C.byref(d))
        if not self.ctx:
            raise HipError(self.L.tts_hip_last_error().decode("utf-8", "replace"))

    def _chk(self, rc):
        if rc != 0:
            raise HipError(self.L.tts_hip_last_error().decode("utf-8", "replace"))

    def load(self, model):
        for t in model.tensors:
            ne = (C.c_int64 * 4)(*(t.ne + [1] * (4 - len(t.ne))))
            raw = np.frombuffer(bytes(t.raw()), dtype=np.uint8)
            self._chk(self.L.tts_hip_upload(self.ctx, t.name.encode(), t.type, len(t.ne), ne, raw.ctypes.data_as(C.c_void_p)))
        self._chk(self.L.tts_hip_finalize(self.ctx, None))

    def encode(self, ids):
        a, ap = _u32(ids)
        n_out = self.L.tts_hip_t5_output_size(self.ctx)
        if n_out < 0:
            raise HipError(self.L.tts_hip_last_error().decode("utf-8", "replace"))
        out = np.empty((a.size, n_out), dtype=np.float32)
        self._chk(self.L.tts_hip_t5_encode(self.ctx, ap, a.size, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def close(self):
        if self.ctx:
            self.L.tts_hip_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def engine_profile(engine, enable):
    """tts_hip_profile on any engine's context (1: every launch, 2: codec launches only, 0: off)"""
    if engine.L.tts_hip_profile(engine.ctx, int(enable)) != 0:
        raise HipError(engine.L.tts_hip_last_error().decode())


def engine_profile_get(engine):
    res = {}
    for k, name in enumerate(KCLASSES):
        st = KStat()
        if engine.L.tts_hip_profile_get(engine.ctx, k, C.byref(st)) != 0:
            raise HipError(engine.L.tts_hip_last_error().decode())
        res[name] = dict(ms_total=st.ms_total, launches=int(st.launches), bytes_total=st.bytes_total, flops_total=st.flops_total)
    return res


class SnacEngine:
    """A SNAC codec context (tts_hip_snac_create): load a synth.SynthSnac / the snac.* tensors of an Orpheus GGUF, decode."""

    def __init__(self, cfg, device=0, flags=0):
        self.L = load_lib()
        self.cfg = cfg
        d = SnacDesc()
        d.struct_size = C.sizeof(SnacDesc)
        d.n_blocks = len(cfg.strides)
        c = cfg.c0
        for i, (s, p) in enumerate(zip(cfg.strides, cfg.paddings)):
            c //= 2
            d.stride[i], d.padding[i], d.groups[i] = s, p, c
        d.n_codebooks = len(cfg.repeats)
        for i, r in enumerate(cfg.repeats):
            d.repeats[i] = r
        d.max_frames, d.flags = cfg.max_frames, flags
        self.ctx = self.L.tts_hip_snac_create(device, C.byref(d))
        if not self.ctx:
            raise HipError(self.L.tts_hip_last_error().decode("utf-8", "replace"))

    def _chk(self, rc):
        if rc != 0:
            raise HipError(self.L.tts_hip_last_error().decode("utf-8", "replace"))

    def load(self, model):
        for t in model.tensors:
            ne = (C.c_int64 * 4)(*(t.ne + [1] * (4 - len(t.ne))))
            raw = np.frombuffer(bytes(t.raw()), dtype=np.uint8)
            self._chk(self.L.tts_hip_upload(self.ctx, t.name.encode(), t.type, len(t.ne), ne, raw.ctypes.data_as(C.c_void_p)))
        self._chk(self.L.tts_hip_finalize(self.ctx, None))

    def decode(self, codes, T, noise=None):
        a, ap = _u32(codes)
        out = np.empty(T * self.cfg.hop, dtype=np.float32)
        nz = None if noise is None else np.ascontiguousarray(noise, dtype=np.float32)
        self._chk(self.L.tts_hip_snac_decode(self.ctx, ap, T, None if nz is None else nz.ctypes.data_as(C.POINTER(C.c_float)),
                                             out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def close(self):
        if self.ctx:
            self.L.tts_hip_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class OrpheusEngine:
    """An Orpheus decoder context (tts_hip_orpheus_create): Llama-3 blocks, one sequence or max_seqs lock-step utterances."""

    def __init__(self, cfg, device=0, flags=0, max_seqs=1):
        self.L = load_lib()
        self.cfg = cfg
        d = OrpheusDesc()
        d.struct_size = C.sizeof(OrpheusDesc)
        d.hidden_size, d.n_layers, d.n_attn_heads, d.n_kv_heads, d.head_dim = cfg.hidden, cfg.layers, cfg.heads, cfg.kv_heads, cfg.head_dim
        d.vocab_size, d.n_ctx, d.rope_base, d.flags = cfg.vocab, cfg.ctx, 0.0, flags
        d.max_seqs = max_seqs
        self.ctx = self.L.tts_hip_orpheus_create(device, C.byref(d))
        if not self.ctx:
            raise HipError(self.L.tts_hip_last_error().decode("utf-8", "replace"))

    def _chk(self, rc):
        if rc != 0:
            raise HipError(self.L.tts_hip_last_error().decode("utf-8", "replace"))

    def debug_read(self, what, max_floats):
        """'l_logits': the logits row the last step left; 'l_k:<layer>' / 'l_v:<layer>': slot 0's cache rows (test / debugging aid)"""
        out = np.empty(max_floats, dtype=np.float32)
        n = self.L.tts_hip_debug_read(self.ctx, what.encode(), out.ctypes.data_as(C.POINTER(C.c_float)), max_floats)
        if n < 0:
            raise HipError(self.err())
        return out[:n].copy()

    def tune(self, key, value):
        """tts_hip_tune: a named tuning / fallback switch (before the first launch)"""
        self._chk(self.L.tts_hip_tune(self.ctx, key.encode(), int(value)))

    def load(self, model):
        for t in model.tensors:
            ne = (C.c_int64 * 4)(*(t.ne + [1] * (4 - len(t.ne))))
            raw = np.frombuffer(bytes(t.raw()), dtype=np.uint8)
            self._chk(self.L.tts_hip_upload(self.ctx, t.name.encode(), t.type, len(t.ne), ne, raw.ctypes.data_as(C.c_void_p)))
        self._chk(self.L.tts_hip_finalize(self.ctx, None))

    def decode(self, ids, pos0):
        """-> (logits [vocab] of the last token, its arg-max)"""
        a, ap = _u32(ids)
        lg = np.empty(self.cfg.vocab, dtype=np.float32)
        tok = C.c_uint32()
        self._chk(self.L.tts_hip_orpheus_decode(self.ctx, ap, a.size, pos0, lg.ctypes.data_as(C.POINTER(C.c_float)), C.byref(tok)))
        return lg, tok.value

    def generate_greedy(self, prompt, max_new, stop_id):
        a, ap = _u32(prompt)
        out = np.zeros(max_new, dtype=np.uint32)
        n = C.c_uint32()
        self._chk(self.L.tts_hip_orpheus_generate_greedy(self.ctx, ap, a.size, max_new, stop_id, out.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(n)))
        return out[:n.value]

    def generate_sampled(self, prompt, max_new, stop_id, uniforms, top_k=50, temperature=1.0, repetition_penalty=1.0, top_p=1.0):
        """sampler::sample on the device (top_k 1..64, top_p >= 1); uniforms [max_new]: the draw of the k-th sampler call"""
        a, ap = _u32(prompt)
        u = np.ascontiguousarray(uniforms, dtype=np.float32)
        assert u.size >= max_new
        out = np.zeros(max_new, dtype=np.uint32)
        n = C.c_uint32()
        sp = Sampling(top_k, top_p, temperature, repetition_penalty)
        self._chk(self.L.tts_hip_orpheus_generate_sampled(self.ctx, ap, a.size, max_new, stop_id, C.byref(sp), u.ctypes.data_as(C.POINTER(C.c_float)),
                                                          out.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(n)))
        return out[:n.value]

    def step_batch(self, slots, ids, pos, want_logits=True):
        """one lock-step forward: row r = token ids[r] of the utterance in cache slot slots[r] at position pos[r] -> (logits [n][vocab] or None, arg-max [n])"""
        s, sp = _u32(slots)
        a, ap = _u32(ids)
        b, bp = _u32(pos)
        n = a.size
        lg = np.empty((n, self.cfg.vocab), dtype=np.float32) if want_logits else None
        tok = np.empty(n, dtype=np.uint32)
        self._chk(self.L.tts_hip_orpheus_step_batch(self.ctx, n, sp, ap, bp, lg.ctypes.data_as(C.POINTER(C.c_float)) if want_logits else None,
                                                    tok.ctypes.data_as(C.POINTER(C.c_uint32))))
        return lg, tok

    def generate_batch(self, prompts, max_new, stop_id, uniforms=None, top_k=50, temperature=1.0, repetition_penalty=1.0, top_p=1.0):
        """tts_hip_orpheus_generate_batch: one id list per utterance; uniforms [n_utt][max_new] selects sampler::sample, None = greedy -> list of id arrays"""
        n = len(prompts)
        cat, cp = _u32(np.concatenate([np.asarray(p, dtype=np.uint32) for p in prompts]))
        lens, lp = _u32(np.array([len(p) for p in prompts], dtype=np.uint32))
        out = np.zeros((n, max_new), dtype=np.uint32)
        cnt = np.zeros(n, dtype=np.uint32)
        spp, up = None, None
        if uniforms is not None:
            u = np.ascontiguousarray(uniforms, dtype=np.float32).reshape(n, max_new)
            sp_ = Sampling(top_k, top_p, temperature, repetition_penalty)
            spp, up = C.byref(sp_), u.ctypes.data_as(C.POINTER(C.c_float))
        self._chk(self.L.tts_hip_orpheus_generate_batch(self.ctx, n, cp, lp, max_new, stop_id, spp, up, out.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                        cnt.ctypes.data_as(C.POINTER(C.c_uint32))))
        return [out[i, :cnt[i]].copy() for i in range(n)]

    def sample_logits(self, logits, uniform, top_k=50, temperature=1.0, repetition_penalty=1.0, top_p=1.0, last_id=-1, rep_count=0):
        """the device sampler on caller-supplied logits [vocab] -> (token, last_id, rep_count)"""
        lg = np.ascontiguousarray(logits, dtype=np.float32)
        assert lg.size == self.cfg.vocab
        sp = Sampling(top_k, top_p, temperature, repetition_penalty)
        li, rc, tok = C.c_int32(last_id), C.c_uint32(rep_count), C.c_uint32()
        self._chk(self.L.tts_hip_orpheus_sample_logits(self.ctx, lg.ctypes.data_as(C.POINTER(C.c_float)), C.byref(sp), uniform, C.byref(li), C.byref(rc), C.byref(tok)))
        return tok.value, li.value, rc.value

    def close(self):
        if self.ctx:
            self.L.tts_hip_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DiaCodes(C.Structure):
    """tts_hip_dia_codes (include/tts_hip.h)"""
    _fields_ = [("bos", C.c_uint32), ("eos", C.c_uint32), ("pad", C.c_uint32), ("max_delay", C.c_uint32), ("delay_pattern", C.c_uint32 * 16)]


class DiaEngine:
    """A Dia context (tts_hip_dia_create): encoder + cross K/V once per sentence, then one decoder step per call."""

    def __init__(self, cfg, device=0, flags=0, cfg_scale=0.0, max_utterances=1):
        self.L = load_lib()
        self.cfg = cfg
        d = DiaDesc()
        d.max_utterances = max_utterances
        d.struct_size = C.sizeof(DiaDesc)
        d.enc_hidden_size, d.enc_layers, d.enc_attn_heads = cfg.enc_hidden, cfg.enc_layers, cfg.enc_heads
        d.dec_hidden_size, d.dec_layers, d.dec_attn_heads, d.dec_kv_heads = cfg.dec_hidden, cfg.dec_layers, cfg.dec_heads, cfg.dec_kv_heads
        d.head_dim, d.n_output_heads, d.output_vocab_size, d.max_ctx, d.max_gen = cfg.head_dim, cfg.n_out, cfg.out_vocab, cfg.max_ctx, cfg.max_gen
        d.cfg_scale, d.flags = cfg_scale, flags
        self.ctx = self.L.tts_hip_dia_create(device, C.byref(d))
        if not self.ctx:
            raise HipError(self.L.tts_hip_last_error().decode("utf-8", "replace"))

    def _chk(self, rc):
        if rc != 0:
            raise HipError(self.L.tts_hip_last_error().decode("utf-8", "replace"))

    def tune(self, key, value):
        """tts_hip_tune: a named tuning / fallback switch (before the first launch)"""
        self._chk(self.L.tts_hip_tune(self.ctx, key.encode(), int(value)))

    def load(self, model, declare_only=False):
        """declare_only: lay the arena out without uploading (the bytes arrive by tts_hip_broadcast_weights_rank / tts_hip_arena_filled)"""
        for t in model.tensors:
            ne = (C.c_int64 * 4)(*(t.ne + [1] * (4 - len(t.ne))))
            raw = None if declare_only else np.frombuffer(bytes(t.raw()), dtype=np.uint8)
            self._chk(self.L.tts_hip_upload(self.ctx, t.name.encode(), t.type, len(t.ne), ne, None if declare_only else raw.ctypes.data_as(C.c_void_p)))
        self._chk(self.L.tts_hip_finalize(self.ctx, None))

    def synchronize(self):
        self._chk(self.L.tts_hip_synchronize(self.ctx))

    def encode(self, tokens, sentence_len, want_states=False):
        a, ap = _u32(tokens)
        assert a.size == self.cfg.max_ctx
        out = np.empty((2, self.cfg.max_ctx, self.cfg.enc_hidden), dtype=np.float32) if want_states else None
        self._chk(self.L.tts_hip_dia_encode(self.ctx, ap, sentence_len, out.ctypes.data_as(C.POINTER(C.c_float)) if want_states else None))
        return out

    def encode_slot(self, slot, tokens, sentence_len):
        a, ap = _u32(tokens)
        assert a.size == self.cfg.max_ctx
        self._chk(self.L.tts_hip_dia_encode_slot(self.ctx, slot, ap, sentence_len, None))

    def step_batch(self, ids, pos, slots=None, want_raw=False):
        """ids [n_utt][n_out], pos [n_utt] -> guided logits [n_utt][n_out][vocab] (and raw [n_utt][2][n_out][vocab])"""
        a, ap = _u32(np.asarray(ids).reshape(-1))
        p, pp = _u32(pos)
        n = p.size
        sl = _u32(slots)[1] if slots is not None else None
        lg = np.empty((n, self.cfg.n_out, self.cfg.out_vocab), dtype=np.float32)
        raw = np.empty((n, 2, self.cfg.n_out, self.cfg.out_vocab), dtype=np.float32) if want_raw else None
        self._chk(self.L.tts_hip_dia_step_batch(self.ctx, n, sl, ap, pp, lg.ctypes.data_as(C.POINTER(C.c_float)),
                                                raw.ctypes.data_as(C.POINTER(C.c_float)) if want_raw else None))
        return (lg, raw) if want_raw else lg

    def generate(self, n_utt, max_gen, delay_pattern, bos, eos, pad, max_delay, uniforms=None, top_k=50, top_p=1.0, temperature=1.0, repetition_penalty=1.0):
        """the whole generation loop on the device (tts_hip_dia_generate) for the encoded slots 0..n_utt-1; uniforms None: sampler::max.
        -> list of [steps][n_out] id arrays (generation order, before adjust_output_tokens)"""
        codes = DiaCodes(bos, eos, pad, max_delay)
        for i, d in enumerate(delay_pattern):
            codes.delay_pattern[i] = int(d)
        out = np.zeros((n_utt, max_gen, self.cfg.n_out), dtype=np.uint32)
        steps = np.zeros(n_utt, dtype=np.uint32)
        sp, up = None, None
        if uniforms is not None:
            u = np.ascontiguousarray(uniforms, dtype=np.float32)
            assert u.size >= max_gen * n_utt * self.cfg.n_out
            sp, up = C.byref(Sampling(top_k, top_p, temperature, repetition_penalty)), u.ctypes.data_as(C.POINTER(C.c_float))
        self._chk(self.L.tts_hip_dia_generate(self.ctx, n_utt, max_gen, C.byref(codes), sp, up, out.ctypes.data_as(C.POINTER(C.c_uint32)),
                                              steps.ctypes.data_as(C.POINTER(C.c_uint32))))
        return [out[u, :int(steps[u])].copy() for u in range(n_utt)]

    def step(self, ids, pos, want_raw=False):
        a, ap = _u32(ids)
        lg = np.empty((self.cfg.n_out, self.cfg.out_vocab), dtype=np.float32)
        raw = np.empty((2, self.cfg.n_out, self.cfg.out_vocab), dtype=np.float32) if want_raw else None
        self._chk(self.L.tts_hip_dia_step(self.ctx, ap, pos, lg.ctypes.data_as(C.POINTER(C.c_float)),
                                          raw.ctypes.data_as(C.POINTER(C.c_float)) if want_raw else None))
        return (lg, raw) if want_raw else lg

    def close(self):
        if self.ctx:
            self.L.tts_hip_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class KokoroEngine:
    """A Kokoro context (tts_hip_kokoro_create): duration graph, then generation graph, per clause."""

    def __init__(self, model, device=0, attn_scale=0.125, tune=None):
        self.L = load_lib()
        self._tune = dict(tune or {})
        cfg = model.cfg
        self.cfg = cfg
        d = KokoroDesc()
        d.struct_size = C.sizeof(KokoroDesc)
        d.n_attn_heads, d.n_recurrence, d.n_dp_layers, d.f0_n_blocks, d.n_conv_layers = cfg.heads, cfg.recurrence, cfg.dp_layers, cfg.f0_blocks, cfg.conv_layers
        d.n_decoder_blocks, d.n_upsamples, d.n_kernels = cfg.decoder_blocks, len(cfg.up_rates), len(cfg.res_kernels)
        d.n_fft, d.hop, d.harmonic_num, d.up_sampling_factor, d.out_conv_padding, d.max_ctx = cfg.n_fft, cfg.hop, cfg.harmonic_num, cfg.up_sampling_factor, 3, cfg.max_ctx
        d.attn_scale, d.upsample_scale = attn_scale, float(np.prod(cfg.up_rates) * cfg.hop)
        d.sample_rate, d.sin_amp, d.noise_std, d.voice_threshold = 24000.0, 0.1, 0.003, 10.0     # kokoro/model.h:219-222
        g = model.geometry
        for i, (st, pd) in enumerate(g["up"]):
            d.up_stride[i], d.up_padding[i] = st, pd
        for i, (st, pd) in enumerate(g["noise"]):
            d.noise_stride[i], d.noise_padding[i] = st, pd
        for i, blk in enumerate(g["res"]):
            for ii, (pd, dl) in enumerate(blk):
                d.res_padding[i][ii], d.res_dilation[i][ii] = pd, dl
        for i, blk in enumerate(g["noise_res"]):
            for ii, (pd, dl) in enumerate(blk):
                d.noise_res_padding[i][ii], d.noise_res_dilation[i][ii] = pd, dl
        self.ctx = self.L.tts_hip_kokoro_create(device, C.byref(d))
        if not self.ctx:
            raise HipError(self.L.tts_hip_last_error().decode("utf-8", "replace"))
        for kk, vv in self._tune.items():   # tts_hip_tune keys (fallback kernel paths), before the first launch
            if self.L.tts_hip_tune(self.ctx, kk.encode(), int(vv)) != 0:
                raise HipError(self.L.tts_hip_last_error().decode("utf-8", "replace"))
        for t in model.tensors:
            ne = (C.c_int64 * 4)(*(t.ne + [1] * (4 - len(t.ne))))
            raw = np.frombuffer(bytes(t.raw()), dtype=np.uint8)
            self._chk(self.L.tts_hip_upload(self.ctx, t.name.encode(), t.type, len(t.ne), ne, raw.ctypes.data_as(C.c_void_p)))
        self._chk(self.L.tts_hip_finalize(self.ctx, None))

    def _chk(self, rc):
        if rc != 0:
            raise HipError(self.L.tts_hip_last_error().decode("utf-8", "replace"))

    def durations(self, tokens, voice):
        a, ap = _u32(tokens)
        lens = np.empty(a.size, dtype=np.float32)
        hid = np.empty((a.size, self.cfg.dp_hidden + self.cfg.style_half), dtype=np.float32)
        fpt = C.POINTER(C.c_float)
        self._chk(self.L.tts_hip_kokoro_durations(self.ctx, ap, a.size, voice.encode(), lens.ctypes.data_as(fpt), hid.ctypes.data_as(fpt)))
        return lens, hid

    def generate(self, tokens, lens, hidden, voice, noise, hsrc_in=None, want_hsrc=False):
        a, ap = _u32(tokens)
        fpt = C.POINTER(C.c_float)
        lens = np.ascontiguousarray(lens, dtype=np.float32)
        hidden = np.ascontiguousarray(hidden, dtype=np.float32)
        noise = np.ascontiguousarray(noise, dtype=np.float32)
        total = int(lens.sum())
        pcm = np.empty(total * self.cfg.up_sampling_factor, dtype=np.float32)
        hs = np.empty((2 * (self.cfg.n_fft // 2 + 1), 2 * total * int(np.prod(self.cfg.up_rates)) + 1), dtype=np.float32)
        hin = None if hsrc_in is None else np.ascontiguousarray(hsrc_in, dtype=np.float32)
        self._chk(self.L.tts_hip_kokoro_generate(self.ctx, ap, a.size, lens.ctypes.data_as(fpt), hidden.ctypes.data_as(fpt), voice.encode(), noise.ctypes.data_as(fpt),
                                                 pcm.ctypes.data_as(fpt), hs.ctypes.data_as(fpt), hin.ctypes.data_as(fpt) if hin is not None else None))
        return (pcm, hs) if want_hsrc else pcm

    def close(self):
        if self.ctx:
            self.L.tts_hip_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

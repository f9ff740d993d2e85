"""Synthetic Parler-TTS + DAC models with the reference's exact GGUF tensor names and KV keys.

There are no real checkpoints in the build/bench environment (no network), so tests and bench run
on seeded random weights of the right architecture.  Names/keys follow
py-gguf/tts_encoders/parler_tts_gguf_encoder.py:85-180 and dac_gguf_encoder.py:7-110 and what the
C++ loaders look up (src/models/parler/model.cpp:4-28,51-108; src/decoder/dac_model.cpp:7-55).
"""
import math
from dataclasses import dataclass, field

import numpy as np

from . import gguf


@dataclass
class Config:
    hidden: int = 1024
    layers: int = 24
    heads: int = 16
    ffn: int = 4096
    out_vocab: int = 1088
    audio_vocab: int = 1024
    n_out: int = 9
    ctx: int = 4096
    max_gen: int = 2580
    enc_len: int = 8
    prompt_vocab: int = 32128
    eos: int = 1024
    bos: int = 1025
    # DAC 44 kHz
    latent: int = 1024
    cb_dim: int = 8
    cb_size: int = 1024
    c0: int = 1536
    strides: tuple = (8, 8, 4, 2)
    seed: int = 0xC0FFEE
    weight_type: int = gguf.F16  # type of the quantisable decoder tensors
    dac_f16: bool = False
    suppress_special: bool = True
    paddings: tuple = field(default=None)

    def __post_init__(self):
        if self.paddings is None:
            self.paddings = tuple(math.ceil(s / 2) for s in self.strides)

    @property
    def hop(self):
        return int(np.prod(self.strides))


def parler_mini(**kw):
    return Config(**kw)


def tiny(**kw):
    base = dict(hidden=256, layers=2, heads=4, ffn=512, out_vocab=80, audio_vocab=64, n_out=4, ctx=128,
                max_gen=96, enc_len=6, prompt_vocab=160, eos=64, bos=65, latent=64, cb_dim=8, cb_size=64,
                c0=96, strides=(4, 2))
    base.update(kw)
    return Config(**base)


def small(**kw):
    """Mid-size twin: real head/vocab/codebook structure (9 heads, V=1088, 4 DAC blocks), thin layers."""
    base = dict(hidden=512, layers=3, heads=8, ffn=1024, ctx=512, max_gen=256, enc_len=8, prompt_vocab=512,
                latent=128, c0=192, strides=(8, 8, 4, 2))
    base.update(kw)
    return Config(**base)


# --------------------------------------------------------------------------------------------------
# numpy block quantisers (same formats as ggml's quantize_row_q{4_0,5_0,8_0}_ref).
# Used only to mint quantised test models; tests cross-check them against oracle/orc_quantize.
# --------------------------------------------------------------------------------------------------
def _f16_bits(x):
    return x.astype("<f2").view("<u2")


class _Pool:
    """Full-depth models at the real widths (tests/test_gpu_full_depth.py): minting 1.6-3.3 G fresh normals (and quantising them) costs minutes of box time, so
    `pooled` models cut every matrix out of one buffer of 2^25 standard normals at a random offset (tiled when the matrix is longer), and quantised matrices are
    minted as random blocks directly (codes uniform, one fp16 scale per block around std / rms(code)).  Small tensors (norms) stay fresh draws."""

    def __init__(self, rng):
        self.rng = rng
        self.buf = rng.standard_normal(1 << 25, dtype=np.float32)

    def normal(self, shape, std):
        n = int(np.prod(shape))
        if n <= self.buf.size:
            off = int(self.rng.integers(0, self.buf.size - n + 1))
            return (self.buf[off:off + n] * np.float32(std)).reshape(shape)
        off = int(self.rng.integers(0, self.buf.size))
        return (np.tile(self.buf, -(-(n + off) // self.buf.size))[off:off + n] * np.float32(std)).reshape(shape)

    def blocks(self, shape, std, ttype):
        nb = int(np.prod(shape)) // 32
        width, rms = {gguf.Q4_0: (18, 4.63), gguf.Q5_0: (22, 9.24), gguf.Q8_0: (34, 73.9)}[ttype]
        out = self.rng.integers(0, 256, size=(nb, width), dtype=np.uint8)
        d = (np.float32(std / rms) * (0.75 + 0.5 * self.rng.random(nb, dtype=np.float32))).astype(np.float16)
        out[:, 0:2] = d.reshape(-1, 1).view(np.uint8)
        if ttype == gguf.Q8_0:   # -128 is not a code quantize_row_q8_0 produces
            body = out[:, 2:]
            body[body == 0x80] = 0x81
        return out.reshape(-1)


def quantize(arr, ttype):
    a = np.ascontiguousarray(arr, dtype=np.float32).reshape(-1, 32)
    nb = a.shape[0]
    if ttype == gguf.Q8_0:
        amax = np.abs(a).max(axis=1)
        d = (amax / np.float32(127.0)).astype(np.float32)
        idv = np.where(d != 0, np.float32(1.0) / np.where(d != 0, d, 1), 0).astype(np.float32)
        x = a * idv[:, None]
        q = np.where(x >= 0, np.floor(x + np.float32(0.5)), np.ceil(x - np.float32(0.5))).astype(np.int8)  # roundf
        out = np.zeros((nb, 34), dtype=np.uint8)
        out[:, 0:2] = _f16_bits(d).reshape(-1, 1).view(np.uint8)
        out[:, 2:] = q.view(np.uint8)
        return out.reshape(-1)
    idx = np.abs(a).argmax(axis=1)
    mx = a[np.arange(nb), idx]
    if ttype == gguf.Q4_0:
        d = (mx / np.float32(-8.0)).astype(np.float32)
        idv = np.where(d != 0, np.float32(1.0) / np.where(d != 0, d, 1), 0).astype(np.float32)
        x = a * idv[:, None]
        q = np.minimum(15, (x + np.float32(8.5)).astype(np.int32).astype(np.int8).astype(np.uint8)).astype(np.uint8)
        out = np.zeros((nb, 18), dtype=np.uint8)
        out[:, 0:2] = _f16_bits(d).reshape(-1, 1).view(np.uint8)
        out[:, 2:] = q[:, :16] | (q[:, 16:] << 4)
        return out.reshape(-1)
    if ttype == gguf.Q5_0:
        d = (mx / np.float32(-16.0)).astype(np.float32)
        idv = np.where(d != 0, np.float32(1.0) / np.where(d != 0, d, 1), 0).astype(np.float32)
        x = a * idv[:, None]
        q = np.minimum(31, (x + np.float32(16.5)).astype(np.int32).astype(np.int8).astype(np.uint8)).astype(np.uint32)
        qh = np.zeros(nb, dtype=np.uint32)
        for j in range(16):
            qh |= ((q[:, j] & 0x10) >> 4) << j
            qh |= ((q[:, j + 16] & 0x10) >> 4) << (j + 16)
        out = np.zeros((nb, 22), dtype=np.uint8)
        out[:, 0:2] = _f16_bits(d).reshape(-1, 1).view(np.uint8)
        out[:, 2:6] = qh.astype("<u4").reshape(-1, 1).view(np.uint8)
        out[:, 6:] = ((q[:, :16] & 0x0F) | ((q[:, 16:] & 0x0F) << 4)).astype(np.uint8)
        return out.reshape(-1)
    raise ValueError(ttype)


def _quantizable(name):
    """parler_is_quanitizable with --quantize-output-heads/-text-embedding/-cross-attn-kv all on
    (examples/quantize/quantize_impl.cpp:51-67)."""
    return not (name.startswith("audio_encoder") or name.endswith("norm.weight") or name.endswith("text_encoding")
                or name.endswith("positional_embed") or name.endswith("norm.bias"))


def _sinusoid(n_pos, dim):
    half = dim // 2
    freq = np.exp(np.arange(half, dtype=np.float64) * -(math.log(10000.0) / half))
    ang = np.arange(n_pos, dtype=np.float64)[:, None] * freq[None, :]
    return np.concatenate([np.cos(ang), np.sin(ang)], axis=1).astype(np.float32)


def _vocab(n, rng):
    """Synthetic unigram vocabulary: specials, printable single characters, then random 2-4 grams."""
    toks = ["<pad>", "</s>", "<unk>"]
    singles = [" "] + [chr(c) for c in range(ord("a"), ord("z") + 1)] + list(".,'!?-") + [chr(c) for c in range(ord("A"), ord("Z") + 1)]
    toks += singles
    seen = set(toks)
    letters = "abcdefghijklmnopqrstuvwxyz"
    while len(toks) < n:
        ln = int(rng.integers(2, 5))
        s = "".join(letters[int(i)] for i in rng.integers(0, 26, ln))
        if rng.random() < 0.4:
            s = " " + s
        if s not in seen:
            seen.add(s)
            toks.append(s)
    toks = toks[:n]
    scores = np.zeros(n, dtype=np.float32)
    for i, t in enumerate(toks):
        scores[i] = 0.0 if i < 3 else -(2.0 + 1.5 * len(t)) + float(rng.random())
    scores[2] = -20.0  # <unk>
    return toks, scores


class TensorDecl:
    """shape-only stand-in for gguf.Tensor (ranks that receive the weights by RCCL broadcast)"""

    def __init__(self, name, ttype, ne):
        self.name, self.type, self.ne = name, int(ttype), [int(x) for x in ne]


class SynthModel:
    """tensors: list[gguf.Tensor] in file order; kv: list of (key, type, value); cfg: Config.
    shapes_only=True builds TensorDecl entries without generating any data."""

    def __init__(self, cfg: Config, shapes_only=False):
        self.cfg = cfg
        self.shapes_only = shapes_only
        rng = np.random.Generator(np.random.Philox(cfg.seed))
        self.rng = rng
        self.tensors = []
        self.f32 = {}  # name -> fp32 array in PyTorch order, AFTER rounding to the stored type where exact
        H, F = cfg.hidden, cfg.ffn

        class _Shape:  # array stand-in that only carries a shape
            def __init__(self, shape):
                self.shape = tuple(shape)

            def __mul__(self, other):
                return self

            __rmul__ = __mul__

            def astype(self, dt):
                return self

            def __setitem__(self, k, v):
                pass

        if shapes_only:
            class _R:
                def standard_normal(self, shape, dtype=None): return _Shape(shape if isinstance(shape, tuple) else (shape,))
                def uniform(self, a, b, shape): return _Shape(shape)
                def integers(self, *a, **k): return np.random.default_rng(0).integers(*a, **k)
                def random(self): return 0.5
            rng = _R()

        def normal(shape, std=0.02, mean=0.0):
            if shapes_only:
                return _Shape(shape)
            return (rng.standard_normal(shape, dtype=np.float32) * np.float32(std) + np.float32(mean)).astype(np.float32)

        def add(name, arr):
            if shapes_only:
                ttype = gguf.F32
                if _quantizable(name):
                    ttype = cfg.weight_type
                elif name.startswith("audio_encoder") and cfg.dac_f16 and not name.endswith("alpha"):
                    ttype = gguf.F16
                self.tensors.append(TensorDecl(name, ttype, list(reversed(arr.shape))))
                return
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            ttype = gguf.F32
            if _quantizable(name):
                ttype = cfg.weight_type
            elif name.startswith("audio_encoder") and cfg.dac_f16 and not name.endswith("alpha"):
                ttype = gguf.F16  # --convert-dac-to-f16 keeps alpha F32 (quantize_impl.cpp:264-266)
            ne = list(reversed(arr.shape))
            if ttype in (gguf.Q4_0, gguf.Q5_0, gguf.Q8_0):
                self.tensors.append(gguf.Tensor(name, ttype, ne, quantize(arr, ttype).tobytes()))
            else:
                self.tensors.append(gguf.Tensor.from_array(name, arr, ttype))

        # ---- voice prompt encoding + DAC (prepare_text_encoding_tensors / prepare_dac_audio_encoder_tensors)
        add("decoder.text_encoding", normal((cfg.enc_len, H), std=0.5))
        chans = [cfg.c0]
        for _ in cfg.strides:
            chans.append(chans[-1] // 2)

        def conv_w(cout, cin, k):
            b = 1.0 / math.sqrt(cin * k)
            return rng.uniform(-b, b, (cout, cin, k)).astype(np.float32)

        add("audio_encoder.initial.bias", rng.uniform(-0.05, 0.05, (cfg.c0,)).astype(np.float32))
        add("audio_encoder.initial.weight", conv_w(cfg.c0, cfg.latent, 7))
        for bi, s in enumerate(cfg.strides):
            cin, cout = chans[bi], chans[bi + 1]
            p = f"audio_encoder.decoder_block.{bi + 1}."
            add(p + "final.alpha", rng.uniform(0.5, 2.0, (1, cin, 1)).astype(np.float32))
            add(p + "final.bias", rng.uniform(-0.05, 0.05, (cout,)).astype(np.float32))
            bnd = 1.0 / math.sqrt(cin * 2)
            add(p + "final.weight", rng.uniform(-bnd, bnd, (cin, cout, 2 * s)).astype(np.float32))
            for r in range(3):
                q = p + f"residual_unit.{r}.res."
                add(q + "initial.alpha", rng.uniform(0.5, 2.0, (1, cout, 1)).astype(np.float32))
                add(q + "initial.bias", rng.uniform(-0.05, 0.05, (cout,)).astype(np.float32))
                add(q + "initial.weight", conv_w(cout, cout, 7))
                add(q + "final.alpha", rng.uniform(0.5, 2.0, (1, cout, 1)).astype(np.float32))
                add(q + "final.bias", rng.uniform(-0.05, 0.05, (cout,)).astype(np.float32))
                add(q + "final.weight", conv_w(cout, cout, 1) * np.float32(0.5))
        add("audio_encoder.final.alpha", rng.uniform(0.5, 2.0, (1, chans[-1], 1)).astype(np.float32))
        add("audio_encoder.final.bias", rng.uniform(-0.05, 0.05, (1,)).astype(np.float32))
        add("audio_encoder.final.weight", conv_w(1, chans[-1], 7))
        for i in range(cfg.n_out):
            p = f"audio_encoder.quantizers.{i}."
            add(p + "codebook.weight", normal((cfg.cb_size, cfg.cb_dim), std=1.0))
            add(p + "out_proj.bias", rng.uniform(-0.05, 0.05, (cfg.latent,)).astype(np.float32))
            add(p + "out_proj.weight", conv_w(cfg.latent, cfg.cb_dim, 1))

        # ---- decoder (prepare_decoder_tensors)
        add("decoder.embed_prompts", normal((cfg.prompt_vocab, H), std=0.02) * np.float32(1.0))
        add("decoder.positional_embed", _sinusoid(cfg.ctx, H) * np.float32(0.02))
        for i in range(cfg.n_out):
            add(f"decoder.embed_tokens.{i}.weight", normal((cfg.out_vocab + 1, H), std=0.02))
        for l in range(cfg.layers):
            p = f"decoder.layers.{l}."
            for blk in ("self_attn", "encoder_attn"):
                for proj in ("k_proj", "v_proj", "q_proj", "out_proj"):
                    add(p + f"{blk}.{proj}.weight", normal((H, H)))
                add(p + f"{blk}_layer_norm.weight", normal((H,), mean=1.0))
                add(p + f"{blk}_layer_norm.bias", normal((H,)))
            add(p + "fc1.weight", normal((F, H)))
            add(p + "fc2.weight", normal((H, F)))
            add(p + "final_layer_norm.weight", normal((H,), mean=1.0))
            add(p + "final_layer_norm.bias", normal((H,)))
        add("decoder.layer_norm.weight", normal((H,), mean=1.0))
        add("decoder.layer_norm.bias", normal((H,)))
        for i in range(cfg.n_out):
            hw = normal((cfg.out_vocab, H), std=0.05)
            if cfg.suppress_special:
                # rows of the special ids (EOS/BOS/pad, >= audio_vocab) are zero: random weights then never
                # emit them, so synthetic generation has a fixed length and every id is a valid DAC code
                hw[cfg.audio_vocab:] = 0.0
            add(f"decoder.lm_heads.{i}.weight.head", hw)

        toks, scores = _vocab(cfg.prompt_vocab, np.random.Generator(np.random.Philox(cfg.seed + 1)))
        self.vocab, self.scores = toks, scores
        U32, STR, ARR, F32T = gguf.T_U32, gguf.T_STR, gguf.T_ARR, gguf.T_F32
        self.kv = [
            ("general.architecture", STR, "parler-tts"),
            ("general.name", STR, "synthetic-parler"),
            ("parler-tts.decoder.encode_length", U32, cfg.enc_len),
            ("dac.up_scaling_factor", U32, cfg.hop),  # written but never read
        ]
        for i, (s, p) in enumerate(zip(cfg.strides, cfg.paddings)):
            self.kv.append((f"dac.dac_layer_stride_{i}", U32, int(s)))
            self.kv.append((f"dac.dac_layer_padding_{i}", U32, int(p)))
        self.kv += [
            ("audio.bos_token_id", U32, cfg.bos),
            ("audio.eos_token_id", U32, cfg.eos),
            ("parler-tts.decoder.hidden_size", U32, H),
            ("parler-tts.decoder.output_heads", U32, cfg.n_out),
            ("parler-tts.decoder.context_length", U32, cfg.ctx),
            ("parler-tts.decoder.attention.head_count", U32, cfg.heads),
            ("parler-tts.decoder.max_generation", U32, cfg.max_gen),
            ("parler-tts.decoder.out_vocab_size", U32, cfg.out_vocab),
            ("parler-tts.decoder.audio_vocab_size", U32, cfg.audio_vocab),
            ("parler-tts.decoder.num_hidden_layers", U32, cfg.layers),
            ("tokenizer.ggml.tokens", ARR, (STR, toks)),
            ("tokenizer.ggml.scores", ARR, (F32T, scores)),
            ("tokenizer.ggml.eos_token_id", U32, 1),
            ("tokenizer.ggml.unknown_token_id", U32, 2),
        ]
        self.by_name = {t.name: t for t in self.tensors}

    def write_gguf(self, path):
        gguf.write(path, self.kv, self.tensors)
        return path


def build(cfg: Config, shapes_only=False) -> SynthModel:
    return SynthModel(cfg, shapes_only=shapes_only)


# --------------------------------------------------------------------------------------------------
# T5 voice-prompt encoder (the model update_conditional_prompt loads; src/models/parler/t5/model.cpp).
# Tensor names / keys as py-gguf/tts_encoders/t5_encoder_gguf_encoder.py:73-90 writes them.
# --------------------------------------------------------------------------------------------------
@dataclass
class T5Config:
    hidden: int = 256
    layers: int = 2
    heads: int = 4           # head size is 64 like flan-t5 (t5/model.h:46)
    ffn: int = 512
    vocab: int = 160         # == the decoder's prompt vocabulary: the Parler runner's tokenizer is reused (model.cpp:513)
    ctx: int = 64            # t5encoder.context_length
    buckets: int = 32        # relative_attn_buckets (t5/model.h:48)
    output_size: int = 256   # == decoder hidden size; a down projection is stored when it differs from `hidden`
    weight_type: int = gguf.F32
    seed: int = 0x75


def t5_tiny(**kw):
    return T5Config(**kw)


def t5_flan_large(**kw):
    """the encoder of parler-tts-mini-v1 (google/flan-t5-large: 24 layers, d_model 1024, 16 heads, d_ff 2816)"""
    base = dict(hidden=1024, layers=24, heads=16, ffn=2816, vocab=32128, ctx=512, output_size=1024)
    base.update(kw)
    return T5Config(**base)


class SynthT5:
    def __init__(self, cfg: T5Config):
        self.cfg = cfg
        rng = np.random.Generator(np.random.Philox(cfg.seed))
        self.tensors = []
        H, F = cfg.hidden, cfg.ffn

        def normal(shape, std):
            return (rng.standard_normal(shape, dtype=np.float32) * np.float32(std)).astype(np.float32)

        def add(name, arr, quantizable=True):
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            ttype = cfg.weight_type if quantizable else gguf.F32
            ne = list(reversed(arr.shape))
            if ttype in (gguf.Q4_0, gguf.Q5_0, gguf.Q8_0):
                self.tensors.append(gguf.Tensor(name, ttype, ne, quantize(arr, ttype).tobytes()))
            else:
                self.tensors.append(gguf.Tensor.from_array(name, arr, ttype))

        if cfg.output_size != H:
            add("t5encoder.down_proj", normal((cfg.output_size, H), 1.0 / math.sqrt(H)))
            add("t5encoder.down_proj_bias", normal((cfg.output_size,), 0.02), quantizable=False)
        add("t5encoder.token_embd", normal((cfg.vocab, H), 1.0))
        add("t5encoder.enc.final_layer_norm", 1.0 + normal((H,), 0.05), quantizable=False)
        for i in range(cfg.layers):
            p = f"t5encoder.enc.blk.{i}."
            if i == 0:
                add(p + "attn_rel_b", normal((cfg.buckets, cfg.heads), 0.5), quantizable=False)
            for nm in ("attn_q", "attn_k", "attn_v", "attn_o"):
                # T5 attention has no 1/sqrt(d) (t5/model.cpp:258: soft_max_ext scale 1.0): keep the logits tame
                add(p + nm, normal((H, H), 0.3 / math.sqrt(H) if nm in ("attn_q", "attn_k") else 1.0 / math.sqrt(H)))
            add(p + "attn_norm", 1.0 + normal((H,), 0.05), quantizable=False)
            add(p + "ffn_up", normal((F, H), 1.0 / math.sqrt(H)))
            add(p + "ffn_gate", normal((F, H), 1.0 / math.sqrt(H)))
            add(p + "ffn_down", normal((H, F), 1.0 / math.sqrt(F)))
            add(p + "ffn_norm", 1.0 + normal((H,), 0.05), quantizable=False)
        U32, STR = gguf.T_U32, gguf.T_STR
        self.kv = [
            ("general.architecture", STR, "t5encoder"),
            ("general.name", STR, "synthetic-t5-encoder"),
            ("t5encoder.block_count", U32, cfg.layers),
            ("t5encoder.embedding_length", U32, H),
            ("t5encoder.attention.head_count", U32, cfg.heads),
            ("t5encoder.context_length", U32, cfg.ctx),
            ("t5encoder.vocab_size", U32, cfg.vocab),
            ("t5encoder.output_size", U32, cfg.output_size),
            ("tokenizer.ggml.bos_token_id", U32, 0),
            ("tokenizer.ggml.eos_token_id", U32, 1),
        ]
        self.by_name = {t.name: t for t in self.tensors}

    def write_gguf(self, path):
        gguf.write(path, self.kv, self.tensors)
        return path


def build_t5(cfg: T5Config) -> SynthT5:
    return SynthT5(cfg)


# --------------------------------------------------------------------------------------------------
# SNAC codec (Orpheus' audio decoder; src/decoder/snac_model.cpp).  Tensor names as
# py-gguf/tts_encoders/orpheus_gguf_encoder.py:89-142 writes them ("snac." + simplify_snac_name), KV keys :214-220.
# --------------------------------------------------------------------------------------------------
@dataclass
class SnacConfig:
    latent: int = 64          # snac_model::embd (768)
    c0: int = 96              # channels after the `up` conv (1024)
    strides: tuple = (4, 2)   # 8, 8, 4, 2
    cb_dim: int = 8
    cb_size: int = 64         # 4096
    repeats: tuple = (4, 2, 1)
    max_frames: int = 64      # snac.max_generation_size (finest-scale tokens)
    seed: int = 0x5AC

    @property
    def paddings(self):
        return tuple((s + 1) // 2 for s in self.strides)   # SNAC DecoderBlock: ConvTranspose1d(k=2s, stride s, padding ceil(s/2))

    @property
    def hop(self):
        return int(np.prod(self.strides))


def snac_tiny(**kw):
    return SnacConfig(**kw)


def snac_24khz(**kw):
    base = dict(latent=768, c0=1024, strides=(8, 8, 4, 2), cb_dim=8, cb_size=4096, max_frames=2580)
    base.update(kw)
    return SnacConfig(**base)


class SynthSnac:
    def __init__(self, cfg: SnacConfig):
        self.cfg = cfg
        rng = np.random.Generator(np.random.Philox(cfg.seed))
        self.tensors = []

        def add(name, arr):
            self.tensors.append(gguf.Tensor.from_array("snac." + name, np.ascontiguousarray(arr, dtype=np.float32), gguf.F32))

        def conv_w(cout, cin, k):
            b = 1.0 / math.sqrt(cin * k)
            return rng.uniform(-b, b, (cout, cin, k)).astype(np.float32)

        def bias(n):
            return rng.uniform(-0.05, 0.05, (n,)).astype(np.float32)

        def alpha(n):
            return rng.uniform(0.5, 2.0, (1, n, 1)).astype(np.float32)

        for i in range(len(cfg.repeats)):
            p = f"quantizers.{i}."
            add(p + "codebook.weight", (rng.standard_normal((cfg.cb_size, cfg.cb_dim), dtype=np.float32)).astype(np.float32))
            add(p + "out_proj.bias", bias(cfg.latent))
            add(p + "out_proj.weight", conv_w(cfg.latent, cfg.cb_dim, 1))
        add("in.weight", conv_w(cfg.latent, 1, 7))          # depthwise: [C][1][7]
        add("in.bias", bias(cfg.latent))
        add("up.weight", conv_w(cfg.c0, cfg.latent, 1))
        add("up.bias", bias(cfg.c0))
        c = cfg.c0
        for li, s in enumerate(cfg.strides):
            cout = c // 2
            p = f"layers.{li}."
            add(p + "alpha", alpha(c))
            bnd = 1.0 / math.sqrt(c * 2)
            add(p + "weight", rng.uniform(-bnd, bnd, (c, cout, 2 * s)).astype(np.float32))
            add(p + "bias", bias(cout))
            add(p + "noise_weight", conv_w(cout, cout, 1) * np.float32(0.3))
            for r in range(3):
                q = p + f"residual_unit.{r}.res."
                add(q + "initial.alpha", alpha(cout))
                add(q + "initial.bias", bias(cout))
                add(q + "initial.weight", conv_w(cout, 1, 7))   # depthwise (groups = channels)
                add(q + "final.alpha", alpha(cout))
                add(q + "final.bias", bias(cout))
                add(q + "final.weight", conv_w(cout, cout, 1) * np.float32(0.5))
            c = cout
        add("alpha_out", alpha(c))
        add("final.weight", conv_w(1, c, 7))
        add("final.bias", bias(1))
        self.c_last = c
        U32 = gguf.T_U32
        self.kv = [("general.architecture", gguf.T_STR, "orpheus"), ("snac.audio_token_channels", U32, len(cfg.repeats)),
                   ("snac.up_sampling_factor", U32, cfg.hop), ("snac.max_generation_size", U32, cfg.max_frames)]
        c = cfg.c0
        for i, (s, pd) in enumerate(zip(cfg.strides, cfg.paddings)):
            c //= 2
            self.kv += [(f"snac.snac_layer_stride_{i}", U32, s), (f"snac.snac_layer_padding_{i}", U32, pd), (f"snac.snac_layer_grouping_{i}", U32, c)]
        self.by_name = {t.name: t for t in self.tensors}

    def write_gguf(self, path):
        gguf.write(path, self.kv, self.tensors)
        return path


def build_snac(cfg: SnacConfig) -> SynthSnac:
    return SynthSnac(cfg)


# --------------------------------------------------------------------------------------------------
# Orpheus decoder (Llama-3 blocks; src/models/orpheus/model.cpp).  Tensor names as
# py-gguf/tts_encoders/orpheus_gguf_encoder.py:118-122,173 writes them, KV keys :190-212.
# --------------------------------------------------------------------------------------------------
@dataclass
class OrpheusConfig:
    hidden: int = 256
    layers: int = 2
    heads: int = 2            # head_dim is 128 (orpheus/model.h:28)
    kv_heads: int = 1
    head_dim: int = 128
    ffn: int = 512
    vocab: int = 200
    ctx: int = 96             # max_context_length + max_generation_size positions of KV cache
    weight_type: int = gguf.F32
    seed: int = 0x0A9


def orpheus_tiny(**kw):
    return OrpheusConfig(**kw)


def orpheus_3b(**kw):
    """canopylabs/orpheus-3b: 28 layers, hidden 3072, 24 heads / 8 kv heads x 128, ffn 8192, vocab 156940"""
    base = dict(hidden=3072, layers=28, heads=24, kv_heads=8, ffn=8192, vocab=156940, ctx=3124)
    base.update(kw)
    return OrpheusConfig(**base)


def llama3_rope_factors(head_dim, base=500000.0, factor=8.0, low=1.0, high=4.0, old_ctx=8192):
    """orpheus_gguf_encoder.py:145-173 (prepare_rope_frequencies)"""
    out = []
    for i in range(0, head_dim, 2):
        freq = 1.0 / (base ** (i / head_dim))
        wavelen = 2 * math.pi / freq
        if wavelen < old_ctx / high:
            out.append(1.0)
        elif wavelen > old_ctx / low:
            out.append(factor)
        else:
            smooth = (old_ctx / wavelen - low) / (high - low)
            out.append(1 / ((1 - smooth) / factor + smooth))
    return np.array(out, dtype=np.float32)


class _Lazy:
    """a matrix of a pooled model: shape + std, minted by add() in the tensor's own type"""

    def __init__(self, shape, std):
        self.shape, self.std = tuple(shape), std


def _add_pooled(tensors, pool, name, lazy, ttype):
    ne = list(reversed(lazy.shape))
    if ttype in (gguf.Q4_0, gguf.Q5_0, gguf.Q8_0):
        tensors.append(gguf.Tensor(name, ttype, ne, pool.blocks(lazy.shape, lazy.std, ttype)))
    else:
        tensors.append(gguf.Tensor.from_array(name, pool.normal(lazy.shape, lazy.std), ttype))


class SynthOrpheus:
    def __init__(self, cfg: OrpheusConfig, pooled=False):
        self.cfg = cfg
        rng = np.random.Generator(np.random.Philox(cfg.seed))
        pool = _Pool(rng) if pooled else None
        self.tensors = []
        H, F, kvH = cfg.hidden, cfg.ffn, cfg.kv_heads * cfg.head_dim

        def normal(shape, std):
            if pooled and len(shape) == 2:
                return _Lazy(shape, std)
            return (rng.standard_normal(shape, dtype=np.float32) * np.float32(std)).astype(np.float32)

        def add(name, arr, quantizable=True):
            if isinstance(arr, _Lazy):
                return _add_pooled(self.tensors, pool, "orpheus." + name, arr, cfg.weight_type if quantizable else gguf.F32)
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            ttype = cfg.weight_type if quantizable else gguf.F32
            ne = list(reversed(arr.shape))
            if ttype in (gguf.Q4_0, gguf.Q5_0, gguf.Q8_0):
                self.tensors.append(gguf.Tensor("orpheus." + name, ttype, ne, quantize(arr, ttype).tobytes()))
            else:
                self.tensors.append(gguf.Tensor.from_array("orpheus." + name, arr, ttype))

        add("embed_tokens", normal((cfg.vocab, H), 1.0))
        for l in range(cfg.layers):
            p = f"layers.{l}."
            add(p + "self_attn.q_proj", normal((cfg.heads * cfg.head_dim, H), 1.0 / math.sqrt(H)))
            add(p + "self_attn.k_proj", normal((kvH, H), 1.0 / math.sqrt(H)))
            add(p + "self_attn.v_proj", normal((kvH, H), 1.0 / math.sqrt(H)))
            add(p + "self_attn.o_proj", normal((H, cfg.heads * cfg.head_dim), 1.0 / math.sqrt(H)))
            add(p + "mlp.gate_proj", normal((F, H), 1.0 / math.sqrt(H)))
            add(p + "mlp.up_proj", normal((F, H), 1.0 / math.sqrt(H)))
            add(p + "mlp.down_proj", normal((H, F), 1.0 / math.sqrt(F)))
            add(p + "input_layernorm", 1.0 + normal((H,), 0.05), quantizable=False)
            add(p + "post_attention_layernorm", 1.0 + normal((H,), 0.05), quantizable=False)
        add("norm", 1.0 + normal((H,), 0.05), quantizable=False)
        add("lm_head", normal((cfg.vocab, H), 1.0 / math.sqrt(H)))
        add("rope_frequencies", llama3_rope_factors(cfg.head_dim), quantizable=False)
        U32 = gguf.T_U32
        self.kv = [("general.architecture", gguf.T_STR, "orpheus"), ("orpheus.layers", U32, cfg.layers), ("orpheus.hidden_size", U32, H),
                   ("orpheus.vocab_size", U32, cfg.vocab), ("orpheus.attn_heads", U32, cfg.heads), ("orpheus.kv_attn_heads", U32, cfg.kv_heads),
                   ("orpheus.head_dim", U32, cfg.head_dim), ("orpheus.kv_hidden_size", U32, kvH), ("orpheus.stopping_token_id", U32, cfg.vocab - 1)]
        self.by_name = {t.name: t for t in self.tensors}

    def write_gguf(self, path):
        gguf.write(path, self.kv, self.tensors)
        return path


def build_orpheus(cfg: OrpheusConfig, pooled=False) -> SynthOrpheus:
    return SynthOrpheus(cfg, pooled=pooled)


class SynthOrpheusFull:
    """One GGUF as the Orpheus converter writes it: orpheus.* + snac.* tensors, byte-pair vocabulary, and the framing
    constants as extension keys so that a small synthetic vocabulary can carry them (the reference hard-codes
    128259 / 128000 / ... / 128266 / 4096, orpheus/model.cpp:8-9,371).  lm_head rows outside the audio-token range are
    zero, so greedy decoding always lands on a valid SNAC code."""

    def __init__(self, ocfg: OrpheusConfig = None, scfg: SnacConfig = None, max_gen=28):
        scfg = scfg or snac_tiny()
        letters = list("abcdefghijklmnopqrstuvwxyz")
        text_vocab = ["<unk>"] + letters + ["Ġ", ":", "th", "he", "the", "Ġthe", "Ġa", "lo", "hel", "hello", "Ġhello", "zo", "zoe"]
        merges = ["t h", "h e", "th e", "Ġ the", "Ġ a", "l o", "he l", "hel lo", "Ġ hello", "z o", "zo e"]
        n_text = len(text_vocab)
        self.specials = dict(pre=[n_text, n_text + 1], app=[n_text + 2, n_text + 3, n_text + 4, n_text + 5], stop=n_text + 6)
        self.audio_offset = n_text + 8
        vocab = self.audio_offset + scfg.cb_size + 3
        ocfg = ocfg or orpheus_tiny(vocab=vocab, ctx=64 + max_gen)
        assert ocfg.vocab == vocab
        self.cfg, self.scfg, self.max_gen = ocfg, scfg, max_gen
        self.orpheus = SynthOrpheus(ocfg)
        head = self.orpheus.by_name["orpheus.lm_head"].to_f32().copy()
        head[: self.audio_offset] = 0.0
        head[self.audio_offset + scfg.cb_size:] = 0.0
        ne = self.orpheus.by_name["orpheus.lm_head"].ne
        if ocfg.weight_type in (gguf.Q4_0, gguf.Q5_0, gguf.Q8_0):
            new_head = gguf.Tensor("orpheus.lm_head", ocfg.weight_type, ne, quantize(head, ocfg.weight_type).tobytes())
        else:
            new_head = gguf.Tensor.from_array("orpheus.lm_head", head, ocfg.weight_type)
        self.orpheus.tensors = [new_head if t.name == "orpheus.lm_head" else t for t in self.orpheus.tensors]
        self.orpheus.by_name = {t.name: t for t in self.orpheus.tensors}
        self.snac = SynthSnac(scfg)
        self.tensors = self.orpheus.tensors + self.snac.tensors
        self.vocab_tokens = text_vocab + [f"<s{i}>" for i in range(vocab - n_text)]
        self.merges = merges
        U32, STR, ARR = gguf.T_U32, gguf.T_STR, gguf.T_ARR
        self.kv = [kv for kv in self.orpheus.kv if kv[0] != "orpheus.stopping_token_id"] + [kv for kv in self.snac.kv if kv[0] != "general.architecture"] + [
            ("orpheus.stopping_token_id", U32, self.specials["stop"]),
            ("orpheus.max_context_length", U32, 64), ("orpheus.max_generation_size", U32, max_gen),
            ("orpheus.audio_token_offset", U32, self.audio_offset), ("orpheus.audio_token_stride", U32, 0),
            ("orpheus.prepended_tokens", ARR, (U32, self.specials["pre"])), ("orpheus.appended_tokens", ARR, (U32, self.specials["app"])),
            ("tokenizer.ggml.tokens", ARR, (STR, self.vocab_tokens)), ("tokenizer.ggml.merges", ARR, (STR, merges)),
            ("tokenizer.ggml.bos_token_id", U32, n_text + 1), ("tokenizer.ggml.eos_token_id", U32, n_text + 7),
        ]
        self.by_name = {t.name: t for t in self.tensors}

    def write_gguf(self, path):
        gguf.write(path, self.kv, self.tensors)
        return path


# --------------------------------------------------------------------------------------------------
# Dia (src/models/dia/model.cpp).  Tensor names / keys as py-gguf/tts_encoders/dia_gguf_encoder.py:72-190 writes them:
# dia.encoder.{embedding,norm,layers.N.{q,k,v,o}_proj,pre_sa_norm,post_sa_norm,gate,up,wo}, dia.decoder.{embeddings.N,
# heads.N,norm,layers.N.{self,cross}_{q,k,v,o}_proj,pre_{sa,ca,mlp}_norm,gate,up,wo}, audio_encoder.* (the DAC codec).
# "dia.decoder.attn_heads" is the number of query heads and "dia.decoder.query_heads" the repeat count of each k/v
# group (the converter writes gqa_query_heads / kv_heads, :169-170; the model uses attn_heads / query_heads groups,
# model.cpp:463,468).  The encoder's hidden size has no key (model.h:68 default 1024); it is the embedding's width.
# --------------------------------------------------------------------------------------------------
DIA_DELAY_PATTERN = (0, 8, 9, 10, 11, 12, 13, 14, 15)   # model.h:84: not stored in the GGUF


@dataclass
class DiaConfig:
    enc_hidden: int = 128
    enc_layers: int = 2
    enc_heads: int = 2          # enc_heads * head_dim == dec_hidden (model.cpp:410)
    enc_ffn: int = 256
    enc_vocab: int = 256        # byte tokens
    dec_hidden: int = 256
    dec_layers: int = 2
    dec_heads: int = 2          # "attn_heads": query heads
    dec_repeat: int = 2         # "query_heads": how many query heads share one k/v group
    dec_ffn: int = 512
    head_dim: int = 128
    n_out: int = 9
    audio_vocab: int = 64       # eos = audio_vocab, pad = +1, bos = +2, output vocab = +4 (model.h:75-79 with 1024)
    max_ctx: int = 24           # dia.encoder.max_context_length
    max_gen: int = 48           # dia.decoder.max_generation_size
    max_delay: int = 15
    weight_type: int = gguf.F32
    seed: int = 0xD1A
    # codec (the Parler generator's DAC builder): latent / codebooks sized for the tiny tests
    latent: int = 64
    cb_dim: int = 8
    c0: int = 64
    strides: tuple = (4, 2)

    @property
    def eos(self): return self.audio_vocab
    @property
    def pad(self): return self.audio_vocab + 1
    @property
    def bos(self): return self.audio_vocab + 2
    @property
    def out_vocab(self): return self.audio_vocab + 4
    @property
    def dec_kv_heads(self): return self.dec_heads // self.dec_repeat
    @property
    def hop(self): return int(np.prod(self.strides))


def dia_tiny(**kw):
    return DiaConfig(**kw)


def dia_1_6b(**kw):
    """nari-labs/Dia-1.6B (model.h:64-84): encoder 12 x 1024 (16 x 128 heads, ffn 4096), decoder 18 x 2048 (16 query heads on
    4 k/v groups x 128, ffn 8192), 9 heads x 1028 logits, 1024 text positions, 3072 audio steps"""
    base = dict(enc_hidden=1024, enc_layers=12, enc_heads=16, enc_ffn=4096, dec_hidden=2048, dec_layers=18, dec_heads=16, dec_repeat=4,
                dec_ffn=8192, audio_vocab=1024, max_ctx=1024, max_gen=3072, latent=1024, c0=1536, strides=(8, 8, 4, 2))
    base.update(kw)
    return DiaConfig(**base)


class SynthDia:
    def __init__(self, cfg: DiaConfig, suppress_special=False, pooled=False):
        self.cfg = cfg
        assert cfg.enc_heads * cfg.head_dim == cfg.dec_hidden and cfg.dec_heads * cfg.head_dim == cfg.dec_hidden
        rng = np.random.Generator(np.random.Philox(cfg.seed))
        pool = _Pool(rng) if pooled else None   # full-depth models: matrices cut out of one buffer of normals (unquantised types only)
        assert not pooled or cfg.weight_type in (gguf.F32, gguf.F16)
        self.tensors = []
        EH, DH, A, kvH = cfg.enc_hidden, cfg.dec_hidden, cfg.dec_heads * cfg.head_dim, cfg.dec_kv_heads * cfg.head_dim

        def normal(shape, std):
            if pooled and len(shape) == 2:
                return pool.normal(shape, std)
            return (rng.standard_normal(shape, dtype=np.float32) * np.float32(std)).astype(np.float32)

        def add(name, arr, quantizable=True):
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            ttype = cfg.weight_type if quantizable else gguf.F32
            ne = list(reversed(arr.shape))
            if ttype in (gguf.Q4_0, gguf.Q5_0, gguf.Q8_0):
                self.tensors.append(gguf.Tensor(name, ttype, ne, quantize(arr, ttype).tobytes()))
            else:
                self.tensors.append(gguf.Tensor.from_array(name, arr, ttype))

        # attention logits are not scaled by 1/sqrt(d) in Dia (soft_max_ext(..., 1.0f, 0), model.cpp:403,586,630): keep
        # q and k small so that the synthetic softmax is not one-hot
        qs = 0.35
        for i in range(cfg.n_out):
            add(f"dia.decoder.embeddings.{i}", normal((cfg.out_vocab, DH), 0.5))
        add("dia.decoder.norm", 1.0 + normal((DH,), 0.05), quantizable=False)
        for i in range(cfg.n_out):
            hw = normal((cfg.out_vocab, DH), 1.0 / math.sqrt(DH))
            if suppress_special:
                hw[cfg.audio_vocab:] = 0.0
            add(f"dia.decoder.heads.{i}", hw)
        for l in range(cfg.dec_layers):
            p = f"dia.decoder.layers.{l}."
            add(p + "pre_sa_norm", 1.0 + normal((DH,), 0.05), quantizable=False)
            add(p + "self_q_proj", normal((A, DH), qs / math.sqrt(DH)))
            add(p + "self_k_proj", normal((kvH, DH), qs / math.sqrt(DH)))
            add(p + "self_v_proj", normal((kvH, DH), 1.0 / math.sqrt(DH)))
            add(p + "self_o_proj", normal((DH, A), 1.0 / math.sqrt(A)))
            add(p + "pre_ca_norm", 1.0 + normal((DH,), 0.05), quantizable=False)
            add(p + "cross_q_proj", normal((A, DH), qs / math.sqrt(DH)))
            add(p + "cross_k_proj", normal((A, EH), qs / math.sqrt(EH)))
            add(p + "cross_v_proj", normal((A, EH), 1.0 / math.sqrt(EH)))
            add(p + "cross_o_proj", normal((DH, A), 1.0 / math.sqrt(A)))
            add(p + "pre_mlp_norm", 1.0 + normal((DH,), 0.05), quantizable=False)
            add(p + "gate", normal((cfg.dec_ffn, DH), 1.0 / math.sqrt(DH)))
            add(p + "up", normal((cfg.dec_ffn, DH), 1.0 / math.sqrt(DH)))
            add(p + "wo", normal((DH, cfg.dec_ffn), 1.0 / math.sqrt(cfg.dec_ffn)))
        add("dia.encoder.embedding", normal((cfg.enc_vocab, EH), 1.0))
        add("dia.encoder.norm", 1.0 + normal((EH,), 0.05), quantizable=False)
        for l in range(cfg.enc_layers):
            p = f"dia.encoder.layers.{l}."
            add(p + "pre_sa_norm", 1.0 + normal((EH,), 0.05), quantizable=False)
            add(p + "q_proj", normal((A, EH), qs / math.sqrt(EH)))
            add(p + "k_proj", normal((A, EH), qs / math.sqrt(EH)))
            add(p + "v_proj", normal((A, EH), 1.0 / math.sqrt(EH)))
            add(p + "o_proj", normal((EH, A), 1.0 / math.sqrt(A)))
            add(p + "post_sa_norm", 1.0 + normal((EH,), 0.05), quantizable=False)
            add(p + "gate", normal((cfg.enc_ffn, EH), 1.0 / math.sqrt(EH)))
            add(p + "up", normal((cfg.enc_ffn, EH), 1.0 / math.sqrt(EH)))
            add(p + "wo", normal((EH, cfg.enc_ffn), 1.0 / math.sqrt(cfg.enc_ffn)))
        # the codec: the Parler generator's DAC tensors and keys (same converter base class, DACEncoder)
        pc = Config(hidden=64, layers=1, heads=1, ffn=64, out_vocab=cfg.out_vocab, audio_vocab=cfg.audio_vocab, n_out=cfg.n_out, ctx=32, max_gen=16,
                    enc_len=2, prompt_vocab=8, eos=cfg.eos, bos=cfg.bos, latent=cfg.latent, cb_dim=cfg.cb_dim, cb_size=cfg.audio_vocab, c0=cfg.c0,
                    strides=cfg.strides, seed=cfg.seed + 1, weight_type=gguf.F32)
        self.dac = SynthModel(pc)
        self.tensors += [t for t in self.dac.tensors if t.name.startswith("audio_encoder.")]
        U32, STR = gguf.T_U32, gguf.T_STR
        self.kv = [("general.architecture", STR, "dia"), ("general.name", STR, "synthetic-dia")]
        self.kv += [kv for kv in self.dac.kv if kv[0].startswith("dac.") or kv[0].startswith("audio.")]
        self.kv += [
            ("dia.attn_head_size", U32, cfg.head_dim), ("dia.eos_token_id", U32, cfg.eos), ("dia.bos_token_id", U32, cfg.bos),
            ("dia.pad_token_id", U32, cfg.pad), ("dia.max_delay", U32, cfg.max_delay),
            ("dia.encoder.max_context_length", U32, cfg.max_ctx), ("dia.encoder.attn_heads", U32, cfg.enc_heads), ("dia.encoder.layers", U32, cfg.enc_layers),
            ("dia.decoder.hidden_size", U32, DH), ("dia.decoder.layers", U32, cfg.dec_layers), ("dia.decoder.output_heads", U32, cfg.n_out),
            ("dia.decoder.attn_heads", U32, cfg.dec_heads), ("dia.decoder.query_heads", U32, cfg.dec_repeat),
            ("dia.decoder.output_vocab_size", U32, cfg.out_vocab), ("dia.decoder.audio_vocab_size", U32, cfg.audio_vocab),
            ("dia.decoder.max_generation_size", U32, cfg.max_gen),
        ]
        self.by_name = {t.name: t for t in self.tensors}

    def write_gguf(self, path):
        gguf.write(path, self.kv, self.tensors)
        return path


def build_dia(cfg: DiaConfig, **kw) -> SynthDia:
    return SynthDia(cfg, **kw)


# --------------------------------------------------------------------------------------------------
# Kokoro (src/models/kokoro/model.cpp).  Tensor names / keys as py-gguf/tts_encoders/kokoro_gguf_encoder.py writes them
# (ALBERT_PARTS :14-38, prepare_lstm_tensor :287-307, prepare_adain_res_block_tensor :309-327, set_gguf_parameters :412-470).
# The phonemizer tables are left out: the acoustic path starts at phoneme ids.
# --------------------------------------------------------------------------------------------------
@dataclass
class KokoroConfig:
    vocab: int = 32
    albert_embd: int = 16        # 128
    hidden: int = 64             # 768
    heads: int = 4               # 12 (the reference's softmax scale stays 0.125 whatever the head size, model.h:196)
    ffn: int = 128               # 2048
    recurrence: int = 2          # 12
    max_ctx: int = 32            # 512
    dp_hidden: int = 32          # 512: duration predictor, text encoder and shared LSTM width
    style_half: int = 8          # 128
    dp_layers: int = 3
    f0_blocks: int = 3
    n_durations: int = 6         # 50 duration bins (sum of sigmoids = the predicted length)
    conv_layers: int = 3
    enc_channels: int = 48       # 1024
    asr_channels: int = 8        # 64
    gen_channels: int = 24       # 512
    decoder_blocks: int = 4
    up_rates: tuple = (10, 6)
    up_kernels: tuple = (20, 12)
    res_kernels: tuple = (3, 7, 11)
    res_dilations: tuple = (1, 3, 5)
    n_fft: int = 20
    hop: int = 5
    harmonic_num: int = 8
    voices: tuple = ("af_test", "bm_test")
    seed: int = 0xC0C0
    forced_frames: int = 0       # > 0 (bench models): the duration head ignores its input and predicts exactly this many frames per token (biases +-20, weights 0)

    @property
    def up_sampling_factor(self):
        return 2 * int(np.prod(self.up_rates)) * self.hop      # 600 in the reference's file (kokoro_gguf_encoder.py:441)


def kokoro_tiny(**kw):
    return KokoroConfig(**kw)


def kokoro_82m(**kw):
    base = dict(vocab=178, albert_embd=128, hidden=768, heads=12, ffn=2048, recurrence=12, max_ctx=512, dp_hidden=512, style_half=128, n_durations=50,
                enc_channels=1024, asr_channels=64, gen_channels=512)
    base.update(kw)
    return KokoroConfig(**base)


class SynthKokoro:
    def __init__(self, cfg: KokoroConfig):
        self.cfg = cfg
        rng = np.random.Generator(np.random.Philox(cfg.seed))
        self.tensors = []
        E, H, F, D, S, C = cfg.albert_embd, cfg.hidden, cfg.ffn, cfg.dp_hidden, cfg.style_half, cfg.dp_hidden

        def normal(shape, std):
            return (rng.standard_normal(shape, dtype=np.float32) * np.float32(std)).astype(np.float32)

        def add(name, arr):
            self.tensors.append(gguf.Tensor.from_array("kokoro." + name, np.ascontiguousarray(arr, dtype=np.float32)))

        def lin(name, out, inp, bias_name=None, std=None):
            add(name, normal((out, inp), std if std is not None else 1.0 / math.sqrt(inp)))
            add(bias_name or name + "_bias", normal((out,), 0.05))

        def norm(name, n, bias_name=None):
            add(name, 1.0 + normal((n,), 0.05))
            add(bias_name or name + "_bias", normal((n,), 0.05))

        def lstm(base, inp, hid):
            for d in ("weights", "reverse_weights"):
                for j in range(4):
                    add(f"{base}.0.{d}.{2 * j}", normal((hid, inp), 1.0 / math.sqrt(inp)))
                    add(f"{base}.0.{d}.{2 * j + 1}", normal((hid, hid), 1.0 / math.sqrt(hid)))
            for d in ("biases", "reverse_biases"):
                for j in range(8):
                    add(f"{base}.0.{d}.{j}", normal((hid,), 0.1))

        def adain_block(base, cin, cout, upsample):
            for k, c in (("norm1", cin), ("norm2", cout)):
                for gb in ("gamma", "beta"):
                    add(f"{base}.{k}_{gb}_weight", normal((c, S), 0.3 / math.sqrt(S)))
                    add(f"{base}.{k}_{gb}_bias", normal((c,), 0.1))
            add(f"{base}.conv1_weight", normal((cout, cin, 3), 1.0 / math.sqrt(3 * cin)))
            add(f"{base}.conv1_bias", normal((cout,), 0.05))
            add(f"{base}.conv2_weight", normal((cout, cout, 3), 1.0 / math.sqrt(3 * cout)))
            add(f"{base}.conv2_bias", normal((cout,), 0.05))
            if upsample:
                add(f"{base}.pool_weight", normal((cin, 1, 3), 0.6))
                add(f"{base}.pool_bias", normal((cin,), 0.05))
            if upsample or cin != cout:
                add(f"{base}.conv1x1_weight", normal((cout, cin, 1), 1.0 / math.sqrt(cin)))

        def gen_res(base, c, k):
            for j in range(3):
                for a in ("1", "2"):
                    for gb in ("gamma", "beta"):
                        add(f"{base}.{j}.{gb}{a}_weight", normal((c, S), 0.3 / math.sqrt(S)))
                        add(f"{base}.{j}.{gb}{a}_bias", normal((c,), 0.1))
                    add(f"{base}.{j}.alpha{a}", rng.uniform(0.5, 2.0, (1, c, 1)).astype(np.float32))
                    add(f"{base}.{j}.convs{a}_weight", normal((c, c, k), 0.7 / math.sqrt(k * c)))
                    add(f"{base}.{j}.convs{a}_bias", normal((c,), 0.05))

        # ---- ALBERT
        add("albert.token_embd", normal((cfg.vocab, E), 1.0))
        add("albert.position_embd", normal((cfg.max_ctx, E), 0.5))
        add("albert.token_type_embd", normal((E,), 0.5))
        norm("albert.norm", E)
        lin("albert.embd", H, E)
        p = "albert.layer.0."
        for nm in ("q", "k", "v", "o"):
            lin(p + nm, H, H)
        lin(p + "ffn", F, H)
        lin(p + "ffn_out", H, F)
        norm(p + "attn_norm", H)
        norm(p + "ffn_norm", H)
        # ---- duration / prosody predictor
        dp = "duration_predictor."
        lin(dp + "encode", D, H)
        for l in range(cfg.dp_layers):
            lstm(f"{dp}layers.{2 * l}.lstm", D + S, D // 2)
            for gb in ("gamma", "beta"):
                add(f"{dp}layers.{2 * l + 1}.{gb}_weight", normal((D, S), 0.3 / math.sqrt(S)))
                add(f"{dp}layers.{2 * l + 1}.{gb}_bias", normal((D,), 0.1))
        lstm(dp + "duration_lstm", D + S, D // 2)
        if cfg.forced_frames > 0:
            add(dp + "duration_proj", np.zeros((cfg.n_durations, D), dtype=np.float32))
            add(dp + "duration_proj_bias", np.where(np.arange(cfg.n_durations) < cfg.forced_frames, 20.0, -20.0).astype(np.float32))
        else:
            lin(dp + "duration_proj", cfg.n_durations, D, std=8.0)   # wide: the tiny model then predicts a spread of lengths
        lstm(dp + "shared_lstm", D + S, D // 2)
        dims = [(D, D, False), (D, D // 2, True), (D // 2, D // 2, False)][:cfg.f0_blocks]
        for br in ("f0", "n"):
            for i, (ci, co, up) in enumerate(dims):
                adain_block(f"{dp}{br}_blocks.{i}", ci, co, up)
            add(f"{dp}{br}_proj_kernel", normal((1, dims[-1][1], 1), (60.0 if br == "f0" else 1.0) / math.sqrt(dims[-1][1])))   # f0 in Hz-like units: some frames voiced (> 10)
            add(f"{dp}{br}_proj_bias", normal((1,), 0.1))
        # ---- text encoder
        add("text_encoder.embedding_weight", normal((cfg.vocab, C), 1.0))
        for l in range(cfg.conv_layers):
            add(f"text_encoder.layers.{l}.weight", normal((C, C, 5), 1.0 / math.sqrt(5 * C)))
            add(f"text_encoder.layers.{l}.bias", normal((C,), 0.05))
            add(f"text_encoder.layers.{l}.gamma", 1.0 + normal((C,), 0.05))
            add(f"text_encoder.layers.{l}.beta", normal((C,), 0.05))
        lstm("text_encoder.lstm", C, C // 2)
        # ---- decoder
        de = "decoder."
        for br in ("f0", "n"):
            add(f"{de}{br}_conv_weight", normal((1, 1, 3), 0.5))
            add(f"{de}{br}_conv_bias", normal((1,), 0.05))
        add(de + "asr_conv_weight", normal((cfg.asr_channels, C, 1), 1.0 / math.sqrt(C)))
        add(de + "asr_conv_bias", normal((cfg.asr_channels,), 0.05))
        CE, CA, CG = cfg.enc_channels, cfg.asr_channels, cfg.gen_channels
        adain_block(de + "encoder_block", C + 2, CE, False)
        for i in range(cfg.decoder_blocks):
            last = i == cfg.decoder_blocks - 1
            adain_block(f"{de}decoder_blocks.{i}", CE + CA + 2, CG if last else CE, last)
        g = de + "generator."
        add(g + "m_source_weight", normal((1, cfg.harmonic_num + 1), 1.0))
        add(g + "m_source_bias", normal((1,), 0.05))
        c = CG
        nb = cfg.n_fft // 2 + 1
        self.geometry = dict(up=[], noise=[], res=[], noise_res=[])
        for i, (u, k) in enumerate(zip(cfg.up_rates, cfg.up_kernels)):
            co = c // 2
            add(f"{g}ups.{i}.weight", normal((c, co, k), 1.0 / math.sqrt(c * k / u)))
            add(f"{g}ups.{i}.bias", normal((co,), 0.05))
            self.geometry["up"].append((u, (k - u) // 2))
            c = co
            if i + 1 < len(cfg.up_rates):
                sf = int(np.prod(cfg.up_rates[i + 1:]))
                nk, ns, npad, rk = sf * 2, sf, (sf + 1) // 2, 7
            else:
                nk, ns, npad, rk = 1, 1, 0, 11
            add(f"{g}noise_blocks.{i}.conv_weight", normal((c, 2 * nb, nk), 0.3 / math.sqrt(2 * nb * nk)))
            add(f"{g}noise_blocks.{i}.conv_bias", normal((c,), 0.05))
            self.geometry["noise"].append((ns, npad))
            gen_res(f"{g}noise_blocks.{i}.resblock", c, rk)
            self.geometry["noise_res"].append([((rk * d - d) // 2, d) for d in cfg.res_dilations])
            for ii, rk2 in enumerate(cfg.res_kernels):
                gen_res(f"{g}resblocks.{i * len(cfg.res_kernels) + ii}", c, rk2)
                self.geometry["res"].append([((rk2 * d - d) // 2, d) for d in cfg.res_dilations])
        add(g + "conv_post_weight", normal((2 * nb, c, 7), 0.5 / math.sqrt(7 * c)))
        add(g + "conv_post_bias", normal((2 * nb,), 0.05))
        for vname in cfg.voices:
            add(f"voice_tensors.{vname}", normal((cfg.max_ctx, 2 * S), 1.0))
        U32, STR, ARR = gguf.T_U32, gguf.T_STR, gguf.T_ARR
        a = "kokoro.duration_predictor.albert."
        gk = "kokoro.decoder.generator."
        self.kv = [("general.architecture", STR, "kokoro"), ("general.name", STR, "synthetic-kokoro"),
                   (a + "context_length", U32, cfg.max_ctx), (a + "layers", U32, 1), (a + "attn_heads", U32, cfg.heads), (a + "hidden_size", U32, H),
                   (a + "recurrence", U32, cfg.recurrence),
                   ("kokoro.duration_predictor.hidden_size", U32, D), ("kokoro.duration_predictor.layers", U32, cfg.dp_layers),
                   ("kokoro.duration_predictor.f0_n_blocks", U32, cfg.f0_blocks), ("kokoro.text_encoder.layers", U32, cfg.conv_layers),
                   (gk + "up_sampling_factor", U32, cfg.up_sampling_factor), (gk + "kernels", U32, len(cfg.res_kernels)), (gk + "upsamples", U32, len(cfg.up_rates)),
                   (gk + "layers", U32, cfg.decoder_blocks), (gk + "padding", U32, 3), (gk + "n_fft", U32, cfg.n_fft), (gk + "hop", U32, cfg.hop)]
        for i, blk in enumerate(self.geometry["noise_res"]):
            for ii, (pd, dl) in enumerate(blk):
                self.kv += [(f"{gk}noise_blocks.{i}.res_block.{ii}.padding", U32, pd), (f"{gk}noise_blocks.{i}.res_block.{ii}.dilation", U32, dl)]
        for i, (st, pd) in enumerate(self.geometry["noise"]):
            self.kv += [(f"{gk}noise_blocks.{i}.stride", U32, st), (f"{gk}noise_blocks.{i}.padding", U32, pd)]
        for i, blk in enumerate(self.geometry["res"]):
            for ii, (pd, dl) in enumerate(blk):
                self.kv += [(f"{gk}res_blocks.{i}.{ii}.padding", U32, pd), (f"{gk}res_blocks.{i}.{ii}.dilation", U32, dl)]
        for i, (st, pd) in enumerate(self.geometry["up"]):
            self.kv += [(f"{gk}up_convs.{i}.padding", U32, pd), (f"{gk}up_convs.{i}.stride", U32, st)]
        self.kv += [("kokoro.voices", ARR, (STR, list(cfg.voices))), ("tokenizer.ggml.tokens", ARR, (STR, [""] + [chr(0x61 + i % 26) for i in range(cfg.vocab - 1)])),
                    ("tokenizer.ggml.eos_token_id", U32, 0), ("tokenizer.ggml.padding_token_id", U32, 0)]
        self.by_name = {t.name: t for t in self.tensors}

    def write_gguf(self, path):
        gguf.write(path, self.kv, self.tensors)
        return path


def build_kokoro(cfg: KokoroConfig) -> SynthKokoro:
    return SynthKokoro(cfg)

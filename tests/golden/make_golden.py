"""Generate tests/golden/*.npz — golden vectors for the oracle.

The reference cannot be built or imported here (its arithmetic lives in the absent ggml submodule,
SURVEY.md §0.1) and it ships no golden vectors (§0.2), so these vectors come from a SECOND,
independent restatement of the same path written with PyTorch CPU primitives in float64
(F.layer_norm / F.linear / softmax / tanh-GELU / F.conv1d / F.conv_transpose1d), on seeded synthetic
weights (tts.cpp_amd/synth.py).  They pin the C oracle against PyTorch's definitions of each
primitive; they do NOT pin it against ggml ("parity unpinned").

Run from the repo root:  python tests/golden/make_golden.py
"""
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as Fn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tts_cpp_amd  # noqa: E402
from tts_cpp_amd import gguf, synth  # noqa: E402

D = torch.float64
D_ = torch.float64


def T(model, name):
    return torch.from_numpy(model.by_name[name].to_f32().astype(np.float64))


class TorchParler:
    """Parler decoder (src/models/parler/model.cpp:520-614) in float64 torch."""

    def __init__(self, model):
        self.m, self.cfg = model, model.cfg
        c = self.cfg
        self.k = [torch.zeros(0, c.hidden, dtype=D) for _ in range(c.layers)]
        self.v = [torch.zeros(0, c.hidden, dtype=D) for _ in range(c.layers)]
        enc = T(model, "decoder.text_encoding")
        self.ck = [Fn.linear(enc, T(model, f"decoder.layers.{l}.encoder_attn.k_proj.weight")) for l in range(c.layers)]
        self.cv = [Fn.linear(enc, T(model, f"decoder.layers.{l}.encoder_attn.v_proj.weight")) for l in range(c.layers)]

    def attend(self, q, K, V, causal_from=None):
        c = self.cfg
        d = c.hidden // c.heads
        S, Tn = q.shape[0], K.shape[0]
        qh = q.view(S, c.heads, d).transpose(0, 1)
        kh = K.view(Tn, c.heads, d).transpose(0, 1)
        vh = V.view(Tn, c.heads, d).transpose(0, 1)
        s = qh @ kh.transpose(1, 2) / d ** 0.5
        if causal_from is not None:
            pos = causal_from + torch.arange(S)
            mask = torch.arange(Tn)[None, :] > pos[:, None]
            s = s.masked_fill(mask[None], float("-inf"))
        p = torch.softmax(s, dim=-1)
        return (p @ vh).transpose(0, 1).reshape(S, c.hidden)

    def decode(self, tokens, pos0, audio):
        c, m = self.cfg, self.m
        if audio:
            x = sum(T(m, f"decoder.embed_tokens.{i}.weight")[int(tokens[i])] for i in range(c.n_out))[None]
        else:
            x = T(m, "decoder.embed_prompts")[torch.as_tensor(np.asarray(tokens, dtype=np.int64))]
        S = x.shape[0]
        x = x + T(m, "decoder.positional_embed")[pos0:pos0 + S]
        for l in range(c.layers):
            p = f"decoder.layers.{l}."
            h = Fn.layer_norm(x, (c.hidden,), T(m, p + "self_attn_layer_norm.weight"), T(m, p + "self_attn_layer_norm.bias"), 1e-5)
            q = Fn.linear(h, T(m, p + "self_attn.q_proj.weight"))
            self.k[l] = torch.cat([self.k[l][:pos0], Fn.linear(h, T(m, p + "self_attn.k_proj.weight"))])
            self.v[l] = torch.cat([self.v[l][:pos0], Fn.linear(h, T(m, p + "self_attn.v_proj.weight"))])
            x = x + Fn.linear(self.attend(q, self.k[l], self.v[l], causal_from=pos0), T(m, p + "self_attn.out_proj.weight"))
            h = Fn.layer_norm(x, (c.hidden,), T(m, p + "encoder_attn_layer_norm.weight"), T(m, p + "encoder_attn_layer_norm.bias"), 1e-5)
            q = Fn.linear(h, T(m, p + "encoder_attn.q_proj.weight"))
            x = x + Fn.linear(self.attend(q, self.ck[l], self.cv[l]), T(m, p + "encoder_attn.out_proj.weight"))
            h = Fn.layer_norm(x, (c.hidden,), T(m, p + "final_layer_norm.weight"), T(m, p + "final_layer_norm.bias"), 1e-5)
            h = Fn.gelu(Fn.linear(h, T(m, p + "fc1.weight")), approximate="tanh")
            x = x + Fn.linear(h, T(m, p + "fc2.weight"))
        h = Fn.layer_norm(x, (c.hidden,), T(m, "decoder.layer_norm.weight"), T(m, "decoder.layer_norm.bias"), 1e-5)
        logits = torch.stack([Fn.linear(h, T(m, f"decoder.lm_heads.{i}.weight.head")) for i in range(c.n_out)])  # [n_out][S][V]
        return logits, h


def snake(x, alpha):
    a = alpha.reshape(-1, 1)
    return x + torch.sin(a * x) ** 2 / a


def torch_dac(model, codes, stages=False):
    """DAC decoder (src/decoder/dac_model.cpp:146-170) in float64 torch. codes [frames][n_out]."""
    c = model.cfg
    codes = torch.as_tensor(np.asarray(codes, dtype=np.int64)).reshape(-1, c.n_out)
    x = 0
    for i in range(c.n_out):
        p = f"audio_encoder.quantizers.{i}."
        e = T(model, p + "codebook.weight")[codes[:, i]].t()[None]  # [1][dim][T]
        x = x + Fn.conv1d(e, T(model, p + "out_proj.weight"), T(model, p + "out_proj.bias"))
    outs = [x[0]]
    x = Fn.conv1d(x, T(model, "audio_encoder.initial.weight"), T(model, "audio_encoder.initial.bias"), padding=3)
    outs.append(x[0])
    for bi, (s, pd) in enumerate(zip(c.strides, c.paddings)):
        p = f"audio_encoder.decoder_block.{bi + 1}."
        x = snake(x[0], T(model, p + "final.alpha"))[None]
        x = Fn.conv_transpose1d(x, T(model, p + "final.weight"), T(model, p + "final.bias"), stride=s, padding=pd)
        for r in range(3):
            q = p + f"residual_unit.{r}.res."
            dil = 3 ** r
            y = snake(x[0], T(model, q + "initial.alpha"))[None]
            y = Fn.conv1d(y, T(model, q + "initial.weight"), T(model, q + "initial.bias"), padding=3 * dil, dilation=dil)
            y = snake(y[0], T(model, q + "final.alpha"))[None]
            y = Fn.conv1d(y, T(model, q + "final.weight"), T(model, q + "final.bias"))
            x = x + y
        outs.append(x[0])
    x = snake(x[0], T(model, "audio_encoder.final.alpha"))[None]
    x = torch.tanh(Fn.conv1d(x, T(model, "audio_encoder.final.weight"), T(model, "audio_encoder.final.bias"), padding=3))
    return (x[0, 0], outs) if stages else x[0, 0]


def t5_bucket(key, query, n_buckets_total=32):
    """t5_runner::set_inputs (src/models/parler/t5/model.cpp:303-316), integer division inside the log included"""
    import math
    n_buckets = n_buckets_total // 2
    max_exact = n_buckets // 2
    den = float(np.float32(math.log(128.0 / max_exact)))  # `float logarithmic_denominator` (:309), used in double arithmetic
    rpos, ab = key - query, abs(key - query)
    if ab < max_exact:
        v = ab
    else:
        v = min(n_buckets - 1, max_exact + int((math.log(ab // max_exact) / den) * max_exact))
    return (n_buckets if rpos > 0 else 0) + v


def torch_t5(model, ids):
    """T5 encoder (t5/model.cpp:216-295) in float64 torch: rms norm eps 1e-6, unscaled attention + relative bias,
    gated tanh-GELU MLP, final norm, optional down projection."""
    c = model.cfg
    n, H, NH = len(ids), c.hidden, c.heads
    d = H // NH
    x = T(model, "t5encoder.token_embd")[torch.from_numpy(np.asarray(ids, dtype=np.int64))]
    relb = T(model, "t5encoder.enc.blk.0.attn_rel_b")  # [buckets][heads]
    bucket = torch.tensor([[t5_bucket(k, q, c.buckets) for k in range(n)] for q in range(n)])  # [query][key]
    bias = relb[bucket].permute(2, 0, 1)  # [heads][query][key]

    def rms(v, w):
        return v * torch.rsqrt((v * v).mean(-1, keepdim=True) + 1e-6) * w

    for l in range(c.layers):
        p = f"t5encoder.enc.blk.{l}."
        cur = rms(x, T(model, p + "attn_norm"))
        q = Fn.linear(cur, T(model, p + "attn_q")).view(n, NH, d).transpose(0, 1)
        k = Fn.linear(cur, T(model, p + "attn_k")).view(n, NH, d).transpose(0, 1)
        v = Fn.linear(cur, T(model, p + "attn_v")).view(n, NH, d).transpose(0, 1)
        att = torch.softmax(q @ k.transpose(1, 2) + bias, dim=-1) @ v
        x = x + Fn.linear(att.transpose(0, 1).reshape(n, H), T(model, p + "attn_o"))
        cur = rms(x, T(model, p + "ffn_norm"))
        up = Fn.gelu(Fn.linear(cur, T(model, p + "ffn_up")), approximate="tanh") * Fn.linear(cur, T(model, p + "ffn_gate"))
        x = x + Fn.linear(up, T(model, p + "ffn_down"))
    x = rms(x, T(model, "t5encoder.enc.final_layer_norm"))
    if "t5encoder.down_proj" in model.by_name:
        x = Fn.linear(x, T(model, "t5encoder.down_proj"), T(model, "t5encoder.down_proj_bias"))
    return x


def torch_snac(model, codes, T, noise):
    """SNAC decoder (src/decoder/snac_model.cpp:86-159) in float64 torch: repeat_interleave'd codebook levels, depthwise /
    pointwise convs, ConvTranspose1d, noise block x + noise * conv1x1(x), snake, tanh."""
    c = model.cfg

    def P(name):
        return T_(model, "snac." + name)

    def snake(x, a):
        a = a.reshape(1, -1, 1)
        return x + torch.sin(a * x) ** 2 / a

    x = None
    off = 0
    for i, rep in enumerate(c.repeats):
        n = T // rep
        ids = torch.from_numpy(np.asarray(codes[off:off + n], dtype=np.int64))
        off += n
        z = P(f"quantizers.{i}.codebook.weight")[ids].t()[None]                    # [1][cb_dim][n]
        z = Fn.conv1d(z, P(f"quantizers.{i}.out_proj.weight"), P(f"quantizers.{i}.out_proj.bias"))
        z = torch.repeat_interleave(z, rep, dim=-1)
        x = z if x is None else x + z
    x = Fn.conv1d(x, P("in.weight"), P("in.bias"), padding=3, groups=c.latent)
    x = Fn.conv1d(x, P("up.weight"), P("up.bias"))
    noff = 0
    for li, (s, pd) in enumerate(zip(c.strides, c.paddings)):
        p = f"layers.{li}."
        x = snake(x, P(p + "alpha"))
        x = Fn.conv_transpose1d(x, P(p + "weight"), P(p + "bias"), stride=s, padding=pd)
        L = x.shape[-1]
        if noise is not None:
            nz = torch.from_numpy(np.asarray(noise[noff:noff + L], dtype=np.float64))[None, None]
            x = x + nz * Fn.conv1d(x, P(p + "noise_weight"))
        noff += L
        ch = x.shape[1]
        for r in range(3):
            q = p + f"residual_unit.{r}.res."
            dil = 3 ** r
            y = snake(x, P(q + "initial.alpha"))
            y = Fn.conv1d(y, P(q + "initial.weight"), P(q + "initial.bias"), padding=3 * dil, dilation=dil, groups=ch)
            y = snake(y, P(q + "final.alpha"))
            y = Fn.conv1d(y, P(q + "final.weight"), P(q + "final.bias"))
            x = x + y
    x = snake(x, P("alpha_out"))
    return torch.tanh(Fn.conv1d(x, P("final.weight"), P("final.bias"), padding=3))[0, 0]


def torch_orpheus(model, prompt, steps):
    """Llama-3 blocks (src/models/orpheus/model.cpp:186-296) in float64 torch with the HF formulation of rotary
    embeddings (inv_freq / llama3 factors, rotate_half) — an independent route to the same numbers as ggml's NEOX rope
    with frequency factors.  Returns last-token logits after the prompt and after each greedy step, and the tokens."""
    c = model.cfg
    H, NH, NKV, hd = c.hidden, c.heads, c.kv_heads, c.head_dim

    def P(name):
        return T_(model, "orpheus." + name)

    inv_freq = 1.0 / (500000.0 ** (torch.arange(0, hd, 2, dtype=D) / hd)) / P("rope_frequencies")

    def rope(x, pos):  # x [n][heads][hd]
        ang = pos[:, None].to(D) * inv_freq[None, :]
        cos, sin = torch.cos(ang)[:, None, :], torch.sin(ang)[:, None, :]
        x0, x1 = x[..., : hd // 2], x[..., hd // 2:]
        return torch.cat([x0 * cos - x1 * sin, x0 * sin + x1 * cos], dim=-1)

    def rms(v, w):
        return v * torch.rsqrt((v * v).mean(-1, keepdim=True) + 1e-5) * w

    ks = [torch.zeros(0, NKV, hd, dtype=D) for _ in range(c.layers)]
    vs = [torch.zeros(0, NKV, hd, dtype=D) for _ in range(c.layers)]

    def forward(ids, pos0):
        n = len(ids)
        pos = torch.arange(pos0, pos0 + n)
        x = P("embed_tokens")[torch.tensor(ids, dtype=torch.long)]
        for l in range(c.layers):
            p = f"layers.{l}."
            cur = rms(x, P(p + "input_layernorm"))
            q = rope(Fn.linear(cur, P(p + "self_attn.q_proj")).view(n, NH, hd), pos)
            k = rope(Fn.linear(cur, P(p + "self_attn.k_proj")).view(n, NKV, hd), pos)
            v = Fn.linear(cur, P(p + "self_attn.v_proj")).view(n, NKV, hd)
            ks[l] = torch.cat([ks[l], k]); vs[l] = torch.cat([vs[l], v])
            kk = ks[l].repeat_interleave(NH // NKV, dim=1)
            vv = vs[l].repeat_interleave(NH // NKV, dim=1)
            sc = torch.einsum("nhd,thd->hnt", q, kk) / hd ** 0.5
            mask = torch.arange(kk.shape[0])[None, :] > pos[:, None]
            sc = sc.masked_fill(mask[None], float("-inf"))
            att = torch.einsum("hnt,thd->nhd", torch.softmax(sc, -1), vv).reshape(n, NH * hd)
            x = x + Fn.linear(att, P(p + "self_attn.o_proj"))
            cur = rms(x, P(p + "post_attention_layernorm"))
            x = x + Fn.linear(Fn.silu(Fn.linear(cur, P(p + "mlp.gate_proj"))) * Fn.linear(cur, P(p + "mlp.up_proj")), P(p + "mlp.down_proj"))
        return Fn.linear(rms(x[-1], P("norm")), P("lm_head"))

    logits = [forward(list(prompt), 0)]
    toks = []
    pos = len(prompt)
    for _ in range(steps):
        toks.append(int(logits[-1].argmax()))
        logits.append(forward([toks[-1]], pos))
        pos += 1
    return torch.stack(logits), toks


def torch_dia(model, tokens, sentence_len, ids_seq, cfg_scale=3.0):
    """Dia (src/models/dia/model.cpp:383-659) in float64 torch, batched over the two streams (text / all-zero), with the HF
    formulation of rotary embeddings (closed-form inv_freq, rotate_half) and boolean masks — an independent route to the
    numbers of the C oracle.  tokens [max_ctx] (zero padded); ids_seq [steps][n_out] decoder inputs.
    Returns encoder states [2][S][EH], guided logits [steps][n_out][V] and raw logits [steps][2][n_out][V]."""
    c = model.cfg
    S, hd, NH, NKV, rep, NO = c.max_ctx, c.head_dim, c.dec_heads, c.dec_kv_heads, c.dec_repeat, c.n_out

    def P(name):
        return T_(model, "dia." + name)

    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, hd, 2, dtype=D) / hd))

    def rope(x, pos):  # x [..., n, heads, hd], pos [n]
        ang = pos.to(D)[:, None] * inv_freq[None, :]
        cos, sin = torch.cos(ang)[:, None, :], torch.sin(ang)[:, None, :]
        x0, x1 = x[..., : hd // 2], x[..., hd // 2:]
        return torch.cat([x0 * cos - x1 * sin, x0 * sin + x1 * cos], dim=-1)

    def rms(v, w):
        return v * torch.rsqrt((v * v).mean(-1, keepdim=True) + 1e-5) * w

    def mlp(cur, p):
        return Fn.linear(Fn.silu(Fn.linear(cur, P(p + "gate"))) * Fn.linear(cur, P(p + "up")), P(p + "wo"))

    # ---- encoder: [2][S][EH]
    ids = torch.stack([torch.tensor(tokens.astype(np.int64)), torch.zeros(S, dtype=torch.long)])
    x = P("encoder.embedding")[ids]
    real = torch.arange(S) < sentence_len
    allowed = real[:, None] == real[None, :]                      # real sees real, pad sees pad (set_inputs :712-721)
    pos = torch.arange(S)
    for l in range(c.enc_layers):
        p = f"encoder.layers.{l}."
        cur = rms(x, P(p + "pre_sa_norm"))
        q = rope(Fn.linear(cur, P(p + "q_proj")).view(2, S, c.enc_heads, hd), pos)
        k = rope(Fn.linear(cur, P(p + "k_proj")).view(2, S, c.enc_heads, hd), pos)
        v = Fn.linear(cur, P(p + "v_proj")).view(2, S, c.enc_heads, hd)
        sc = torch.einsum("bnhd,bthd->bhnt", q, k).masked_fill(~allowed[None, None], float("-inf"))   # no 1/sqrt(d) in Dia
        att = torch.einsum("bhnt,bthd->bnhd", torch.softmax(sc, -1), v).reshape(2, S, c.enc_heads * hd)
        x = x + Fn.linear(att, P(p + "o_proj"))
        x = x + mlp(rms(x, P(p + "post_sa_norm")), p)
    enc = rms(x, P("encoder.norm"))

    # ---- cross K/V: keys only for the sentence (the other cache rows are zero), values for every position
    ck, cv = [], []
    for l in range(c.dec_layers):
        p = f"decoder.layers.{l}."
        k = rope(Fn.linear(enc, P(p + "cross_k_proj")).view(2, S, NH, hd), pos)
        ck.append(k * real[None, :, None, None].to(D))
        cv.append(Fn.linear(enc, P(p + "cross_v_proj")).view(2, S, NH, hd))

    ks = [torch.zeros(2, 0, NKV, hd, dtype=D) for _ in range(c.dec_layers)]
    vs = [torch.zeros(2, 0, NKV, hd, dtype=D) for _ in range(c.dec_layers)]
    guided, raws = [], []
    for step, ids_t in enumerate(ids_seq):
        pt = torch.tensor([step])
        x = sum(P(f"decoder.embeddings.{i}")[int(ids_t[i])] for i in range(NO))[None, None, :].repeat(2, 1, 1)   # [2][1][DH]
        for l in range(c.dec_layers):
            p = f"decoder.layers.{l}."
            cur = rms(x, P(p + "pre_sa_norm"))
            q = rope(Fn.linear(cur, P(p + "self_q_proj")).view(2, 1, NH, hd), pt)
            k = rope(Fn.linear(cur, P(p + "self_k_proj")).view(2, 1, NKV, hd), pt)
            v = Fn.linear(cur, P(p + "self_v_proj")).view(2, 1, NKV, hd)
            ks[l] = torch.cat([ks[l], k], dim=1); vs[l] = torch.cat([vs[l], v], dim=1)
            kk, vv = ks[l].repeat_interleave(rep, dim=2), vs[l].repeat_interleave(rep, dim=2)
            sc = torch.einsum("bnhd,bthd->bhnt", q, kk)
            att = torch.einsum("bhnt,bthd->bnhd", torch.softmax(sc, -1), vv).reshape(2, 1, NH * hd)
            x = x + Fn.linear(att, P(p + "self_o_proj"))
            cur = rms(x, P(p + "pre_ca_norm"))
            q = rope(Fn.linear(cur, P(p + "cross_q_proj")).view(2, 1, NH, hd), pt)
            sc = torch.einsum("bnhd,bthd->bhnt", q, ck[l])                        # all S positions, no mask
            att = torch.einsum("bhnt,bthd->bnhd", torch.softmax(sc, -1), cv[l]).reshape(2, 1, NH * hd)
            x = x + Fn.linear(att, P(p + "cross_o_proj"))
            x = x + mlp(rms(x, P(p + "pre_mlp_norm")), p)
        h = rms(x[:, 0], P("decoder.norm"))
        raw = torch.stack([Fn.linear(h, P(f"decoder.heads.{i}")) for i in range(NO)], dim=1)   # [2][NO][V]
        raws.append(raw)
        guided.append(raw[0] + cfg_scale * (raw[0] - raw[1]))
    return enc, torch.stack(guided), torch.stack(raws)


def torch_kokoro(model, tokens, voice_name, noise, attn_scale=0.125, only_durations=False, dbg=None, hsrc_in=None):
    """Kokoro (src/models/kokoro/model.cpp) in float64 torch through library modules — nn.LSTM, F.layer_norm, F.instance_norm,
    F.conv1d / conv_transpose1d (groups, output_padding), F.interpolate (nearest, linear), torch.stft / torch.istft — i.e. the
    PyTorch definitions the C oracle restates by hand.  Returns lengths, duration hidden states, F0 / N curves and the audio."""
    c = model.cfg
    E, H, D, S = c.albert_embd, c.hidden, c.dp_hidden, c.style_half

    def P(name):
        return T_(model, "kokoro." + name)

    def make_lstm(base, inp, hid):
        l = torch.nn.LSTM(inp, hid, batch_first=True, bidirectional=True).double()
        with torch.no_grad():
            for sfx, wn, bn in (("", "weights", "biases"), ("_reverse", "reverse_weights", "reverse_biases")):
                getattr(l, "weight_ih_l0" + sfx).copy_(torch.cat([P(f"{base}.0.{wn}.{2 * j}") for j in range(4)]))
                getattr(l, "weight_hh_l0" + sfx).copy_(torch.cat([P(f"{base}.0.{wn}.{2 * j + 1}") for j in range(4)]))
                getattr(l, "bias_ih_l0" + sfx).copy_(torch.cat([P(f"{base}.0.{bn}.{2 * j}") for j in range(4)]))
                getattr(l, "bias_hh_l0" + sfx).copy_(torch.cat([P(f"{base}.0.{bn}.{2 * j + 1}") for j in range(4)]))
        return l

    n = len(tokens)
    ids = torch.tensor(np.asarray(tokens, dtype=np.int64))
    voice = P("voice_tensors." + voice_name)
    style_p, style_d = voice[n - 3, S:], voice[n - 3, :S]
    # ---- ALBERT
    x = P("albert.token_embd")[ids] + P("albert.position_embd")[:n] + P("albert.token_type_embd")
    x = Fn.layer_norm(x, (E,), P("albert.norm"), P("albert.norm_bias"), eps=1e-12)
    x = Fn.linear(x, P("albert.embd"), P("albert.embd_bias"))
    L0 = "albert.layer.0."
    hs = H // c.heads
    for _ in range(c.recurrence):
        q = Fn.linear(x, P(L0 + "q"), P(L0 + "q_bias")).view(n, c.heads, hs)
        k = Fn.linear(x, P(L0 + "k"), P(L0 + "k_bias")).view(n, c.heads, hs)
        v = Fn.linear(x, P(L0 + "v"), P(L0 + "v_bias")).view(n, c.heads, hs)
        att = torch.einsum("hnt,thd->nhd", torch.softmax(torch.einsum("nhd,thd->hnt", q, k) * attn_scale, -1), v).reshape(n, H)
        x = Fn.layer_norm(Fn.linear(att, P(L0 + "o"), P(L0 + "o_bias")) + x, (H,), P(L0 + "ffn_norm"), P(L0 + "ffn_norm_bias"), eps=1e-12)
        ff = Fn.linear(Fn.gelu(Fn.linear(x, P(L0 + "ffn"), P(L0 + "ffn_bias")), approximate="tanh"), P(L0 + "ffn_out"), P(L0 + "ffn_out_bias"))
        x = Fn.layer_norm(ff + x, (H,), P(L0 + "attn_norm"), P(L0 + "attn_norm_bias"), eps=1e-12)
    # ---- prosody predictor
    dp = "duration_predictor."
    cur = torch.cat([Fn.linear(x, P(dp + "encode"), P(dp + "encode_bias")), style_p.expand(n, S)], dim=1)
    for l in range(c.dp_layers):
        y, _ = make_lstm(f"{dp}layers.{2 * l}.lstm", D + S, D // 2)(cur[None])
        gamma = Fn.linear(style_p, P(f"{dp}layers.{2 * l + 1}.gamma_weight"), P(f"{dp}layers.{2 * l + 1}.gamma_bias"))
        beta = Fn.linear(style_p, P(f"{dp}layers.{2 * l + 1}.beta_weight"), P(f"{dp}layers.{2 * l + 1}.beta_bias"))
        y = Fn.layer_norm(y[0], (D,), eps=1e-5)
        cur = torch.cat([(1 + gamma) * y + beta, style_p.expand(n, S)], dim=1)
    hidden = cur
    y, _ = make_lstm(dp + "duration_lstm", D + S, D // 2)(cur[None])
    dur = torch.sigmoid(Fn.linear(y[0], P(dp + "duration_proj"), P(dp + "duration_proj_bias"))).sum(-1)
    lens = torch.clamp(torch.floor(dur + 0.5), 1, 50)     # roundf for positive sums
    if only_durations:
        return lens, hidden
    # ---- alignment + F0 / N
    tok_of = torch.repeat_interleave(torch.arange(n), lens.long())
    T = len(tok_of)
    en = hidden[tok_of]
    sh, _ = make_lstm(dp + "shared_lstm", D + S, D // 2)(en[None])
    sh = sh[0].t()[None]                                   # [1][D][T]

    def adain(x_, style, base, k):
        g = Fn.linear(style, P(f"{base}.{k}_gamma_weight"), P(f"{base}.{k}_gamma_bias"))
        b = Fn.linear(style, P(f"{base}.{k}_beta_weight"), P(f"{base}.{k}_beta_bias"))
        return (1 + g)[None, :, None] * Fn.instance_norm(x_, eps=1e-5) + b[None, :, None]

    def ada_block(x_, style, base):
        up = ("kokoro." + base + ".pool_weight") in model.by_name
        y_ = Fn.leaky_relu(adain(x_, style, base, "norm1"), 0.2)
        if up:
            y_ = Fn.conv_transpose1d(y_, P(base + ".pool_weight"), P(base + ".pool_bias"), stride=2, padding=1, output_padding=1, groups=y_.shape[1])
        y_ = Fn.conv1d(y_, P(base + ".conv1_weight"), P(base + ".conv1_bias"), padding=1)
        y_ = Fn.leaky_relu(adain(y_, style, base, "norm2"), 0.2)
        y_ = Fn.conv1d(y_, P(base + ".conv2_weight"), P(base + ".conv2_bias"), padding=1)
        s_ = x_
        if ("kokoro." + base + ".conv1x1_weight") in model.by_name:
            if up:
                s_ = Fn.interpolate(s_, scale_factor=2, mode="nearest")
            s_ = Fn.conv1d(s_, P(base + ".conv1x1_weight"))
        return (y_ + s_) / math.sqrt(2.0)

    curves = []
    for br in ("f0", "n"):
        cur = sh
        for i in range(c.f0_blocks):
            cur = ada_block(cur, style_p, f"{dp}{br}_blocks.{i}")
        curves.append(Fn.conv1d(cur, P(f"{dp}{br}_proj_kernel"), P(f"{dp}{br}_proj_bias"))[0, 0])
    f0c, nc = curves
    # ---- text encoder
    t = P("text_encoder.embedding_weight")[ids].t()[None]
    for l in range(c.conv_layers):
        t = Fn.conv1d(t, P(f"text_encoder.layers.{l}.weight"), P(f"text_encoder.layers.{l}.bias"), padding=2)
        t = Fn.layer_norm(t.transpose(1, 2), (t.shape[1],), P(f"text_encoder.layers.{l}.gamma"), P(f"text_encoder.layers.{l}.beta"), eps=1e-5).transpose(1, 2)
        t = Fn.leaky_relu(t, 0.2)
    tl, _ = make_lstm("text_encoder.lstm", D, D // 2)(t.transpose(1, 2))
    asr = tl[0][tok_of].t()[None]                          # [1][C][T]
    # ---- decoder
    de = "decoder."
    f0d = Fn.conv1d(f0c[None, None], P(de + "f0_conv_weight"), P(de + "f0_conv_bias"), stride=2, padding=1)
    nd = Fn.conv1d(nc[None, None], P(de + "n_conv_weight"), P(de + "n_conv_bias"), stride=2, padding=1)
    cur = ada_block(torch.cat([asr, f0d, nd], dim=1), style_d, de + "encoder_block")
    asr_res = Fn.conv1d(asr, P(de + "asr_conv_weight"), P(de + "asr_conv_bias"))
    for i in range(c.decoder_blocks):
        cur = ada_block(torch.cat([cur, asr_res, f0d, nd], dim=1), style_d, f"{de}decoder_blocks.{i}")
    if dbg is not None:
        dbg["asr"], dbg["dec_out"] = asr[0], cur[0]
    # ---- generator: harmonic source
    g = de + "generator."
    up = int(np.prod(c.up_rates)) * c.hop
    NHm, L2 = c.harmonic_num + 1, f0c.shape[0]
    LS = L2 * up
    harm = torch.arange(1, NHm + 1, dtype=D_)[:, None]
    rad = torch.remainder(f0c[None, :] * harm / 24000.0, 1.0)
    phase = torch.cumsum(rad, dim=1) * (up * 2 * math.pi)
    phase = Fn.interpolate(phase[None], scale_factor=up, mode="linear", align_corners=False)[0]
    f0u = Fn.interpolate(f0c[None, None], scale_factor=up, mode="nearest")[0, 0]
    voiced = f0u > 10.0
    nz = torch.from_numpy(np.asarray(noise, dtype=np.float64)).view(NHm, LS)
    sine = torch.sin(phase) * (voiced * 0.1)[None] + torch.where(voiced, torch.tensor(0.003, dtype=D_), torch.tensor(0.1 / 3.0, dtype=D_))[None] * nz
    har = torch.tanh(Fn.linear(sine.t(), P(g + "m_source_weight"), P(g + "m_source_bias"))[:, 0])
    N, hop = c.n_fft, c.hop
    win = torch.sin(math.pi * torch.arange(N, dtype=D_) / N) ** 2
    spec = torch.stft(har, N, hop, N, window=win, center=True, pad_mode="reflect", return_complex=True)     # [11][F]
    hsrc = torch.cat([spec.abs(), spec.angle()], dim=0)[None]
    hsrc_own = hsrc[0]
    if hsrc_in is not None:
        hsrc = torch.from_numpy(np.asarray(hsrc_in, dtype=np.float64))[None]
    if dbg is not None:
        dbg["sine"], dbg["har"], dbg["hsrc"] = sine, har, hsrc[0]

    def gen_res(x_, base, pads, dils):
        for j in range(3):
            def ad(y_, a):
                gm = Fn.linear(style_d, P(f"{base}.{j}.gamma{a}_weight"), P(f"{base}.{j}.gamma{a}_bias"))
                bt = Fn.linear(style_d, P(f"{base}.{j}.beta{a}_weight"), P(f"{base}.{j}.beta{a}_bias"))
                y_ = (1 + gm)[None, :, None] * Fn.instance_norm(y_, eps=1e-5) + bt[None, :, None]
                al = P(f"{base}.{j}.alpha{a}")
                return y_ + torch.sin(al * y_) ** 2 / al
            y_ = Fn.conv1d(ad(x_, "1"), P(f"{base}.{j}.convs1_weight"), P(f"{base}.{j}.convs1_bias"), padding=pads[j], dilation=dils[j])
            y_ = Fn.conv1d(ad(y_, "2"), P(f"{base}.{j}.convs2_weight"), P(f"{base}.{j}.convs2_bias"), padding=pads[0])
            x_ = x_ + y_
        return x_

    geo = model.geometry
    nk = len(c.res_kernels)
    for i in range(len(c.up_rates)):
        cur = Fn.leaky_relu(cur, 0.1)
        cur = Fn.conv_transpose1d(cur, P(f"{g}ups.{i}.weight"), P(f"{g}ups.{i}.bias"), stride=geo["up"][i][0], padding=geo["up"][i][1])
        if i == len(c.up_rates) - 1:
            cur = Fn.pad(cur, (1, 0), mode="reflect")
        xs = Fn.conv1d(hsrc, P(f"{g}noise_blocks.{i}.conv_weight"), P(f"{g}noise_blocks.{i}.conv_bias"), stride=geo["noise"][i][0], padding=geo["noise"][i][1])
        xs = gen_res(xs, f"{g}noise_blocks.{i}.resblock", [p_ for p_, _ in geo["noise_res"][i]], [d_ for _, d_ in geo["noise_res"][i]])
        cur = cur + xs
        cur = sum(gen_res(cur, f"{g}resblocks.{i * nk + ii}", [p_ for p_, _ in geo["res"][i * nk + ii]], [d_ for _, d_ in geo["res"][i * nk + ii]]) for ii in range(nk)) / nk
        if dbg is not None:
            dbg[f"gen_stage{i}"] = cur[0]
    cur = Fn.conv1d(Fn.leaky_relu(cur, 0.01), P(g + "conv_post_weight"), P(g + "conv_post_bias"), padding=3)[0]
    nb = N // 2 + 1
    mag, ph = torch.exp(cur[:nb]), torch.sin(cur[nb:])
    if dbg is not None:
        dbg["post"] = torch.cat([mag, ph])
    out_len = T * c.up_sampling_factor
    y = torch.istft(torch.polar(mag, ph), N, hop, N, window=win, center=True, length=out_len)
    # torch divides by the window envelope of its F frames; the reference by compute_window_squared_sum (util.cpp:203-217), which
    # adds out_len / hop + N / 2 / hop frames: rescale
    Fr = mag.shape[1]
    w2 = (win ** 2).numpy()
    env_t, env_r = np.zeros(out_len + 2 * N), np.zeros(out_len + 2 * N)
    for f in range(Fr):
        env_t[f * hop:f * hop + N] += w2
    for f in range(out_len // hop + (N // 2) // hop):
        env_r[f * hop:f * hop + N] += w2
    half = N // 2
    y = y * torch.from_numpy(env_t[half:half + out_len] / env_r[half:half + out_len])
    return lens, hidden, f0c, nc, y, hsrc_own


def T_(model, name):
    return torch.from_numpy(model.by_name[name].to_f32().astype(np.float64))


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    km = synth.build_kokoro(synth.kokoro_tiny())
    ktoks = np.array([0, 5, 9, 3, 16, 7, 21, 0], dtype=np.uint32)
    knoise = np.random.default_rng(21).random(9 * 50 * km.cfg.up_sampling_factor, dtype=np.float32)   # the tests draw the same stream (seed 21) instead of storing it
    with torch.no_grad():
        klens, _ = torch_kokoro(km, ktoks, "af_test", None, only_durations=True)   # lengths first: they size the noise
        kn = knoise[:9 * int(klens.sum()) * km.cfg.up_sampling_factor]
        _, _, _, _, _, khs = torch_kokoro(km, ktoks, "af_test", kn)
        # the audio is computed from the fp32-rounded conditioning that is stored, so that an implementation fed the same
        # conditioning can be compared sample for sample (the phase channels are discontinuous at +-pi)
        klens, khid, kf0, kn_, kpcm, _ = torch_kokoro(km, ktoks, "af_test", kn, hsrc_in=khs.numpy().astype(np.float32))
    np.savez_compressed(os.path.join(out_dir, "tiny_kokoro.npz"), tokens=ktoks, noise_seed=np.int32(21), lens=klens.numpy().astype(np.float32), hidden=khid.numpy().astype(np.float32),
                        f0=kf0.numpy().astype(np.float32), n=kn_.numpy().astype(np.float32), pcm=kpcm.numpy().astype(np.float32), hsrc=khs.numpy().astype(np.float32))
    print("wrote tiny_kokoro.npz", "lens", klens.numpy())
    dm = synth.build_dia(synth.dia_tiny())
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from oracle import dia_tokenize
    dtoks, dn = dia_tokenize("[S1] Hi there [S2] ok", dm.cfg.max_ctx)
    rd = np.random.default_rng(11)
    dids = rd.integers(0, dm.cfg.audio_vocab, (5, dm.cfg.n_out)).astype(np.uint32)
    dids[0] = dm.cfg.bos
    with torch.no_grad():
        denc, dguided, draw = torch_dia(dm, dtoks, dn, dids)
    np.savez_compressed(os.path.join(out_dir, "tiny_dia.npz"), tokens=dtoks, sentence_len=np.int32(dn), ids=dids, enc=denc.numpy().astype(np.float32),
                        logits=dguided.numpy().astype(np.float32), raw=draw.numpy().astype(np.float32))
    print("wrote tiny_dia.npz")
    om = synth.build_orpheus(synth.orpheus_tiny())
    op = np.random.default_rng(5).integers(0, om.cfg.vocab, 9).astype(np.uint32)
    with torch.no_grad():
        olog, otoks = torch_orpheus(om, op, 6)
    np.savez_compressed(os.path.join(out_dir, "tiny_orpheus.npz"), prompt=op, logits=olog.numpy().astype(np.float32), tokens=np.array(otoks, dtype=np.uint32))
    print("wrote tiny_orpheus.npz")
    sn = synth.build_snac(synth.snac_tiny())
    rs = np.random.default_rng(99)
    Tn = 12
    sn_codes = np.concatenate([rs.integers(0, sn.cfg.cb_size, Tn // r) for r in sn.cfg.repeats]).astype(np.uint32)
    nlen, L = 0, Tn
    for s_ in sn.cfg.strides:
        L *= s_
        nlen += L
    sn_noise = rs.standard_normal(nlen).astype(np.float32)
    with torch.no_grad():
        np.savez_compressed(os.path.join(out_dir, "tiny_snac.npz"), codes=sn_codes, T=np.int32(Tn), noise=sn_noise,
                            pcm_noise=torch_snac(sn, sn_codes, Tn, sn_noise).numpy().astype(np.float32),
                            pcm_clean=torch_snac(sn, sn_codes, Tn, None).numpy().astype(np.float32))
    print("wrote tiny_snac.npz")
    t5 = synth.build_t5(synth.t5_tiny())
    t5p = synth.build_t5(synth.t5_tiny(output_size=192, seed=0x76))  # with the down projection
    ids = np.random.default_rng(77).integers(3, 160, 40).astype(np.uint32)
    with torch.no_grad():
        np.savez_compressed(os.path.join(out_dir, "tiny_t5.npz"), ids=ids, out=torch_t5(t5, ids).numpy().astype(np.float32),
                            out_proj=torch_t5(t5p, ids[:23]).numpy().astype(np.float32))
    print("wrote tiny_t5.npz")
    cfg = synth.tiny(weight_type=gguf.F32)
    model = synth.build(cfg)
    rng = np.random.default_rng(1234)
    prompt = rng.integers(3, cfg.prompt_vocab, 7).astype(np.uint32)
    n_steps = 5
    audio_ids = rng.integers(0, cfg.audio_vocab, (n_steps, cfg.n_out)).astype(np.uint32)
    tp = TorchParler(model)
    with torch.no_grad():
        _, h0 = tp.decode(prompt, 0, audio=False)
        logits, hidden = [], []
        for s in range(n_steps):
            lg, h = tp.decode(audio_ids[s], len(prompt) + s, audio=True)
            logits.append(lg[:, 0, :].numpy())
            hidden.append(h[0].numpy())
        k0 = tp.k[0].numpy()
        v1 = tp.v[cfg.layers - 1].numpy()
        codes = rng.integers(0, cfg.cb_size, (9, cfg.n_out)).astype(np.uint32)
        pcm, stages = torch_dac(model, codes, stages=True)
    np.savez_compressed(
        os.path.join(out_dir, "tiny_f32.npz"),
        prompt=prompt, audio_ids=audio_ids, prompt_hidden=h0.numpy().astype(np.float32),
        logits=np.stack(logits).astype(np.float32), hidden=np.stack(hidden).astype(np.float32),
        k_layer0=k0.astype(np.float32), v_last=v1.astype(np.float32),
        codes=codes, pcm=pcm.numpy().astype(np.float32),
        **{f"dac_stage{i}": s.numpy().astype(np.float32) for i, s in enumerate(stages)},
    )
    print("wrote tiny_f32.npz", {k: v.shape for k, v in np.load(os.path.join(out_dir, "tiny_f32.npz")).items()})


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Golden vectors from implementations this repository did NOT write (build container only; the .npz files are committed, this
script never travels to a GPU box and nothing under tests/ imports transformers or /root/reference at test time).

The oracle (oracle/*.c) restates the reference's graphs; its float results were so far pinned only to tests/golden/make_golden.py, a
second restatement by the same author.  Here every model family is instantiated from the library the reference's converters
(/root/reference/py-gguf/tts_encoders/*.py) convert FROM — Hugging Face `transformers` — with tiny seeded dimensions, run in float64,
and its weights are exported under the converters' tensor names and layout rules into the same container the GGUF writer takes.
tests/test_upstream_golden.py feeds those tensors to the oracle and compares.

  family            upstream class (the converter's source)                      converter rules followed (file:line)
  orpheus           transformers LlamaForCausalLM, llama3 rope scaling            orpheus_gguf_encoder.py:118-122 (names), :145-173 (rope factors)
  t5 encoder        transformers T5EncoderModel (gated-gelu = flan)               t5_encoder_gguf_encoder.py:62-80
  dac               transformers DacModel decoder + residual VQ from_codes        dac_gguf_encoder.py:7-35 (names), :43-110; weight norm folded by the
                                                                                  reference's own tensor_util.get_regularized_weight (imported from
                                                                                  /root/reference/py-gguf/tts_encoders/tensor_util.py, torch only)
  dia               transformers DiaForConditionalGeneration (encoder, decoder,  dia_gguf_encoder.py:74-129
                    logits_dense)
  prompt tokenizer  tokenizers.models.Unigram behind T5's pre-tokenizer           parler_tts_gguf_encoder.py:187-202
  orpheus tokenizer tokenizers.models.BPE, byte-level                             orpheus_gguf_encoder.py:231-242
  delay pattern     MusicgenForCausalLM.build / apply_delay_pattern_mask          (no converter rule: generation logic, model.cpp:734-785)
  kokoro albert     transformers AlbertModel (kokoro's `bert` is one)             kokoro_gguf_encoder.py:14-37, :274-287
  kokoro stages     torch.nn.LSTM, F.instance_norm, F.conv_transpose1d(groups,    kokoro_gguf_encoder.py:289-309 (LSTM gate split); kokoro/model.cpp:35-134,
  + snac depthwise  output_padding), torch.stft / istft, F.conv1d(groups)        util.cpp:111-133, 203-217; decoder/snac_model.cpp:86-110
  parler decoder    transformers MusicgenForCausalLM (Parler-TTS' decoder is a    parler_tts_gguf_encoder.py:112-130
                    fork of it: same modules and parameter names; parler_tts
                    itself is not installed here)

Intentional divergences of the REFERENCE from these upstream models, each visible in the numbers this script prints and stored in the
fixtures (the tests assert them):
  * T5 relative position buckets: t5/model.cpp:303-316 takes log of the INTEGER quotient |d| / max_exact, HF of the float quotient — equal
    for |d| < 8 and at |d| = 8, 16, 32, 64, different elsewhere.  The fixture holds HF's output for a 7-token input (identical buckets),
    and for a 24-token input both HF's own output and HF's output with the reference's bucket formula patched in.
  * GELU: the reference evaluates tanh-GELU (ggml_gelu) where Parler-TTS' config asks for erf-GELU; the Musicgen twin is configured with
    gelu_pytorch_tanh so that the comparison isolates everything else, and the erf variant's distance is recorded.
  * Dia cross-attention: the reference ropes the cross-attention query (decoder position, dia/model.cpp:606) and keys (encoder position, :489) —
    what the `dia` package its converter imports did; transformers' DiaCrossAttention applies no rope.  The oracle has a switch for it
    (orc_dia_model.no_cross_rope): with the switch it equals transformers to 2e-5, without it (the reference's graph) it does not.
  * BPE prompts with doubled spaces: upstream emits the extra space as its own token, the reference drops it (tokenizer.cpp:209-296 splits at
    spaces and discards empty pieces).
  * delay pattern at the END of a generation: upstream knows max_length in advance and feeds pad to codebook k for its last 8 - k positions; the
    reference keeps feeding what it sampled (it stops on EOS / max_generation instead).  The frames both keep are the same.
  * snake: HF's Snake1d divides by (alpha + 1e-9), snake_1d (src/util.cpp:96-101) by alpha: 1e-9 relative, below fp32 resolution.
Run:  python tests/golden/make_upstream_golden.py        (writes tests/golden/upstream_*.npz)
"""
import importlib.util
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF_UTIL = "/root/reference/py-gguf/tts_encoders/tensor_util.py"


def npy(t):
    return t.detach().to(torch.float64).cpu().numpy()


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KB)")


# ---------------------------------------------------------------------------------------------------------------------------------
def make_orpheus():
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(1001)
    kw = dict(vocab_size=200, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
              head_dim=128, max_position_embeddings=131072, rms_norm_eps=1e-5, attention_bias=False, mlp_bias=False, tie_word_embeddings=False)
    rope = dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0, original_max_position_embeddings=8192, rope_theta=500000.0)
    try:
        cfg = LlamaConfig(**kw, rope_parameters=rope)
    except TypeError:
        cfg = LlamaConfig(**kw, rope_theta=500000.0, rope_scaling={k: v for k, v in rope.items() if k != "rope_theta"})
    model = LlamaForCausalLM(cfg).double().eval()
    with torch.no_grad():   # norms away from 1 so that their placement matters
        for n, p in model.named_parameters():
            if "norm" in n:
                p.copy_(1.0 + 0.1 * torch.randn_like(p))
    tensors = {}
    for name, param in model.model.named_parameters():              # orpheus_gguf_encoder.py:118-121
        tensors[f"orpheus.{name[:-7]}"] = npy(param).astype(np.float32)
    tensors["orpheus.lm_head"] = npy(model.lm_head.weight).astype(np.float32)   # :122
    # :145-173 prepare_rope_frequencies, as written there
    base, dim = 500000.0, 128
    freqs = 1.0 / (base ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))
    factor, low, high, old = 8.0, 1.0, 4.0, 8192
    lw, hw = old / low, old / high
    rf = []
    for f in freqs:
        wl = 2 * math.pi / f
        if wl < hw:
            rf.append(1)
        elif wl > lw:
            rf.append(factor)
        else:
            smooth = (old / wl - low) / (high - low)
            rf.append(1 / ((1 - smooth) / factor + smooth))
    tensors["orpheus.rope_frequencies"] = np.array([float(x) for x in rf], dtype=np.float32)

    # the converter exports fp32: run the upstream model on the fp32-rounded weights, in float64
    with torch.no_grad():
        for name, param in model.named_parameters():
            param.copy_(param.to(torch.float32).to(torch.float64))
    rng = np.random.default_rng(7)
    n = 2304   # beyond 2048 positions: the llama3 low-frequency scaling is visible in the logits
    ids = rng.integers(0, 200, n)
    with torch.no_grad():
        out = model(torch.tensor(ids[None]), output_hidden_states=True)
    logits = npy(out.logits[0])
    hidden = npy(out.hidden_states[-1][0])   # after the final norm
    save("upstream_orpheus.npz", ids=ids.astype(np.uint32), logits_last=logits[-1], logits_at_40=logits[40], hidden_rows=hidden[[0, 40, n - 1]],
         hidden_row_index=np.array([0, 40, n - 1]), cfg=np.array([64, 2, 2, 1, 128, 128, 200]),
         **{"t:" + k: v for k, v in tensors.items()})


# ---------------------------------------------------------------------------------------------------------------------------------
def reference_bucket(rel, n_buckets_total=32):
    """t5/model.cpp:303-316 as written (rpos = i - ii = key - query)"""
    n_buckets = n_buckets_total // 2
    max_exact = n_buckets // 2
    den = float(np.float32(math.log(128.0 / max_exact)))   # `float logarithmic_denominator` (:309): at |d| = 16..23 the quotient
    out = np.zeros_like(rel)                                # log(2) / den * 8 = 1.99999999 truncates to 1, not 2
    for idx, r in np.ndenumerate(rel):
        ab = abs(int(r))
        v = ab if ab < max_exact else min(n_buckets - 1, max_exact + int((math.log(ab // max_exact) / den) * max_exact))
        out[idx] = (n_buckets if r > 0 else 0) + v
    return out


def make_t5():
    from transformers import T5Config, T5EncoderModel
    from transformers.models.t5 import modeling_t5

    torch.manual_seed(1002)
    cfg = T5Config(vocab_size=120, d_model=128, d_kv=64, d_ff=256, num_layers=2, num_heads=2, relative_attention_num_buckets=32,   # head size 64 (flan-t5)
                   relative_attention_max_distance=128, feed_forward_proj="gated-gelu", layer_norm_epsilon=1e-6, dropout_rate=0.0)
    model = T5EncoderModel(cfg).double().eval()
    proj = torch.nn.Linear(128, 64).double()   # enc_to_dec_proj (t5_encoder_gguf_encoder.py:63-65)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "layer_norm" in n:
                p.copy_(1.0 + 0.1 * torch.randn_like(p))
            if "relative_attention_bias" in n:
                p.copy_(torch.randn_like(p))
        for p in list(model.parameters()) + list(proj.parameters()):
            p.copy_(p.to(torch.float32).to(torch.float64))
    enc = model.encoder
    t = {"t5encoder.down_proj": proj.weight, "t5encoder.down_proj_bias": proj.bias, "t5encoder.token_embd": enc.embed_tokens.weight,
         "t5encoder.enc.final_layer_norm": enc.final_layer_norm.weight}
    for i, layer in enumerate(enc.block):                                                             # :66-80
        if i == 0:
            t[f"t5encoder.enc.blk.{i}.attn_rel_b"] = layer.layer[0].SelfAttention.relative_attention_bias.weight
        sa, ff = layer.layer[0].SelfAttention, layer.layer[1].DenseReluDense
        t[f"t5encoder.enc.blk.{i}.attn_q"], t[f"t5encoder.enc.blk.{i}.attn_k"] = sa.q.weight, sa.k.weight
        t[f"t5encoder.enc.blk.{i}.attn_v"], t[f"t5encoder.enc.blk.{i}.attn_o"] = sa.v.weight, sa.o.weight
        t[f"t5encoder.enc.blk.{i}.attn_norm"] = layer.layer[0].layer_norm.weight
        t[f"t5encoder.enc.blk.{i}.ffn_up"], t[f"t5encoder.enc.blk.{i}.ffn_gate"] = ff.wi_0.weight, ff.wi_1.weight
        t[f"t5encoder.enc.blk.{i}.ffn_down"] = ff.wo.weight
        t[f"t5encoder.enc.blk.{i}.ffn_norm"] = layer.layer[1].layer_norm.weight
    tensors = {k: npy(v).astype(np.float32) for k, v in t.items()}

    def run(ids):
        with torch.no_grad():
            h = model(torch.tensor(ids[None])).last_hidden_state
            return npy(proj(h)[0])

    rng = np.random.default_rng(8)
    ids7, ids24 = rng.integers(3, 120, 7), rng.integers(3, 120, 24)
    out7, out24_hf = run(ids7), run(ids24)
    # the same upstream model with the reference's bucket formula patched in (everything else HF's)
    attn = modeling_t5.T5Attention
    orig = attn._relative_position_bucket

    def patched(relative_position, bidirectional=True, num_buckets=32, max_distance=128):
        # HF: relative_position = memory_position - query_position (key - query).  The reference's rpos = i - ii with i the KEY and ii the
        # QUERY position (pos_bucket[i * n + ii] lands on kq's (ne0 = key, ne1 = query) after build_t5_pos_bias' permute): the same sign
        rel = relative_position.cpu().numpy()
        return torch.tensor(reference_bucket(rel, num_buckets), dtype=torch.long)

    attn._relative_position_bucket = staticmethod(patched)
    try:
        out24_ref = run(ids24)
        out7_ref = run(ids7)
    finally:
        attn._relative_position_bucket = orig
    print(f"t5: |HF - HF with reference buckets| 7 tokens {np.abs(out7 - out7_ref).max():.2e}, 24 tokens {np.abs(out24_hf - out24_ref).max():.2e} "
          f"(max |out| {np.abs(out24_hf).max():.2f})")
    save("upstream_t5.npz", ids7=ids7.astype(np.uint32), ids24=ids24.astype(np.uint32), out7=out7, out24_hf=out24_hf, out24_refbuckets=out24_ref,
         cfg=np.array([120, 128, 64, 256, 2, 2, 64]), **{"t:" + k: v for k, v in tensors.items()})


# ---------------------------------------------------------------------------------------------------------------------------------
def make_dac(name="upstream_dac.npz", hidden=96, strides=(4, 2), latent=64, seed=1003, frames=11):
    """hidden 96 / strides (4, 2): thin, every layer on the generic kernels.  hidden 192 / strides (2, 2) (upstream_dac_b3.npz): a
    96-channel class, i.e. the device's fused bf16 x 3 residual unit and bf16 x 3 transposed conv, against upstream directly."""
    from transformers import DacConfig, DacModel

    spec = importlib.util.spec_from_file_location("ref_tensor_util", REF_UTIL)   # the reference's own weight-norm folding (torch only)
    ref_util = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_util)

    torch.manual_seed(seed)
    strides = list(strides)
    cfg = DacConfig(encoder_hidden_size=8, downsampling_ratios=[2, 4], decoder_hidden_size=hidden, upsampling_ratios=strides, n_codebooks=4,
                    codebook_size=64, codebook_dim=8, hidden_size=latent, sampling_rate=44100)
    model = DacModel(cfg).double().eval()
    dec, qz = model.decoder, model.quantizer
    # descript-audio-codec checkpoints (what dac_gguf_encoder.py reads) carry weight_g / weight_v: old-style weight norm on every conv
    convs = [dec.conv1, dec.conv2] + [q.out_proj for q in qz.quantizers]
    for blk in dec.block:
        convs += [blk.conv_t1] + [c for ru in (blk.res_unit1, blk.res_unit2, blk.res_unit3) for c in (ru.conv1, ru.conv2)]
    for c in convs:
        torch.nn.utils.weight_norm(c)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("weight_g"):
                p.mul_(0.5 + torch.rand_like(p))        # g != |v|: the fold matters
            if n.endswith("alpha"):
                p.copy_(0.5 + 1.5 * torch.rand_like(p))
            if n.endswith("bias"):
                p.copy_(0.1 * torch.randn_like(p))
    # tensor names: the converter's part maps (dac_gguf_encoder.py:7-35) applied to the same ROLES in HF's module tree
    t = {}

    def folded(module, prefix, modules):
        return ref_util.get_regularized_weight(modules, prefix + ".weight_g")   # dac_gguf_encoder.py:53-56

    mods = {n: m for n, m in dec.named_modules()}
    t["audio_encoder.initial.weight"], t["audio_encoder.initial.bias"] = folded(dec.conv1, "conv1", mods), dec.conv1.bias
    for bi, blk in enumerate(dec.block):
        p = f"audio_encoder.decoder_block.{bi + 1}."
        t[p + "final.alpha"] = blk.snake1.alpha
        t[p + "final.weight"], t[p + "final.bias"] = folded(blk.conv_t1, f"block.{bi}.conv_t1", mods), blk.conv_t1.bias
        for r, ru in enumerate((blk.res_unit1, blk.res_unit2, blk.res_unit3)):
            q = p + f"residual_unit.{r}.res."
            t[q + "initial.alpha"], t[q + "final.alpha"] = ru.snake1.alpha, ru.snake2.alpha
            t[q + "initial.weight"], t[q + "initial.bias"] = folded(ru.conv1, f"block.{bi}.res_unit{r + 1}.conv1", mods), ru.conv1.bias
            t[q + "final.weight"], t[q + "final.bias"] = folded(ru.conv2, f"block.{bi}.res_unit{r + 1}.conv2", mods), ru.conv2.bias
    t["audio_encoder.final.alpha"] = dec.snake1.alpha
    t["audio_encoder.final.weight"], t["audio_encoder.final.bias"] = folded(dec.conv2, "conv2", mods), dec.conv2.bias
    qmods = {n: m for n, m in qz.named_modules()}
    for i, q in enumerate(qz.quantizers):                                      # dac_gguf_encoder.py:82-97: audio_encoder.<quantizer name>
        p = f"audio_encoder.quantizers.{i}."
        t[p + "codebook.weight"] = q.codebook.weight
        t[p + "out_proj.weight"], t[p + "out_proj.bias"] = folded(q.out_proj, f"quantizers.{i}.out_proj", qmods), q.out_proj.bias
    tensors = {k: npy(v).astype(np.float32) for k, v in t.items()}
    # upstream forward on the exported (fp32-rounded, folded) weights: remove the weight norm and write them back
    for c in convs:
        torch.nn.utils.remove_weight_norm(c)
    with torch.no_grad():
        for p_ in model.parameters():
            p_.copy_(p_.to(torch.float32).to(torch.float64))
    rng = np.random.default_rng(9)
    codes = rng.integers(0, 64, (frames, 4))
    stages = {}
    with torch.no_grad():
        z = qz.from_codes(torch.tensor(codes.T[None]))[0]
        stages["stage0"] = npy(z[0])
        h = dec.conv1(z)
        stages["stage1"] = npy(h[0])
        for bi, blk in enumerate(dec.block):
            h = blk(h)
            stages[f"stage{2 + bi}"] = npy(h[0])
        pcm = npy(dec.tanh(dec.conv2(dec.snake1(h)))[0, 0])
        pcm_api = npy(model.decode(audio_codes=torch.tensor(codes.T[None])).audio_values.reshape(-1))
    assert np.abs(pcm - pcm_api).max() < 1e-12, "stage-by-stage walk differs from DacModel.decode"
    if frames > 64:   # the long fixture: stages in fp32 (they are compared at 1e-5 of their range)
        stages = {k: v.astype(np.float32) for k, v in stages.items()}
    save(name, codes=codes.astype(np.uint32), pcm=pcm, strides=np.array(strides), cfg=np.array([latent, 8, 64, hidden, 4]), **stages,
         **{"t:" + k: v for k, v in tensors.items()})


# ---------------------------------------------------------------------------------------------------------------------------------
def make_parler():
    from transformers import MusicgenDecoderConfig, MusicgenForCausalLM

    torch.manual_seed(1004)
    H, NCB, V, L, F, HEADS, ENC, PV = 128, 4, 80, 2, 256, 2, 6, 160   # head size 64: what the device kernels are specialised for

    def build(act):
        torch.manual_seed(1004)
        cfg = MusicgenDecoderConfig(vocab_size=V, hidden_size=H, num_hidden_layers=L, ffn_dim=F, num_attention_heads=HEADS, num_codebooks=NCB,
                                    max_position_embeddings=128, activation_function=act, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0,
                                    layerdrop=0.0, scale_embedding=False, audio_channels=1, pad_token_id=V, bos_token_id=V + 1, tie_word_embeddings=False)
        m = MusicgenForCausalLM(cfg).double().eval()
        with torch.no_grad():
            for n, p in m.named_parameters():
                if "layer_norm" in n and n.endswith("weight"):
                    p.copy_(1.0 + 0.1 * torch.randn_like(p))
                elif "layer_norm" in n:
                    p.copy_(0.1 * torch.randn_like(p))
                else:
                    p.copy_(0.08 * torch.randn_like(p))
            for p in m.parameters():
                p.copy_(p.to(torch.float32).to(torch.float64))
        return m

    model = build("gelu_pytorch_tanh")
    dec = model.model.decoder
    gen = torch.Generator().manual_seed(5)
    embed_prompts = (0.05 * torch.randn(PV, H, generator=gen, dtype=torch.float64)).to(torch.float32).to(torch.float64)
    text_enc = (0.5 * torch.randn(ENC, H, generator=gen, dtype=torch.float64)).to(torch.float32).to(torch.float64)
    t = {"decoder.embed_prompts": embed_prompts, "decoder.text_encoding": text_enc}
    t["decoder.positional_embed"] = dec.embed_positions.weights                       # parler_tts_gguf_encoder.py:116-117
    for name, param in dec.named_parameters():                                      # :118-120
        t[f"decoder.{name}"] = param
    for name, param in model.lm_heads.named_parameters():                           # :121-128
        t[f"decoder.lm_heads.{name}.head"] = param
    tensors = {k: npy(v).astype(np.float32) for k, v in t.items()}

    rng = np.random.default_rng(10)
    prompt = rng.integers(3, PV, 5)
    steps = 6
    audio = rng.integers(0, 64, (steps, NCB))       # ids fed at every audio step (teacher forcing; no delay pattern here: the graph only)

    def run(m):
        """prompt embeddings in front of the summed codebook embeddings, positions 0 .. n - 1 over the whole sequence (what Parler-TTS'
        forward does with prompt_hidden_states).  MusicgenDecoder.forward cannot express that (with inputs_embeds it gives every token
        position 0), so the embedding sum and the position add are done here and everything after them — the decoder layers, the final
        LayerNorm, the heads — is upstream's code."""
        d = m.model.decoder
        with torch.no_grad():
            pe = embed_prompts[torch.tensor(prompt)]
            ae = sum(d.embed_tokens[c](torch.tensor(audio[:, c])) for c in range(NCB))
            x = torch.cat([pe, ae], 0)[None]
            T = x.shape[1]
            x = x + d.embed_positions.weights[:T].to(x.dtype)[None]
            mask = torch.full((T, T), float("-inf"), dtype=x.dtype).triu(1)[None, None]
            for layer in d.layers:
                x = layer(x, attention_mask=mask, encoder_hidden_states=text_enc[None])
                x = x[0] if isinstance(x, tuple) else x
            hid = d.layer_norm(x)[0]
            logits = torch.stack([head(hid) for head in m.lm_heads], 0)            # [NCB][T][V]
        return npy(hid), npy(logits)

    def run_api(m):
        """audio ids only, through MusicgenForCausalLM's own forward (embedding sum and positions upstream's too)"""
        with torch.no_grad():
            ids = torch.tensor(audio.T.reshape(1 * NCB, -1))       # (batch * codebooks, T)
            out = m(input_ids=ids, encoder_hidden_states=text_enc[None])
        return npy(out.logits.reshape(1, NCB, audio.shape[0], V)[0])               # [NCB][T][V]

    hid, logits = run(model)
    # distance of the erf-GELU variant (what the Parler-TTS config asks for) from the tanh-GELU the reference evaluates
    m2 = build("gelu")
    m2.load_state_dict(model.state_dict())
    _, logits_erf = run(m2)
    print(f"parler twin: |tanh-GELU - erf-GELU| logits {np.abs(logits - logits_erf).max():.2e} (max |logit| {np.abs(logits).max():.2f})")
    logits_api = run_api(model)
    save("upstream_parler.npz", prompt=prompt.astype(np.uint32), audio=audio.astype(np.uint32), hidden=hid, logits=logits, logits_erf_gelu=logits_erf,
         logits_audio_only_api=logits_api,
         cfg=np.array([H, L, HEADS, F, V, NCB, ENC, PV]), **{"t:" + k: v for k, v in tensors.items()})


# ---------------------------------------------------------------------------------------------------------------------------------
def make_dia():
    """transformers DiaForConditionalGeneration (encoder, decoder, logits_dense) at tiny dims (2 + 2 layers, 2 query heads on 1 k/v group in the decoder), float64.
    Both encoder streams of the reference's classifier-free-guidance batch (the text, all zeros) with NO padding (sentence length =
    max_context_length): there the reference's block mask (model.cpp:712-721) and upstream's padding mask are both "everything sees
    everything".  The decoder is run teacher-forced over T steps (causal mask = the incremental cache path), both streams with the same ids."""
    from transformers import DiaConfig, DiaDecoderConfig, DiaEncoderConfig, DiaForConditionalGeneration

    torch.manual_seed(1007)
    # head size 64 keeps the fixture at ~1.5 MB; the graph is the same at 128 (tiny_dia.npz and the GPU tests run the device's 128)
    EH, EL, ENH, EF, DH, DL, DNH, DKV, DF, HD, NO, AV, S, G = 64, 2, 2, 128, 128, 2, 2, 1, 192, 64, 9, 48, 24, 48
    V = AV + 4
    enc = DiaEncoderConfig(max_position_embeddings=S, num_hidden_layers=EL, hidden_size=EH, num_attention_heads=ENH, num_key_value_heads=ENH, head_dim=HD,
                           intermediate_size=EF, norm_eps=1e-5, vocab_size=256, hidden_act="silu")
    dec = DiaDecoderConfig(max_position_embeddings=G, num_hidden_layers=DL, hidden_size=DH, intermediate_size=DF, num_attention_heads=DNH,
                           num_key_value_heads=DKV, head_dim=HD, cross_num_attention_heads=DNH, cross_head_dim=HD, cross_num_key_value_heads=DNH,
                           cross_hidden_size=EH, norm_eps=1e-5, vocab_size=V, hidden_act="silu", num_channels=NO,
                           pad_token_id=AV + 1, eos_token_id=AV, bos_token_id=AV + 2)
    cfg = DiaConfig(encoder_config=enc, decoder_config=dec, pad_token_id=AV + 1, eos_token_id=AV, bos_token_id=AV + 2,
                    delay_pattern=[0, 8, 9, 10, 11, 12, 13, 14, 15])
    model = DiaForConditionalGeneration(cfg).double().eval()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "norm" in n:                                  # away from 1: their placement matters
                p.copy_(1.0 + 0.1 * torch.randn_like(p))
            elif "embed" in n:
                p.copy_(0.5 * torch.randn_like(p))
            elif "q_proj" in n or "k_proj" in n:             # attention logits are not scaled by 1/sqrt(d) in Dia: keep the softmax soft
                p.copy_(0.35 / math.sqrt(p.shape[1]) * torch.randn_like(p))
            else:
                p.copy_(torch.randn_like(p) / math.sqrt(p.shape[1]))
        for p in model.parameters():                         # the converter exports fp32: upstream runs on the fp32-rounded weights, in float64
            p.copy_(p.to(torch.float32).to(torch.float64))
    sd = {k: npy(v).astype(np.float32) for k, v in model.state_dict().items()}
    t = {}
    # dia_gguf_encoder.py:108-129 (encoder), :74-106 (decoder).  The converter reads the `dia` package's DenseGeneral tensors [in, heads, head_dim]
    # and writes reshape(in, -1).T = [heads * head_dim][in]: exactly a torch Linear weight, which is what the transformers port stores.
    t["dia.encoder.embedding"] = sd["model.encoder.embedding.weight"]
    t["dia.encoder.norm"] = sd["model.encoder.norm.weight"]
    for l in range(EL):
        a, b = f"model.encoder.layers.{l}.", f"dia.encoder.layers.{l}."
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            t[b + n] = sd[a + f"self_attention.{n}.weight"]
        t[b + "pre_sa_norm"], t[b + "post_sa_norm"] = sd[a + "pre_sa_norm.weight"], sd[a + "post_sa_norm.weight"]
        gu = sd[a + "mlp.gate_up_proj.weight"]              # wi_fused[:, 0] = gate, [:, 1] = up (:116-121); the port chunks the output the same way
        t[b + "gate"], t[b + "up"], t[b + "wo"] = gu[:EF], gu[EF:], sd[a + "mlp.down_proj.weight"]
    emb = sd["model.decoder.embeddings.embed.weight"]       # one table with per-channel offsets = the package's nine embeddings stacked
    for i in range(NO):
        t[f"dia.decoder.embeddings.{i}"] = emb[i * V:(i + 1) * V]
        t[f"dia.decoder.heads.{i}"] = sd["logits_dense.weight"][i * V:(i + 1) * V]     # logits_dense[:, i].T (:83-87)
    t["dia.decoder.norm"] = sd["model.decoder.norm.weight"]
    for l in range(DL):
        a, b = f"model.decoder.layers.{l}.", f"dia.decoder.layers.{l}."
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            t[b + "self_" + n] = sd[a + f"self_attention.{n}.weight"]
            t[b + "cross_" + n] = sd[a + f"cross_attention.{n}.weight"]
        for n in ("pre_sa_norm", "pre_ca_norm", "pre_mlp_norm"):
            t[b + n] = sd[a + n + ".weight"]
        gu = sd[a + "mlp.gate_up_proj.weight"]
        t[b + "gate"], t[b + "up"], t[b + "wo"] = gu[:DF], gu[DF:], sd[a + "mlp.down_proj.weight"]

    rng = np.random.default_rng(11)
    text = rng.integers(1, 256, S)                           # S byte tokens, none of them the pad id 0
    T = 20
    ids = rng.integers(0, AV, (T, NO))
    ids[0, :] = AV + 2                                       # step 0: bos on every channel (model.cpp:724-726 feeds both streams the same ids)
    with torch.no_grad():
        enc_in = torch.tensor(np.stack([text, np.zeros(S, dtype=np.int64)]))
        enc_out = model.model.encoder(input_ids=enc_in, attention_mask=torch.ones(2, S, dtype=torch.long)).last_hidden_state
        dec_in = torch.tensor(np.stack([ids, ids]))
        dec_out = model.model.decoder(input_ids=dec_in, encoder_hidden_states=enc_out,
                                      encoder_attention_mask=torch.ones(2, S, dtype=torch.long)).last_hidden_state
        raw = model.logits_dense(dec_out).view(2, T, NO, V)
    save("upstream_dia.npz", text=text.astype(np.uint32), ids=ids.astype(np.uint32), enc_out=npy(enc_out), raw_logits=npy(raw).transpose(1, 0, 2, 3),
         cfg=np.array([EH, EL, ENH, EF, DH, DL, DNH, DNH // DKV, DF, HD, NO, AV, S, G]), **{"t:" + k: v for k, v in t.items()})
    print("dia: encoder", tuple(enc_out.shape), "raw logits", tuple(raw.shape), "max |logit|", float(raw.abs().max()))


# ---------------------------------------------------------------------------------------------------------------------------------
def make_unigram():
    """The prompt tokenizer.  parler_tts_gguf_encoder.py:187-202 converts FROM a Hugging Face `tokenizers` Unigram model (the T5 fast
    tokenizer of the Parler-TTS repositories): pieces with U+2581 replaced by a space, scores from the tokenizer's JSON, unk / eos ids.  The twin
    here is a seeded 300-piece Unigram model behind T5's pre-tokenizer (WhitespaceSplit + Metaspace, prefix space always); its ids for a set of
    sentences are the fixture, the vocabulary is exported by the converter's rules."""
    from tokenizers import Tokenizer, models, pre_tokenizers

    rng = np.random.default_rng(5)
    letters = "abcdefghijklmnopqrstuvwxyz"
    sp = "\u2581"
    vocab = [("<pad>", 0.0), ("</s>", 0.0), ("<unk>", 0.0), (sp, -2.5)]
    vocab += [(ch, float(-4 - 3 * rng.random())) for ch in letters] + [(sp + ch, float(-3.5 - 3 * rng.random())) for ch in letters]
    seen = {p for p, _ in vocab}
    while len(vocab) < 300:
        w = "".join(rng.choice(list(letters), int(rng.integers(2, 6))))
        if rng.random() < 0.5:
            w = sp + w
        if w not in seen:
            seen.add(w)
            vocab.append((w, float(-5 - 6 * rng.random())))
    tok = Tokenizer(models.Unigram(vocab, unk_id=2, byte_fallback=False))
    tok.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.WhitespaceSplit(), pre_tokenizers.Metaspace(replacement=sp, prepend_scheme="always")])
    pieces = [p for p, _ in vocab]
    texts = []
    for i in range(40):                                  # words biased towards the multi-letter pieces so that the segmentation has choices
        words = []
        for _ in range(int(rng.integers(1, 9))):
            w = ""
            while len(w) < int(rng.integers(1, 10)):
                w += (pieces[int(rng.integers(4, 300))].replace(sp, "") if rng.random() < 0.6 else letters[int(rng.integers(0, 26))])
            words.append(w)
        texts.append((" " if i % 5 else "  ").join(words))    # every fifth sentence with doubled spaces (the reference collapses them, tokenizer.h:22)
    texts += ["a?b", "zz!!zz", "caf\u00e9 ab"]            # characters outside the vocabulary: <unk>, consecutive ones fused
    ids = [tok.encode(t).ids for t in texts]
    flat = np.array([x for row in ids for x in row], dtype=np.uint32)
    save("upstream_unigram.npz", pieces=np.array([p.replace(sp, " ") for p in pieces]),      # :193-194
         scores=np.array([sc for _, sc in vocab], dtype=np.float32), unk=np.array(2), eos=np.array(1),   # :195-200
         texts=np.array(texts), ids_flat=flat, ids_len=np.array([len(r) for r in ids], dtype=np.int64))
    print("unigram:", len(texts), "sentences,", flat.size, "ids; e.g.", texts[0], "->", tok.encode(texts[0]).tokens)


# ---------------------------------------------------------------------------------------------------------------------------------
def make_bpe():
    """Orpheus' prompt tokenizer.  orpheus_gguf_encoder.py:231-242 copies `model.vocab` (in id order) and `model.merges` out of a Hugging Face
    byte-level BPE tokenizer.json.  The twin: a 400-token byte-level BPE trained by `tokenizers` on seeded pseudo-words; its ids for sentences
    of letters and single spaces are the fixture.  (The reference cuts a prompt at spaces only, tokenizer.cpp:209-296, where Llama-3's
    pre-tokenizer regex also cuts at digits and punctuation: sentences with those are outside what the two share.)"""
    import json

    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers

    rng = np.random.default_rng(9)
    syll = ["ka", "to", "mi", "re", "sun", "lo", "ve", "da", "ri", "no", "ta", "shi", "be", "qu", "zo", "el", "an", "ing", "er", "st"]

    def word():
        return "".join(syll[int(rng.integers(0, len(syll)))] for _ in range(int(rng.integers(1, 5))))

    corpus = [" ".join(word() for _ in range(int(rng.integers(3, 12)))) for _ in range(400)]
    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=True)
    tok.decoder = decoders.ByteLevel()
    tok.train_from_iterator(corpus, trainers.BpeTrainer(vocab_size=400, initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), show_progress=False))
    model = json.loads(tok.to_str())["model"]
    tokens = list(model["vocab"].keys())                                            # :238
    assert [model["vocab"][t] for t in tokens] == list(range(len(tokens)))
    merges = [" ".join(pair) if not isinstance(pair, str) else pair for pair in model["merges"]]   # :240
    texts = [" ".join(word() for _ in range(int(rng.integers(1, 10)))) for _ in range(40)]
    texts += ["k", "zzzz qqq"]                       # a single byte, letters the merges never saw together
    ids = [tok.encode(t).ids for t in texts]
    flat = np.array([x for row in ids for x in row], dtype=np.uint32)
    # a doubled space: upstream keeps the first space as a token of its own ("Ġ"), the reference's split drops empty pieces — a divergence of the
    # reference, stored with upstream's ids so that the test can state it
    dbl = "sunlo  vedari"
    save("upstream_bpe.npz", tokens=np.array(tokens), merges=np.array(merges), texts=np.array(texts), ids_flat=flat,
         ids_len=np.array([len(r) for r in ids], dtype=np.int64), doubled_space_text=np.array(dbl),
         doubled_space_ids=np.array(tok.encode(dbl).ids, dtype=np.uint32), space_id=np.array(tok.token_to_id("\u0120")))
    print("bpe:", len(tokens), "tokens,", len(merges), "merges; e.g.", texts[0], "->", tok.encode(texts[0]).tokens)


# ---------------------------------------------------------------------------------------------------------------------------------
def make_delay():
    """The delay pattern of the nine codebooks and its undoing — index logic, so the comparison is exact.  Upstream:
    MusicgenForCausalLM.build_delay_pattern_mask / apply_delay_pattern_mask (Parler-TTS inherits both) and generate()'s final
    `output_ids[output_ids != pad]` reshape.  Reference: the ids fed per step (parler/model.cpp:778-785) and adjust_output_tokens (:734-760).
    A seeded stream of "sampled" tokens stands for the model."""
    from transformers import MusicgenDecoderConfig, MusicgenForCausalLM

    K, AV = 9, 64
    bos = AV + 2
    cfg = MusicgenDecoderConfig(vocab_size=AV + 4, hidden_size=16, num_hidden_layers=1, ffn_dim=16, num_attention_heads=1, num_codebooks=K,
                                max_position_embeddings=64, audio_channels=1, pad_token_id=bos, bos_token_id=bos, tie_word_embeddings=False)
    m = MusicgenForCausalLM(cfg)
    steps = 30
    rng = np.random.default_rng(13)
    samples = rng.integers(0, AV, (steps, K))                       # what the sampler returned after step s (1-based s = row s - 1)
    start = torch.full((K, 1), bos, dtype=torch.long)
    _, mask = m.build_delay_pattern_mask(start, pad_token_id=bos, max_length=steps + 1)
    seq = torch.cat([start, torch.tensor(samples.T)], 1)            # position 0 = the bos column, position s = the sample of step s
    fed = m.apply_delay_pattern_mask(seq, mask)                     # what generate() feeds back, position by position
    frames = fed[fed != bos].reshape(K, -1)                         # generate()'s un-delay: drop every pad / bos
    save("upstream_delay.npz", samples=samples.astype(np.uint32), fed=fed.numpy().T.astype(np.uint32), mask=mask.numpy().T.astype(np.int64),
         frames=frames.numpy().T.astype(np.uint32), cfg=np.array([K, AV, bos, steps]))
    print("delay:", tuple(fed.shape), "->", tuple(frames.shape))


# ---------------------------------------------------------------------------------------------------------------------------------
def make_albert():
    """Kokoro's text model.  kokoro_gguf_encoder.py:274-287 walks `model.bert` — a transformers AlbertModel — with the names of ALBERT_PARTS
    (:14-37): one shared layer applied num_hidden_layers times.  The twin has the dims of tts_cpp_amd.synth.kokoro_tiny (so that the fixture's
    tensors can replace the synthetic ones of that model) and transformers' own softmax scale 1/sqrt(head size) — the reference hard-codes
    0.125 = 1/sqrt(64), Kokoro-82M's head size (model.h:196), which the oracle takes as a parameter."""
    from transformers import AlbertConfig, AlbertModel

    torch.manual_seed(1011)
    V, E, H, NH, F, REC, CTX = 32, 16, 64, 4, 128, 2, 32
    cfg = AlbertConfig(vocab_size=V, embedding_size=E, hidden_size=H, num_hidden_layers=REC, num_hidden_groups=1, num_attention_heads=NH,
                       intermediate_size=F, inner_group_num=1, hidden_act="gelu_new", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                       max_position_embeddings=CTX, type_vocab_size=2, layer_norm_eps=1e-12)
    model = AlbertModel(cfg, add_pooling_layer=False).double().eval()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "LayerNorm.weight" in n or "layer_norm.weight" in n:
                p.copy_(1.0 + 0.1 * torch.randn_like(p))
            elif n.endswith("bias"):
                p.copy_(0.1 * torch.randn_like(p))
            elif "embeddings" in n:
                p.copy_(0.7 * torch.randn_like(p))
            else:
                p.copy_(torch.randn_like(p) / math.sqrt(p.shape[1]))
        for p in model.parameters():
            p.copy_(p.to(torch.float32).to(torch.float64))
    parts = {"embeddings.word_embeddings.weight": "token_embd", "embeddings.position_embeddings.weight": "position_embd",          # ALBERT_PARTS :14-37
             "embeddings.LayerNorm.weight": "norm", "embeddings.LayerNorm.bias": "norm_bias", "encoder.embedding_hidden_mapping_in.weight": "embd",
             "encoder.embedding_hidden_mapping_in.bias": "embd_bias", "full_layer_layer_norm.weight": "attn_norm", "full_layer_layer_norm.bias": "attn_norm_bias",
             "attention.query.weight": "q", "attention.query.bias": "q_bias", "attention.key.weight": "k", "attention.key.bias": "k_bias",
             "attention.value.weight": "v", "attention.value.bias": "v_bias", "attention.dense.weight": "o", "attention.dense.bias": "o_bias",
             "attention.LayerNorm.weight": "ffn_norm", "attention.LayerNorm.bias": "ffn_norm_bias", "ffn.weight": "ffn", "ffn.bias": "ffn_bias",
             "ffn_output.weight": "ffn_out", "ffn_output.bias": "ffn_out_bias"}
    layer = "encoder.albert_layer_groups.0.albert_layers.0."
    t = {}
    for name, param in model.named_parameters():                                                  # :279-287
        if name in parts:
            t["kokoro.albert." + parts[name]] = npy(param).astype(np.float32)
        elif layer in name and name[len(layer):] in parts:
            t["kokoro.albert.layer.0." + parts[name[len(layer):]]] = npy(param).astype(np.float32)
        elif name == "embeddings.token_type_embeddings.weight":
            t["kokoro.albert.token_type_embd"] = npy(param).astype(np.float32)[0, :]
    assert len(t) == 23, sorted(t)
    rng = np.random.default_rng(17)
    ids = rng.integers(0, V, 19)
    with torch.no_grad():
        out = model(input_ids=torch.tensor(ids[None]), attention_mask=torch.ones(1, ids.size, dtype=torch.long),
                    token_type_ids=torch.zeros(1, ids.size, dtype=torch.long)).last_hidden_state[0]
    save("upstream_albert.npz", ids=ids.astype(np.uint32), out=npy(out), cfg=np.array([V, E, H, NH, F, REC, CTX]), **{"t:" + k: v for k, v in t.items()})
    print("albert:", tuple(out.shape), "max |out|", float(out.abs().max()))


# ---------------------------------------------------------------------------------------------------------------------------------
def make_kokoro_stages():
    """Stage fixtures for the parts of the Kokoro graph beyond ALBERT and for SNAC's depthwise block, each computed by ONE PyTorch module or
    functional in float64 (nothing of this repository in between) and stored with the tensors under the names / split rules of the converter:
      lstm      torch.nn.LSTM(bidirectional): weight_ih / weight_hh / bias_ih / bias_hh of both directions, split into the four gate blocks
                i, f, g, o and interleaved ih, hh as kokoro_gguf_encoder.py:289-309 (prepare_lstm_tensor) does       <-> kokoro/model.cpp:35-86
      ada       AdainResBlk1d with upsampling, from its constituents: F.instance_norm + nn.Linear style affine (1 + gamma) x + beta, leaky relu 0.2,
                F.conv_transpose1d(groups = C, k 3, stride 2, padding 1, output_padding 1) — the fork's ggml_conv_transpose_1d(.., 2, 1, 1, 1, C)
                call —, F.conv1d k 3, shortcut = nearest x2 + conv1x1, (y + s) / sqrt 2                                  <-> kokoro/model.cpp:88-134
      stft      torch.stft(center, reflect, onesided) magnitude / angle and torch.istft of (exp, sin)-shaped inputs       <-> util.cpp:111-133, 203-217
      snac_dw   F.conv1d(groups = C, k 7, dilation d, padding 3 d): SNAC's depthwise residual conv                     <-> decoder/snac_model.cpp:86-110
    The istft output is rescaled from torch's window envelope (its F frames) to the reference's compute_window_squared_sum (out_len / hop +
    n_fft / 2 / hop frames), the one documented difference between the two definitions."""
    import torch.nn.functional as Fn
    torch.manual_seed(1013)
    out = {}
    # ---- lstm
    L, IN, HID = 11, 12, 5
    lstm = torch.nn.LSTM(IN, HID, batch_first=True, bidirectional=True).double().eval()
    with torch.no_grad():
        for p in lstm.parameters():
            p.copy_((0.5 * torch.randn_like(p)).to(torch.float32).to(torch.float64))
    x = torch.randn(L, IN, dtype=torch.float64).to(torch.float32).to(torch.float64)
    with torch.no_grad():
        y, _ = lstm(x[None])
    out["lstm_x"], out["lstm_y"], out["lstm_dims"] = npy(x).astype(np.float32), npy(y[0]), np.array([L, IN, HID])
    for name, param in lstm.named_parameters():                      # prepare_lstm_tensor, layer 0
        data = npy(param).astype(np.float32)
        blocks = [data[i * (data.shape[0] // 4):(i + 1) * (data.shape[0] // 4)] for i in range(4)]
        part = ("reverse_" if "reverse" in name else "") + ("weights" if "weight" in name else "biases")
        for i, d in enumerate(blocks):
            out[f"t:stage.lstm.0.{part}.{i * 2 if '_ih_' in name else i * 2 + 1}"] = d
    # ---- ada block (upsampling, with a 1x1 shortcut)
    C, CO, S, LA = 6, 10, 7, 9
    style = torch.randn(S, dtype=torch.float64).to(torch.float32).to(torch.float64)
    xa = torch.randn(1, C, LA, dtype=torch.float64).to(torch.float32).to(torch.float64)
    P = {}
    def mk(name, *shape, scale=0.4):
        P[name] = (scale * torch.randn(*shape, dtype=torch.float64)).to(torch.float32).to(torch.float64)
        return P[name]
    for k, ch in (("norm1", C), ("norm2", CO)):
        mk(f"{k}_gamma_weight", ch, S); mk(f"{k}_gamma_bias", ch); mk(f"{k}_beta_weight", ch, S); mk(f"{k}_beta_bias", ch)
    mk("pool_weight", C, 1, 3); mk("pool_bias", C)
    mk("conv1_weight", CO, C, 3); mk("conv1_bias", CO); mk("conv2_weight", CO, CO, 3); mk("conv2_bias", CO); mk("conv1x1_weight", CO, C, 1)
    def adain(v, k):
        g = Fn.linear(style, P[f"{k}_gamma_weight"], P[f"{k}_gamma_bias"]); b = Fn.linear(style, P[f"{k}_beta_weight"], P[f"{k}_beta_bias"])
        return (1 + g)[None, :, None] * Fn.instance_norm(v, eps=1e-5) + b[None, :, None]
    ya = Fn.leaky_relu(adain(xa, "norm1"), 0.2)
    ya = Fn.conv_transpose1d(ya, P["pool_weight"], P["pool_bias"], stride=2, padding=1, output_padding=1, groups=C)
    ya = Fn.conv1d(ya, P["conv1_weight"], P["conv1_bias"], padding=1)
    ya = Fn.conv1d(Fn.leaky_relu(adain(ya, "norm2"), 0.2), P["conv2_weight"], P["conv2_bias"], padding=1)
    sa = Fn.conv1d(Fn.interpolate(xa, scale_factor=2, mode="nearest"), P["conv1x1_weight"])
    out["ada_x"], out["ada_style"], out["ada_y"], out["ada_dims"] = npy(xa[0]).astype(np.float32), npy(style).astype(np.float32), npy((ya + sa)[0] / math.sqrt(2.0)), np.array([C, CO, S, LA])
    for k, v in P.items():
        out["t:stage.ada." + k] = npy(v).astype(np.float32)
    # ---- stft / istft
    N, hop, LS = 20, 5, 120
    win = torch.sin(math.pi * torch.arange(N, dtype=torch.float64) / N) ** 2
    sig = torch.randn(LS, dtype=torch.float64).to(torch.float32).to(torch.float64)
    spec = torch.stft(sig, N, hop, N, window=win, center=True, pad_mode="reflect", return_complex=True)
    out["stft_x"], out["stft_win"], out["stft_mag"], out["stft_ph"], out["stft_dims"] = npy(sig).astype(np.float32), npy(win).astype(np.float32), npy(spec.abs()), npy(spec.angle()), np.array([N, hop, LS])
    mag = torch.exp(0.3 * torch.randn(N // 2 + 1, spec.shape[1], dtype=torch.float64)).to(torch.float32).to(torch.float64)
    ph = torch.sin(torch.randn(N // 2 + 1, spec.shape[1], dtype=torch.float64)).to(torch.float32).to(torch.float64)
    ph[0], ph[-1] = 0.0, 0.0                                                   # a real signal has real DC / Nyquist bins
    out_len = LS
    yi = torch.istft(torch.polar(mag, ph), N, hop, N, window=win, center=True, length=out_len)
    w2 = npy(win) ** 2
    env_t, env_r = np.zeros(out_len + 2 * N), np.zeros(out_len + 2 * N)
    for f in range(spec.shape[1]):
        env_t[f * hop:f * hop + N] += w2
    for f in range(out_len // hop + (N // 2) // hop):                          # compute_window_squared_sum (util.cpp:203-217)
        env_r[f * hop:f * hop + N] += w2
    half = N // 2
    out["istft_mag"], out["istft_ph"] = npy(mag).astype(np.float32), npy(ph).astype(np.float32)
    out["istft_y"] = npy(yi) * env_t[half:half + out_len] / env_r[half:half + out_len]
    # ---- SNAC depthwise conv
    CD, LD, DIL = 8, 40, 3
    wd = (0.4 * torch.randn(CD, 1, 7, dtype=torch.float64)).to(torch.float32).to(torch.float64)
    bd = (0.4 * torch.randn(CD, dtype=torch.float64)).to(torch.float32).to(torch.float64)
    xd = torch.randn(1, CD, LD, dtype=torch.float64).to(torch.float32).to(torch.float64)
    out["dw_x"], out["dw_w"], out["dw_b"], out["dw_y"], out["dw_dims"] = npy(xd[0]).astype(np.float32), npy(wd[:, 0]).astype(np.float32), npy(bd).astype(np.float32), npy(Fn.conv1d(xd, wd, bd, padding=3 * DIL, dilation=DIL, groups=CD)[0]), np.array([CD, LD, DIL])
    save("upstream_kokoro_stages.npz", **out)
    print("kokoro stages:", {k: v.shape for k, v in out.items() if not k.startswith("t:")})


if __name__ == "__main__":
    which = sys.argv[1:] or ["orpheus", "t5", "dac", "dac_b3", "parler", "dia", "unigram", "bpe", "delay", "albert", "kokoro_stages"]
    for w in which:
        {"orpheus": make_orpheus, "t5": make_t5, "dac": make_dac, "parler": make_parler, "dia": make_dia, "unigram": make_unigram, "bpe": make_bpe, "delay": make_delay, "albert": make_albert, "kokoro_stages": make_kokoro_stages,
         "dac_b3": lambda: make_dac("upstream_dac_b3.npz", hidden=192, strides=(2, 2), seed=1005, frames=300)}[w]()

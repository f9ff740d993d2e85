/*
 * tts_oracle.h — CPU restatement ("oracle") of the TTS.cpp Parler-TTS + DAC hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (tts.cpp_amd/) may include,
 * link or call this.  Allowed users: tests/, __graft_entry__.smoke(), bench.py's
 * cpu_baseline leg.
 *
 * PARITY STATUS.  The reference's kernels live in the un-vendored ggml fork (github.com/mmwillet/ggml, branch
 * support-for-tts, commit pin unknown; /root/reference/ggml is empty) and the reference ships no golden vectors
 * (SURVEY.md §0.1, §0.2), so nothing here can be checked against the reference's own floats.  What IS pinned, and to what:
 *   - the sampler restatement against the real reference src/sampler.cpp, compiled from where it lies into
 *     oracle/_ref/ (oracle/Makefile, tests/test_sampler.py): bit-exact;
 *   - the arithmetic of every graph whose upstream model is importable in the build container, against that model in
 *     float64 — the implementations the reference's converters convert FROM, weights exported under the converters'
 *     naming and layout rules (tests/golden/make_upstream_golden.py -> tests/golden/upstream_*.npz,
 *     tests/test_upstream_golden.py): transformers' LlamaForCausalLM (Orpheus), T5EncoderModel, DacModel (weight norm
 *     folded by the reference's own tensor_util.py), MusicgenForCausalLM (Parler's decoder; delay pattern and un-delay
 *     exact), DiaForConditionalGeneration, AlbertModel (Kokoro's text model), and `tokenizers` Unigram / BPE for the
 *     prompt tokenizers; every place where the REFERENCE departs from those upstreams is asserted as such (T5 buckets,
 *     tanh-GELU, Dia's rope in cross-attention, Kokoro's fixed softmax scale, doubled spaces in BPE prompts);
 *   - every tensor primitive against PyTorch CPU primitives (tests/golden/tiny_*.npz, tests/golden/make_golden.py).
 * "parity unpinned" remains true of: ggml's kernel-level rounding (fp16 rounding of activations in front of F16
 * weights, the fp16-indexed GELU table — stated as upstream knowledge), the fork's own ops behind Kokoro's vocoder
 * (kokoro_oracle.c's header), SNAC, and Kokoro beyond its ALBERT stage.
 *
 * All file:line citations are relative to /root/reference.
 */
#ifndef TTS_ORACLE_H
#define TTS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ggml type ids used by the reference's GGUF files (examples/quantize/README.md:48-55) */
enum { ORC_F32 = 0, ORC_F16 = 1, ORC_Q4_0 = 2, ORC_Q5_0 = 6, ORC_Q8_0 = 8 };

#define ORC_MAX_LAYERS 64
#define ORC_MAX_HEADS  16

/* ---- scalar helpers -------------------------------------------------------------- */
float    orc_h2f(uint16_t h);
uint16_t orc_f2h(float f); /* round-to-nearest-even, like GGML_FP32_TO_FP16 */

/* ---- ggml block formats (upstream ggml-quants.c knowledge) ---------- */
size_t orc_row_bytes(int type, int64_t n);                          /* bytes of n elements */
int    orc_dequantize(int type, const void *src, float *dst, int64_t n);
int    orc_quantize(int type, const float *src, void *dst, int64_t n); /* *_ref quantizers */

/* y[r][n] = sum_k W[n][k] * x[r][k]      (ggml_mul_mat(W, x): W ne=[K,N], x ne=[K,R])
 * act_mode: 0 = keep activations fp32 (weights dequantised);
 *           1 = ggml CPU semantics: activations converted to the weight type's
 *               vec_dot_type first (F16 -> fp16 rounding, Q* -> Q8_0 blocks + integer dot). */
void orc_mul_mat(int type, const void *W, int64_t K, int64_t N, const float *x, int64_t R,
                 float *y, int act_mode);

/* ---- Parler decoder (src/models/parler/model.cpp:387-457,520-643) ------------------ */
typedef struct {
    int type;           /* ORC_* */
    const void *data;   /* raw GGUF bytes, rows of ne0 */
} orc_w;

typedef struct {
    orc_w q, k, v, o;               /* self_attn.{q,k,v,out}_proj.weight  [H x H]   */
    const float *sa_ln_w, *sa_ln_b; /* self_attn_layer_norm                          */
    orc_w cq, ck, cv, co;           /* encoder_attn.*                                 */
    const float *ca_ln_w, *ca_ln_b; /* encoder_attn_layer_norm                        */
    orc_w fc1, fc2;                 /* [F x H], [H x F]                               */
    const float *f_ln_w, *f_ln_b;   /* final_layer_norm                               */
} orc_parler_layer;

typedef struct {
    int32_t H, L, n_heads, F, V, n_out, n_ctx, E, use_cross;
    int32_t act_mode;   /* see orc_mul_mat */
    int32_t gelu_mode;  /* 0 = fp32 tanh-GELU; 1 = ggml's fp16-table GELU (upstream ggml knowledge) */
    orc_w embed_prompts;            /* [prompt_vocab][H]                               */
    orc_w embed_tokens[ORC_MAX_HEADS]; /* [V+1][H] each (model.cpp:397-403)            */
    orc_w lm_heads[ORC_MAX_HEADS];  /* [V][H] each (model.cpp:441-457)                 */
    const float *pos_embed;         /* [>=n_ctx][H]  F32 (quantize allow-list keeps it F32) */
    const float *text_encoding;     /* [E][H]        F32                               */
    const float *ln_w, *ln_b;       /* decoder.layer_norm                              */
    orc_parler_layer layers[ORC_MAX_LAYERS];
} orc_parler_model;

typedef struct orc_parler_state orc_parler_state;

orc_parler_state *orc_parler_state_new(const orc_parler_model *m);
void              orc_parler_state_free(orc_parler_state *s);
/* model.cpp:110-173  prep_cross_key_values */
void orc_parler_prep_cross(const orc_parler_model *m, orc_parler_state *s);
/* model.cpp:648-693 decode(): S tokens at positions pos0..pos0+S-1.
 *   audio=0: `tokens` are S text-prompt ids (embed_prompts rows)
 *   audio=1: S must be 1 and `tokens` holds n_out codebook ids (model.cpp:394-403,783-785)
 * logits_out (may be NULL): ggml layout ne=[V, S, n_out]  ->  [n_out][S][V]  (model.cpp:455-456)
 * hidden_out (may be NULL): final-layer-normed hidden state [S][H] (for per-stage parity). */
void orc_parler_decode(const orc_parler_model *m, orc_parler_state *s, int audio,
                       const uint32_t *tokens, int S, uint32_t pos0, float *logits_out,
                       float *hidden_out);
/* read back one layer's cache rows for tests: K[t][H], V[t][H] (both position-major here) */
void orc_parler_get_kv(const orc_parler_state *s, int layer, int n_pos, float *k_out, float *v_out);

/* ---- T5 voice-prompt encoder (src/models/parler/t5/model.cpp:179-320) ----------------- */
typedef struct {
    orc_w q, k, v, o;        /* attn_{q,k,v,o}   [H x H]                    (:244-246,264) */
    const float *attn_norm;  /* rms norm weight  [H]                        (:179-185)     */
    orc_w wi_0, wi_1, wo;    /* ffn_up, ffn_gate [F x H], ffn_down [H x F]  (:272-275)     */
    const float *mlp_norm;
} orc_t5_layer;
typedef struct {
    int32_t H, L, n_heads, F, n_buckets, out_size;
    int32_t act_mode, gelu_mode;
    orc_w embd;                 /* token_embd [vocab][H] (:229)                                  */
    const float *rel_bias;      /* relative attention bias [n_buckets][n_heads] (layer 0, shared) */
    const float *out_norm;      /* final rms norm                                                 */
    orc_w down_proj;            /* [out_size x H] or data == NULL (:285-291)                      */
    const float *down_proj_bias;
    orc_t5_layer layers[ORC_MAX_LAYERS];
} orc_t5_model;
/* relative position bucket of (query i, key ii) exactly as t5_runner::set_inputs fills it (:303-316),
 * integer division inside the log included */
uint32_t orc_t5_bucket(int i, int ii, int n_buckets_total);
/* t5_runner::run (:321-357): ids [n] -> out [n][out_size] */
void orc_t5_encode(const orc_t5_model *m, const uint32_t *ids, int n, float *out);

/* ---- generation loop helpers (model.cpp:715-792) ------------------------------------ */
/* next decoder input ids after `step` completed audio steps (model.cpp:778-785) */
void orc_parler_next_ids(int n_out, int step, const uint32_t *last_outputs, const uint8_t *eos_seen,
                         uint32_t bos, uint32_t eos, uint32_t *next_ids);
/* adjust_output_tokens (model.cpp:734-760), quirks included. returns number of tokens written */
size_t orc_parler_adjust_output_tokens(const uint32_t *tokens, size_t n_tokens, int n_out,
                                       uint32_t audio_vocab, uint32_t eos, uint32_t *filtered);

/* ---- sampler (src/sampler.cpp) ------------------------------------------------------- */
typedef struct {
    uint32_t n_output_heads, vocab_size, top_k;
    float temperature, top_p, repetition_penalty;
    int do_sample;
    int32_t last_token_ids[ORC_MAX_HEADS];
    uint32_t repetition_counts[ORC_MAX_HEADS];
    int rep_initialised;
} orc_sampler;
void orc_sampler_init(orc_sampler *s, uint32_t n_heads, uint32_t vocab);
void orc_sampler_reset(orc_sampler *s);                                   /* sampler.cpp:71-80  */
void orc_sampler_max(const orc_sampler *s, const float *logits, uint32_t *out); /* :185-204 */
/* full sample(): `uniforms` supplies one U[0,1) draw per head in head order (the reference draws
 * from an unseeded std::minstd_rand, sampler.cpp:47-48 — the draw is injected here so the
 * arithmetic around it can be compared). logits are mutated in place exactly as the reference does. */
void orc_sampler_sample(orc_sampler *s, float *logits, const float *uniforms, uint32_t *out);

/* ---- DAC decoder (src/decoder/dac_model.cpp:100-170, general_neural_audio_codec.cpp:133-172) */
typedef struct {
    const float *in_alpha, *in_w, *in_b;   /* res.initial: snake alpha [C], conv k7 W [C][C][7], b [C] */
    const float *out_alpha, *out_w, *out_b;/* res.final:   snake alpha [C], conv k1 W [C][C][1], b [C] */
} orc_dac_res;
typedef struct {
    int32_t stride, padding, cin, cout;
    const float *alpha;                     /* [cin]                                   */
    const float *w;                         /* ConvTranspose1d weight [cin][cout][2*stride] */
    const float *b;                         /* [cout]                                  */
    orc_dac_res res[3];
} orc_dac_block;
typedef struct {
    int32_t n_codebooks, codebook_dim, codebook_size, latent; /* 9, 8, 1024, 1024 */
    const float *codebook[ORC_MAX_HEADS];   /* [codebook_size][codebook_dim]           */
    const float *out_proj_w[ORC_MAX_HEADS]; /* [latent][codebook_dim][1]               */
    const float *out_proj_b[ORC_MAX_HEADS]; /* [latent]                                */
    int32_t c0;                             /* channels after initial conv (1536)      */
    const float *init_w, *init_b;           /* conv k7 [c0][latent][7], [c0]           */
    int32_t n_blocks;
    orc_dac_block blocks[8];
    const float *final_alpha, *final_w, *final_b; /* snake [c_last], conv k7 [1][c_last][7], [1] */
    /* 1: the conv kernels are F16 tensors (quantize --convert-dac-to-f16, quantize_impl.cpp:264-266): ggml_conv_1d goes
     * through an fp16 im2col and conv_transpose_1d_f16_f32 converts its source to fp16 (upstream ggml), so every conv
     * input is rounded to fp16 after snake; accumulation stays fp32.  Weight pointers then hold the exact F16 values. */
    int32_t f16_conv;
} orc_dac_model;

/* codes: frame-major [frames][n_codebooks] (dac_model.cpp:112); pcm_out: frames*prod(strides).
 * stage_out (optional): if stage>=0 the activation after that stage is copied there:
 *   stage 0 = quantizer sum [latent][T], 1 = after initial conv, 2..1+n_blocks = after block i. */
int64_t orc_dac_decode(const orc_dac_model *m, const uint32_t *codes, int frames, float *pcm_out,
                       int stage, float *stage_out);

/* ---- Orpheus decoder: Llama-3 blocks (src/models/orpheus/model.cpp:122-125,186-296) ----------------
 * rms norm eps 1e-5, q/k/v projections, NEOX-style RoPE with per-frequency factors (ggml_rope_ext(..., head_size,
 * mode 2, 0, 500000, 1, 0, 1, 0, 0) :189-191,248-251), grouped-query attention (kv head = q head / (n_heads/n_kv_heads):
 * the reference stores K/V repeat-interleaved 3x in its cache, :197-221), silu(gate x) * (up x) MLP, lm_head on the
 * LAST token only (:287-290). */
typedef struct {
    orc_w q, k, v, o, gate, up, down;
    const float *input_norm, *post_norm;
} orc_orpheus_layer;
typedef struct {
    int32_t H, L, n_heads, n_kv_heads, head_dim, F, V, n_ctx;
    int32_t act_mode;
    orc_w embd, head;               /* embed_tokens [V][H], lm_head [V][H] */
    const float *out_norm;
    const float *rope_freqs;        /* [head_dim/2] frequency factors ("orpheus.rope_frequencies") */
    orc_orpheus_layer layers[ORC_MAX_LAYERS];
} orc_orpheus_model;
typedef struct orc_orpheus_state orc_orpheus_state;
orc_orpheus_state *orc_orpheus_state_new(const orc_orpheus_model *m);
void               orc_orpheus_state_free(orc_orpheus_state *s);
/* orpheus_runner::decode (:298-325): n tokens at positions pos0..pos0+n-1; logits_out [V] of the last token;
 * hidden_out (may be NULL) [n][H] final-normed hidden states */
void orc_orpheus_decode(const orc_orpheus_model *m, orc_orpheus_state *s, const uint32_t *tokens, int n, uint32_t pos0,
                        float *logits_out, float *hidden_out);

/* ---- Dia (src/models/dia/model.cpp) -----------------------------------------------------------------------------
 * Encoder (:383-440): byte-token embedding, rms norm eps 1e-5 (:352-357), q/k/v to n_heads x head_dim (wider than the
 * encoder's hidden size), NEOX rope (ggml_rope(..., head_size, 2): base 10000) on q and k, softmax with scale 1.0 and
 * the block mask of set_inputs (:712-721: real tokens see real tokens, pad positions see pad positions), silu(gate)*up
 * MLP, final norm.  It always runs max_ctx positions and two streams: the text and an all-zero "unconditional" one.
 * Cross K/V (:505-541): K only for the first prompt_size positions (rope'd with their positions; the rest of the
 * cache keeps its cleared zeros), V for all max_ctx positions.
 * Decoder step (:543-659): sum of the n_out codebook embeddings, per layer self-attention (k/v on kv_heads groups,
 * rope on q and k, scale 1.0, no mask over the cached positions), cross-attention over ALL max_ctx encoder positions
 * (q rope'd with the decoder position, scale 1.0, no mask), silu MLP; final norm, n_out heads, and the cfg_scale map
 * (util.cpp:175-200): cond + scale * (cond - uncond); its "> max_output -> -inf" line is overwritten by the next
 * statement in the reference, so no logit is masked. */
typedef struct {
    orc_w q, k, v, o, gate, up, out;
    const float *sa_norm, *mlp_norm;
} orc_dia_enc_layer;
typedef struct {
    orc_w sq, sk, sv, so, cq, ck, cv, co, gate, up, out;
    const float *sa_norm, *ca_norm, *mlp_norm;
} orc_dia_dec_layer;
typedef struct {
    int32_t enc_H, enc_L, enc_heads, enc_F;
    int32_t dec_H, dec_L, dec_heads, dec_kv_heads, dec_F;
    int32_t head_dim, n_out, V, max_ctx, max_gen;
    int32_t act_mode;
    float   cfg_scale;
    orc_w enc_embd;                      /* dia.encoder.embedding [256][enc_H] */
    const float *enc_norm, *dec_norm;
    orc_w dec_embd[ORC_MAX_HEADS];       /* dia.decoder.embeddings.N [V][dec_H] */
    orc_w heads[ORC_MAX_HEADS];          /* dia.decoder.heads.N [V][dec_H] */
    orc_dia_enc_layer enc[ORC_MAX_LAYERS];
    orc_dia_dec_layer dec[ORC_MAX_LAYERS];
    /* 0: the reference's graph — rope on the cross-attention query (decoder position, model.cpp:606) and on the cross keys (encoder position,
     * :489), as the `dia` package the converter imports did.  1: no rope in cross-attention, which is what Hugging Face's DiaModel computes
     * (the later upstream code): the switch exists so that the rest of the graph can be pinned to that implementation
     * (tests/test_upstream_golden.py::test_dia_against_transformers_dia) with the difference isolated and asserted. */
    int32_t no_cross_rope;
} orc_dia_model;
typedef struct orc_dia_state orc_dia_state;
orc_dia_state *orc_dia_state_new(const orc_dia_model *m);
void           orc_dia_state_free(orc_dia_state *s);
/* tokens [max_ctx]: the sentence bytes followed by zeros (tokenize_sentence :661-705); fills the cross K/V of both
 * streams; enc_out (may be NULL) [2][max_ctx][enc_H] final-normed encoder states */
void orc_dia_encode(const orc_dia_model *m, orc_dia_state *s, const uint32_t *tokens, int sentence_len, float *enc_out);
/* one decoder step at position pos with the n_out ids of the previous step; logits_out [n_out][V] after cfg_scale;
 * raw_out (may be NULL) [2][n_out][V] conditional / unconditional logits */
void orc_dia_step(const orc_dia_model *m, orc_dia_state *s, const uint32_t *ids, uint32_t pos, float *logits_out, float *raw_out);
/* dia_runner::check_stopping (:767-785): may overwrite ids with eos / pad; *delay_steps carries dctx->delay_steps (-1 at
 * the start); returns 1 when generation stops */
int orc_dia_check_stopping(uint32_t *ids, int n_out, const uint32_t *delay_pattern, uint32_t max_delay, uint32_t eos, uint32_t pad,
                           uint32_t current_position, uint32_t max_generation_size, int *delay_steps);
/* dia_runner::adjust_output_tokens (:787-808) */
size_t orc_dia_adjust_output_tokens(const uint32_t *tokens, size_t size, int n_out, const uint32_t *delay_pattern, uint32_t max_delay,
                                    uint32_t audio_vocab, uint32_t *filtered);

/* ---- SNAC decoder (src/decoder/snac_model.cpp:86-159, general_neural_audio_codec.cpp:133-172) --------
 * Same layer builder as DAC with three differences: the codebooks run at 1/4, 1/2 and 1x the latent rate and are
 * repeat-interleaved up (:97-101), the first conv of the model and of every residual unit is depthwise (groups = channels),
 * and every layer adds noise * conv1x1(x) after its transposed conv (gnac.cpp:155-159). */
typedef struct {
    const float *in_alpha, *in_w, *in_b;   /* snake alpha [C], depthwise conv k7 W [C][1][7], b [C] */
    const float *out_alpha, *out_w, *out_b;/* snake alpha [C], conv k1 W [C][C][1], b [C]           */
} orc_snac_res;
typedef struct {
    int32_t stride, padding, cin, cout;
    const float *alpha, *w, *b;             /* snake, ConvTranspose1d [cin][cout][2*stride], bias   */
    const float *noise_w;                   /* conv k1 [cout][cout][1], no bias                     */
    orc_snac_res res[3];
} orc_snac_block;
typedef struct {
    int32_t n_codebooks, codebook_dim, codebook_size, latent;
    int32_t repeats[4];                     /* 4, 2, 1 (snac_model.h:17)                            */
    const float *codebook[4], *out_proj_w[4], *out_proj_b[4];
    const float *in_w, *in_b;               /* depthwise conv k7 [latent][1][7]                     */
    int32_t c0;
    const float *up_w, *up_b;               /* conv k1 [c0][latent][1]                              */
    int32_t n_blocks;
    orc_snac_block blocks[8];
    const float *final_alpha, *final_w, *final_b;
} orc_snac_model;
/* codes: level-major as snac_runner::set_inputs lays them out (:161-178): T/repeats[0] ids of level 0, then
 * T/repeats[1] of level 1, ...; noise: per layer l, L_l floats (L_l = T * prod(stride_0..l)), concatenated (:131-137),
 * or NULL for no noise; pcm_out: T * prod(strides). */
int64_t orc_snac_decode(const orc_snac_model *m, const uint32_t *codes, int T, const float *noise, float *pcm_out);
void orc_conv1d_dw(const float *x, int C, int64_t L, const float *w, const float *b, int K, int pad, int dil, float *y);

/* primitives exposed for unit tests */
void orc_conv1d(const float *x, int cin, int64_t L, const float *w, const float *b, int cout, int K,
                int pad, int dil, float *y);
void orc_conv_transpose1d(const float *x, int cin, int64_t L, const float *w, const float *b, int cout,
                          int K, int stride, int pad, float *y);
void orc_snake(float *x, int C, int64_t L, const float *alpha);
void orc_layer_norm(const float *x, int H, const float *w, const float *b, float *y);
float orc_gelu(float x, int mode);

int orc_set_threads(int n); /* OpenMP threads used by mul_mat / conv; returns the value in effect */

#ifdef __cplusplus
}
#endif
#endif

/*
 * tts_hip.h — C ABI of the MI355X (gfx950) compute shim underneath TTS.cpp's
 * tts_generation_runner API.  Plain pointers and sizes only; no C++/torch types.
 *
 * What this replaces in the reference (file:line under /root/reference):
 *   - the ggml graph build + backend-sched execution inside
 *       parler_tts_runner::decode            src/models/parler/model.cpp:648-693
 *       parler_tts_runner::build_parler_graph                      :520-614
 *       parler_tts_model::prep_cross_key_values                    :110-173
 *       parler_kv_cache_init / parler_build_kv_store               :339-385, :420-439
 *       dac_runner::run / build_dac_graph    src/decoder/dac_model.cpp:146-212
 *   - the weight buffer owned by tts_model (tts_model::set_tensor, src/tts_model.cpp:157-164)
 *   - runner_context's device/backend state (src/tts_model.h:16-44)
 * The host C++ runner (tts.cpp_amd/host) keeps tokenisation, the AR loop, sampling and the
 * delay pattern exactly where the reference has them and calls into this ABI once per
 * decode step and once per codec decode.
 *
 * Conventions: every int-returning call returns 0 on success, non-zero on failure and
 * leaves a message retrievable by tts_hip_last_error().  The reference has no error codes
 * (TTS_ABORT -> abort(), src/util.cpp:14-22); the C++ wrapper aborts on non-zero to match.
 * A context is single-threaded from the caller's point of view (one HIP stream + one device
 * per context), mirroring "one runner per worker thread" (examples/server/server.cpp:316-321).
 */
#ifndef TTS_HIP_H
#define TTS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ggml tensor type ids as they appear in GGUF tensor infos (examples/quantize/README.md:48-55) */
enum tts_hip_type {
    TTS_HIP_F32  = 0,
    TTS_HIP_F16  = 1,
    TTS_HIP_Q4_0 = 2,
    TTS_HIP_Q5_0 = 6,
    TTS_HIP_Q8_0 = 8,
};

#define TTS_HIP_MAX_DAC_BLOCKS 8

/* Hyper-parameters that the reference reads from GGUF KV pairs
 * (parler_tts_model::prep_constants model.cpp:51-108, dac_model::prep_constants/prep_layers
 * dac_model.cpp:15-55).  Shapes that the reference takes from tensor shapes (FFN width,
 * prompt vocab, DAC channel counts) are inferred from the uploaded tensors at finalize. */
typedef struct tts_hip_desc {
    uint32_t struct_size;         /* sizeof(tts_hip_desc), ABI guard */
    /* Parler decoder */
    uint32_t hidden_size;         /* parler-tts.decoder.hidden_size          (1024) */
    uint32_t n_layers;            /* parler-tts.decoder.num_hidden_layers    (24)   */
    uint32_t n_attn_heads;        /* parler-tts.decoder.attention.head_count (16)   */
    uint32_t n_output_heads;      /* parler-tts.decoder.output_heads         (9)    */
    uint32_t output_vocab_size;   /* parler-tts.decoder.out_vocab_size       (1088) */
    uint32_t max_ctx_length;      /* parler-tts.decoder.context_length       (4096) */
    uint32_t n_encode_length;     /* parler-tts.decoder.encode_length (voice prompt tokens) */
    uint32_t use_cross_attn;      /* generation_configuration.use_cross_attn (common.h:59) */
    /* DAC */
    uint32_t dac_n_blocks;        /* 4 (dac_model.h:33) */
    uint32_t dac_stride[TTS_HIP_MAX_DAC_BLOCKS];   /* dac.dac_layer_stride_i  */
    uint32_t dac_padding[TTS_HIP_MAX_DAC_BLOCKS];  /* dac.dac_layer_padding_i */
    uint32_t dac_max_frames;      /* parler-tts.decoder.max_generation (2580) */
    /* engine knobs (extensions; the reference is max_seqs=1, kv F32) */
    uint32_t max_seqs;            /* utterances decoded in lock-step on this device */
    uint32_t kv_type;             /* TTS_HIP_F32 (reference, model.h:138-139) or TTS_HIP_F16 */
    uint32_t gelu_mode;           /* 1 = ggml CPU's fp16-table GELU semantics, 0 = fp32 tanh GELU */
    uint32_t flags;               /* TTS_HIP_FLAG_* */
    uint32_t kv_positions;        /* positions kept per sequence in the self-attention cache; 0 = max_ctx_length
                                     (reference, model.cpp:368-369).  Generation stops at position
                                     max_generation (check_stopping, model.cpp:720-722), so max_generation suffices. */
} tts_hip_desc;

#define TTS_HIP_FLAG_NO_GRAPH   1u  /* launch kernels eagerly instead of replaying a captured hipGraph */
#define TTS_HIP_FLAG_VALU_GEMM  2u  /* use the scalar-FMA reference GEMV kernels instead of MFMA (debug/parity) */
#define TTS_HIP_FLAG_NO_DAC     4u  /* context carries no audio codec */
#define TTS_HIP_FLAG_NO_PARLER  8u  /* context carries only the audio codec */
#define TTS_HIP_FLAG_DAC_F32    32u /* F16 codec tensors: compute with fp32 activations (exact-fp32 MFMA) instead of ggml's
                                       fp16 im2col x fp16 kernel semantics */
#define TTS_HIP_FLAG_DEQUANT_Q  16u /* decode Q4_0/Q5_0/Q8_0 matrices to fp32 at upload (fp32 activations) instead of
                                       the integer path with Q8_0-quantised activations (ggml's CPU semantics) */

typedef struct tts_hip_ctx tts_hip_ctx;

/* ---- lifecycle ------------------------------------------------------------------------- */
int          tts_hip_device_count(void);
tts_hip_ctx *tts_hip_create(int device, const tts_hip_desc *desc);  /* NULL on failure */
void         tts_hip_destroy(tts_hip_ctx *ctx);
const char  *tts_hip_last_error(void);
const char  *tts_hip_version(void);

/* ---- weights -----------------------------------------------------------------------------
 * One call per GGUF tensor, with the tensor's GGUF name, e.g.
 * "decoder.layers.3.self_attn.q_proj.weight", "audio_encoder.decoder_block.2.residual_unit.0.res.initial.weight"
 * (== runner->assign_weight(name, tensor), src/models/loaders.cpp:79-88, parler/model.cpp:500-508).
 * ne[] is the GGUF order (ne[0] fastest).  host_data may be NULL: the tensor is then only
 * declared (its bytes arrive later through the arena, see tts_hip_arena_*). Unknown names are
 * ignored with return value 0 and a warning, like the reference (model.cpp:506). */
int tts_hip_upload(tts_hip_ctx *ctx, const char *name, int type, int n_dims, const int64_t *ne,
                   const void *host_data);

/* Lays the weight arena out (fusing q/k/v and the lm heads into single matrices), moves the
 * uploaded tensors into it, allocates KV cache + workspaces, and — when weights are present —
 * runs prep_cross_key_values (model.cpp:110-173).  `external_arena` may be NULL (the shim
 * allocates) or a device pointer of at least tts_hip_arena_bytes() bytes owned by the caller
 * (e.g. a torch uint8 tensor that is then broadcast with RCCL). */
size_t tts_hip_arena_bytes(tts_hip_ctx *ctx); /* valid after all uploads, before or after finalize */
int    tts_hip_finalize(tts_hip_ctx *ctx, void *external_arena);
void  *tts_hip_arena_ptr(tts_hip_ctx *ctx);
/* After the arena of a declare-only context was filled by a collective: mark weights present.
 * (cross K/V live inside the arena, so they arrive with the broadcast.) */
int    tts_hip_arena_filled(tts_hip_ctx *ctx);

/* The one collective of the path: RCCL broadcast of the finished weight arena (weights + precomputed cross
 * K/V) from ctxs[root] to every other context, so that a host serving G devices parses and uploads the GGUF once — where the
 * reference's server re-parses the file for every worker (examples/server/server.cpp:316-321).  All n contexts belong to this
 * process, one per DISTINCT device, each finalized with its own arena of the same tts_hip_arena_bytes(); the non-root contexts
 * were filled declare-only (tts_hip_upload with host_data == NULL) and are marked ready here (tts_hip_arena_filled).
 * Single process: ncclCommInitAll over the contexts' devices, ncclGroupStart, one ncclBroadcast(uint8, arena bytes) per context on
 * its own stream, ncclGroupEnd, stream syncs, communicators destroyed.  n == 1 is a no-op that only validates.  librccl.so is
 * opened on first use (it is not a load-time dependency of this library). */
int    tts_hip_broadcast_weights(tts_hip_ctx **ctxs, int n, int root);
/* One-process-per-GPU form of the same broadcast: rank 0 obtains a 128-byte id (tts_hip_comm_unique_id) and hands it to the other
 * ranks by whatever the launcher offers (bench.py: torch.distributed); every rank then calls tts_hip_broadcast_weights_rank. */
int    tts_hip_comm_unique_id(void *id128);
int    tts_hip_broadcast_weights_rank(tts_hip_ctx *ctx, const void *id128, int rank, int world, int root);
/* update_conditional_prompt (model.cpp:510-518): replace the voice-prompt encoding
 * [n_tokens][hidden] (fp32, host) and recompute the cross K/V. n_tokens <= n_encode_length cap given at create. */
int    tts_hip_parler_set_text_encoding(tts_hip_ctx *ctx, const float *enc, uint32_t n_tokens);

/* ---- Parler decoder -------------------------------------------------------------------- */
/* pctx->reset + cache reuse (model.cpp:312-322,848-856): forget all sequences' positions. */
int tts_hip_parler_reset(tts_hip_ctx *ctx);
/* decode() with audio_generation=false (model.cpp:648-693 on a batch_from_sentence batch :473-498):
 * run `n` text-prompt ids of sequence `seq` at positions pos0..pos0+n-1 and append their K/V.
 * The prompt logits are never consumed by the reference (sampler only runs on audio steps,
 * model.cpp:774-776), so none are produced. */
int tts_hip_parler_prefill(tts_hip_ctx *ctx, uint32_t seq, const uint32_t *text_ids, uint32_t n, uint32_t pos0);
/* The same for n sequences in as few forwards as possible (extension): ids = the prompts concatenated, lens[n];
 * seqs[n] (NULL = 0..n-1) and pos0[n] (NULL = all 0).  Equivalent to n tts_hip_parler_prefill calls. */
int tts_hip_parler_prefill_batch(tts_hip_ctx *ctx, uint32_t n, const uint32_t *seqs, const uint32_t *text_ids,
                                 const uint32_t *lens, const uint32_t *pos0);
/* decode() with audio_generation=true for n_seqs sequences in lock-step:
 *   ids   [n_seqs][n_output_heads]  codebook ids fed this step (model.cpp:394-403,778-785)
 *   pos   [n_seqs]                  absolute position of each sequence (model.cpp:784)
 *   seqs  [n_seqs] or NULL          cache slot of each row (NULL = 0..n_seqs-1)
 *   logits_out [n_seqs][n_output_heads][output_vocab_size] fp32 host memory (model.cpp:455-456,682-683)
 * Blocks until the logits are in logits_out. */
int tts_hip_parler_step(tts_hip_ctx *ctx, uint32_t n_seqs, const uint32_t *ids, const uint32_t *pos,
                        const uint32_t *seqs, float *logits_out);
/* Same step, but sampler::max (src/sampler.cpp:185-204, first maximum wins, no repetition
 * penalty) is evaluated on the device; tokens_out [n_seqs][n_output_heads]. */
int tts_hip_parler_step_greedy(tts_hip_ctx *ctx, uint32_t n_seqs, const uint32_t *ids, const uint32_t *pos,
                               const uint32_t *seqs, uint32_t *tokens_out);
/* Device-resident greedy generation of `n_steps` audio steps for n_seqs sequences whose prompts
 * were prefetched with tts_hip_parler_prefill: the delay-pattern feed (model.cpp:778-785) and the
 * EOS bookkeeping (check_stopping :715-732) run on the device, the host synchronises once.
 *   start_pos [n_seqs]: position of the first audio step (prompt length)
 *   tokens_out [n_steps][n_seqs][n_output_heads] sampled ids, still delayed (pctx->output_tokens)
 *   steps_done [n_seqs] (may be NULL): number of steps executed before all heads saw EOS
 * bos/eos: audio.bos_token_id / audio.eos_token_id. */
int tts_hip_parler_generate_greedy(tts_hip_ctx *ctx, uint32_t n_seqs, const uint32_t *start_pos,
                                   uint32_t n_steps, uint32_t bos, uint32_t eos, uint32_t *tokens_out,
                                   uint32_t *steps_done);

/* sampler::sample on the device (src/sampler.cpp:3-69: softmax :82-116, topk :152-183, topp :118-150) for
 * output_vocab_size <= 2048.  repetition_penalty != 1 keeps sampler::last_token_ids / repetition_counts per
 * (sequence, head) on the device (reset at the start of a generation, sampler.cpp:71-80).  top_k == 0 or >= vocab disables top-k, top_p >= 1
 * disables top-p; the order of the fp32 sums follows the reference (see sample_kernel).  The U[0,1) draws are the
 * caller's (the reference draws them from std::minstd_rand, sampler.cpp:47-48). */
typedef struct tts_hip_sampling {
    uint32_t top_k;
    float    top_p;
    float    temperature;
    float    repetition_penalty;   /* 1 = off */
} tts_hip_sampling;
/* tts_hip_parler_generate_greedy with sampler::sample instead of sampler::max.
 *   uniforms [n_steps][n_seqs][n_output_heads]: draw for head h of sequence s at its k-th sampler call */
int tts_hip_parler_generate_sampled(tts_hip_ctx *ctx, uint32_t n_seqs, const uint32_t *start_pos,
                                    uint32_t n_steps, uint32_t bos, uint32_t eos, const tts_hip_sampling *sampling,
                                    const float *uniforms, uint32_t *tokens_out, uint32_t *steps_done);
/* The device sampler alone on caller-supplied logits [n_rows][n_output_heads][output_vocab_size]
 * (uniforms [n_rows][n_output_heads]) -> tokens_out [n_rows][n_output_heads]; for parity tests.
 * last_ids / rep_counts [n_rows][n_output_heads]: the repetition state, read and updated in place (may be NULL when
 * repetition_penalty == 1). */
int tts_hip_sample_logits(tts_hip_ctx *ctx, uint32_t n_rows, const float *logits, const tts_hip_sampling *sampling,
                          const float *uniforms, int32_t *last_ids, uint32_t *rep_counts, uint32_t *tokens_out);

/* ---- T5 voice-prompt encoder (src/models/parler/t5/model.cpp) ---------------------------------
 * What parler_tts_runner::update_conditional_prompt runs (model.cpp:510-518): text_encoder_from_file ->
 * t5_runner::generate -> prep_cross_key_values.  A T5 context is its own tts_hip_ctx: create, tts_hip_upload
 * every "t5encoder.*" tensor of the encoder GGUF (names: t5/model.cpp:3-18), tts_hip_finalize(ctx, NULL),
 * tts_hip_t5_encode, then hand the result to the decoder context with tts_hip_parler_set_text_encoding. */
typedef struct tts_hip_t5_desc {
    uint32_t struct_size;      /* sizeof(tts_hip_t5_desc) */
    uint32_t hidden_size;      /* t5encoder.embedding_length      (t5/model.cpp:129-132) */
    uint32_t n_layers;         /* t5encoder.block_count           (:124-127) */
    uint32_t n_attn_heads;     /* t5encoder.attention.head_count  (:134-137); head size is 64 (t5/model.h:46) */
    uint32_t max_ctx_length;   /* t5encoder.context_length        (:139-142) */
    uint32_t n_buckets;        /* relative_attn_buckets, 32 (t5/model.h:48); 0 = 32 */
    uint32_t output_size;      /* t5encoder.output_size (:160-163), 0 = take it from the tensors */
    uint32_t gelu_mode;        /* as tts_hip_desc */
    uint32_t flags;            /* TTS_HIP_FLAG_VALU_GEMM | TTS_HIP_FLAG_DEQUANT_Q */
} tts_hip_t5_desc;
tts_hip_ctx *tts_hip_t5_create(int device, const tts_hip_t5_desc *desc);
/* t5_runner::run (t5/model.cpp:321-357): ids [n_tokens] (the caller appends EOS, :361-362) ->
 * out [n_tokens][output size] fp32 (final rms norm, then down_proj + bias when the GGUF has them) */
int tts_hip_t5_encode(tts_hip_ctx *ctx, const uint32_t *ids, uint32_t n_tokens, float *out);
int tts_hip_t5_output_size(tts_hip_ctx *ctx);   /* after tts_hip_finalize / tts_hip_arena_bytes; < 0 on error */

/* ---- Orpheus decoder: Llama-3 blocks (src/models/orpheus/model.cpp) --------------------------------------
 * Device side of orpheus_runner::decode (:298-325): create, tts_hip_upload every "orpheus.*" tensor (names :11-60),
 * tts_hip_finalize(ctx, NULL), then tts_hip_orpheus_decode per call of decode() — the prompt batch first, one token
 * per step after.  Tokenizer (BPE), prompt framing (:341-356), sampler and the 7-ids-per-frame -> SNAC level mapping
 * (:358-376) stay with the host. */
typedef struct tts_hip_orpheus_desc {
    uint32_t struct_size;
    uint32_t hidden_size;      /* orpheus.hidden_size (3072)   */
    uint32_t n_layers;         /* orpheus.layers (28)          */
    uint32_t n_attn_heads;     /* orpheus.attn_heads (24)      */
    uint32_t n_kv_heads;       /* orpheus.kv_attn_heads (8)    */
    uint32_t head_dim;         /* orpheus.head_dim (128)       */
    uint32_t vocab_size;       /* orpheus.vocab_size (156940)  */
    uint32_t n_ctx;            /* KV positions: max_context_length + max_generation_size (1024 + 2100, :176-177) */
    float    rope_base;        /* 500000 (:190,250); 0 = that  */
    uint32_t flags;            /* TTS_HIP_FLAG_VALU_GEMM | TTS_HIP_FLAG_DEQUANT_Q */
    uint32_t max_seqs;         /* KV-cache slots for lock-step utterances (tts_hip_orpheus_generate_batch); 0 / 1: one sequence, as the reference's runner */
} tts_hip_orpheus_desc;
tts_hip_ctx *tts_hip_orpheus_create(int device, const tts_hip_orpheus_desc *desc);
/* n tokens at positions pos0 .. pos0+n-1 (KV cache appended); logits_out [vocab_size] of the LAST token (:287-290).
 * tokens_out (may be NULL): sampler::max of those logits (first maximum wins), evaluated on the device. */
int tts_hip_orpheus_decode(tts_hip_ctx *ctx, const uint32_t *ids, uint32_t n, uint32_t pos0, float *logits_out, uint32_t *token_out);
/* greedy generate_from_batch (:378-392 with sampler::max): decode the prompt, then feed back the arg-max until
 * stop_id or max_new ids; returns the count in *n_out */
int tts_hip_orpheus_generate_greedy(tts_hip_ctx *ctx, const uint32_t *prompt, uint32_t n_prompt, uint32_t max_new, uint32_t stop_id,
                                    uint32_t *tokens_out, uint32_t *n_out);
/* the same loop with sampler::sample (orpheus/model.cpp:389-398, src/sampler.cpp:3-69) on the device: top_k in 1..64 (the default
 * generation_configuration, include/common.h:45-55, has top_k 50, top_p 1), any top_p > 0, temperature and repetition penalty.
 * The reference sorts all 156 940 logits on the host at every step; here two kernels select the top_k candidates in the
 * reference's order (value descending; equal values by index) and run the softmax / inverse-CDF scan in the reference's
 * operation order.  top_p < 1 (nucleus sampling, sampler.cpp:22-27, 118-150): a third kernel first accumulates the softmax
 * total over the whole vocabulary in index order, as the reference's fp32 sum is, and the picks keep their full-vocabulary
 * probabilities (about 0.65 ms per token more).  uniforms [max_new]: the U[0,1) draw of the k-th sampler call (the caller's
 * std::minstd_rand).  top_k 0 or > 64 is refused: sample on the host from tts_hip_orpheus_decode's logits. */
int tts_hip_orpheus_generate_sampled(tts_hip_ctx *ctx, const uint32_t *prompt, uint32_t n_prompt, uint32_t max_new, uint32_t stop_id,
                                     const tts_hip_sampling *sampling, const float *uniforms, uint32_t *tokens_out, uint32_t *n_out);
/* Lock-step utterances (the reference runs N independent workers with a model copy each, examples/server/server.cpp:225-321): one forward
 * carries one row per live utterance, row r in cache slot slots[r] at position pos[r].  step_batch: logits_out [n][vocab_size] and / or tokens_out [n]
 * (sampler::max per row) may be NULL.  generate_batch: generate_from_batch (orpheus/model.cpp:378-392) for n_utt utterances at once — prompts are the
 * utterances' ids back to back (n_prompt[u] each), tokens_out [n_utt][max_new], n_out [n_utt]; sampling NULL = sampler::max, else sampler::sample with
 * uniforms [n_utt][max_new] (utterance u's k-th sampler call draws uniforms[u * max_new + k]) and one repetition state per utterance.  Every utterance
 * receives exactly the ids its own one-sequence generation produces; finished utterances leave the step. */
int tts_hip_orpheus_step_batch(tts_hip_ctx *ctx, uint32_t n, const uint32_t *slots, const uint32_t *ids, const uint32_t *pos, float *logits_out, uint32_t *tokens_out);
int tts_hip_orpheus_generate_batch(tts_hip_ctx *ctx, uint32_t n_utt, const uint32_t *prompts, const uint32_t *n_prompt, uint32_t max_new, uint32_t stop_id,
                                   const tts_hip_sampling *sampling, const float *uniforms, uint32_t *tokens_out, uint32_t *n_out);
/* the device sampler alone on caller-supplied logits [vocab_size], for parity tests; last_id / rep_count: sampler::last_token_ids /
 * repetition_counts, read and updated in place (may be NULL when repetition_penalty == 1) */
int tts_hip_orpheus_sample_logits(tts_hip_ctx *ctx, const float *logits, const tts_hip_sampling *sampling, float uniform, int32_t *last_id,
                                  uint32_t *rep_count, uint32_t *token_out);

/* ---- Dia encoder + decoder step (src/models/dia/model.cpp) --------------------------------------------------------
 * Device side of dia_runner::decode (:730-757): create, tts_hip_upload every "dia.*" tensor (names
 * py-gguf/tts_encoders/dia_gguf_encoder.py:72-131), tts_hip_finalize(ctx, NULL), tts_hip_dia_encode once per sentence
 * (build_dia_encoder :383-440 + build_dia_cross_kv_store :505-541, both streams), then tts_hip_dia_step per token.
 * Tokenisation (:661-705), the sampler, check_stopping / the delay pattern (:767-808) stay with the host; the codec is
 * the DAC context (tts_hip_create with TTS_HIP_FLAG_NO_PARLER, "audio_encoder.*"). */
typedef struct tts_hip_dia_desc {
    uint32_t struct_size;
    uint32_t enc_hidden_size;   /* width of dia.encoder.embedding (1024; model.h:68, no GGUF key)        */
    uint32_t enc_layers;        /* dia.encoder.layers (12)                                               */
    uint32_t enc_attn_heads;    /* dia.encoder.attn_heads (16): heads x head_dim == dec_hidden_size       */
    uint32_t dec_hidden_size;   /* dia.decoder.hidden_size (2048)                                        */
    uint32_t dec_layers;        /* dia.decoder.layers (18)                                               */
    uint32_t dec_attn_heads;    /* dia.decoder.attn_heads (16) query heads                               */
    uint32_t dec_kv_heads;      /* attn_heads / dia.decoder.query_heads (16 / 4 = 4 k/v groups, :463,468) */
    uint32_t head_dim;          /* dia.attn_head_size (128)                                              */
    uint32_t n_output_heads;    /* dia.decoder.output_heads (9)                                          */
    uint32_t output_vocab_size; /* dia.decoder.output_vocab_size (1028)                                  */
    uint32_t max_ctx;           /* dia.encoder.max_context_length (1024): the encoder always runs all of it */
    uint32_t max_gen;           /* dia.decoder.max_generation_size (3072) self-attention cache positions  */
    float    cfg_scale;         /* dia.cfg_scale (3.0, model.h:82); 0 = that                             */
    uint32_t flags;             /* TTS_HIP_FLAG_VALU_GEMM | TTS_HIP_FLAG_DEQUANT_Q */
    uint32_t max_utterances;    /* extension: utterances decoded in lock-step (each is the reference's batch of 2 guidance streams,
                                   dia/model.cpp:330-341); 0 = 1.  Rows of a step = 2 x utterances */
} tts_hip_dia_desc;
tts_hip_ctx *tts_hip_dia_create(int device, const tts_hip_dia_desc *desc);
/* tokens [max_ctx]: the sentence bytes then zeros; the all-zero second stream is added here (:700-703).  Fills the cross
 * K/V of every decoder layer (K only for the first sentence_len positions, the rest zero as in a freshly cleared cache)
 * and resets nothing else: self-attention cache rows are overwritten as positions are decoded.
 * enc_out (may be NULL): [2][max_ctx][enc_hidden_size] final-normed encoder states. */
int tts_hip_dia_encode(tts_hip_ctx *ctx, const uint32_t *tokens, uint32_t sentence_len, float *enc_out);
/* one decoder step at position pos with the n_output_heads ids of the previous step (both streams get the same ids).
 * logits_out [n_output_heads][output_vocab_size]: cond + cfg_scale * (cond - uncond) (util.cpp:194-196);
 * raw_out (may be NULL): [2][n_output_heads][output_vocab_size] conditional, unconditional. */
int tts_hip_dia_step(tts_hip_ctx *ctx, const uint32_t *ids, uint32_t pos, float *logits_out, float *raw_out);
/* Lock-step utterances (BASELINE config 3: batch 32 = 4 utterances per GPU x 8 GPUs; SURVEY.md §8e: M = 8 rows per GPU).  The
 * reference decodes one utterance = 2 guidance rows per graph (:330-341, :705-721); here utterance slot u owns rows 2u (text) and
 * 2u+1 (all-zero twin) of every cache, tts_hip_dia_encode_slot fills its cross K/V, and one step carries the rows of n_utt slots:
 * ids [n_utt][n_output_heads], pos [n_utt] (a slot that has finished keeps stepping on its last position; its rows are ignored by
 * the host), slots [n_utt] or NULL (= 0..n_utt-1), logits_out [n_utt][n_output_heads][vocab] guided, raw_out (may be NULL)
 * [n_utt][2][n_output_heads][vocab].  tts_hip_dia_encode / _step are the slot-0, one-utterance forms. */
int tts_hip_dia_encode_slot(tts_hip_ctx *ctx, uint32_t slot, const uint32_t *tokens, uint32_t sentence_len, float *enc_out);
int tts_hip_dia_step_batch(tts_hip_ctx *ctx, uint32_t n_utt, const uint32_t *slots, const uint32_t *ids, const uint32_t *pos,
                           float *logits_out, float *raw_out);
/* The whole generate_from_batch loop (dia/model.cpp:806-870) on the device for utterance slots 0..n_utt-1 (encoded with
 * tts_hip_dia_encode_slot): per step check_stopping (:767-785: EOS on head 0 or position max_gen - max_delay starts the countdown that
 * forces EOS / PAD into the delayed heads), the decoder step, guidance, sampler::sample (sampling != NULL; sample_kernel, any top_k /
 * top_p / temperature / repetition penalty at this vocabulary) or sampler::max (NULL), and the delay-pattern feedback (:795-803) run
 * as one captured graph; the host looks at the done flags every few steps.  uniforms [max_gen][n_utt][n_output_heads]: the draw for
 * head h of utterance u at its k-th sampler call (ignored for sampler::max).  tokens_out [n_utt][max_gen][n_output_heads]: the
 * sampled ids in generation order (the runner's output_tokens before adjust_output_tokens); steps_out [n_utt]: sampler calls made. */
typedef struct tts_hip_dia_codes {
    uint32_t bos, eos, pad;        /* dia.bos_token_id 1026, eos 1024, pad 1025 (model.h:70-72) */
    uint32_t max_delay;            /* model.h:74 (15) */
    uint32_t delay_pattern[16];    /* per output head (model.h:66: 0,8,9,...,15) */
} tts_hip_dia_codes;
int tts_hip_dia_generate(tts_hip_ctx *ctx, uint32_t n_utt, uint32_t max_gen, const tts_hip_dia_codes *codes, const tts_hip_sampling *sampling,
                         const float *uniforms, uint32_t *tokens_out, uint32_t *steps_out);

/* ---- Kokoro (src/models/kokoro/model.cpp) ----------------------------------------------------------------------------
 * Device side of kokoro_duration_runner::run (:1069-1123) and kokoro_runner::run (:1277-1325): create, tts_hip_upload every
 * "kokoro.*" tensor (names py-gguf/tts_encoders/kokoro_gguf_encoder.py), tts_hip_finalize(ctx, NULL), then per clause
 * tts_hip_kokoro_durations (ALBERT + prosody predictor + duration head) and tts_hip_kokoro_generate (alignment ... iSTFT).
 * The phonemizer, the tokenizer, clause chunking (:1340-1388) and the source-noise draws (set_inputs :1255) stay with the host.
 * First version: plain fp32 kernels (csrc/kokoro_kernels.h). */
typedef struct tts_hip_kokoro_desc {
    uint32_t struct_size;
    uint32_t n_attn_heads;       /* kokoro.duration_predictor.albert.attn_heads (12) */
    uint32_t n_recurrence;       /* ...albert.recurrence (12): the one shared layer is applied this many times */
    uint32_t n_dp_layers;        /* kokoro.duration_predictor.layers (3) */
    uint32_t f0_n_blocks;        /* kokoro.duration_predictor.f0_n_blocks (3) */
    uint32_t n_conv_layers;      /* kokoro.text_encoder.layers (3) */
    uint32_t n_decoder_blocks;   /* kokoro.decoder.generator.layers (4) */
    uint32_t n_upsamples;        /* kokoro.decoder.generator.upsamples (2) */
    uint32_t n_kernels;          /* kokoro.decoder.generator.kernels (3) */
    uint32_t n_fft, hop;         /* kokoro.decoder.generator.{n_fft,hop} (20, 5) */
    uint32_t harmonic_num;       /* 8 (model.h:218) */
    uint32_t up_sampling_factor; /* kokoro.decoder.generator.up_sampling_factor (600 samples per duration frame) */
    uint32_t out_conv_padding;   /* kokoro.decoder.generator.padding (3) */
    uint32_t max_ctx;            /* ...albert.context_length (512) tokens per call */
    float    attn_scale;         /* 0.125 (model.h:196: not derived from the head size) */
    float    upsample_scale;     /* 300 (model.h:195) */
    float    sample_rate, sin_amp, noise_std, voice_threshold;   /* 24000, 0.1, 0.003, 10 (model.h:219-222) */
    uint32_t up_stride[4], up_padding[4];             /* kokoro.decoder.generator.up_convs.i.{stride,padding} */
    uint32_t noise_stride[4], noise_padding[4];       /* ...noise_blocks.i.{stride,padding} */
    uint32_t res_padding[16][3], res_dilation[16][3]; /* ...res_blocks.i.j.{padding,dilation} */
    uint32_t noise_res_padding[4][3], noise_res_dilation[4][3];   /* ...noise_blocks.i.res_block.j.{padding,dilation} */
    uint32_t flags;
} tts_hip_kokoro_desc;
tts_hip_ctx *tts_hip_kokoro_create(int device, const tts_hip_kokoro_desc *desc);
/* tokens [n] (bos, phoneme ids, eos); voice: the name after "kokoro.voice_tensors."; lens_out [n] = clamp(round(sum of the
 * duration sigmoids), 1, 50) (:1035-1037); hidden_out [n][duration hidden + style half] (:1029-1031) */
int tts_hip_kokoro_durations(tts_hip_ctx *ctx, const uint32_t *tokens, uint32_t n, const char *voice, float *lens_out, float *hidden_out);
/* lens [n]: whole numbers (the predicted ones, or forced); hidden [n][duration hidden + style half] from the call above;
 * noise [(harmonic_num + 1) * total * up_sampling_factor] uniform draws, total = sum(lens); pcm_out [total * up_sampling_factor].
 * hsrc_out / hsrc_in (may be NULL) [2 * (n_fft / 2 + 1)][2 * total * upsample_scale / hop + 1]: the STFT conditioning read out /
 * replaced (its phase channels wrap at +-pi, see oracle/kokoro_oracle.c). */
int tts_hip_kokoro_generate(tts_hip_ctx *ctx, const uint32_t *tokens, uint32_t n, const float *lens, const float *hidden, const char *voice, const float *noise,
                            float *pcm_out, float *hsrc_out, const float *hsrc_in);

/* ---- SNAC codec (src/decoder/snac_model.cpp; Orpheus' audio decoder) -------------------------------
 * A SNAC context is its own tts_hip_ctx: create, tts_hip_upload every "snac.*" tensor (names:
 * py-gguf/tts_encoders/orpheus_gguf_encoder.py:89-142), tts_hip_finalize(ctx, NULL), tts_hip_snac_decode. */
typedef struct tts_hip_snac_desc {
    uint32_t struct_size;
    uint32_t n_blocks;                          /* snac_model::n_layers (4) */
    uint32_t stride[TTS_HIP_MAX_DAC_BLOCKS];    /* snac.snac_layer_stride_i   (snac_model.cpp:26-48) */
    uint32_t padding[TTS_HIP_MAX_DAC_BLOCKS];   /* snac.snac_layer_padding_i  */
    uint32_t groups[TTS_HIP_MAX_DAC_BLOCKS];    /* snac.snac_layer_grouping_i: must equal the layer's channel count (depthwise, gnac.cpp:136-140) */
    uint32_t n_codebooks;                       /* snac.audio_token_channels (3) */
    uint32_t repeats[4];                        /* 4, 2, 1 (snac_model.h:17) */
    uint32_t max_frames;                        /* snac.max_generation_size: finest-level tokens per call */
    uint32_t flags;                             /* TTS_HIP_FLAG_VALU_GEMM */
} tts_hip_snac_desc;
tts_hip_ctx *tts_hip_snac_create(int device, const tts_hip_snac_desc *desc);
/* snac_runner::run (snac_model.cpp:180-208).  codes: level-major ids as set_inputs lays them out (:161-178): T/repeats[0]
 * of level 0, T/repeats[1] of level 1, ...; noise: per layer l, T * prod(stride_0..l) floats, concatenated (:131-137) —
 * the reference draws them from an unseeded normal generator (:177), here they are the caller's; NULL = no noise;
 * pcm_out: T * prod(strides) fp32 samples. */
int tts_hip_snac_decode(tts_hip_ctx *ctx, const uint32_t *codes, uint32_t T, const float *noise, float *pcm_out);

/* ---- DAC codec --------------------------------------------------------------------------- */
/* dac_runner::run (dac_model.cpp:172-212): codes [frames][n_output_heads] (frame-major),
 * pcm_out: frames * prod(strides) fp32 samples in host memory.  Blocks until done. */
int tts_hip_dac_decode(tts_hip_ctx *ctx, const uint32_t *codes, uint32_t frames, float *pcm_out);
/* The same for n utterances in one pass (extension; the reference decodes one utterance per dac_runner::run).
 * codes: the utterances' code frames concatenated [sum(frames)][n_output_heads]; frames[n]; pcm_out: the PCM of
 * the utterances concatenated (frames[i] * prod(strides) samples each).  Results are identical to n single calls. */
int tts_hip_dac_decode_batch(tts_hip_ctx *ctx, const uint32_t *codes, const uint32_t *frames, uint32_t n, float *pcm_out);

/* ---- introspection (tests, bench) -------------------------------------------------------- */
/* Copy an internal buffer to the host.  what: "hidden" (final-normed hidden of the last forward,
 * [rows][H]), "k:<layer>:<seq>" / "v:<layer>:<seq>" (cache rows [n_pos][H] as fp32),
 * "dac:<stage>" (activation after DAC stage, see oracle stage numbering; requires
 * tts_hip_set_debug(ctx,1) before the decode); Orpheus contexts: "l_logits" (the logits row the last
 * step left), "l_k:<layer>" / "l_v:<layer>" (cache slot 0 of a layer, [n_ctx][kv width]).
 * Returns number of floats written or <0. */
int64_t tts_hip_debug_read(tts_hip_ctx *ctx, const char *what, float *out, size_t max_floats);
int     tts_hip_set_debug(tts_hip_ctx *ctx, int on);

/* Per-kernel-class timing with HIP events on the context's stream.  While enabled, forwards are
 * launched eagerly and every launch of every class is bracketed by an event pair. */
enum tts_hip_kclass {
    TTS_HIP_K_EMBED = 0,        /* embed_rows_kernel */
    TTS_HIP_K_LN = 1,           /* ln_rows_kernel (LayerNorm as its own launch, > 8 rows) */
    TTS_HIP_K_GEMM_QKV = 2,     /* [LN +] fused q/k/v projection + KV-cache append */
    TTS_HIP_K_ATTN_SELF = 3,    /* attn_kernel (+ attn_combine_kernel) over the self-attention cache */
    TTS_HIP_K_GEMM_ATTN_OUT = 4,/* self_attn.out_proj + residual */
    TTS_HIP_K_GEMM_CROSS_Q = 5, /* [LN +] encoder_attn.q_proj */
    TTS_HIP_K_ATTN_CROSS = 6,   /* attn_kernel over the voice-prompt K/V */
    TTS_HIP_K_GEMM_CROSS_OUT = 7,/* encoder_attn.out_proj + residual */
    TTS_HIP_K_GEMM_FC1 = 8,     /* [LN +] fc1 + GELU */
    TTS_HIP_K_GEMM_FC2 = 9,     /* fc2 + residual */
    TTS_HIP_K_GEMM_HEADS = 10,  /* [LN +] 9 lm heads */
    TTS_HIP_K_SAMPLE = 11,      /* argmax_kernel + feed_kernel */
    TTS_HIP_K_GEMM_OTHER = 12,  /* cross K/V precompute */
    TTS_HIP_K_DAC_EMBED = 13,   /* dac_embed_kernel */
    TTS_HIP_K_DAC_CONV7 = 14,   /* conv1d k=7 (initial + residual units) */
    TTS_HIP_K_DAC_CONV1 = 15,   /* conv1d k=1 (+ residual add) */
    TTS_HIP_K_DAC_CONVT = 16,   /* ConvTranspose1d upsampling */
    TTS_HIP_K_DAC_FINAL = 17,   /* final Cout=1 conv + tanh */
    TTS_HIP_K_DAC_RESUNIT = 18, /* resunit_b3_kernel: one residual unit (snake, k=7 conv, snake, k=1 conv, + x) in one launch */
    TTS_HIP_K_KOKORO_CONV = 19, /* Kokoro's stride-1 "same" convolutions on conv1d_mfma_kernel (exact-fp32 MFMA) */
    TTS_HIP_K_COUNT = 20
};
typedef struct tts_hip_kstat {
    double   ms_total;      /* summed event-elapsed time */
    uint64_t launches;
    double   bytes_total;   /* ALGORITHMIC bytes (weights + activations + cache rows each launch must touch) */
    double   flops_total;   /* algorithmic flops */
} tts_hip_kstat;
/* Arithmetic of the codec convolutions of this context (bench.py prices each kernel family against the pipe it runs on): bit 0 = the
 * k = 7 convs of the wide classes as bf16 x 3 split products (TTS_HIP_DAC_BF16X3), bit 1 = residual units of 96 / 192 channels as one
 * launch, bf16 x 3 (TTS_HIP_DAC_FUSE), bit 2 = transposed convs as bf16 x 3 (TTS_HIP_DAC_CONVT_B3), bit 5 (32) = the wide classes keep their activations as bf16 x 3 split planes, so
 * their k = 1 convs are bf16 x 3 products too (TTS_HIP_DAC_PLANES; otherwise exact-fp32 MFMA); 8 = F16 tensors (fp16 im2col, fp16
 * MFMA); 16 = scalar-FMA cross-check kernels; 0 = exact-fp32 MFMA throughout or no codec. */
int tts_hip_dac_arith(tts_hip_ctx *ctx);
int tts_hip_profile(tts_hip_ctx *ctx, int enable);        /* 1: every launch, forwards run eagerly; 2: only the launches that are
                                                              never graph-captured (the DAC), decoder steps keep replaying their
                                                              hipGraph; 0: off.  Enabling clears the counters. */
int tts_hip_profile_get(tts_hip_ctx *ctx, int kclass, tts_hip_kstat *out);
const char *tts_hip_kclass_name(int kclass);

/* ---- continuous batching: a generation that admits new utterances while others are running -----------------------------------------
 * Replaces, for a serving loop, "form a batch from what is queued, run it to the end" (examples/server/server.cpp:126-158 is the queue this
 * widens; parler/model.cpp:762-792 the loop).  The context needs max_seqs >= n_slots + 1: cache slot n_slots pads the lock-step forward.
 *   begin   fixes n_slots, the per-utterance step budget max_steps (token buffer [max_steps][n_slots + 1][heads]), bos / eos and the sampler
 *           (sp == NULL: sampler::max; else sampler::sample on the device as in tts_hip_parler_generate_sampled)
 *   admit   n utterances into free slots: prompts concatenated in ids, lens[i] ids each (prefilled as one side batch, positions from 0);
 *           uniforms [n][max_steps][heads] for a sampled stream (the host-drawn std::minstd_rand sequence of each utterance), else NULL
 *   run     n_steps lock-step decode steps over the live utterances; then *n_finished slots whose check_stopping() fired (EOS on every head,
 *           position == the cached positions, or max_steps reached) are reported with their step counts and become free
 *   collect the tokens [steps][heads] of a finished slot (before the slot is admitted again)
 * An utterance's tokens are those of a solo tts_hip_parler_generate_* run of the same prompt (tests/test_gpu_runner.py). */
int tts_hip_parler_stream_begin(tts_hip_ctx *ctx, uint32_t n_slots, uint32_t max_steps, uint32_t bos, uint32_t eos, const tts_hip_sampling *sp);
int tts_hip_parler_stream_admit(tts_hip_ctx *ctx, uint32_t n, const uint32_t *slots, const uint32_t *ids, const uint32_t *lens, const float *uniforms);
int tts_hip_parler_stream_run(tts_hip_ctx *ctx, uint32_t n_steps, uint32_t *n_finished, uint32_t *finished_slots, uint32_t *finished_steps);
int tts_hip_parler_stream_collect(tts_hip_ctx *ctx, uint32_t slot, uint32_t steps, uint32_t *tokens_out);
int tts_hip_parler_stream_end(tts_hip_ctx *ctx);

/* Tuning and fallback switches by name (profiles/ harnesses and the fallback parity test; call between tts_hip_create and the first
 * launch).  Unknown key: -1.  Not an environment variable on purpose: a deployment cannot flip a kernel path by accident. */
int tts_hip_tune(tts_hip_ctx *ctx, const char *key, int value);

/* stream / device handles for callers that need to order their own work (torch interop) */
void *tts_hip_stream(tts_hip_ctx *ctx);
int   tts_hip_synchronize(tts_hip_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* TTS_HIP_H */

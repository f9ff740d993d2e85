/*
 * tts_c.h — C ABI over the C++ runner API (tts.cpp_amd/host/common.h), for language bindings
 * (ctypes / cgo / JNI / N-API) and the parity tests.
 *
 * Each entry point corresponds to one call the reference's applications make on its C++ API
 * (file:line under /root/reference):
 *   tts_c_runner_from_file  -> runner_from_file()                      src/models/loaders.h:19-20, loaders.cpp:34-95
 *   tts_c_generate          -> tts_generation_runner::generate()       include/common.h:93, parler/model.cpp:838-858
 *   tts_c_sampling_rate     -> tts_runner::sampling_rate               include/common.h:70
 *   tts_c_arch              -> runner->loader.get().arch               examples/perf_battery/perf_battery.cpp:116
 *   tts_c_free              -> ~tts_generation_runner
 * plus test hooks for the host pieces that have no device dependency (tokenizer, sampler, GGUF reader).
 * Errors: the reference aborts (TTS_ABORT, src/util.cpp:14-22); through this ABI the same conditions are
 * returned as NULL / non-zero with tts_c_last_error().
 */
#ifndef TTS_C_H
#define TTS_C_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct tts_c_runner tts_c_runner;

/* generation_configuration (include/common.h:45-66) */
typedef struct tts_c_config {
    const char *voice;
    int         top_k;
    float       temperature;
    float       repetition_penalty;
    int         use_cross_attn;
    int         max_tokens;
    float       top_p;
    int         sample;
    uint64_t    seed; /* extension: 0 = unseeded like the reference (src/sampler.cpp:47) */
} tts_c_config;

void          tts_c_default_config(tts_c_config *cfg);
tts_c_runner *tts_c_runner_from_file(const char *path, int n_threads, const tts_c_config *cfg, int cpu_only);
/* *data stays owned by the runner and valid until the next generate (dac_model.cpp:190-191) */
int           tts_c_generate(tts_c_runner *r, const char *text, const tts_c_config *cfg, const float **data, size_t *n_outputs);
/* Extension: n utterances in lock-step on the runner's device (parler_runner::generate_batch).  texts[n];
 * data[i] / n_outputs[i] receive each utterance's PCM (runner-owned, valid until the next generate*). The runner
 * must have been loaded with TTS_HIP_MAX_SEQS >= n in the environment. */
int           tts_c_generate_batch(tts_c_runner *r, const char *const *texts, int n, const tts_c_config *cfg, const float **data,
                                   size_t *n_outputs);
/* Extension: any number of utterances through ONE continuous-batching session of the runner (tts_generation_runner::generate_stream): at most
 * max_seqs - 1 generate at a time, and a row freed by an utterance that finishes is refilled from texts[] at the next look-in point (every 32
 * decode steps) instead of idling until the longest one is done.  Same outputs as n generate() calls; data[i] valid until the next call. */
int           tts_c_generate_stream(tts_c_runner *r, const char *const *texts, int n, const tts_c_config *cfg, const float **data,
                                    size_t *n_outputs);
/* Placement of the NEXT tts_c_runner_from_file on the calling thread (host/common.h tts_load_options): device (< 0: TTS_HIP_DEVICE or
 * 0), lock-step KV slots (0: TTS_HIP_MAX_SEQS or 1), declare_only != 0: lay the model out without uploading its bytes — the weights
 * then arrive in tts_hip_arena_ptr(tts_c_runner_device_context(r)) by a collective and tts_hip_arena_filled() marks them present. */
void          tts_c_set_load_options(int device, int max_seqs, int declare_only);
/* the same plus share_with (tts_load_options::share_with): a loaded runner of the same model on the same device whose weight arena the
 * next runner uses instead of uploading its own (own KV cache, own stream, own voice prompt once it is updated); the reference's server
 * loads the file once per worker (examples/server/server.cpp:316-321).  Lifetime: the arena is reference counted in the device library, so
 * the runners may be freed in any order — the memory goes with the last of them.  Both calls apply to ONE load: tts_c_runner_from_file
 * resets the thread's options to the defaults when it returns. */
void          tts_c_set_load_options_ex(int device, int max_seqs, int declare_only, tts_c_runner *share_with);
/* the runner's tts_hip_ctx* (include/tts_hip.h: arena, profiling), or NULL when the architecture keeps several contexts */
void         *tts_c_runner_device_context(tts_c_runner *r);
/* the loaded runner's own tokenizer (Parler: unigram ids + EOS as batch_from_sentence builds them); returns the id count */
int           tts_c_runner_tokenize(tts_c_runner *r, const char *text, uint32_t *out, int cap);
float         tts_c_sampling_rate(tts_c_runner *r);
const char   *tts_c_arch(tts_c_runner *r);
void          tts_c_free(tts_c_runner *r);
const char   *tts_c_last_error(void);

/* ---- test hooks ---------------------------------------------------------------------------------- */
/* which = 0: prompt ids of the last generate (tokenised + EOS); 1: sampled ids, still delayed
 * (pctx->output_tokens).  Returns the count (copies at most cap). */
int tts_c_last_tokens(tts_c_runner *r, int which, uint32_t *out, int cap);
/* unigram tokenizer from the GGUF vocabulary + EOS, as batch_from_sentence builds it (model.cpp:473-498); a GGUF with
 * tokenizer.ggml.merges gets the byte-pair tokenizer instead (src/tokenizer.cpp:265-296; ids only, no framing) */
int tts_c_tokenize(const char *gguf_path, const char *text, uint32_t *out, int cap);

typedef struct tts_c_sampler_cfg {
    uint32_t n_output_heads, vocab_size, top_k;
    float    temperature, top_p, repetition_penalty;
    int      do_sample;
    uint64_t seed;
} tts_c_sampler_cfg;
/* sampler::sample (src/sampler.cpp:3-69); uniforms != NULL injects the per-head U[0,1) draws */
int tts_c_sampler_sample(const tts_c_sampler_cfg *cfg, const int32_t *last_ids, const uint32_t *counts, float *logits,
                         const float *uniforms, uint32_t *out);

/* tts_generation_runner::update_conditional_prompt (common.h:91; parler/model.cpp:510-518): encode `prompt` with the
 * T5 encoder GGUF at text_encoder_path and make it the voice prompt of every later generate().  which == 2 in
 * tts_c_last_tokens returns the ids it was encoded from. */
int tts_c_update_conditional_prompt(tts_c_runner *r, const char *text_encoder_path, const char *prompt);

/* ---- device pool: the server's worker pool (examples/server/server.cpp:126-330,885-895) with one worker per device
 * and dynamic lock-step batching of the queued requests (tts.cpp_amd/host/device_pool.h).  No HTTP. -------------- */
typedef struct tts_c_pool tts_c_pool;
/* n_workers workers (server --n-parallelism), worker w on devices[w % n_devices] (NULL: device 0); each worker decodes
 * up to max_batch compatible queued requests together, waiting batch_window_ms for more after the first. */
tts_c_pool *tts_c_pool_create(const char *model_path, int n_workers, const int *devices, int n_devices, int max_batch,
                              int batch_window_ms, const tts_c_config *load_cfg);
/* server --text-encoder-path: the T5 GGUF that CONDITIONAL_PROMPT tasks use; applies to pools created afterwards by
 * this thread (NULL / "" = none) */
void tts_c_pool_set_text_encoder(const char *path);
/* pool_options::continuous for pools created afterwards by this thread: a worker keeps one generation session per run of compatible requests
 * and admits queued requests into rows that free up while the others are still generating (continuous batching), instead of running each
 * batch to its end.  tts_c_pool_admitted_in_flight: how many requests entered a session that way. */
void     tts_c_pool_set_continuous(int on);
/* pool_options::continuous_yield_ms for pools created afterwards by this thread (default 2000): a continuous session stops admitting compatible
 * requests once a request it cannot take (another model, incompatible sampling parameters) has waited this long at the head of the queue */
void     tts_c_pool_set_continuous_yield_ms(int ms);
uint64_t tts_c_pool_admitted_in_flight(tts_c_pool *pool);
int  tts_c_pool_submit(tts_c_pool *pool, const char *text, const tts_c_config *cfg);   /* task id, < 0 on error */
/* CONDITIONAL_PROMPT task (server.cpp:263-271) fanned out to every worker; wait on the id like any task (no audio:
 * tts_c_pool_wait returns 0 with n_outputs == 0 on success, 1 + tts_c_last_error() otherwise) */
int  tts_c_pool_conditional_prompt(tts_c_pool *pool, const char *prompt);
/* blocks until the task is done (timeout_ms < 0: forever).  0 = audio in data[0..n_outputs), valid until
 * tts_c_pool_release(id); 1 = finished without audio (tts_c_last_error says why); -1 = not finished */
int  tts_c_pool_wait(tts_c_pool *pool, int id, int timeout_ms, const float **data, size_t *n_outputs, int *batch_size, int *worker);
void tts_c_pool_release(tts_c_pool *pool, int id);
void tts_c_pool_stats(tts_c_pool *pool, uint64_t *tasks, uint64_t *batches, uint64_t *largest_batch, uint64_t *timed_out);
/* how the pool loaded its models: RCCL weight broadcasts performed (one per model when the workers span more than one device) and
 * workers that use their device's existing weight arena instead of uploading the file again */
void tts_c_pool_load_stats(tts_c_pool *pool, int *weight_broadcasts, int *shared_arena_loads);
void tts_c_pool_free(tts_c_pool *pool);

int tts_c_gguf_summary(const char *path, uint64_t *n_tensors, uint64_t *n_kv, uint64_t *data_offset, char *arch, int arch_cap);
int tts_c_gguf_tensor(const char *path, int index, char *name, int name_cap, int *type, int64_t ne[4], uint64_t *checksum);

/* ---- Dia host logic, callable without a device (host/dia_runner.h; reference src/models/dia/model.cpp) -------------
 * tokenize_sentence :661-705: out[max_ctx] = the sentence bytes ([S1]/[S2] -> 1/2) then zeros; returns the sentence length or -1 */
int tts_c_dia_tokenize(const char *sentence, uint32_t max_ctx, uint32_t *out);
/* check_stopping :767-785 with the default 9-head delay pattern; ids[9] may be rewritten; returns 1 when generation stops */
int tts_c_dia_check_stopping(uint32_t *ids, uint32_t eos, uint32_t pad, uint32_t max_delay, uint32_t current_position, uint32_t max_generation_size,
                             int *delay_steps);
/* adjust_output_tokens :787-808: tokens[n_steps][9] -> filtered frames [..][9]; returns the number of ids written */
int64_t tts_c_dia_adjust_output_tokens(const uint32_t *tokens, uint64_t n_ids, uint32_t audio_vocab, uint32_t max_delay, uint32_t *filtered);

/* ---- Kokoro host logic, callable without a device (host/kokoro_runner.h; reference src/tokenizer.cpp:159-177,
 * src/models/kokoro/model.cpp:1340-1388) ------------------------------------------------------------------------------
 * single_pass_tokenizer::tokenize over the given vocabulary; returns the number of ids (out may be NULL to count) */
int tts_c_single_pass_tokenize(const char *const *vocab, int n_vocab, const char *text, uint32_t *out, int cap);
/* the clause / chunk split of kokoro_runner::generate (:1420-1446) for a phoneme string: out receives, per chunk, its length
 * followed by its ids (bos ... eos); returns the number of uint32 written (or needed, when larger than cap) */
int tts_c_kokoro_chunks(const char *const *vocab, int n_vocab, const char *phonemes, uint32_t max_ctx, uint32_t space_token_id, uint32_t *out, int cap);
/* the state of the reference's noise engine (random_uniform_gen, src/util.cpp:65-71: std::default_random_engine = minstd_rand0) after k more draws:
 * kokoro_runner::generate_batch hands every clause its stretch of the one stream this way (host/kokoro_runner.h) */
uint32_t tts_c_minstd0_jump(uint32_t state, uint64_t k);
/* n draws of that engine through std::uniform_real_distribution<float>(0, 1) from `state`, drawn as `threads` parallel stretches (what the runner does for a
 * clause's source noise); returns the state afterwards */
uint32_t tts_c_minstd0_uniform(uint32_t state, uint64_t n, float *out, uint32_t threads);

/* ---- the quantize tool (examples/quantize/quantize_impl.h:5-15: quantization_params + quantize_gguf) ------------
 * Host only.  quantize_type is the ggml type number (F16 1, Q4_0 2, Q5_0 6, Q8_0 8; quantize.cpp:11-20). */
typedef struct tts_c_quantization_params {
    uint32_t n_threads;
    int      quantize_type;
    int      quantize_output_heads;
    int      quantize_text_embeddings;
    int      quantize_cross_attn_kv;
    int      convert_dac_to_f16;
    int      convert_non_quantizable_to_f16;
} tts_c_quantization_params;
int tts_c_quantize_gguf(const char *ifile, const char *ofile, const tts_c_quantization_params *params);  /* 0 / -1 */
/* the allow-list decision for one tensor name: 0 copied, 1 quantised to quantize_type, 2 converted to F16, -1 error */
int tts_c_quantize_decision(const char *arch, const char *tensor_name, int n_dims, const tts_c_quantization_params *params);
/* one call of the row quantisers (n values, n % 32 == 0 for the block types); returns bytes written or -1 */
int64_t tts_c_quantize_rows(int type, const float *src, void *dst, int64_t n_per_row, int64_t nrows, uint32_t n_threads);

#ifdef __cplusplus
}
#endif
#endif

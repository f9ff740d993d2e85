// common.h — public C++ API of the MI355X-native engine.
//
// Source-compatible with the reference's include/common.h (/root/reference/include/common.h:13-101):
// the applications (examples/cli/cli.cpp:79-95, examples/perf_battery/perf_battery.cpp:102-116,
// examples/server/server.cpp:261-298) only touch the names declared here, so they build against this
// header unchanged.  What differs is underneath: no ggml types — weights are handed to the runner as
// gguf_tensor_view records (our own GGUF reader, host/gguf.h) and the compute goes through the C ABI in
// include/tts_hip.h.
#pragma once

#include <cstddef>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <string_view>
#include <vector>

// tts_response (common.h:13-17): `data` points into a buffer owned by the runner and stays valid until
// the next generate() on that runner (dac_model.cpp:190-191); callers never free it.
struct tts_response {
    float *  data        = nullptr;
    size_t   n_outputs   = 0;
    uint32_t hidden_size = 0;  // only meaningful for encoder outputs (t5); kept for layout compatibility
};

enum tts_arch {
    PARLER_TTS_ARCH = 0,
    KOKORO_ARCH     = 1,
    DIA_ARCH        = 2,
    ORPHEUS_ARCH    = 3,
};

extern const std::map<std::string, tts_arch> SUPPORTED_ARCHITECTURES;
extern const std::map<tts_arch, std::string> ARCHITECTURE_NAMES;

// generation_configuration (common.h:45-66): same field names, defaults and constructor argument order.
struct generation_configuration {
    generation_configuration(std::string voice = "", int top_k = 50, float temperature = 1.0f,
                             float repetition_penalty = 1.0f, bool use_cross_attn = true,
                             std::string espeak_voice_id = "", int max_tokens = 0, float top_p = 1.0f,
                             bool sample = true)
        : use_cross_attn(use_cross_attn), temperature(temperature), repetition_penalty(repetition_penalty),
          top_p(top_p), top_k(top_k), max_tokens(max_tokens), voice(std::move(voice)), sample(sample),
          espeak_voice_id(std::move(espeak_voice_id)) {}

    bool        use_cross_attn;
    float       temperature;
    float       repetition_penalty;
    float       top_p;
    int         top_k;
    int         max_tokens;
    std::string voice;
    bool        sample;
    std::string espeak_voice_id;
    // ---- extensions (not in the reference) -------------------------------------------------------
    // The reference seeds std::minstd_rand from std::random_device on every sample() (sampler.cpp:47),
    // so sampled output is irreproducible; seed != 0 makes it reproducible here.
    uint64_t seed = 0;
};

struct tts_runner {
    float sampling_rate   = 44100.0f;  // common.h:70
    bool  supports_voices = false;
    virtual ~tts_runner() = default;
};

struct tts_model_loader;
struct gguf_file;

// One tensor of the GGUF file as handed to assign_weight: what the reference passes as ggml_tensor&
// (loaders.cpp:79-88) reduced to what a loader needs.
struct gguf_tensor_view {
    const char *  name;
    int           type;     // ggml type id (F32=0, F16=1, Q4_0=2, Q5_0=6, Q8_0=8)
    int           n_dims;
    int64_t       ne[4];    // ne[0] fastest
    const void *  data;     // points into the mapped file
    size_t        nbytes;
};

struct tts_generation_runner : tts_runner {
    const std::reference_wrapper<const tts_model_loader> loader;
    std::shared_ptr<gguf_file>                           buf;  // keeps the mapping alive (reference: unique_ptr<llama_mmap>)
    explicit tts_generation_runner(const tts_model_loader & loader);
    ~tts_generation_runner() override;

    virtual void assign_weight(const char * name, const gguf_tensor_view & tensor) = 0;
    virtual void prepare_post_load()                                              = 0;
    virtual std::vector<std::string_view> list_voices();
    virtual void update_conditional_prompt(const char * file_path, const char * prompt);
    virtual void generate(const char * sentence, tts_response & output, const generation_configuration & config) = 0;

    // ---- extension (not in the reference, which generates one utterance per call) -----------------------------
    // n utterances in one call; outputs[i].data stays valid until the next generate/generate_batch on this runner.
    // The default decodes them one after the other; a runner that can decode in lock-step on its device overrides
    // it (parler_runner) and reports how many utterances one call may carry.
    virtual void     generate_batch(const std::vector<std::string> & sentences, std::vector<tts_response> & outputs,
                                    const generation_configuration & config);
    virtual uint32_t batch_capacity() const { return UINT32_MAX; }
    // the device context (tts_hip_ctx*) that holds this runner's weight arena, or nullptr when the runner cannot hand its weights to
    // another runner (tts_load_options::share_with / tts_hip_broadcast_weights)
    virtual void *   device_context() const { return nullptr; }

    // ---- extension: continuous batching — a session that takes utterances in while others are still generating ---------------------
    // The reference's server hands one task at a time to a worker that owns a whole model (examples/server/server.cpp:126-158, 236-271);
    // generate_batch widened that to "form a batch from the queue, run it to the end".  A session keeps the lock-step forward full instead:
    // stream_submit() enters an utterance into a free row at the next look-in point (every 32 decode steps), stream_step() runs one such
    // interval and hands back the utterances whose check_stopping() fired inside it, already decoded to audio.  An utterance's audio is that
    // of a generate() call of its own.  A runner without the extension reports 0 capacity and the callers fall back to generate_batch.
    struct stream_result {
        size_t       ticket = 0;   // the caller's handle, as given to stream_submit
        tts_response audio;        // valid until the next stream_step / stream_end of this runner
    };
    virtual uint32_t stream_capacity() const { return 0; }            // utterances a session can hold at once (0: not supported)
    virtual void     stream_begin(const generation_configuration & config);
    virtual uint32_t stream_free() const { return 0; }                // free rows right now
    virtual uint32_t stream_live() const { return 0; }                // utterances generating or waiting for their codec pass
    virtual void     stream_submit(size_t ticket, const std::string & sentence);
    virtual void     stream_step(std::vector<stream_result> & finished);
    virtual void     stream_end();
    // any number of sentences through one session (more than batch_capacity() is fine); outputs[i].data valid until the next call on this runner
    void             generate_stream(const std::vector<std::string> & sentences, std::vector<tts_response> & outputs,
                                     const generation_configuration & config);

  protected:
    std::vector<std::vector<float>> batch_store_;  // audio of the default generate_batch
};

// loaders.h:8-20
struct tts_model_loader {
    explicit tts_model_loader(const char * arch, bool is_test = false);
    const char * const arch;
    const bool         is_test;
    virtual std::unique_ptr<tts_generation_runner> from_file(gguf_file * meta, int n_threads, bool cpu_only,
                                                             const generation_configuration & config) const = 0;

  protected:
    ~tts_model_loader() = default;
};

// Same call the reference applications make (loaders.h:19-20).  `cpu_only` was "do not use Metal" in the
// reference; here cpu_only == true selects device 0 and cpu_only == false reads TTS_HIP_DEVICE — the
// engine has no CPU path, so loading without an MI355X aborts like any other fatal error in the reference
// (TTS_ABORT, util.cpp:14-22).  "test:<arch>" loads a weightless test backend (loaders.cpp:37-44).
std::unique_ptr<tts_generation_runner> runner_from_file(const char * fname, int n_threads,
                                                        const generation_configuration & config, bool cpu_only = true);

// ---- extension: where and how the NEXT runner_from_file on the calling thread places its model.  The reference has one runner per
// process-wide backend; a host that serves several devices (device_pool) needs per-load placement, and the process environment is
// not a channel for it (setenv races with every getenv in the process).  Unset fields fall back to TTS_HIP_DEVICE /
// TTS_HIP_MAX_SEQS as read at load time, then to device 0 / one sequence.
struct tts_load_options {
    int  device       = -1;
    int  max_seqs     = 0;
    // declare_only: the tensors' shapes are declared to the device and the arena is laid out, but no bytes are uploaded: the weights
    // arrive by tts_hip_broadcast_weights (RCCL) from the runner that parsed the file.
    bool declare_only = false;
    // share_with: a loaded runner of the same model on the same device; this runner uses its weight arena (own KV cache, own stream).
    const tts_generation_runner * share_with = nullptr;
    // continuous batching (generate_stream): look-in intervals a finished utterance may wait for a codec group of 64 to fill before its partial group is
    // decoded anyway (1: latency first, the default; more: larger codec passes under steady load — a pass of 64 costs little more than a pass of 8)
    int  stream_codec_hold = 1;
};
tts_load_options & tts_thread_load_options();   // thread_local; runner_from_file reads it, the caller resets it afterwards
int      tts_load_device();                     // resolved values for the loaders
uint32_t tts_load_max_seqs();

[[noreturn]] void tts_abort(const char * file, int line, const char * fmt, ...);
#define TTS_ABORT(...) tts_abort(__FILE__, __LINE__, __VA_ARGS__)
#define TTS_ASSERT(x) do { if (!(x)) TTS_ABORT("TTS_ASSERT(%s) failed\n", #x); } while (0)

// kokoro_runner.h — Kokoro generation runner on top of the HIP shim (include/tts_hip.h).
//
// Mirrors kokoro_runner (/root/reference/src/models/kokoro/model.h:420-469, model.cpp:1277-1458): split the phoneme string
// into clauses that fit the context, tokenise one UTF-8 symbol at a time, and per clause run the duration graph, draw the
// source noise, run the generation graph and append the audio.  The ggml graphs are replaced by tts_hip_kokoro_durations /
// tts_hip_kokoro_generate.
//
// Not here: the phonemizer (reference: src/models/kokoro/phonemizer.cpp, 1.2k lines of rules / dictionary / espeak glue).
// The reference turns text into phonemes first (generate :1409-1417); this runner takes the phoneme string itself, i.e. what
// text_to_phonemes would have returned.
#pragma once
#include <memory>
#include <random>
#include <string>
#include <unordered_set>
#include <vector>

#include "../../include/tts_hip.h"
#include "common.h"

extern const struct kokoro_model_loader final : tts_model_loader {
    explicit kokoro_model_loader();
    std::unique_ptr<tts_generation_runner> from_file(gguf_file * meta, int n_threads, bool cpu_only,
                                                     const generation_configuration & config) const override;
} kokoro_loader;

// single_pass_tokenizer (src/tokenizer.h:58-74, tokenizer.cpp:159-177): at every position the SHORTEST vocabulary entry that is
// a prefix of the remaining text wins (lengths are tried from 1 upwards); nothing matches -> id 0 and one byte is skipped
struct single_pass_tokenizer {
    explicit single_pass_tokenizer(std::vector<std::string> tokens);
    size_t                   max_size = 0;
    uint32_t                 unknown_id = 0;
    std::vector<std::string> tokens;
    void tokenize(const std::string & text, std::vector<uint32_t> & token_ids) const;
};

struct kokoro_hparams {  // defaults kokoro/model.h:180-222
    uint32_t bos_token_id = 0, eos_token_id = 0, space_token_id = 16;
    uint32_t max_context_length = 512, n_attn_heads = 12, n_layers = 1, n_recurrence = 12;
    uint32_t f0_n_blocks = 3, n_duration_prediction_layers = 3, n_conv_layers = 3;
    uint32_t n_kernels = 3, n_upsamples = 2, n_decoder_blocks = 4, out_conv_padding = 3, true_n_fft = 20, stft_hop = 5, harmonic_num = 8;
    uint32_t up_sampling_factor = 600;
    float    upsample_scale = 300.0f, scale = 0.125f, sin_amp = 0.1f, noise_std = 0.003f, voice_threshold = 10.0f, sample_rate = 24000.0f;
    uint32_t up_stride[4] = {0}, up_padding[4] = {0}, noise_stride[4] = {0}, noise_padding[4] = {0};
    uint32_t res_padding[16][3] = {{0}}, res_dilation[16][3] = {{0}}, noise_res_padding[4][3] = {{0}}, noise_res_dilation[4][3] = {{0}};
    std::vector<std::string> voices;
};

// tokenize_chunks (model.cpp:1340-1388): clauses -> bos + ids + eos lists no longer than the context
std::vector<std::vector<uint32_t>> kokoro_tokenize_chunks(const kokoro_hparams & hp, const single_pass_tokenizer & tok, std::vector<std::string> clauses);

struct kokoro_runner final : tts_generation_runner {
    kokoro_runner(const kokoro_hparams & hp, single_pass_tokenizer * tok, int device, const std::string & voice);
    ~kokoro_runner() override;

    void assign_weight(const char * name, const gguf_tensor_view & tensor) override;
    void prepare_post_load() override;
    void generate(const char * phonemes, tts_response & output, const generation_configuration & config) override;
    std::vector<std::string_view> list_voices() override;

    // ---- extension: utterance-level concurrency inside one GPU --------------------------------------------------------------------
    // A Kokoro synthesis is a chain of short launches (LSTM recurrences, AdaIN, iSTFT): one context leaves most of the GPU idle.  The
    // reference's answer is N workers with a model each (examples/server/server.cpp:225-321); here generate_batch runs the clauses of
    // n utterances through up to `lanes_max` device contexts on their own streams (a host thread each) that all read ONE weight
    // arena.  Every utterance's audio is bit for bit that of the same utterances given to generate() one after the other: the source
    // noise is one minstd stream in the reference (random_uniform_gen, util.cpp:65-71), so clause i takes the stretch of the stream that
    // follows clause i - 1's (the engine is jumped ahead, x -> a^k x mod m, once the earlier clauses' durations are known).
    void     generate_batch(const std::vector<std::string> & sentences, std::vector<tts_response> & outputs, const generation_configuration & config) override;
    void *   device_context() const override { return ctx; }
    uint32_t lanes_max = 4;   // tts_load_options::max_seqs / TTS_HIP_MAX_SEQS when given, TTS_KOKORO_LANES; 1: generate_batch = generate in a loop

    std::vector<uint32_t> last_prompt_tokens;   // every clause's ids of the last generate, concatenated
    std::vector<float>    last_lengths;
    bool                  phoneme_notice_given = false;

    kokoro_hparams                         hp;
    std::unique_ptr<single_pass_tokenizer> tokenizer;
    tts_hip_ctx *                          ctx = nullptr;
    std::string                            voice;
    std::unordered_set<std::string>        uploaded_voices;
    uint32_t                               duration_hidden = 0, style_half = 0;   // from the tensor shapes (model.h:197,206 defaults 512 / 128)
    std::vector<float>                     pcm;
    std::default_random_engine             noise_engine;   // random_uniform_gen's engine (util.cpp:65-71): default seed, never reseeded

  private:
    struct tensor_decl { std::string name; int type, n_dims; int64_t ne[4]; };
    std::vector<tensor_decl>   decls;               // what assign_weight saw: a further lane declares the same tensors on the same arena
    std::vector<tts_hip_ctx *> lanes;               // lanes[0] == ctx; the others are created by the first generate_batch that needs them
    tts_hip_kokoro_desc        desc{};
    int                        device = 0;
    tts_hip_ctx *              share_ctx = nullptr; // tts_load_options::share_with: this runner reads another runner's arena
    bool                       declare_only = false;
    tts_hip_ctx * lane(size_t i);
    void run(const std::vector<uint32_t> & tokens);
};

// the state of std::minstd_rand0 after k more draws (x -> 16807^k x mod 2^31 - 1)
uint32_t minstd0_jump(uint32_t state, uint64_t k);
// out[0 .. n) = the draws an engine in `state` gives through uniform_real_distribution<float>(0, 1), drawn as `threads` parallel stretches
void minstd0_draw_uniform(uint32_t state, size_t n, float * out, unsigned threads);

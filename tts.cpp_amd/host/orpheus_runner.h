// orpheus_runner.h — Orpheus generation runner on top of the HIP shim (include/tts_hip.h).
//
// Mirrors orpheus_runner (/root/reference/src/models/orpheus/model.h:104-140, model.cpp:341-448): frame the prompt
// (fixed leading / trailing ids, optional "voice: " prefix), byte-pair tokenise, autoregress one token at a time through
// the Llama-3 decoder until the stopping token, regroup every 7 ids into the three SNAC levels, decode with SNAC.
// The ggml graphs inside decode() and snac_runner::run() are replaced by tts_hip_orpheus_* / tts_hip_snac_*.
#pragma once
#include <array>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include "../../include/tts_hip.h"
#include "common.h"
#include "sampler.h"
#include "tokenizer.h"

extern const struct orpheus_model_loader final : tts_model_loader {
    explicit orpheus_model_loader();
    std::unique_ptr<tts_generation_runner> from_file(gguf_file * meta, int n_threads, bool cpu_only,
                                                     const generation_configuration & config) const override;
} orpheus_loader;

struct orpheus_hparams {  // defaults = canopylabs/orpheus-3b (orpheus/model.h:24-37) + hubertsiuzdak/snac_24khz (snac_model.h:10-21)
    uint32_t vocab_size = 156940, n_attn_heads = 24, n_kv_attn_heads = 8, head_size = 128, hidden_size = 3072, kv_hidden_size = 1024;
    uint32_t n_layers = 28, max_context_length = 1024, max_generation_size = 2100, stopping_token_id = 128258;
    uint32_t eos_token_id = 128001, bos_token_id = 128000;
    uint32_t audio_heads = 3;
    uint32_t heads[7] = {0, 1, 2, 2, 1, 2, 2};
    // "undocumented constants" the reference hard-codes (model.cpp:8-9, 371).  Extension: a GGUF may override them with
    // orpheus.{prepended_tokens,appended_tokens,audio_token_offset,audio_token_stride} (synthetic test models do).
    std::vector<uint32_t> prepended_tokens{128259, 128000};
    std::vector<uint32_t> appended_tokens{128009, 128260, 128261, 128257};
    uint32_t audio_token_offset = 128266, audio_token_stride = 4096;
    // SNAC
    uint32_t snac_layers = 4, snac_heads = 3, snac_up = 512, snac_max_generation = 2580;
    uint32_t snac_stride[TTS_HIP_MAX_DAC_BLOCKS] = {0}, snac_padding[TTS_HIP_MAX_DAC_BLOCKS] = {0}, snac_groups[TTS_HIP_MAX_DAC_BLOCKS] = {0};
    uint32_t snac_repeats[3] = {4, 2, 1};
};

struct orpheus_runner final : tts_generation_runner {
    orpheus_runner(const orpheus_hparams & hp, bpe_tokenizer * tok, int device);
    ~orpheus_runner() override;

    void assign_weight(const char * name, const gguf_tensor_view & tensor) override;
    void prepare_post_load() override;
    void generate(const char * sentence, tts_response & output, const generation_configuration & config) override;
    // extension: lock-step utterances (tts_hip_orpheus_generate_batch): every utterance's audio is that of a generate() call of its own, made in order
    void     generate_batch(const std::vector<std::string> & sentences, std::vector<tts_response> & outputs, const generation_configuration & config) override;
    uint32_t batch_capacity() const override { return max_seqs; }
    std::vector<std::string_view> list_voices() override;

    void decode_audio(const std::vector<uint32_t> & output_tokens, std::vector<float> & audio);   // prepare_output_tokens + SNAC
    // pieces exposed for tests
    std::vector<uint32_t>              batch_from_sentence(const std::string & sentence, const std::string & voice) const;  // model.cpp:341-356
    std::vector<std::vector<uint32_t>> prepare_output_tokens(const std::vector<uint32_t> & output_tokens) const;           // :358-376
    std::vector<uint32_t> last_prompt_tokens, last_output_tokens;
    std::vector<std::vector<uint32_t>> last_batch_tokens;   // generate_batch: the ids of every utterance

    orpheus_hparams                hp;
    std::unique_ptr<bpe_tokenizer> tokenizer;
    sampler                        smp;
    uint32_t                       max_seqs = 1;     // cache slots of the decoder context (tts_load_options::max_seqs, at most 64)
    tts_hip_ctx *                  lm = nullptr;     // Llama-3 decoder context
    tts_hip_ctx *                  snac = nullptr;   // SNAC codec context
    std::vector<float>             pcm, logits;
    std::default_random_engine     noise_engine;     // random_normal_gen's engine (util.cpp:74-80): default seed, never reseeded
    std::normal_distribution<float> noise_dist{0.0f, 1.0f};
};

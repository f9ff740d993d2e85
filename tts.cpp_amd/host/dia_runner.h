// dia_runner.h — Dia generation runner on top of the HIP shim (include/tts_hip.h).
//
// Mirrors dia_runner (/root/reference/src/models/dia/model.h:187-216, model.cpp:661-858): byte tokenisation with the
// [S1] / [S2] speaker tags, one encoder pass over the padded text and an all-zero "unconditional" twin, an autoregressive
// loop over nine delayed codebook heads with classifier-free guidance inside every step, the end-of-sequence countdown of
// check_stopping, un-delay, DAC.  The ggml graphs inside decode() and dac_runner::run() are replaced by tts_hip_dia_* and
// tts_hip_dac_decode; tokenisation, sampling and the stopping logic stay on the host as in the reference.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../../include/tts_hip.h"
#include "common.h"
#include "sampler.h"

extern const struct dia_model_loader final : tts_model_loader {
    explicit dia_model_loader();
    std::unique_ptr<tts_generation_runner> from_file(gguf_file * meta, int n_threads, bool cpu_only,
                                                     const generation_configuration & config) const override;
} dia_loader;

struct dia_hparams {  // defaults = nari-labs/Dia-1.6B (dia/model.h:64-84)
    uint32_t n_output_heads = 9, n_encoder_layers = 12, n_decoder_layers = 18, encoder_hidden_size = 1024, decoder_hidden_size = 2048;
    uint32_t encoder_attn_heads = 16, decoder_attn_heads = 16, decoder_query_heads = 4, head_size = 128;
    uint32_t eos_token_id = 1024, pad_token_id = 1025, bos_token_id = 1026, output_vocab_size = 1028, audio_vocab_size = 1024;
    uint32_t max_generation_size = 3072, max_encoder_context_length = 1024, max_delay = 15;
    float    cfg_scale = 3.0f;
    std::vector<uint32_t> delay_pattern{0, 8, 9, 10, 11, 12, 13, 14, 15};   // model.h:84: not a GGUF key
    uint32_t dac_n_layers = 4;
    uint32_t dac_stride[TTS_HIP_MAX_DAC_BLOCKS] = {0}, dac_padding[TTS_HIP_MAX_DAC_BLOCKS] = {0};
    uint32_t up_sampling_factor = 512;
};

// host logic of the runner as free functions of the hyper-parameters (no device behind them; the runner and the CPU tests call these)
uint32_t dia_tokenize_sentence(const dia_hparams & hp, std::string sentence, std::vector<uint32_t> & tokens);                               // model.cpp:661-705
bool     dia_check_stopping(const dia_hparams & hp, std::vector<uint32_t> & audio_tokens, uint32_t current_position, uint32_t max_generation_size,
                            int & delay_steps);                                                                                             // :767-785
void     dia_adjust_output_tokens(const dia_hparams & hp, const std::vector<uint32_t> & output_tokens, std::vector<uint32_t> & filtered);   // :787-808

struct dia_runner final : tts_generation_runner {
    dia_runner(const dia_hparams & hp, int device);
    ~dia_runner() override;

    void assign_weight(const char * name, const gguf_tensor_view & tensor) override;
    void prepare_post_load() override;
    void generate(const char * sentence, tts_response & output, const generation_configuration & config) override;
    // extension (BASELINE config 3: 4 utterances per GPU): n utterances in lock-step, each the reference's batch of two guidance streams
    // (model.cpp:330-341); per-utterance sampler, check_stopping countdown and un-delay; one batched DAC pass.  Needs max_seqs >= n at
    // load time (tts_load_options / TTS_HIP_MAX_SEQS).  Greedy results equal n separate generate() calls.
    void generate_batch(const std::vector<std::string> & sentences, std::vector<tts_response> & outputs,
                        const generation_configuration & config) override;
    uint32_t batch_capacity() const override { return max_seqs; }
    uint32_t max_seqs = 1;
    std::vector<std::vector<uint32_t>> last_batch_tokens;

    std::vector<uint32_t> last_prompt_tokens, last_output_tokens;

    dia_hparams        hp;
    sampler            smp;
    tts_hip_ctx *      lm = nullptr;    // encoder + decoder context
    tts_hip_ctx *      dac = nullptr;   // codec context
    std::vector<float> pcm, logits;
};

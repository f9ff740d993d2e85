// parler_runner.h — Parler-TTS generation runner on top of the HIP shim (include/tts_hip.h).
//
// Mirrors parler_tts_runner (/root/reference/src/models/parler/model.h:187-225, model.cpp:473-498,
// 704-858): tokenise -> text-prompt decode -> AR loop with the delay pattern, host sampling and EOS
// tracking -> un-delay -> DAC.  The ggml graph build/compute inside decode() and dac_runner::run() is
// replaced by calls into the C ABI; everything the reference keeps on the host stays on the host.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../../include/tts_hip.h"
#include "common.h"
#include "sampler.h"
#include "tokenizer.h"

extern const struct parler_model_loader final : tts_model_loader {
    explicit parler_model_loader();
    std::unique_ptr<tts_generation_runner> from_file(gguf_file * meta, int n_threads, bool cpu_only,
                                                     const generation_configuration & config) const override;
} parler_loader;

struct parler_hparams {  // defaults = Parler TTS Mini v1 (model.h:66-83)
    uint32_t n_output_heads = 9, n_encode_length = 0, hidden_size = 1024, max_ctx_length = 4096, n_attn_heads = 16;
    uint32_t output_vocab_size = 1088, eos_token_id = 1024, audio_vocab_size = 1024, max_generation_size = 2580;
    uint32_t n_layers = 24, bos_token_id = 1025;
    uint32_t dac_n_layers = 4;  // dac_model.h:33
    uint32_t dac_stride[TTS_HIP_MAX_DAC_BLOCKS] = {0}, dac_padding[TTS_HIP_MAX_DAC_BLOCKS] = {0};
    uint32_t up_sampling_factor = 512;
};

struct parler_runner final : tts_generation_runner {
    parler_runner(const parler_hparams & hp, unigram_tokenizer * tok, int device, bool use_cross_attn);
    ~parler_runner() override;

    void assign_weight(const char * name, const gguf_tensor_view & tensor) override;
    void prepare_post_load() override;
    void generate(const char * sentence, tts_response & output, const generation_configuration & config) override;
    void update_conditional_prompt(const char * file_path, const char * prompt) override;

    // Extension (the reference generates one utterance per call; its only multi-utterance construct is the
    // server's pool of full replicas, examples/server/server.cpp:885-895): n utterances decoded in lock-step on
    // this runner's device — one pass over the weights per step for all of them, one batched DAC pass.
    // outputs[i].data points into a runner-owned buffer valid until the next generate/generate_batch.
    // Needs max_seqs >= n (TTS_HIP_MAX_SEQS at load time).  Results equal n separate generate() calls.
    void generate_batch(const std::vector<std::string> & sentences, std::vector<tts_response> & outputs,
                        const generation_configuration & config) override;
    uint32_t batch_capacity() const override { return max_seqs; }
    // continuous batching (common.h): max_seqs - 1 rows (one cache slot pads the lock-step forward), tts_hip_parler_stream_* underneath
    uint32_t stream_capacity() const override { return max_seqs > 1 ? max_seqs - 1 : 0; }
    void     stream_begin(const generation_configuration & config) override;
    uint32_t stream_free() const override { return (uint32_t) st_free.size(); }
    uint32_t stream_live() const override { return st_live + (uint32_t) st_codec.size(); }
    void     stream_submit(size_t ticket, const std::string & sentence) override;
    void     stream_step(std::vector<stream_result> & finished) override;
    void     stream_end() override;
    void *   device_context() const override { return ctx; }
    bool          declare_only = false;   // tts_load_options at load time: no weight bytes uploaded by this runner
    tts_hip_ctx * share_ctx = nullptr;    // ... and whose arena it uses instead (same device)
    uint32_t max_seqs = 1;
    std::vector<std::vector<uint32_t>> last_batch_tokens;  // per utterance, still delayed

    // pieces exposed for tests
    void                 adjust_output_tokens(const std::vector<uint32_t> & output_tokens, std::vector<uint32_t> & filtered) const;
    std::vector<uint32_t> last_output_tokens;  // pctx->output_tokens of the last generate (still delayed)
    std::vector<uint32_t> last_prompt_tokens;

    parler_hparams                     hp;
    std::unique_ptr<unigram_tokenizer> tokenizer;
    sampler                            smp;
    tts_hip_ctx *                      ctx = nullptr;
    bool                               use_cross_attn;
    int                                device_id = 0;
    std::vector<uint32_t>              last_conditional_tokens;  // ids the voice prompt was encoded from (tests)
    std::vector<float>                 pcm;     // runner-owned output buffer (dctx->buf_output)
    std::vector<float>                 logits;

  private:
    // session state of the continuous batching
    struct pending { size_t ticket; uint32_t slot; std::vector<uint32_t> prompt; };
    struct decoded { size_t ticket; std::vector<uint32_t> frames; };   // un-delayed codes waiting for a codec pass
    bool                        st_on = false;
    generation_configuration    st_cfg{};
    uint32_t                    st_live = 0, st_max_steps = 0;
    uint32_t                    st_codec_hold = 1;   // tts_load_options::stream_codec_hold at load time
    uint32_t                    st_codec_held = 0;   // stream_step calls the oldest undecoded finished utterance has waited for a codec group to fill
    std::vector<uint32_t>       st_free;            // free cache slots
    std::vector<size_t>         st_ticket;          // slot -> ticket
    std::vector<uint32_t>       st_start;           // slot -> prompt length
    std::vector<pending>        st_wait;            // submitted, not yet admitted (admitted as one side batch by the next stream_step)
    std::vector<decoded>        st_codec;
    std::vector<std::vector<float>> st_pcm;         // audio handed out by the last stream_step
};

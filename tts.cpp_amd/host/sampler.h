// sampler.h — per-head logits -> token ids on the host, the same arithmetic as
// /root/reference/src/sampler.cpp (greedy `max`, softmax with temperature / repetition penalty, top-k by
// full sort, top-p trimming, inverse-CDF draw).  Extension: a seed (the reference seeds from
// std::random_device on every call, sampler.cpp:47).
#pragma once
#include <cstdint>
#include <random>
#include <vector>

struct sampler {
    uint32_t n_output_heads = 9;
    uint32_t eos_token_id = 1024;
    uint32_t vocab_size = 1088;
    float    temperature = 1.0f;
    uint32_t top_k = 0;
    float    top_p = 1.0f;
    float    repetition_penalty = 1.0f;
    bool     do_sample = true;
    uint64_t seed = 0;       // 0 = std::random_device per call, like the reference
    uint64_t n_calls = 0;

    std::vector<int32_t>  last_token_ids;
    std::vector<uint32_t> repetition_counts;

    void reset();
    void sample(float * logits, std::vector<uint32_t> & output_tokens);
    // the n_output_heads U[0,1) draws of the next sample() call (std::minstd_rand + uniform_real_distribution as
    // sampler.cpp:47-50); advances n_calls exactly like sample() does, so a caller can draw ahead for a
    // device-resident loop and get the token stream sample() would have produced
    void draw_uniforms(float * u);
    // test hook: same as sample() but the per-head uniform draws are supplied
    void sample_with_uniforms(float * logits, const float * uniforms, std::vector<uint32_t> & output_tokens);
    void max(const float * logits, std::vector<uint32_t> & out) const;

  private:
    float penalised(float v, uint32_t head) const;
    void  softmax(float * logits, const std::vector<std::vector<size_t>> & picks, const std::vector<uint32_t> & max_idx) const;
    std::vector<std::vector<size_t>> topk(const float * logits, bool performed_softmax) const;
    void topp(const float * logits, std::vector<std::vector<size_t>> & picks, std::vector<float> & max_head_probs) const;
};

// ref_sampler_wrap.cpp — extern "C" shim around the REAL reference sampler.
//
// TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it #includes the reference
// header from where it lies (/root/reference/src/sampler.h) and is linked against
// /root/reference/src/sampler.cpp compiled in place by oracle/Makefile.  The output goes to
// oracle/_ref/libref_sampler.so (git-ignored, travels to the GPU box as a built artefact).
// sampler.cpp is the only translation unit of the reference's hot path that builds without the
// absent ggml submodule (it includes nothing but <vector>/<random>/<numeric>/<algorithm>).
#include "sampler.h"

#include <cstring>

extern "C" {

struct ref_sampler_cfg {
    uint32_t n_output_heads, vocab_size, top_k;
    float    temperature, top_p, repetition_penalty;
    int      do_sample;
};

static void apply(sampler & s, const ref_sampler_cfg * c) {
    s.n_output_heads     = c->n_output_heads;
    s.vocab_size         = c->vocab_size;
    s.top_k              = c->top_k;
    s.temperature        = c->temperature;
    s.top_p              = c->top_p;
    s.repetition_penalty = c->repetition_penalty;
    s.do_sample          = c->do_sample != 0;
}

// sampler::max (greedy), with optional repetition state injected
void ref_sampler_max(const ref_sampler_cfg * c, const int32_t * last_ids, const uint32_t * counts,
                     float * logits, uint32_t * out) {
    sampler s;
    apply(s, c);
    if (last_ids) {
        s.last_token_ids.assign(last_ids, last_ids + c->n_output_heads);
        s.repetition_counts.assign(counts, counts + c->n_output_heads);
    }
    std::vector<uint32_t> o;
    s.max(logits, o);
    std::memcpy(out, o.data(), o.size() * sizeof(uint32_t));
}

// sampler::sample with do_sample=false (what generate() runs under greedy)
void ref_sampler_sample_greedy(const ref_sampler_cfg * c, float * logits, uint32_t * out) {
    sampler s;
    apply(s, c);
    s.do_sample = false;
    s.reset();
    std::vector<uint32_t> o;
    s.sample(logits, o);
    std::memcpy(out, o.data(), o.size() * sizeof(uint32_t));
}

// The deterministic part of sampler::sample before the random draw:
// max -> [softmax] -> [topk] -> [softmax] -> [topp].  Mutates logits in place exactly as
// sample() does; returns per-head pick lists (flattened, n_picks per head) and max_head_probs.
// Mirrors the call sequence of sampler.cpp:18-41 using the reference's own member functions.
int ref_sampler_distribution(const ref_sampler_cfg * c, const int32_t * last_ids, const uint32_t * counts,
                             float * logits, uint32_t * picks_out, uint32_t * n_picks_out,
                             float * max_head_probs_out) {
    sampler s;
    apply(s, c);
    s.reset();
    if (last_ids && c->repetition_penalty != 1.0f) {
        s.last_token_ids.assign(last_ids, last_ids + c->n_output_heads);
        s.repetition_counts.assign(counts, counts + c->n_output_heads);
    }
    std::vector<uint32_t> max_vals;
    std::vector<float>    max_head_probs;
    s.max(logits, max_vals);
    std::vector<std::vector<size_t>> picks;
    bool performed_softmax = false;
    if (s.top_p < 1.0) {
        s.softmax(logits, picks, max_vals);
        performed_softmax = true;
    }
    if (s.top_k > 0 && s.top_k < s.vocab_size) {
        picks = s.topk(logits, performed_softmax);
    }
    if (s.top_p >= 1.0) {
        s.softmax(logits, picks, max_vals);
        performed_softmax = true;
    }
    if (s.top_p < 1.0) {
        s.topp(logits, picks, max_head_probs);
    }
    for (uint32_t i = 0; i < c->n_output_heads; i++) {
        uint32_t n = picks.empty() ? 0 : (uint32_t) picks[i].size();
        n_picks_out[i] = n;
        for (uint32_t j = 0; j < n; j++) picks_out[i * c->vocab_size + j] = (uint32_t) picks[i][j];
        max_head_probs_out[i] = max_head_probs.empty() ? 1.0f : max_head_probs[i];
    }
    return 0;
}

}  // extern "C"

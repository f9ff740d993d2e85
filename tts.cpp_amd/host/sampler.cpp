#include "sampler.h"

#include <algorithm>
#include <cmath>
#include <numeric>

void sampler::reset() {  // sampler.cpp:71-80
    if (repetition_penalty != 1.0f) {
        last_token_ids.assign(n_output_heads, -1);
        repetition_counts.assign(n_output_heads, 0);
    }
}

float sampler::penalised(float v, uint32_t head) const {
    return (float) ((double) v / std::pow((double) repetition_penalty, (double) repetition_counts[head]));
}

void sampler::max(const float * logits, std::vector<uint32_t> & out) const {  // sampler.cpp:185-204
    const bool rep = repetition_penalty != 1.0f && !last_token_ids.empty();
    for (uint32_t h = 0; h < n_output_heads; h++) {
        float    best = -INFINITY;
        uint32_t id = 0;
        const float * row = logits + (size_t) h * vocab_size;
        for (uint32_t i = 0; i < vocab_size; i++) {
            float v = row[i];
            if (rep && last_token_ids[h] == (int32_t) i) v = penalised(v, h);
            if (v > best) { best = v; id = i; }  // first maximum wins
        }
        out.push_back(id);
    }
}

void sampler::softmax(float * logits, const std::vector<std::vector<size_t>> & picks, const std::vector<uint32_t> & max_idx) const {
    const bool nucleus = !picks.empty(), rep = repetition_penalty != 1.0f, temp = temperature != 1.0f;
    for (uint32_t h = 0; h < n_output_heads; h++) {
        float * row = logits + (size_t) h * vocab_size;
        float   top = row[max_idx[h]];
        if (rep && last_token_ids[h] == (int32_t) max_idx[h]) top = penalised(top, h);
        if (temp) top /= temperature;
        const size_t n = nucleus ? picks[h].size() : vocab_size;
        float        total = 0.0f;
        for (size_t j = 0; j < n; j++) {
            const size_t i = nucleus ? picks[h][j] : j;
            float        v = row[i];
            if (rep && last_token_ids[h] == (int32_t) i) v = penalised(v, h);
            if (temp) v /= temperature;
            v = expf(v - top);
            total += v;
            row[i] = v;
        }
        for (size_t j = 0; j < n; j++) {
            const size_t i = nucleus ? picks[h][j] : j;
            row[i] = row[i] / total;
        }
    }
}

std::vector<std::vector<size_t>> sampler::topk(const float * logits, bool performed_softmax) const {
    const bool rep = repetition_penalty != 1.0f;
    std::vector<std::vector<size_t>> out;
    for (uint32_t h = 0; h < n_output_heads; h++) {
        std::vector<size_t> order(vocab_size);
        std::iota(order.begin(), order.end(), 0);
        const float * row = logits + (size_t) h * vocab_size;
        if (top_k <= vocab_size) {
            // std::sort with the reference's comparator semantics (sampler.cpp:167-179), so equal keys land
            // wherever libstdc++'s introsort puts them in the reference too
            std::sort(order.begin(), order.end(), [&](size_t a, size_t b) {
                float va = row[a], vb = row[b];
                if (!performed_softmax) {
                    if (rep && last_token_ids[h] == (int32_t) a) va = penalised(va, h);
                    else if (rep && last_token_ids[h] == (int32_t) b) vb = penalised(vb, h);
                }
                return va > vb;
            });
            order.resize(top_k);
        }
        out.push_back(std::move(order));
    }
    return out;
}

void sampler::topp(const float * logits, std::vector<std::vector<size_t>> & picks, std::vector<float> & max_head_probs) const {
    if (picks.empty()) {
        for (uint32_t h = 0; h < n_output_heads; h++) {
            std::vector<size_t> order(vocab_size);
            std::iota(order.begin(), order.end(), 0);
            const float * row = logits + (size_t) h * vocab_size;
            std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return row[a] > row[b]; });
            picks.push_back(std::move(order));
        }
    }
    for (uint32_t h = 0; h < n_output_heads; h++) {
        const float * row = logits + (size_t) h * vocab_size;
        float mass = 0.0f;
        int   keep = -1;
        for (size_t j = 0; j < picks[h].size(); j++) {
            mass += row[picks[h][j]];
            if (mass >= top_p) { keep = (int) j + 1; break; }
        }
        max_head_probs.push_back(std::min(mass, top_p));
        if (keep > 0) picks[h].resize((size_t) keep);
    }
}

void sampler::sample_with_uniforms(float * logits, const float * uniforms, std::vector<uint32_t> & output_tokens) {
    if (!do_sample) { max(logits, output_tokens); return; }
    std::vector<uint32_t> max_idx;
    std::vector<float>    max_head_probs;
    max(logits, max_idx);
    std::vector<std::vector<size_t>> picks;
    bool nucleus = false, did_softmax = false;
    if (top_p < 1.0f) { softmax(logits, picks, max_idx); did_softmax = true; }
    if (top_k > 0 && top_k < vocab_size) { picks = topk(logits, did_softmax); nucleus = true; }
    if (top_p >= 1.0f) { softmax(logits, picks, max_idx); did_softmax = true; }
    if (top_p < 1.0f) { topp(logits, picks, max_head_probs); nucleus = true; }
    const bool rep = repetition_penalty != 1.0f;
    if (rep && (last_token_ids.empty() || repetition_counts.empty())) reset();
    for (uint32_t h = 0; h < n_output_heads; h++) {
        const float * row = logits + (size_t) h * vocab_size;
        const float   target = top_p < 1.0f ? uniforms[h] * max_head_probs[h] : uniforms[h];
        const size_t  n = nucleus ? picks[h].size() : vocab_size;
        float         cum = 0.0f;
        size_t        chosen = n ? (nucleus ? picks[h][n - 1] : n - 1) : 0;
        for (size_t j = 0; j < n; j++) {
            const size_t i = nucleus ? picks[h][j] : j;
            cum += row[i];
            // the reference's third clause `j >= picks[i].size() - 1` indexes an empty vector when neither
            // top-k nor top-p is active (undefined behaviour, sampler.cpp:57); here it means "last candidate"
            if (target <= cum || j + 1 >= n) { chosen = i; break; }
        }
        if (rep) {
            if (last_token_ids[h] != (int32_t) chosen) repetition_counts[h] = 0;
            last_token_ids[h] = (int32_t) chosen;
            repetition_counts[h] += 1;
        }
        output_tokens.push_back((uint32_t) chosen);
    }
}

void sampler::draw_uniforms(float * u) {
    std::minstd_rand gen(seed ? (std::minstd_rand::result_type) (seed * 0x9E3779B97F4A7C15ull + ++n_calls) : std::random_device{}());
    std::uniform_real_distribution<float> dist(0.0f, 1.0f);
    for (uint32_t h = 0; h < n_output_heads; h++) u[h] = dist(gen);
}

void sampler::sample(float * logits, std::vector<uint32_t> & output_tokens) {
    if (!do_sample) { max(logits, output_tokens); return; }
    std::vector<float> u(n_output_heads);
    draw_uniforms(u.data());
    sample_with_uniforms(logits, u.data(), output_tokens);
}

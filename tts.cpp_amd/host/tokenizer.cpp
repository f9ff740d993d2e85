#include "tokenizer.h"

#include <algorithm>
#include <cmath>

#include "gguf.h"

unigram_tokenizer::unigram_tokenizer(const std::vector<std::string> & vocab, std::vector<float> sc, uint32_t unk)
    : scores(std::move(sc)), unk_token(unk) {
    unk_token_score = unk < scores.size() ? scores[unk] : 0.0f;
    trie.emplace_back();
    for (uint32_t id = 0; id < vocab.size(); id++) {
        int32_t cur = 0;
        for (char ch : vocab[id]) {
            auto it = trie[cur].next.find(ch);
            if (it == trie[cur].next.end()) {
                trie.emplace_back();
                const int32_t nn = (int32_t) trie.size() - 1;
                trie[cur].next.emplace(ch, nn);
                cur = nn;
            } else {
                cur = it->second;
            }
        }
        // duplicates: the reference fills an unordered_map<string, id> (later ids overwrite earlier ones,
        // tokenizer.cpp:136-139) and then walks it; last writer wins here too
        trie[cur].token = (int32_t) id;
    }
}

static size_t utf8_len(char c) {
    static const size_t lookup[16] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 3, 4};
    return lookup[static_cast<uint8_t>(c) >> 4];
}

static bool is_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r'; }

void unigram_tokenizer::tokenize(const std::string & text, std::vector<uint32_t> & tokens) const {
    std::string norm;
    if (dedupe_spaces) {  // " " + regex_replace(text, "\\s{2,}", " ")
        norm.reserve(text.size() + 1);
        norm.push_back(' ');
        for (size_t i = 0; i < text.size();) {
            if (is_space(text[i])) {
                size_t j = i;
                while (j < text.size() && is_space(text[j])) j++;
                if (j - i >= 2) norm.push_back(' ');
                else norm.push_back(text[i]);
                i = j;
            } else {
                norm.push_back(text[i++]);
            }
        }
    } else {
        norm = text;
    }
    const size_t n = norm.size();
    struct best { uint32_t token; size_t from; float score; };
    std::vector<best> dp(n + 1, best{unk_token, 0, -INFINITY});
    dp[0] = best{unk_token, 0, 0.0f};

    for (size_t off = 0; off < n;) {
        const size_t step = std::min(utf8_len(norm[off]), n - off);
        const float  base = dp[off].score;
        bool         covered = false;  // some vocabulary entry spans exactly this code point
        int32_t      cur = 0;
        for (size_t end = off; end < n; end++) {
            auto it = trie[cur].next.find(norm[end]);
            if (it == trie[cur].next.end()) break;
            cur = it->second;
            if (trie[cur].token >= 0) {
                const size_t len = end + 1 - off;
                if (len == step) covered = true;
                const float s = base + scores[(size_t) trie[cur].token];
                if (s > dp[off + len].score) dp[off + len] = best{(uint32_t) trie[cur].token, off, s};
            }
        }
        if (!covered) {
            const float s = base + unk_token_score;
            if (s > dp[off + step].score) dp[off + step] = best{unk_token, off, s};
        }
        off += step;
    }

    // walk back from the end; consecutive unknowns collapse into one
    bool   prev_unknown = false;
    size_t at = n;
    for (;;) {
        const best & b = dp[at];
        const bool   unknown = b.token == unk_token;
        if (!(prev_unknown && unknown)) tokens.push_back(b.token);
        if (b.from == 0) break;
        prev_unknown = unknown;
        at = b.from;
    }
    std::reverse(tokens.begin(), tokens.end());
}

unigram_tokenizer * unigram_tokenizer_from_gguf(const gguf_file & meta) {
    const gguf_value * toks = meta.get("tokenizer.ggml.tokens");
    const gguf_value * sc = meta.get("tokenizer.ggml.scores");
    const gguf_value * unk = meta.get("tokenizer.ggml.unknown_token_id");
    if (!toks || !sc || !unk) TTS_ABORT("GGUF file lacks tokenizer.ggml.{tokens,scores,unknown_token_id}\n");
    if (toks->arr_s.size() != sc->arr_n) TTS_ABORT("tokenizer vocabulary and score arrays differ in length\n");
    std::vector<float> scores((const float *) sc->arr_data, (const float *) sc->arr_data + sc->arr_n);
    auto * t = new unigram_tokenizer(toks->arr_s, std::move(scores), (uint32_t) unk->u);
    if (auto e = meta.get("tokenizer.ggml.eos_token_id")) t->eos_token = (uint32_t) e->u;
    return t;
}

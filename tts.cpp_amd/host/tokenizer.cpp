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

// ---- byte-pair tokenizer (Orpheus) ---------------------------------------------------------------------------
void bpe_tokenizer::tokenize(const std::string & text, std::vector<uint32_t> & token_ids) const {
    bool   space_prior = false;
    size_t i = 0;
    while (i < text.size()) {
        if (text[i] == ' ') { space_prior = true; i++; continue; }
        size_t j = text.find(' ', i);
        if (j == std::string::npos) j = text.size();
        const std::string chunk = text.substr(i, j - i);
        piece(space_prior ? "\xC4\xA0" + chunk : chunk, token_ids);   // "Ġ"
        i = j;
    }
}

void bpe_tokenizer::piece(const std::string & chunk, std::vector<uint32_t> & token_ids) const {
    auto whole = tokens_to_ids.find(chunk);
    if (whole != tokens_to_ids.end()) { token_ids.push_back(whole->second); return; }
    // symbols = UTF-8 characters; links make merged-away symbols skippable
    struct sym { size_t pos, size; int prev, next; bool live; };
    std::vector<sym> syms;
    for (size_t i = 0; i < chunk.size();) {
        size_t n = 1;
        while (i + n < chunk.size() && ((unsigned char) chunk[i + n] & 0xC0) == 0x80) n++;
        syms.push_back({i, n, (int) syms.size() - 1, -1, true});
        i += n;
    }
    for (size_t k = 0; k + 1 < syms.size(); k++) syms[k].next = (int) k + 1;
    auto str = [&](int k) { return chunk.substr(syms[(size_t) k].pos, syms[(size_t) k].size); };
    // lowest rank first; equal ranks: the pair whose left symbol starts first (bpe_merge_comp, tokenizer.cpp:231-233)
    for (;;) {
        int best = -1, best_rank = 0;
        for (int k = 0; k >= 0 && k < (int) syms.size(); k = syms[(size_t) k].next) {
            const int nx = syms[(size_t) k].next;
            if (nx < 0) break;
            auto r = ranks.find(str(k) + " " + str(nx));
            if (r != ranks.end() && (best < 0 || r->second < best_rank)) { best = k; best_rank = r->second; }
        }
        if (best < 0) break;
        sym & a = syms[(size_t) best];
        sym & b = syms[(size_t) a.next];
        a.size += b.size;
        b.live = false;
        a.next = b.next;
        if (a.next >= 0) syms[(size_t) a.next].prev = best;
    }
    for (int k = 0; k >= 0 && k < (int) syms.size(); k = syms[(size_t) k].next) {
        auto it = tokens_to_ids.find(str(k));
        token_ids.push_back(it == tokens_to_ids.end() ? 0u : it->second);
        if (syms[(size_t) k].next < 0) break;
    }
}

bpe_tokenizer * bpe_tokenizer_from_gguf(const gguf_file & meta) {
    const gguf_value * toks = meta.get("tokenizer.ggml.tokens");
    const gguf_value * merges = meta.get("tokenizer.ggml.merges");
    const gguf_value * eos = meta.get("tokenizer.ggml.eos_token_id");
    const gguf_value * bos = meta.get("tokenizer.ggml.bos_token_id");
    if (!toks) TTS_ABORT("The 'tokenizer.ggml.tokens' key must be set in order to support BPE tokenization.\n");
    if (!merges) TTS_ABORT("The 'tokenizer.ggml.merges' key must be set in order to support BPE tokenization.\n");
    if (!eos) TTS_ABORT("The 'tokenizer.ggml.eos_token_id' key must be set in order to support BPE tokenization.\n");
    if (!bos) TTS_ABORT("The 'tokenizer.ggml.bos_token_id' key must be set in order to support BPE tokenization.\n");
    auto * t = new bpe_tokenizer;
    t->bos_token_id = (uint32_t) bos->u;
    t->eos_token_id = (uint32_t) eos->u;
    for (size_t i = 0; i < toks->arr_s.size(); i++) t->tokens_to_ids[toks->arr_s[i]] = (uint32_t) i;
    for (size_t i = 0; i < merges->arr_s.size(); i++) {
        const std::string & m = merges->arr_s[i];
        const size_t sp = m.find(' ');
        if (sp == std::string::npos || m.find(' ', sp + 1) != std::string::npos)
            TTS_ABORT("Invalid pair, '%s', found in BPE merges, 'tokenizer.ggml.merges', at index %zu.\n", m.c_str(), i);
        t->ranks[m] = (int) i;
    }
    return t;
}

// tokenizer.h — unigram (SentencePiece-style) tokenizer used for the Parler text prompt.
// Behaviour follows /root/reference/src/tokenizer.cpp:49-127 (Viterbi over a character trie,
// " " + text with runs of >=2 whitespace collapsed, unknown-run joining); vocabulary and scores come
// from the GGUF keys tokenizer.ggml.{tokens,scores,unknown_token_id,eos_token_id} (tokenizer.cpp:130-157).
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

struct gguf_file;

struct unigram_tokenizer {
    std::vector<float> scores;
    uint32_t           unk_token = 0;
    float              unk_token_score = 0.0f;
    uint32_t           eos_token = 1;
    bool               dedupe_spaces = true;

    unigram_tokenizer(const std::vector<std::string> & vocab, std::vector<float> scores, uint32_t unk_token);
    void tokenize(const std::string & text, std::vector<uint32_t> & tokens) const;

  private:
    struct node {
        int32_t                           token = -1;
        std::unordered_map<char, int32_t> next;
    };
    std::vector<node> trie;  // node 0 is the root
};

unigram_tokenizer * unigram_tokenizer_from_gguf(const gguf_file & meta);

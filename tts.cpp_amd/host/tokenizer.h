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

// Byte-pair tokenizer used by Orpheus (reference: bpe_tokenizer, /root/reference/src/tokenizer.h:141-150,
// tokenizer.cpp:209-296).  Behaviour: the text is cut at single spaces; every piece after the first space is
// prefixed with "Ġ" (the flag is never cleared, tokenizer.cpp:265-275); a piece that is a vocabulary entry is emitted
// whole; otherwise it is split into UTF-8 characters and adjacent pairs are merged lowest rank first, ties by
// position (tokenizer.cpp:231-263); a symbol that is not in the vocabulary maps to id 0 (operator[] on the map, :286).
struct bpe_tokenizer {
    std::unordered_map<std::string, uint32_t> tokens_to_ids;
    std::unordered_map<std::string, int>      ranks;   // "left right" -> merge rank
    uint32_t bos_token_id = 0, eos_token_id = 0;

    void tokenize(const std::string & text, std::vector<uint32_t> & token_ids) const;

  private:
    void piece(const std::string & chunk, std::vector<uint32_t> & token_ids) const;
};

// tokenizer.ggml.{tokens,merges,bos_token_id,eos_token_id} (tokenizer.cpp:298-331); every key is required
bpe_tokenizer * bpe_tokenizer_from_gguf(const gguf_file & meta);

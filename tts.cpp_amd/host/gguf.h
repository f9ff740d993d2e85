// gguf.h — self-contained GGUF v2/v3 reader over an mmap'd file (the reference goes through ggml's
// gguf_init_from_file + llama_mmap, src/models/loaders.cpp:45-68; neither is vendored).
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"

enum gguf_vtype : uint32_t {
    GGUF_U8 = 0, GGUF_I8, GGUF_U16, GGUF_I16, GGUF_U32, GGUF_I32, GGUF_F32, GGUF_BOOL, GGUF_STR, GGUF_ARR,
    GGUF_U64, GGUF_I64, GGUF_F64
};

struct gguf_value {
    gguf_vtype               type = GGUF_U32;
    gguf_vtype               elem_type = GGUF_U32;  // arrays
    uint64_t                 u = 0;                 // integers / bool
    double                   f = 0;                 // floats
    std::string              s;                     // strings
    std::vector<std::string> arr_s;                 // string arrays
    const void *             arr_data = nullptr;    // numeric arrays: points into the mapping
    uint64_t                 arr_n = 0;
};

struct gguf_file {
    std::string                                 path;
    void *                                      map = nullptr;
    size_t                                      map_size = 0;
    std::vector<uint8_t>                        owned;  // OLLAMA_NO_MMAP: file read into memory instead (loaders.cpp:45)
    uint32_t                                    version = 0;
    std::unordered_map<std::string, gguf_value> kv;
    struct kv_span { std::string key; size_t begin, end; };  // byte range of one key/value record in the file
    std::vector<kv_span>                        kv_order;      // file order (a rewriter copies records verbatim)
    std::vector<gguf_tensor_view>               tensors;
    std::vector<std::string>                    tensor_names;  // owns the name strings
    size_t                                      data_offset = 0;

    ~gguf_file();
    const uint8_t * base() const { return map ? (const uint8_t *) map : owned.data(); }
    static std::shared_ptr<gguf_file> open(const char * path, std::string & err);

    int find_key(const std::string & key) const { return kv.count(key) ? 1 : -1; }
    const gguf_value * get(const std::string & key) const {
        auto it = kv.find(key);
        return it == kv.end() ? nullptr : &it->second;
    }
    // first present key of a list of aliases (search_for_gguf_keys, src/util.cpp:55-64)
    const gguf_value * get_any(std::initializer_list<const char *> keys) const;
    bool               get_u32(std::initializer_list<const char *> keys, uint32_t & out) const;
};

size_t gguf_type_row_bytes(int type, int64_t n);

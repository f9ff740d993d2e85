// quantize.h — GGUF → GGUF weight quantisation (the reference's `quantize` tool: examples/quantize/quantize_impl.h:5-15,
// quantize_impl.cpp:181-293).  Host-only: it rewrites a file, no device work.
#pragma once
#include <cstddef>
#include <cstdint>

// ggml_type numbers the tool accepts (examples/quantize/quantize.cpp:11-20)
enum tts_qtype : int { TTS_QTYPE_F32 = 0, TTS_QTYPE_F16 = 1, TTS_QTYPE_Q4_0 = 2, TTS_QTYPE_Q5_0 = 6, TTS_QTYPE_Q8_0 = 8 };

struct quantization_params {
    uint32_t n_threads = 1;
    int      quantize_type = TTS_QTYPE_Q4_0;
    bool     quantize_output_heads = false;
    bool     quantize_text_embeddings = false;
    bool     quantize_cross_attn_kv = false;
    bool     convert_dac_to_f16 = false;
    bool     convert_non_quantizable_to_f16 = false;
};

// Rewrites `ifile` as `ofile`: every tensor the architecture's allow-list names is converted from F32 to
// `quantize_type`; all key/values are kept and general.quantization_{version,type} are set.  Aborts (TTS_ABORT) on a
// tensor that is on the allow-list but not F32, or on an architecture without an allow-list.
void quantize_gguf(const char * ifile, const char * ofile, const quantization_params & params);

// which tensors `quantize_gguf` converts, exposed for tests: 0 = copied as is, 1 = quantize_type, 2 = F16
int quantize_decision(const char * arch, const char * tensor_name, int n_dims, const quantization_params & params);

// ggml's reference row quantisers (quantize_row_{q4_0,q5_0,q8_0}_ref, fp32 → fp16 round-to-nearest-even) over
// `nrows` rows of `n_per_row` floats, rows split over `n_threads`; returns the bytes written
size_t quantize_rows(int type, const float * src, void * dst, int64_t n_per_row, int64_t nrows, uint32_t n_threads);

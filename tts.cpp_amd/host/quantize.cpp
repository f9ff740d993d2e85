// quantize.cpp — GGUF → GGUF weight quantisation, host only.
//
// Behaviour follows the reference's tool (examples/quantize/quantize_impl.cpp): the per-architecture allow-lists
// (:14-80), "quantised tensors must start as F32" (:248-253, :266-271), every key/value carried over plus
// general.quantization_{version,type} (:203-206), tensors written in file order, each padded to the alignment
// (:289-290).  What is different underneath: ggml is not available, so the block formats are produced by our own
// restatement of ggml's reference row quantisers, and the file is produced by a streaming writer (sizes are known
// from the shapes, so the header is written once instead of being patched in after the data).
#include "quantize.h"

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <string_view>
#include <thread>
#include <vector>

#include "common.h"
#include "gguf.h"

namespace {

constexpr int      QK = 32;
constexpr uint32_t QNT_VERSION = 2;  // GGML_QNT_VERSION of the ggml generation the reference builds against

bool starts_with(std::string_view s, std::string_view p) { return s.substr(0, p.size()) == p; }
bool ends_with(std::string_view s, std::string_view p) { return s.size() >= p.size() && s.substr(s.size() - p.size()) == p; }
bool contains(std::string_view s, std::string_view p) { return s.find(p) != std::string_view::npos; }

// ---- allow-lists -----------------------------------------------------------------------------------------------
bool kokoro_f16_compatible(std::string_view n) {  // quantize_impl.cpp:14-18
    return !contains(n, "voice_tensors") && !contains(n, "bias") && !contains(n, "gamma") && !contains(n, "beta") &&
           !contains(n, "alpha") && !ends_with(n, "embd") && !ends_with(n, "norm");
}

bool kokoro_quantizable(std::string_view n) {  // quantize_impl.cpp:20-40
    if (!kokoro_f16_compatible(n)) return false;
    if (starts_with(n, "kokoro.albert") || starts_with(n, "kokoro.text_encoder.lstm")) return true;
    constexpr std::string_view prefix = "kokoro.duration_predictor.";
    if (!starts_with(n, prefix)) return false;
    std::string_view part = n.substr(prefix.size());
    part = part.substr(0, part.find('.'));
    for (std::string_view p : {"duration_proj", "encode", "shared_lstm", "duration_lstm", "layers"})
        if (part == p) return true;
    return false;
}

bool dia_quantizable(std::string_view n, const quantization_params & p) {  // quantize_impl.cpp:42-49
    bool q = !starts_with(n, "audio_encoder") && !ends_with(n, "norm");
    if (!p.quantize_output_heads) q = q && !starts_with(n, "dia.decoder.heads");
    return q;
}

bool parler_quantizable(std::string_view n, const quantization_params & p) {  // quantize_impl.cpp:51-67
    bool q = !starts_with(n, "audio_encoder") && !ends_with(n, "norm.weight") && !ends_with(n, "text_encoding") &&
             !ends_with(n, "positional_embed") && !ends_with(n, "norm.bias");
    if (!p.quantize_output_heads) q = q && !ends_with(n, "weight.head");
    if (!p.quantize_text_embeddings) q = q && !ends_with(n, "embed_prompts");
    if (!p.quantize_cross_attn_kv) q = q && !ends_with(n, "encoder_attn.k_proj.weight") && !ends_with(n, "encoder_attn.v_proj.weight");
    return q;
}

// Extension: the reference's tool has no Orpheus list (its is_quantizable aborts, :77-78) although BASELINE's Orpheus
// configuration is Q4_0.  The Llama-3 matrices are the 2-D orpheus.* tensors; norms, the rope frequency factors and
// the SNAC codec stay as they are.  The output head follows --quantize-output-heads like the other architectures.
bool orpheus_quantizable(std::string_view n, int n_dims, const quantization_params & p) {
    if (!starts_with(n, "orpheus.") || n_dims < 2) return false;
    if (ends_with(n, "norm") || ends_with(n, "rope_frequencies")) return false;
    if (!p.quantize_output_heads && ends_with(n, "lm_head")) return false;
    return true;
}

// ---- fp32 → fp16, round to nearest even (what ggml_fp32_to_fp16_row produces) -------------------------------------
uint16_t f16_bits(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint16_t sign = (uint16_t) ((x >> 16) & 0x8000u);
    const uint32_t a = x & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint16_t) (sign | 0x7e00u);   // NaN
    if (a >= 0x47800000u) return (uint16_t) (sign | 0x7c00u);  // |f| >= 2^16 (and inf)
    if (a >= 0x38800000u) {                                    // normal half: exponent >= -14
        const uint32_t m = a - 0x38000000u;                    // rebias 127 → 15
        uint32_t       r = m >> 13;
        const uint32_t rem = m & 0x1fffu;
        if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r++;  // a carry into the exponent is the right answer (→ inf at 65520)
        return (uint16_t) (sign | r);
    }
    if (a < 0x33000000u) return sign;                          // below half of the smallest subnormal
    const int      e = (int) (a >> 23);                        // 102 .. 112
    const uint32_t mant = (a & 0x7fffffu) | 0x800000u;
    const int      shift = 126 - e;                            // half subnormals count units of 2^-24
    uint32_t       r = mant >> shift;
    const uint32_t rem = mant & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (r & 1u))) r++;
    return (uint16_t) (sign | r);
}

// ---- ggml reference block quantisers (quantize_row_q4_0_ref / q5_0_ref / q8_0_ref; formats in SURVEY.md A.3) -----
void signed_absmax(const float * x, float & max) {
    float amax = 0.0f;
    max = 0.0f;
    for (int j = 0; j < QK; j++)
        if (amax < fabsf(x[j])) { amax = fabsf(x[j]); max = x[j]; }
}

void block_q4_0(const float * x, uint8_t * out) {
    float max;
    signed_absmax(x, max);
    const float    d = max / -8.0f, id = d != 0.0f ? 1.0f / d : 0.0f;
    const uint16_t dh = f16_bits(d);
    memcpy(out, &dh, 2);
    for (int j = 0; j < QK / 2; j++) {
        const uint8_t lo = (uint8_t) (int8_t) (x[j] * id + 8.5f), hi = (uint8_t) (int8_t) (x[j + QK / 2] * id + 8.5f);
        out[2 + j] = (uint8_t) ((lo > 15 ? 15 : lo) | ((hi > 15 ? 15 : hi) << 4));
    }
}

void block_q5_0(const float * x, uint8_t * out) {
    float max;
    signed_absmax(x, max);
    const float    d = max / -16.0f, id = d != 0.0f ? 1.0f / d : 0.0f;
    const uint16_t dh = f16_bits(d);
    memcpy(out, &dh, 2);
    uint32_t high = 0;
    for (int j = 0; j < QK / 2; j++) {
        uint8_t lo = (uint8_t) (int8_t) (x[j] * id + 16.5f), hi = (uint8_t) (int8_t) (x[j + QK / 2] * id + 16.5f);
        if (lo > 31) lo = 31;
        if (hi > 31) hi = 31;
        out[6 + j] = (uint8_t) ((lo & 0x0f) | ((hi & 0x0f) << 4));
        high |= (uint32_t) (lo >> 4) << j;
        high |= (uint32_t) (hi >> 4) << (j + QK / 2);
    }
    memcpy(out + 2, &high, 4);
}

void block_q8_0(const float * x, uint8_t * out) {
    float amax = 0.0f;
    for (int j = 0; j < QK; j++) amax = fmaxf(amax, fabsf(x[j]));
    const float    d = amax / 127.0f, id = d != 0.0f ? 1.0f / d : 0.0f;
    const uint16_t dh = f16_bits(d);
    memcpy(out, &dh, 2);
    for (int j = 0; j < QK; j++) out[2 + j] = (uint8_t) (int8_t) roundf(x[j] * id);
}

void quantize_row(int type, const float * src, uint8_t * dst, int64_t n) {
    switch (type) {
        case TTS_QTYPE_F16: {
            for (int64_t i = 0; i < n; i++) { const uint16_t h = f16_bits(src[i]); memcpy(dst + 2 * i, &h, 2); }
            break;
        }
        case TTS_QTYPE_Q4_0: for (int64_t b = 0; b < n / QK; b++) block_q4_0(src + b * QK, dst + b * 18); break;
        case TTS_QTYPE_Q5_0: for (int64_t b = 0; b < n / QK; b++) block_q5_0(src + b * QK, dst + b * 22); break;
        case TTS_QTYPE_Q8_0: for (int64_t b = 0; b < n / QK; b++) block_q8_0(src + b * QK, dst + b * 34); break;
        default: memcpy(dst, src, (size_t) n * 4); break;
    }
}

// ---- output file --------------------------------------------------------------------------------------------------
struct out_file {
    FILE * f = nullptr;
    size_t pos = 0;
    explicit out_file(const char * path) : f(fopen(path, "wb")) {
        if (!f) TTS_ABORT("cannot open '%s' for writing\n", path);
    }
    ~out_file() { if (f) fclose(f); }
    void put(const void * p, size_t n) {
        if (n && fwrite(p, 1, n, f) != n) TTS_ABORT("write failed\n");  // the reference fails fast on write errors too (:230)
        pos += n;
    }
    template <typename T> void val(T v) { put(&v, sizeof(T)); }
    void str(std::string_view s) { val<uint64_t>(s.size()); put(s.data(), s.size()); }
    void pad_to(size_t align) {
        static const char zeros[64] = {0};
        while (pos % align) put(zeros, std::min(sizeof(zeros), align - pos % align));
    }
    void close() {
        if (fclose(f) != 0) { f = nullptr; TTS_ABORT("write failed\n"); }
        f = nullptr;
    }
};

std::string u32_record(std::string_view key, uint32_t v) {
    std::string    r;
    const uint64_t n = key.size();
    const uint32_t t = GGUF_U32;
    r.append((const char *) &n, 8).append(key).append((const char *) &t, 4).append((const char *) &v, 4);
    return r;
}

}  // namespace

size_t quantize_rows(int type, const float * src, void * dst, int64_t n_per_row, int64_t nrows, uint32_t n_threads) {
    const size_t row_bytes = gguf_type_row_bytes(type, n_per_row);
    if (row_bytes == 0) TTS_ABORT("quantisation type '%d' is not supported\n", type);
    if (type != TTS_QTYPE_F16 && type != TTS_QTYPE_F32 && n_per_row % QK != 0)
        TTS_ABORT("a row of %lld values cannot be split into blocks of %d\n", (long long) n_per_row, QK);
    // rows are independent, so the split over threads cannot change the result; hand out chunks of >= 16K values
    // like the reference does (quantize_impl.cpp:100-105)
    const int64_t chunk_rows = std::max<int64_t>(1, (32 * 512 + n_per_row - 1) / n_per_row);
    const int64_t n_chunks = (nrows + chunk_rows - 1) / chunk_rows;
    const int     threads = (int) std::max<int64_t>(1, std::min<int64_t>(n_threads, n_chunks));
    std::atomic<int64_t> next{0};
    auto work = [&]() {
        for (;;) {
            const int64_t c = next.fetch_add(1);
            if (c >= n_chunks) return;
            const int64_t r1 = std::min(nrows, (c + 1) * chunk_rows);
            for (int64_t r = c * chunk_rows; r < r1; r++)
                quantize_row(type, src + r * n_per_row, (uint8_t *) dst + (size_t) r * row_bytes, n_per_row);
        }
    };
    if (threads == 1) {
        work();
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++) pool.emplace_back(work);
        for (auto & t : pool) t.join();
    }
    return row_bytes * (size_t) nrows;
}

int quantize_decision(const char * arch, const char * tensor_name, int n_dims, const quantization_params & p) {
    const std::string_view n{tensor_name};
    const auto             it = SUPPORTED_ARCHITECTURES.find(arch);
    if (it == SUPPORTED_ARCHITECTURES.end()) TTS_ABORT("%s failed. The architecture '%s' is not supported.\n", __func__, arch);
    bool q = false;
    switch (it->second) {
        case PARLER_TTS_ARCH: q = parler_quantizable(n, p); break;
        case DIA_ARCH: q = dia_quantizable(n, p); break;
        case KOKORO_ARCH: q = kokoro_quantizable(n); break;
        case ORPHEUS_ARCH: q = orpheus_quantizable(n, n_dims, p); break;
    }
    if (q) return 1;
    if ((p.convert_non_quantizable_to_f16 && kokoro_f16_compatible(n)) ||
        (p.convert_dac_to_f16 && starts_with(n, "audio_encoder") && !ends_with(n, "alpha")))
        return 2;  // quantize_impl.cpp:264-265
    return 0;
}

void quantize_gguf(const char * ifile, const char * ofile, const quantization_params & params) {
    std::string err;
    auto        in = gguf_file::open(ifile, err);
    if (!in) TTS_ABORT("%s\n", err.c_str());
    std::string arch = "parler-tts";  // only Parler files may lack the key (quantize_impl.cpp:188-192)
    if (auto a = in->get("general.architecture")) arch = a->s;
    if (!SUPPORTED_ARCHITECTURES.count(arch)) TTS_ABORT("%s failed. The architecture '%s' is not supported.\n", __func__, arch.c_str());
    const int qtype = params.quantize_type;
    if (gguf_type_row_bytes(qtype, QK) == 0 || qtype == TTS_QTYPE_F32)
        TTS_ABORT("ERROR: quantization type '%d' is not supported by this build (F16, Q4_0, Q5_0, Q8_0).\n", qtype);

    // pass 1: decide every tensor's new type and size
    struct plan { int type; size_t nbytes; int n_dims; int64_t rows; };
    std::vector<plan> plans;
    plans.reserve(in->tensors.size());
    for (const gguf_tensor_view & t : in->tensors) {
        int nd = 1;
        for (int d = 1; d < 4; d++)
            if (t.ne[d] > 1) nd = d + 1;  // what ggml_n_dims reports: trailing 1s dropped
        const int     what = quantize_decision(arch.c_str(), t.name, nd, params);
        const int64_t rows = t.ne[1] * t.ne[2] * t.ne[3];
        plan          p{t.type, t.nbytes, nd, rows};
        if (what != 0) {
            if (t.type != TTS_QTYPE_F32)
                TTS_ABORT("ERROR: All %s tensors must be transformed from 32bit floats. Tensor, '%s', has improper type, '%d'\n",
                          what == 1 ? "quantized" : "converted", t.name, t.type);
            p.type = what == 1 ? qtype : TTS_QTYPE_F16;
            if (p.type != TTS_QTYPE_F16 && t.ne[0] % QK != 0)
                TTS_ABORT("ERROR: Tensor '%s' has rows of %lld values, not a multiple of the block size %d\n", t.name, (long long) t.ne[0], QK);
            p.nbytes = gguf_type_row_bytes(p.type, t.ne[0]) * (size_t) rows;
        }
        plans.push_back(p);
    }

    // key/values: the input's records verbatim and in order, the two quantisation keys replaced in place or appended
    size_t align = 32;
    if (auto a = in->get("general.alignment")) align = a->u ? (size_t) a->u : 32;
    const std::string rec_version = u32_record("general.quantization_version", QNT_VERSION);
    const std::string rec_type = u32_record("general.quantization_type", (uint32_t) qtype);
    bool              have_version = false, have_type = false;
    for (const auto & s : in->kv_order) {
        have_version |= s.key == "general.quantization_version";
        have_type |= s.key == "general.quantization_type";
    }

    out_file out(ofile);
    out.put("GGUF", 4);
    out.val<uint32_t>(3);
    out.val<uint64_t>(in->tensors.size());
    out.val<uint64_t>(in->kv_order.size() + !have_version + !have_type);
    for (const auto & s : in->kv_order) {
        if (s.key == "general.quantization_version") out.put(rec_version.data(), rec_version.size());
        else if (s.key == "general.quantization_type") out.put(rec_type.data(), rec_type.size());
        else out.put(in->base() + s.begin, s.end - s.begin);
    }
    if (!have_version) out.put(rec_version.data(), rec_version.size());
    if (!have_type) out.put(rec_type.data(), rec_type.size());
    size_t offset = 0;
    for (size_t i = 0; i < plans.size(); i++) {
        const gguf_tensor_view & t = in->tensors[i];
        out.str(t.name);
        out.val<uint32_t>((uint32_t) plans[i].n_dims);
        for (int d = 0; d < plans[i].n_dims; d++) out.val<uint64_t>((uint64_t) t.ne[d]);
        out.val<uint32_t>((uint32_t) plans[i].type);
        out.val<uint64_t>(offset);
        offset += (plans[i].nbytes + align - 1) / align * align;
    }
    out.pad_to(align);

    // pass 2: tensor data, converted one tensor at a time
    std::vector<uint8_t> work;
    for (size_t i = 0; i < plans.size(); i++) {
        const gguf_tensor_view & t = in->tensors[i];
        if (plans[i].type == t.type) {
            out.put(t.data, t.nbytes);
        } else {
            if (work.size() < plans[i].nbytes) work.resize(plans[i].nbytes);
            quantize_rows(plans[i].type, (const float *) t.data, work.data(), t.ne[0], plans[i].rows, params.n_threads);
            out.put(work.data(), plans[i].nbytes);
        }
        fprintf(stdout, "At tensor: '%s' with new size: %zu bytes\n", t.name, plans[i].nbytes);
        out.pad_to(align);
    }
    out.close();
}

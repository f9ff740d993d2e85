// tts_hip.hip — implementation of include/tts_hip.h for MI355X (gfx950).
//
// Owns: the device weight arena (tts_model's backend buffer, /root/reference/src/tts_model.cpp:157-169),
// the self-attention KV cache (parler_kv_cache, src/models/parler/model.cpp:339-385), the cross K/V
// (prep_cross_key_values :110-173), one HIP stream, and the captured hipGraphs that replace the
// per-step ggml graph rebuild (build_parler_graph :520-614 + ggml_backend_sched_alloc_graph :674).
// No CPU fallback: every entry point fails if the device is unavailable.
#include "../../include/tts_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <mutex>
#include <string>
#include <vector>

#include "dac_kernels.h"
#include "dac_b3_kernels.h"
#include "parler_kernels.h"
#include "gemm_tile_kernels.h"
#include "gemv_stream_kernels.h"
#include "t5_kernels.h"
#include "llama_kernels.h"
#include "dia_kernels.h"
#include "gemv_kernels.h"
#include "kokoro_kernels.h"
#define LLAMA_GREEDY_CHUNK 8

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";
static int set_err(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return -1;
}
#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) return set_err("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
    } while (0)
#define CHK(expr)                       \
    do {                                \
        int _r = (expr);                \
        if (_r != 0) return _r;         \
    } while (0)

extern "C" const char *tts_hip_last_error(void) { return g_err; }
extern "C" const char *tts_hip_version(void) { return "tts_hip 0.1 (gfx950)"; }

// ------------------------------------------------------------------------------------------------
// host-side format helpers (GGUF block formats)
// ------------------------------------------------------------------------------------------------
static float h2f_host(uint16_t h) {
    const uint32_t sign = (uint32_t) (h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1F, man = h & 0x3FF, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {
            exp = 113;
            while ((man & 0x400) == 0) { man <<= 1; exp--; }
            bits = sign | (exp << 23) | ((man & 0x3FF) << 13);
        }
    } else if (exp == 31) bits = sign | 0x7F800000u | (man << 13);
    else bits = sign | ((exp + 112) << 23) | (man << 13);
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

static size_t type_row_bytes(int type, int64_t n) {
    switch (type) {
        case TTS_HIP_F32: return (size_t) n * 4;
        case TTS_HIP_F16: return (size_t) n * 2;
        case TTS_HIP_Q4_0: return (size_t) (n / 32) * 18;
        case TTS_HIP_Q5_0: return (size_t) (n / 32) * 22;
        case TTS_HIP_Q8_0: return (size_t) (n / 32) * 34;
        default: return 0;
    }
}

// exact dequantisation of a GGUF tensor to fp32 on the host (round-1 handling of Q4_0/Q5_0/Q8_0 and
// of fp16 tensors that the kernels want in fp32)
static int dequant_to_f32(int type, const void *src, float *dst, int64_t n) {
    const uint8_t *p = (const uint8_t *) src;
    if (type == TTS_HIP_F32) { memcpy(dst, src, (size_t) n * 4); return 0; }
    if (type == TTS_HIP_F16) {
        const uint16_t *h = (const uint16_t *) src;
        for (int64_t i = 0; i < n; i++) dst[i] = h2f_host(h[i]);
        return 0;
    }
    if (n % 32) return -1;
    for (int64_t b = 0; b < n / 32; b++, dst += 32) {
        uint16_t dh;
        memcpy(&dh, p, 2);
        const float d = h2f_host(dh);
        if (type == TTS_HIP_Q4_0) {
            const uint8_t *qs = p + 2;
            for (int j = 0; j < 16; j++) {
                dst[j] = (float) ((int) (qs[j] & 0xF) - 8) * d;
                dst[j + 16] = (float) ((int) (qs[j] >> 4) - 8) * d;
            }
            p += 18;
        } else if (type == TTS_HIP_Q5_0) {
            uint32_t qh;
            memcpy(&qh, p + 2, 4);
            const uint8_t *qs = p + 6;
            for (int j = 0; j < 16; j++) {
                const int b0 = (qh >> j) & 1, b1 = (qh >> (j + 16)) & 1;
                dst[j] = (float) ((int) ((qs[j] & 0xF) | (b0 << 4)) - 16) * d;
                dst[j + 16] = (float) ((int) ((qs[j] >> 4) | (b1 << 4)) - 16) * d;
            }
            p += 22;
        } else if (type == TTS_HIP_Q8_0) {
            const int8_t *qs = (const int8_t *) (p + 2);
            for (int j = 0; j < 32; j++) dst[j] = (float) qs[j] * d;
            p += 34;
        } else return -1;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
#define TTS_HIP_Q8I 100  // device-side: int8 block integers [N][K] followed by fp16 block scales [N][K/32]

struct Tensor {
    int type = 0;  // type as stored on the device (F32, F16 or TTS_HIP_Q8I)
    int n_dims = 0;
    int64_t ne[4] = {1, 1, 1, 1};
    size_t nbytes = 0;
    int src_type = 0;  // ggml type the tensor had in the GGUF
    void *tmp = nullptr;  // device staging copy until finalize
    bool has_data = false;
    int64_t nelem() const { return ne[0] * ne[1] * ne[2] * ne[3]; }
};

struct W {  // a matrix living in the arena
    size_t off = 0;
    size_t soff = 0;  // TTS_HIP_Q8I: block scales
    int type = 0;
    int64_t K = 0, N = 0;
    bool src_q4 = false;            // every stacked source tensor was Q4_0 in the GGUF
    const uint8_t *q4 = nullptr;    // TTS_HIP_Q4_NATIVE: the 4-bit codes repacked next to the int8 expansion (gemv_q4_rows_kernel)
};

struct PLayer {
    W qkv, o, cq, ck, cv, co, fc1, fc2;
    size_t sa_w = 0, sa_b = 0, ca_w = 0, ca_b = 0, f_w = 0, f_b = 0;
};

struct DRes { size_t in_alpha, in_w, in_b, out_alpha, out_w, out_b; };
struct DBlock { int stride, padding, cin, cout; size_t alpha, w, b; DRes res[3]; };

struct CopyItem { size_t dst; std::string src; size_t src_off = 0; size_t bytes = 0; };  // bytes == 0: the whole tensor

struct ProfEv { hipEvent_t a, b; int kclass; };

// Codec activation buffers of a device.  Codec passes of one device take turns (g_dac_pass_mutex: a pass fills the chip), so every
// context of the device works in the same buffers instead of holding its own three activation buffers (197 KB per frame and utterance of
// a pass each: 12.6 GB per context for 64 x 248 frames, 38.6 GB at 1016 frames); they are freed when the device's last codec context goes.
constexpr int GRAPH_KEY_ROWS = 8192;   // captured decode steps are keyed mode * GRAPH_KEY_ROWS + rows (run_step, drop_gen_graphs)

struct DacBuffers {
    float *dbuf[3] = {nullptr, nullptr, nullptr};
    size_t dbuf_elems = 0;       // capacity of each buffer in floats
    size_t cap_codes = 0;        // ids d_codes holds (frames summed over a batch, padded to the longest, x codebooks)
    float *dplanes = nullptr;    // second planes buffer of the wide classes (the first lives in dbuf[2])
    uint32_t *d_codes = nullptr;
    float *h_pcm = nullptr;
    size_t h_pcm_elems = 0;
    int users = 0;
};
static DacBuffers g_dac_buffers[64];

struct tts_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t dac_stream = nullptr;  // low-priority queue for the codec (NULL: same stream)
    tts_hip_desc d{};
    std::map<std::string, Tensor> tensors;
    bool planned = false, finalized = false, weights_present = false;
    bool has_parler = false, has_dac = false;

    // arena
    char *arena = nullptr;
    size_t arena_bytes = 0;
    bool arena_external = false;
    std::vector<CopyItem> copies;

    // parler model
    int H = 0, L = 0, NH = 0, F = 0, V = 0, NO = 0, NCTX = 0, E = 0, ECAP = 0, PV = 0, EROWS = 0, NPOS = 0, KVPOS = 0;
    W embed_prompts, embed_tokens, heads;
    size_t pos_embed = 0, text_enc = 0, ln_w = 0, ln_b = 0, cross_kv = 0;
    // The voice-prompt encoding and the cross K/V computed from it are the only arena entries that change after load
    // (tts_hip_parler_set_text_encoding).  Contexts of one device may share an arena (tts_hip_finalize(ctx, arena of another context)),
    // so a context that gets a new prompt moves both into allocations of its own and the arena stays immutable: a sibling in the middle
    // of a generation keeps reading the prompt it started with (the reference keeps a whole model per worker, server.cpp:316-321).
    char *cond_text_enc = nullptr, *cond_cross_kv = nullptr;   // private copies, or NULL = the arena's
    char *text_enc_ptr() const { return cond_text_enc ? cond_text_enc : arena + text_enc; }
    char *cross_kv_ptr() const { return cond_cross_kv ? cond_cross_kv : arena + cross_kv; }
    std::vector<PLayer> layers;

    // dac model
    int d_ncb = 0, d_cbsize = 0, d_cbdim = 0, d_latent = 0, d_c0 = 0, d_clast = 0, d_up = 1;
    size_t d_codebook = 0, d_projw = 0, d_projb = 0, d_initw = 0, d_initb = 0, d_falpha = 0, d_fw = 0, d_fb = 0;
    std::vector<DBlock> dblocks;

    // runtime buffers
    int RMAX = 0;
    void *kcache = nullptr, *vcache = nullptr;  // [L][max_seqs][NCTX][H]
    float *x = nullptr, *q = nullptr, *att = nullptr, *u32 = nullptr, *logits = nullptr, *part = nullptr, *dbg = nullptr;
    _Float16 *u16 = nullptr, *xn16 = nullptr, *att16 = nullptr;
    int ln_fuse_max = 8;  // rows up to which LayerNorm stays fused in the GEMM prologue
    float *partials = nullptr;  // [8][RMAX][H] split-K slabs of the residual GEMMs
    int8_t *aq = nullptr;       // Q8_0-quantised activation rows [RMAX][max(H,F)]
    float *ad = nullptr;        // their block scales
    bool all_q8i = false;       // every decoder matrix is on the integer path (all GEMMs go through run_qgemm)
    int q_fuse_max = 16;        // rows up to which the integer GEMM quantises its own activations
    bool q4_native = false;     // TTS_HIP_Q4_NATIVE (with TTS_HIP_GEMV_ROWS; default on for Orpheus contexts): Q4_0 matrices are read as 4-bit codes
    std::vector<void *> q4_bufs;
    bool q4_rms = true;         // TTS_HIP_Q4_RMS=0: the rms norms in front of the q/k/v and gate|up projections keep their own launches
    bool q4_silu = true;        // TTS_HIP_Q4_SILU=0: gate|up, silu * up and the down projection stay three launches
    bool q4_rope = true;        // TTS_HIP_Q4_ROPE=0: the Llama q/k/v projection keeps its separate rope + cache-append launch
    bool q4_lds = true;         // TTS_HIP_Q4_LDS=0: Q4_0 row products stay on gemv_q4_rows_kernel (one feature per wave, activations from L2)
    bool gemv_stream = true;    // TTS_HIP_GEMV_STREAM=0: <= 16-row F16 GEMMs of the Dia step stay on gemm16_kernel (gemv_stream_kernels.h otherwise)
    bool llama_graph = false;   // TTS_HIP_LLAMA_GRAPH (default on for Orpheus contexts): the greedy step as one captured graph
    bool gemv_rows = false;     // TTS_HIP_GEMV_ROWS (default on for Orpheus contexts): 1..4 rows go through the streaming one-wave-per-feature kernels (gemv_kernels.h)
    // ---- Orpheus decoder context (tts_hip_orpheus_create) ----
    bool has_llama = false;
    tts_hip_orpheus_desc lm{};
    struct LLayer { size_t in_norm = 0, post_norm = 0; W qkv, o, gu, down; };
    std::vector<LLayer> l_layers;
    size_t l_embd = 0, l_out_norm = 0, l_ropef = 0;
    W l_head;
    int l_V = 0, l_Vpad = 0, l_kvH = 0, l_ksplit = 1;
    float *l_x = nullptr, *l_xn = nullptr, *l_qkv = nullptr, *l_att = nullptr, *l_gu = nullptr, *l_g = nullptr, *l_logits = nullptr, *l_parts = nullptr;
    float *l_kc = nullptr, *l_vc = nullptr;
    uint32_t *l_ids = nullptr, *l_pos = nullptr, *l_tok = nullptr;
    int l_pending = 0;
    // ---- Dia context (tts_hip_dia_create) ----
    bool has_dia = false;
    tts_hip_dia_desc dia{};
    struct DiaEnc { size_t sa_norm = 0, mlp_norm = 0; W qkv, o, gu, out; };
    struct DiaDec { size_t sa_norm = 0, ca_norm = 0, mlp_norm = 0; W sqkv, so, cq, ckv, co, gu, out; };
    std::vector<DiaEnc> di_enc;
    std::vector<DiaDec> di_dec;
    size_t di_enc_embd = 0, di_enc_norm = 0, di_dec_norm = 0, di_embd[16] = {0};
    W di_heads;
    int di_EH = 0, di_EF = 0, di_DF = 0, di_A = 0, di_kvH = 0, di_V = 0, di_Vpad = 0, di_ksplit = 1, di_pending = 0, di_evocab = 0;
    float *di_ex = nullptr, *di_exn = nullptr, *di_eqkv = nullptr, *di_eatt = nullptr, *di_egu = nullptr, *di_eg = nullptr, *di_ek = nullptr, *di_ev = nullptr;
    float *di_ckv = nullptr, *di_ck = nullptr, *di_cv = nullptr, *di_k = nullptr, *di_v = nullptr;
    float *di_x = nullptr, *di_xn = nullptr, *di_qkv = nullptr, *di_q = nullptr, *di_att = nullptr, *di_gu = nullptr, *di_g = nullptr, *di_parts = nullptr;
    float *di_logits = nullptr, *di_guided = nullptr;
    uint32_t *di_tok = nullptr, *di_epos = nullptr, *di_eseq = nullptr, *di_kbeg = nullptr, *di_kend = nullptr;
    uint32_t *di_ids = nullptr, *di_pos = nullptr, *di_seq = nullptr, *di_cend = nullptr;
    // device-resident generation loop (tts_hip_dia_generate): sampled ids [U][NO], countdown [U] / done [U] / sampler call [U], history [U][G][NO]
    uint32_t *di_stok = nullptr, *di_loop = nullptr, *di_hist = nullptr;
    _Float16 *di_e16 = nullptr;   // [2 * max_ctx][max(EH, A, EF)] the encoder activations rounded to fp16 for gemm_tile_kernel
    struct { const void *uni = nullptr, *pen = nullptr; tts_hip_sampling sp{}; int mode = -1; uint32_t U = 0, max_gen = 0; tts_hip_dia_codes codes{}; } di_baked;
    int di_U = 1;                        // utterance slots (rows = 2 per slot)
    std::vector<uint8_t> di_slot_encoded;   // tts_hip_dia_encode_slot has run for the slot
    uint32_t *h_di = nullptr;            // pinned staging: ids / pos / seq of a step
    // ---- Kokoro context (tts_hip_kokoro_create) ----
    bool has_kokoro = false;
    tts_hip_kokoro_desc ko{};
    struct KTensor { size_t off = 0; int64_t ne[4] = {1, 1, 1, 1}; };
    std::unordered_map<std::string, KTensor> k_tensors;   // every "kokoro.*" tensor, fp32, by GGUF name
    // ---- SNAC codec context (tts_hip_snac_create) ----
    bool has_snac = false;
    tts_hip_snac_desc snac{};
    struct SRes { size_t in_alpha = 0, in_w = 0, in_b = 0, out_alpha = 0, out_w = 0, out_b = 0; };
    struct SBlock { int stride = 0, padding = 0, cin = 0, cout = 0; size_t alpha = 0, w = 0, b = 0, noise_w = 0; SRes res[3]; };
    std::vector<SBlock> sblocks;
    size_t s_codebook = 0, s_projw = 0, s_projb = 0, s_inw = 0, s_inb = 0, s_upw = 0, s_upb = 0, s_falpha = 0, s_fw = 0, s_fb = 0;
    int s_latent = 0, s_c0 = 0, s_cbdim = 0, s_cbsize = 0, s_up = 1, s_clast = 0;
    float *sbuf[3] = {nullptr, nullptr, nullptr};
    float *s_noise = nullptr;
    uint32_t *s_codes = nullptr;
    bool snac_packed = false;
    // ---- T5 voice-prompt encoder context (tts_hip_t5_create) ----
    bool has_t5 = false;
    tts_hip_t5_desc t5{};
    struct T5Layer { size_t attn_norm = 0, mlp_norm = 0; W qkv, o, wi, wo; };
    std::vector<T5Layer> t5_layers;
    size_t t5_embd = 0, t5_relb = 0, t5_out_norm = 0, t5_down_b = 0;
    W t5_down;
    bool t5_has_down = false, t5_has_down_b = false;
    int t5_vocab = 0, t5_out = 0;
    int *t5_bucket = nullptr;          // bucket of (key - query) + (n_ctx - 1), host-computed with the reference's arithmetic
    float *t5_x = nullptr, *t5_qkv = nullptr, *t5_att = nullptr, *t5_ug = nullptr, *t5_g = nullptr, *t5_y = nullptr;
    uint32_t *t5_ids = nullptr;
    tts_hip_sampling smp{};     // parameters baked into the captured MODE_GEN_SAMPLE graphs
    float *d_uniforms = nullptr;  // [calls][R][n_out] host-drawn U[0,1) for sample_kernel
    unsigned long long *l_cand = nullptr;   // Orpheus sampler: [TOPK_PARTS][TOPK_MAXK] stage-1 survivors
    uint32_t *l_smp = nullptr;              // Orpheus sampler: [0] last token (int32), [1] repetition count, [2] sampler call index
    struct { const void *uni = nullptr, *pen = nullptr; uint32_t k = 0; float temp = 0; } l_smp_baked;   // what the captured sampled step holds
    double *d_pen = nullptr;      // pow(repetition_penalty, count) table (host-evaluated)
    int pen_len = 0;
    int32_t *d_last = nullptr;    // [RMAX][n_out] sampler::last_token_ids
    uint32_t *d_repc = nullptr;   // [RMAX][n_out] sampler::repetition_counts
    size_t uniforms_cap = 0;
    uint32_t g_bos = 0xFFFFFFFFu, g_eos = 0xFFFFFFFFu;  // ids baked into the captured feed kernel
    int pending_parts = 0;      // slabs waiting to be folded into x by the next LayerNorm launch
    int ln_waves = 1;           // rows (waves) per LayerNorm workgroup
    int ksplit_big = 4;         // K slices for K >= 4096 residual GEMMs
    uint32_t *d_ids = nullptr, *d_pos = nullptr, *d_seq = nullptr, *d_tok = nullptr, *d_step = nullptr, *d_steps_done = nullptr;
    uint32_t *d_gather = nullptr;   // scratch of the row compaction: map [R] + ids [R][n_out] + pos / seq / step [R] each
    int gen_total = 0;              // utterances of the generation loop under way (rows of the forward <= this after a compaction)
    bool gen_compact = true;        // TTS_HIP_GEN_COMPACT=0: finished utterances keep idling in the lock-step forward
    uint32_t *d_tokens_out = nullptr;
    size_t tokens_out_cap = 0;
    uint8_t *d_eos = nullptr;
    // pinned staging
    uint32_t *h_ids = nullptr, *h_pos = nullptr, *h_seq = nullptr, *h_tok = nullptr;
    float *h_logits = nullptr;
    std::vector<uint32_t> host_pos;  // positions per row of the forward being enqueued (for byte accounting)

    // dac: the activation buffers are the device's (g_dac_buffers)
    size_t dac_frame_elems = 0;  // largest activation per frame over all stages (C * L / frames)
    size_t dac_cap_frames = 0;   // SNAC: frames its own buffers hold
    uint32_t *d_frames = nullptr;
    size_t d_frames_cap = 0;
    bool debug = false;
    std::map<int, std::vector<float>> dac_dbg;
    std::map<size_t, float *> packed;  // arena offset of a conv weight -> its MFMA-tile-packed copy
    std::set<size_t> packed_c192;      // ... and the k = 7 weights of 192-channel units packed as one 192-channel tile (TTS_HIP_DAC_C192=1)
    int dac_c192 = 0;
    std::set<size_t> packed_direct;    // ... of those, the k = 1 weights packed as [cin][cout] for conv1x1_direct_kernel
    std::map<size_t, _Float16 *> packed16;  // same, fp16 images (dac_f16)
    std::map<size_t, __bf16 *> packed_b3;   // k = 7 conv weights as three bf16 planes (dac_b3, experiment)
    std::map<size_t, __bf16 *> packed_ru;   // residual unit (keyed by its k = 7 weight) -> stage stream of resunit_b3_kernel
    std::map<size_t, __bf16 *> packed_p;    // conv weight -> bf16 planes in stage order for conv_b3p_kernel (k = 7: 64-channel tiles, k = 1: 128-channel tiles)
    int dac_k1_variant = 1;     // TTS_HIP_DAC_K1_VARIANT: tiles of the k = 1 convs on planes: 0 = 128 ch x 256 pos (21.9 ms per pass); 1 = 256 x 256 where the channels divide by 256, 128 x 512 elsewhere (20.4 ms: fewer re-reads of the operands from L2)
    int dac_p_variant = 0;      // TTS_HIP_DAC_P_VARIANT: tile shape of the k = 7 convs on planes (0 = 4 waves, one LDS buffer, two workgroups per CU: measured best, profiles/r03/tap7_call16.txt)
    int dac_tap7 = 1;           // TTS_HIP_DAC_TAP7=0: the k = 7 convs on planes keep the tap-pair k-steps (8 slots for 7 taps) instead of one tap per k-step
    int dac_planes = 1;         // TTS_HIP_DAC_PLANES=0: the wide classes (channels % 128 == 0, no fused unit) keep fp32 activations and stage snake + split per tile
    bool dac_buf_user = false;  // counted in g_dac_buffers[device].users
    std::map<size_t, __bf16 *> packed_ct;   // transposed conv weight -> bf16 planes of convt_b3_kernel
    int dac_convt_b3 = 1;       // TTS_HIP_DAC_CONVT_B3=0: the transposed convs stay on the exact-fp32 MFMA kernel
    int dac_fuse = 1;           // TTS_HIP_DAC_FUSE=0: residual units at 96 / 192 channels stay two launches (k = 7 conv, k = 1 conv + residual)
    int dac_b3_variant = 2;     // TTS_HIP_DAC_B3_VARIANT: tile shape of the 64-channel class of the experiment
    int dac_b3 = 2;             // TTS_HIP_DAC_BF16X3 (default 2 since round 3; 0 = exact-fp32 MFMA convs): k = 7 convs of F32 tensors as six bf16 MFMAs per product (conv1d_mfma_b3_kernel); 1 = the layers with 64-channel tiles (measured, tested), 2 = also the 96-channel tile (written after the GPU budget of round 2 was spent: never run)
    bool kk_lstm_split = true;  // TTS_HIP_KOKORO_LSTM_SPLIT=0: the bidirectional LSTMs through the one-workgroup-per-direction kernel
    char *kk_pool = nullptr;    // Kokoro scratch pool (KScratch): grows to the largest call
    size_t kk_pool_cap = 0, kk_pool_next = 0;
    int *kk_stuck = nullptr;    // set by kk_lstm_split_kernel when a granule never arrives (bounded spin)
    bool kk_mfma = true;        // TTS_HIP_KOKORO_MFMA=0: every Kokoro convolution through the one-thread-per-output kernel
    int dac_group = 64;         // TTS_HIP_DAC_GROUP: utterances per codec pass (16: 451, 32: 458, 64: 461, 128: 460, 384: 462 audio-s/s at 3 x 384)
    bool dac_conv1_direct = true;   // TTS_HIP_DAC_CONV1_DIRECT=0: the 96- / 192-channel k=1 convs stay on conv1d_mfma_kernel<1,...>
    int dac_pad = 0;            // TTS_HIP_DAC_PAD=1: activation rows at the padded stride of dac_row_stride (measured: no effect, profiles/r02/dac_row_stride.log)
    int dac_variant = 20;       // TTS_HIP_DAC_VARIANT (tuning; 20 = 96-channel class on 128-position tiles, the one variant that measured faster): position-tile variant of the k = 7 conv kernel per channel-tile class
    int dac_alpha_tab = 1;      // 0: the codec kernels read snake's alpha from memory instead of an LDS table (smaller footprint); 2: table unless it costs a resident workgroup
    int dac_prio = 2;           // TTS_HIP_DAC_PRIO: waves of the k = 7 conv raise their issue priority for the staging phase of a chunk (2, measured 1.2 % faster twice), for the MFMA phase (1, slower), static per workgroup (3 / 4, neutral / slower), 0 = never
    int dac_lds_reserve_kb = 0; // LDS the codec kernels leave free per CU for another context's decoder workgroups
    int gemm_rows_per_wg = 0;   // > 0: forwards with more rows split them over workgroups of this many rows (16/32/64)
    int gemm_ngs_max = 16;      // cap on parallel row-group wave sets per GEMM workgroup (LDS = 4 KB x waves x RB)
    bool attn_short = true;     // TTS_HIP_ATTN_SHORT=0: cross-attention through the general kernel
    int tile_min_rows = 33;     // forwards with at least this many rows take the LDS-tiled GEMM (gemm_tile_kernels.h); 0 = never
    int tile_force = -1;        // TTS_HIP_TILE_FORCE: tile shape index for every tiled GEMM (tuning)
    int tile_force_ks = 0;      // TTS_HIP_TILE_KS: k slices for the residual GEMMs (tuning)
    const void *aq_src = nullptr;  // activation rows whose Q8_0 blocks already sit in aq / ad (written by the producing kernel)
    int attn_split_max = 8;     // TTS_HIP_ATTN_SPLIT: key splits of the decode attention of the Llama / Dia steps (1 = off)
    float *attn_part = nullptr; // [rows][heads][splits][130] partial softmax results
    size_t attn_part_cap = 0;   // in (row, head, split) triples
    int tile_deep = 1;          // TTS_HIP_TILE_DEEP=0: always the 64-wide k-tiles / 4 buffers form (half the LDS per workgroup)
    bool dac_f16 = false;      // every codec conv kernel arrived as F16: fp16 im2col x fp16 kernel, fp32 accumulate (ggml)
    bool dac_packed = false;

    // graphs
    std::map<int, hipGraphExec_t> graphs;

    // profiling
    bool prof = false;       // full per-launch event timing: forwards run eagerly
    bool prof_light = false; // event pairs only around launches that are never graph-captured (the DAC)
    std::vector<ProfEv> prof_events;
    bool prof_cur = false;
    tts_hip_kstat kstat[TTS_HIP_K_COUNT]{};
    int attn_nsplit_override = 0;
    // batch-1 chain (<= 4 rows; timeline in profiles/r03/b1_chain.txt):
    bool b1_fc2_split = true;        // TTS_HIP_B1_FC2_SPLIT=0: fc2 stays 64 workgroups x 128 KB of weights (7.4 us) instead of 256 x 32 KB writing four K-slice slabs
    bool b1_defer_combine = true;    // TTS_HIP_B1_DEFER_COMBINE=0: the split-T self-attention folds its partials itself (last workgroup, +3.3 us) instead of out_proj's prologue
    long long *b1_stamps = nullptr;  // TTS_HIP_B1_STAMPS=1: 16 s_memrealtime stamps per launch of a <= 4-row forward (debug_read "stamps", profiles/b1_chain.py)
    int b1_stamp_slot = 0;
    bool attn_fused = true;          // TTS_HIP_ATTN_FUSED=0: split-T self-attention keeps its separate combine launch and small batches stay unsplit
    uint32_t *attn_cnt = nullptr;    // [RMAX][heads] arrival counters of the fused combine (attn_kernel)
};

static const char *KNAMES[TTS_HIP_K_COUNT] = {"embed", "ln", "gemm_qkv", "attn_self", "gemm_attn_out", "gemm_cross_q", "attn_cross",
                                              "gemm_cross_out", "gemm_fc1", "gemm_fc2", "gemm_heads", "sample", "gemm_other",
                                              "dac_embed", "dac_conv7", "dac_conv1", "dac_convt", "dac_final", "dac_resunit", "kokoro_conv_mfma"};
extern "C" const char *tts_hip_kclass_name(int k) { return (k >= 0 && k < TTS_HIP_K_COUNT) ? KNAMES[k] : "?"; }

extern "C" int tts_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { set_err("hipGetDeviceCount failed (no ROCm device visible)"); return 0; }
    return n;
}

extern "C" tts_hip_ctx *tts_hip_create(int device, const tts_hip_desc *desc) {
    if (!desc || desc->struct_size != sizeof(tts_hip_desc)) { set_err("tts_hip_create: bad desc (struct_size mismatch)"); return nullptr; }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { set_err("tts_hip_create: no HIP device available; this library has no CPU fallback"); return nullptr; }
    if (device < 0 || device >= n) { set_err("tts_hip_create: device %d out of range (%d devices)", device, n); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { set_err("hipSetDevice(%d) failed", device); return nullptr; }
    tts_hip_ctx *c = new tts_hip_ctx;
    c->device = device;
    c->d = *desc;
    if (c->d.max_seqs == 0) c->d.max_seqs = 1;
    if (c->d.kv_type != TTS_HIP_F16) c->d.kv_type = TTS_HIP_F32;
    c->has_parler = !(desc->flags & TTS_HIP_FLAG_NO_PARLER);
    c->has_dac = !(desc->flags & TTS_HIP_FLAG_NO_DAC);
    {
        // the decoder's short dependent kernels get the high-priority queue, the codec's chip-filling convolutions the
        // low one: when contexts share a GPU, one context's step is not held up behind another's conv workgroups
        int least = 0, greatest = 0;
        const char *e = getenv("TTS_HIP_STREAM_PRIO");
        const bool prio = !(e && atoi(e) == 0) && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest;
        hipError_t rc = prio ? hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, greatest)
                             : hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (rc != hipSuccess) { set_err("hipStreamCreate failed"); delete c; return nullptr; }
        if (prio && hipStreamCreateWithPriority(&c->dac_stream, hipStreamNonBlocking, least) != hipSuccess) c->dac_stream = nullptr;
    }
    const char *ns = getenv("TTS_HIP_ATTN_NSPLIT");
    if (ns) c->attn_nsplit_override = atoi(ns);
    if (const char *e = getenv("TTS_HIP_ATTN_FUSED")) c->attn_fused = atoi(e) != 0;
    if (const char *e = getenv("TTS_HIP_B1_FC2_SPLIT")) c->b1_fc2_split = atoi(e) != 0;
    if (const char *e = getenv("TTS_HIP_B1_DEFER_COMBINE")) c->b1_defer_combine = atoi(e) != 0;
    const char *lf = getenv("TTS_HIP_LN_FUSE_MAX");
    if (lf) c->ln_fuse_max = std::max(0, std::min(32, atoi(lf)));
    if (const char *e = getenv("TTS_HIP_LN_WAVES")) c->ln_waves = std::max(1, std::min(4, atoi(e)));
    if (const char *e = getenv("TTS_HIP_KSPLIT_BIG")) c->ksplit_big = std::max(1, std::min(8, atoi(e)));
    if (const char *e = getenv("TTS_HIP_DAC_VARIANT")) c->dac_variant = atoi(e);
    if (const char *e = getenv("TTS_HIP_DAC_PAD")) c->dac_pad = atoi(e);
    if (const char *e = getenv("TTS_HIP_DAC_C192")) c->dac_c192 = atoi(e);
    if (const char *e = getenv("TTS_HIP_DAC_CONV1_DIRECT")) c->dac_conv1_direct = atoi(e) != 0;
    if (const char *e = getenv("TTS_HIP_KOKORO_MFMA")) c->kk_mfma = atoi(e) != 0;
    if (const char *e = getenv("TTS_HIP_KOKORO_LSTM_SPLIT")) c->kk_lstm_split = atoi(e) != 0;
    if (const char *e = getenv("TTS_HIP_DAC_GROUP")) c->dac_group = std::max(1, atoi(e));
    if (const char *e = getenv("TTS_HIP_DAC_ALPHA_TAB")) c->dac_alpha_tab = std::max(0, std::min(2, atoi(e)));
    if (const char *e = getenv("TTS_HIP_DAC_PRIO")) c->dac_prio = atoi(e);
    if (const char *e = getenv("TTS_HIP_DAC_BF16X3")) c->dac_b3 = std::max(0, atoi(e));
    if (const char *e = getenv("TTS_HIP_DAC_B3_VARIANT")) c->dac_b3_variant = atoi(e);
    if (const char *e = getenv("TTS_HIP_DAC_FUSE")) c->dac_fuse = atoi(e);
    if (const char *e = getenv("TTS_HIP_DAC_CONVT_B3")) c->dac_convt_b3 = atoi(e);
    if (const char *e = getenv("TTS_HIP_DAC_PLANES")) c->dac_planes = atoi(e);
    if (const char *e = getenv("TTS_HIP_DAC_TAP7")) c->dac_tap7 = atoi(e);
    if (const char *e = getenv("TTS_HIP_GEN_COMPACT")) c->gen_compact = atoi(e) != 0;
    if (const char *e = getenv("TTS_HIP_DAC_P_VARIANT")) c->dac_p_variant = atoi(e);
    if (const char *e = getenv("TTS_HIP_DAC_K1_VARIANT")) c->dac_k1_variant = atoi(e);
    if (const char *e = getenv("TTS_HIP_DAC_LDS_RESERVE_KB")) c->dac_lds_reserve_kb = std::max(0, std::min(96, atoi(e)));
    if (const char *e = getenv("TTS_HIP_GEMM_ROWS_PER_WG")) { const int v = atoi(e); c->gemm_rows_per_wg = v <= 0 ? 0 : (v <= 16 ? 16 : (v <= 32 ? 32 : 64)); }
    if (const char *e = getenv("TTS_HIP_GEMM_NGS_MAX")) c->gemm_ngs_max = std::max(1, std::min(16, atoi(e)));
    if (const char *e = getenv("TTS_HIP_ATTN_SHORT")) c->attn_short = atoi(e) != 0;
    if (const char *e = getenv("TTS_HIP_TILE_MIN_ROWS")) c->tile_min_rows = std::max(0, atoi(e));
    if (const char *e = getenv("TTS_HIP_TILE_FORCE")) c->tile_force = atoi(e);
    if (const char *e = getenv("TTS_HIP_TILE_KS")) c->tile_force_ks = atoi(e);
    if (const char *e = getenv("TTS_HIP_TILE_DEEP")) c->tile_deep = atoi(e);
    if (const char *e = getenv("TTS_HIP_ATTN_SPLIT")) c->attn_split_max = std::max(1, std::min(16, atoi(e)));
    if (const char *e = getenv("TTS_HIP_Q_FUSE_MAX")) c->q_fuse_max = std::max(0, std::min(16, atoi(e)));
    if (const char *e = getenv("TTS_HIP_GEMV_ROWS")) c->gemv_rows = atoi(e) != 0;
    if (const char *e = getenv("TTS_HIP_LLAMA_GRAPH")) c->llama_graph = atoi(e) != 0;
    if (const char *e = getenv("TTS_HIP_GEMV_STREAM")) c->gemv_stream = atoi(e) != 0;
    if (const char *e = getenv("TTS_HIP_Q4_LDS")) c->q4_lds = atoi(e) != 0;
    if (const char *e = getenv("TTS_HIP_Q4_ROPE")) c->q4_rope = atoi(e) != 0;
    if (const char *e = getenv("TTS_HIP_Q4_SILU")) c->q4_silu = atoi(e) != 0;
    if (const char *e = getenv("TTS_HIP_Q4_RMS")) c->q4_rms = atoi(e) != 0;
    if (const char *e = getenv("TTS_HIP_Q4_NATIVE")) c->q4_native = atoi(e) != 0;
    return c;
}

static void free_dev(void *p) { if (p) (void) hipFree(p); }
static std::mutex g_dac_pass_mutex[64];
static void dac_buffers_release(tts_hip_ctx *c) {
    if (!c->dac_buf_user) return;
    std::lock_guard<std::mutex> lock(g_dac_pass_mutex[(unsigned) c->device % 64]);
    DacBuffers &B = g_dac_buffers[(unsigned) c->device % 64];
    c->dac_buf_user = false;
    if (--B.users > 0) return;
    for (int i = 0; i < 3; i++) { free_dev(B.dbuf[i]); B.dbuf[i] = nullptr; }
    free_dev(B.dplanes); B.dplanes = nullptr;
    free_dev(B.d_codes); B.d_codes = nullptr;
    if (B.h_pcm) { (void) hipHostFree(B.h_pcm); B.h_pcm = nullptr; }
    B.dbuf_elems = B.cap_codes = B.h_pcm_elems = 0;
    B.users = 0;
}

extern "C" void tts_hip_destroy(tts_hip_ctx *c) {
    if (!c) return;
    (void) hipSetDevice(c->device);
    (void) hipStreamSynchronize(c->stream);
    for (auto &g : c->graphs) (void) hipGraphExecDestroy(g.second);
    for (auto &t : c->tensors) free_dev(t.second.tmp);
    if (!c->arena_external) free_dev(c->arena);
    free_dev(c->kcache); free_dev(c->vcache); free_dev(c->x); free_dev(c->q); free_dev(c->att); free_dev(c->u32);
    free_dev(c->u16); free_dev(c->xn16); free_dev(c->att16); free_dev(c->partials); free_dev(c->aq); free_dev(c->ad); free_dev(c->d_uniforms); free_dev(c->l_cand); free_dev(c->l_smp); free_dev(c->d_pen); free_dev(c->d_last); free_dev(c->d_repc);
    free_dev(c->l_x); free_dev(c->l_xn); free_dev(c->l_qkv); free_dev(c->l_att); free_dev(c->l_gu); free_dev(c->l_g); free_dev(c->l_logits); free_dev(c->l_parts);
    free_dev(c->attn_part); free_dev(c->kk_stuck); free_dev(c->kk_pool);
    free_dev(c->l_kc); free_dev(c->l_vc); free_dev(c->l_ids); free_dev(c->l_pos); free_dev(c->l_tok);
    for (void *p : c->q4_bufs) free_dev(p);
    for (float *p : {c->di_ex, c->di_exn, c->di_eqkv, c->di_eatt, c->di_egu, c->di_eg, c->di_ek, c->di_ev, c->di_ckv, c->di_ck, c->di_cv, c->di_k, c->di_v, c->di_x,
                     c->di_xn, c->di_qkv, c->di_q, c->di_att, c->di_gu, c->di_g, c->di_parts, c->di_logits, c->di_guided})
        free_dev(p);
    for (uint32_t *p : {c->di_tok, c->di_epos, c->di_eseq, c->di_kbeg, c->di_kend, c->di_ids, c->di_pos, c->di_seq, c->di_cend, c->di_stok, c->di_loop, c->di_hist}) free_dev(p);
    free_dev(c->di_e16);
    for (int i = 0; i < 3; i++) free_dev(c->sbuf[i]);
    free_dev(c->s_noise); free_dev(c->s_codes);
    free_dev(c->t5_bucket); free_dev(c->t5_x); free_dev(c->t5_qkv); free_dev(c->t5_att); free_dev(c->t5_ug); free_dev(c->t5_g); free_dev(c->t5_y); free_dev(c->t5_ids); free_dev(c->logits); free_dev(c->part); free_dev(c->attn_cnt); free_dev(c->b1_stamps); free_dev(c->dbg); free_dev(c->d_ids); free_dev(c->d_pos);
    free_dev(c->d_seq); free_dev(c->d_gather); free_dev(c->d_tok); free_dev(c->d_step); free_dev(c->d_steps_done); free_dev(c->d_tokens_out);
    free_dev(c->d_eos); free_dev(c->d_frames);
    dac_buffers_release(c);
    for (auto &pw : c->packed) free_dev(pw.second);
    for (auto &pw : c->packed16) free_dev(pw.second);
    for (auto &pw : c->packed_b3) free_dev(pw.second);
    free_dev(c->cond_text_enc); free_dev(c->cond_cross_kv);
    for (auto &pw : c->packed_ru) free_dev(pw.second);
    for (auto &pw : c->packed_ct) free_dev(pw.second);
    for (auto &pw : c->packed_p) free_dev(pw.second);
    if (c->h_ids) (void) hipHostFree(c->h_ids);
    if (c->h_pos) (void) hipHostFree(c->h_pos);
    if (c->h_seq) (void) hipHostFree(c->h_seq);
    if (c->h_tok) (void) hipHostFree(c->h_tok);
    if (c->h_logits) (void) hipHostFree(c->h_logits);
    if (c->h_di) (void) hipHostFree(c->h_di);
    for (auto &e : c->prof_events) { (void) hipEventDestroy(e.a); (void) hipEventDestroy(e.b); }
    (void) hipStreamDestroy(c->stream);
    if (c->dac_stream) (void) hipStreamDestroy(c->dac_stream);
    delete c;
}

// ------------------------------------------------------------------------------------------------
// upload
// ------------------------------------------------------------------------------------------------
static bool ends_with(const std::string &s, const char *suf) {
    const size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}
static bool starts_with(const std::string &s, const char *pre) { return s.compare(0, strlen(pre), pre) == 0; }

// Which uploaded tensors stay fp16 on the device: the big decoder matrices and embedding tables.
// Norm vectors, positional table, text encoding and the whole DAC are kept fp32.
// decoder matrices that are only ever used through mul_mat (not get_rows): eligible for the integer path
static bool is_t5_matmul(const std::string &name) {
    return starts_with(name, "t5encoder.") && (ends_with(name, ".attn_q") || ends_with(name, ".attn_k") || ends_with(name, ".attn_v") ||
                                                ends_with(name, ".attn_o") || ends_with(name, ".ffn_up") || ends_with(name, ".ffn_gate") ||
                                                ends_with(name, ".ffn_down") || name == "t5encoder.down_proj");
}
static bool is_llama_matmul(const std::string &name) {
    return starts_with(name, "orpheus.") && (ends_with(name, "_proj") || name == "orpheus.lm_head");
}
static bool is_dia_matmul(const std::string &name) {
    return starts_with(name, "dia.") && (ends_with(name, "_proj") || ends_with(name, ".gate") || ends_with(name, ".up") || ends_with(name, ".wo") ||
                                         name.find(".heads.") != std::string::npos);
}
static bool is_matmul_weight(const std::string &name) {
    if (is_t5_matmul(name) || is_llama_matmul(name) || is_dia_matmul(name)) return true;
    return starts_with(name, "decoder.") && (ends_with(name, "_proj.weight") || ends_with(name, "fc1.weight") ||
                                              ends_with(name, "fc2.weight") || ends_with(name, "weight.head"));
}

// Q4_0 / Q5_0 / Q8_0 blocks -> int8 block integers + fp16 scales (exact: the integers of the block formats)
static int expand_q_to_i8(int type, const void *src, int64_t n, int8_t *q, uint16_t *d) {
    const uint8_t *p = (const uint8_t *) src;
    for (int64_t b = 0; b < n / 32; b++, q += 32) {
        memcpy(&d[b], p, 2);
        if (type == TTS_HIP_Q4_0) {
            const uint8_t *qs = p + 2;
            for (int j = 0; j < 16; j++) { q[j] = (int8_t) ((int) (qs[j] & 0xF) - 8); q[j + 16] = (int8_t) ((int) (qs[j] >> 4) - 8); }
            p += 18;
        } else if (type == TTS_HIP_Q5_0) {
            uint32_t qh;
            memcpy(&qh, p + 2, 4);
            const uint8_t *qs = p + 6;
            for (int j = 0; j < 16; j++) {
                q[j] = (int8_t) ((int) ((qs[j] & 0xF) | (((qh >> j) & 1) << 4)) - 16);
                q[j + 16] = (int8_t) ((int) ((qs[j] >> 4) | (((qh >> (j + 16)) & 1) << 4)) - 16);
            }
            p += 22;
        } else if (type == TTS_HIP_Q8_0) {
            memcpy(q, p + 2, 32);
            p += 34;
        } else return -1;
    }
    return 0;
}

static bool keeps_f16(const std::string &name) {
    if (is_t5_matmul(name) || is_llama_matmul(name) || is_dia_matmul(name)) return true;
    if (!starts_with(name, "decoder.")) return false;
    if (name.find("layer_norm") != std::string::npos) return false;
    if (name == "decoder.positional_embed" || name == "decoder.text_encoding") return false;
    return true;
}

extern "C" int tts_hip_upload(tts_hip_ctx *c, const char *name_c, int type, int n_dims, const int64_t *ne, const void *host) {
    if (!c || !name_c || !ne) return set_err("tts_hip_upload: null argument");
    if (c->finalized) return set_err("tts_hip_upload(%s): context already finalized", name_c);
    HIPCHK(hipSetDevice(c->device));
    std::string name(name_c);
    if (c->has_kokoro) {
        if (!starts_with(name, "kokoro.")) return 0;     // kokoro_runner::assign_weight asserts the prefix (kokoro/model.cpp:1327-1332)
    } else if (c->has_dia) {
        if (!starts_with(name, "dia.")) return 0;        // "audio_encoder.*" belongs to the codec context (dia/model.cpp:892-898)
    } else if (c->has_llama) {
        if (!starts_with(name, "orpheus.")) return 0;    // "snac.*" belongs to the codec context (orpheus/model.cpp:430-438)
    } else if (c->has_snac) {
        if (!starts_with(name, "snac.")) return 0;       // the Orpheus GGUF also carries "orpheus.*" (orpheus/model.cpp)
        if (name.find(".in_proj") != std::string::npos) return 0;
    } else if (c->has_t5) {
        if (!starts_with(name, "t5encoder.")) return 0;  // assign_to_t5_encoder ignores other top levels (t5/model.cpp:107-109)
    } else if (!starts_with(name, "decoder.") && !starts_with(name, "audio_encoder.")) {
        fprintf(stderr, "tts_hip: ignoring unhandled tensor '%s'\n", name_c);  // model.cpp:506
        return 0;
    }
    if (starts_with(name, "decoder.") && !c->has_parler) return 0;
    if (starts_with(name, "audio_encoder.") && !c->has_dac) return 0;
    if (name.find(".in_proj") != std::string::npos) return 0;  // unused quantizer input projection (gnac.cpp:126-130)
    if (n_dims < 1 || n_dims > 4) return set_err("tts_hip_upload(%s): n_dims=%d", name_c, n_dims);
    Tensor t;
    t.n_dims = n_dims;
    for (int i = 0; i < n_dims; i++) t.ne[i] = ne[i];
    const int64_t n = t.nelem();
    const size_t src_bytes = type_row_bytes(type, t.ne[0]) * (size_t) (n / t.ne[0]);
    if (src_bytes == 0) return set_err("tts_hip_upload(%s): unsupported ggml type %d", name_c, type);
    const bool keep16 = (type == TTS_HIP_F16) && keeps_f16(name);
    const bool quant = type == TTS_HIP_Q4_0 || type == TTS_HIP_Q5_0 || type == TTS_HIP_Q8_0;
    if (c->has_t5 && starts_with(name, "t5encoder.")) { /* falls through to the common storage rules */ }
    const bool q8i = quant && is_matmul_weight(name) && n_dims == 2 && (t.ne[0] % 256 == 0) && !(c->d.flags & TTS_HIP_FLAG_DEQUANT_Q) &&
                     !(c->d.flags & TTS_HIP_FLAG_VALU_GEMM);
    t.type = q8i ? TTS_HIP_Q8I : (keep16 ? TTS_HIP_F16 : TTS_HIP_F32);
    t.nbytes = q8i ? (size_t) n + (size_t) (n / 32) * 2 : (size_t) n * (keep16 ? 2 : 4);
    t.src_type = type;
    t.has_data = host != nullptr;
    if (host) {
        HIPCHK(hipMalloc(&t.tmp, t.nbytes));
        // every failure below releases the staging allocation (a server that retries loads must not accumulate device memory)
        struct TmpGuard { void *&p; bool armed = true; ~TmpGuard() { if (armed) { free_dev(p); p = nullptr; } } } guard{t.tmp};
        if (q8i) {
            std::vector<int8_t> q((size_t) n);
            std::vector<uint16_t> d((size_t) (n / 32));
            if (expand_q_to_i8(type, host, n, q.data(), d.data()) != 0) return set_err("tts_hip_upload(%s): block expansion failed", name_c);
            HIPCHK(hipMemcpy(t.tmp, q.data(), (size_t) n, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy((char *) t.tmp + n, d.data(), (size_t) (n / 32) * 2, hipMemcpyHostToDevice));
        } else if (keep16 || type == TTS_HIP_F32) {
            HIPCHK(hipMemcpy(t.tmp, host, t.nbytes, hipMemcpyHostToDevice));
        } else {
            std::vector<float> f((size_t) n);
            if (dequant_to_f32(type, host, f.data(), n) != 0) return set_err("tts_hip_upload(%s): dequantisation failed", name_c);
            HIPCHK(hipMemcpy(t.tmp, f.data(), t.nbytes, hipMemcpyHostToDevice));
        }
        guard.armed = false;
    }
    auto it = c->tensors.find(name);
    if (it != c->tensors.end()) free_dev(it->second.tmp);
    c->tensors[name] = t;
    c->planned = false;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// arena planning
// ------------------------------------------------------------------------------------------------
struct Planner {
    tts_hip_ctx *c;
    size_t cur = 0;
    std::string err;
    size_t alloc(size_t bytes) {
        cur = (cur + 255) & ~(size_t) 255;
        const size_t o = cur;
        cur += bytes;
        return o;
    }
    const Tensor *get(const std::string &n) {
        auto it = c->tensors.find(n);
        if (it == c->tensors.end()) { if (err.empty()) err = "missing tensor '" + n + "'"; return nullptr; }
        return &it->second;
    }
    size_t place(const std::string &n) {
        const Tensor *t = get(n);
        if (!t) return 0;
        const size_t o = alloc(t->nbytes);
        c->copies.push_back({o, n});
        return o;
    }
    size_t place_f32(const std::string &n) {
        const Tensor *t = get(n);
        if (t && t->type != TTS_HIP_F32 && err.empty()) err = "tensor '" + n + "' must be fp32";
        return place(n);
    }
    W mat(const std::string &n) { return fused({n}); }
    // several same-shaped matrices stacked along N
    W fused(const std::vector<std::string> &names, int pad_rows_to = 1) {
        W w;
        for (size_t i = 0; i < names.size(); i++) {
            const Tensor *t = get(names[i]);
            if (!t) return w;
            if (i == 0) {
                w.type = t->type; w.K = t->ne[0]; w.N = 0;
                w.src_q4 = true;
                cur = (cur + 255) & ~(size_t) 255;
                w.off = cur;
            } else if (t->type != w.type || t->ne[0] != w.K) {
                if (err.empty()) err = "tensors fused with '" + names[0] + "' differ in type/shape: '" + names[i] + "'";
                return w;
            }
            const size_t main_bytes = t->type == TTS_HIP_Q8I ? (size_t) t->nelem() : t->nbytes;
            w.src_q4 = w.src_q4 && t->src_type == TTS_HIP_Q4_0;
            c->copies.push_back({cur, names[i], 0, main_bytes});
            cur += main_bytes;
            w.N += t->nelem() / t->ne[0];
        }
        // rows up to a multiple of pad_rows_to exist in the arena (uninitialised: their outputs are never read)
        const int64_t n_pad = (w.N + pad_rows_to - 1) / pad_rows_to * pad_rows_to - w.N;
        if (n_pad) cur += (size_t) n_pad * w.K * (w.type == TTS_HIP_Q8I ? 1 : (w.type == TTS_HIP_F16 ? 2 : 4));
        if (w.type == TTS_HIP_Q8I) {  // the block scales of the stacked matrices, [N_total][K/32] fp16
            cur = (cur + 255) & ~(size_t) 255;
            w.soff = cur;
            for (auto &nm : names) {
                const Tensor *t = get(nm);
                const size_t sb = (size_t) (t->nelem() / 32) * 2;
                c->copies.push_back({cur, nm, (size_t) t->nelem(), sb});
                cur += sb;
            }
            cur += (size_t) n_pad * (w.K / 32) * 2;
        }
        w.N += n_pad;
        return w;
    }
};

static int plan(tts_hip_ctx *c) {
    if (c->planned) return 0;
    Planner P{c};
    c->copies.clear();
    const tts_hip_desc &d = c->d;
    if (c->has_parler) {
        c->H = d.hidden_size; c->L = d.n_layers; c->NH = d.n_attn_heads; c->NO = d.n_output_heads;
        c->V = d.output_vocab_size; c->NCTX = d.max_ctx_length; c->E = d.n_encode_length;
        c->KVPOS = d.kv_positions ? (int) std::min<uint32_t>(d.kv_positions, d.max_ctx_length) : (int) d.max_ctx_length;
        if (c->H <= 0 || c->L <= 0 || c->NH <= 0 || c->NO <= 0 || c->V <= 0 || c->NCTX <= 0) return set_err("plan: incomplete Parler hyper-parameters in desc");
        if (c->H / c->NH != 64 || c->H % c->NH) return set_err("plan: head size %d unsupported (kernels are specialised for 64, Parler-Mini/Large)", c->H / c->NH);
        if (c->H % 16 || c->V % 16) return set_err("plan: hidden size and vocab must be multiples of 16");
        if (c->NO > 16) return set_err("plan: at most 16 output heads");
        c->ECAP = std::max(c->E, 512);  // max_encode_length, model.h:69
        c->embed_prompts = P.mat("decoder.embed_prompts");
        c->PV = (int) c->embed_prompts.N;
        { const Tensor *t = P.get("decoder.positional_embed"); c->NPOS = t ? (int) (t->nelem() / t->ne[0]) : 0; }
        c->pos_embed = P.place_f32("decoder.positional_embed");
        c->ln_w = P.place_f32("decoder.layer_norm.weight");
        c->ln_b = P.place_f32("decoder.layer_norm.bias");
        std::vector<std::string> en, hn;
        for (int i = 0; i < c->NO; i++) {
            en.push_back("decoder.embed_tokens." + std::to_string(i) + ".weight");
            hn.push_back("decoder.lm_heads." + std::to_string(i) + ".weight.head");
        }
        c->embed_tokens = P.fused(en);
        c->EROWS = (int) (c->embed_tokens.N / c->NO);
        c->heads = P.fused(hn);
        c->layers.assign(c->L, PLayer{});
        for (int l = 0; l < c->L; l++) {
            const std::string p = "decoder.layers." + std::to_string(l) + ".";
            PLayer &y = c->layers[l];
            y.sa_w = P.place_f32(p + "self_attn_layer_norm.weight");
            y.sa_b = P.place_f32(p + "self_attn_layer_norm.bias");
            y.qkv = P.fused({p + "self_attn.q_proj.weight", p + "self_attn.k_proj.weight", p + "self_attn.v_proj.weight"});
            y.o = P.mat(p + "self_attn.out_proj.weight");
            if (d.use_cross_attn) {
                y.ca_w = P.place_f32(p + "encoder_attn_layer_norm.weight");
                y.ca_b = P.place_f32(p + "encoder_attn_layer_norm.bias");
                y.cq = P.mat(p + "encoder_attn.q_proj.weight");
                y.ck = P.mat(p + "encoder_attn.k_proj.weight");
                y.cv = P.mat(p + "encoder_attn.v_proj.weight");
                y.co = P.mat(p + "encoder_attn.out_proj.weight");
            }
            y.f_w = P.place_f32(p + "final_layer_norm.weight");
            y.f_b = P.place_f32(p + "final_layer_norm.bias");
            y.fc1 = P.mat(p + "fc1.weight");
            y.fc2 = P.mat(p + "fc2.weight");
        }
        c->F = c->layers.empty() ? 0 : (int) c->layers[0].fc1.N;
        c->all_q8i = c->heads.type == TTS_HIP_Q8I;
        for (const PLayer &y : c->layers) {
            c->all_q8i = c->all_q8i && y.qkv.type == TTS_HIP_Q8I && y.o.type == TTS_HIP_Q8I && y.fc1.type == TTS_HIP_Q8I && y.fc2.type == TTS_HIP_Q8I;
            if (d.use_cross_attn) c->all_q8i = c->all_q8i && y.cq.type == TTS_HIP_Q8I && y.co.type == TTS_HIP_Q8I;
        }
        if (d.use_cross_attn) {
            // voice-prompt encoding gets ECAP rows so that update_conditional_prompt fits (model.cpp:129-136)
            const Tensor *t = P.get("decoder.text_encoding");
            if (t) {
                if ((int) (t->nelem() / t->ne[0]) != c->E && P.err.empty()) P.err = "decoder.text_encoding rows != n_encode_length";
                c->text_enc = P.alloc((size_t) c->ECAP * c->H * 4);
                c->copies.push_back({c->text_enc, "decoder.text_encoding"});
            }
            c->cross_kv = P.alloc((size_t) c->L * 2 * c->ECAP * c->H * 4);
        }
    }
    if (c->has_llama) {
        // orpheus_model (orpheus/model.h:24-52, tensor names orpheus/model.cpp:11-60)
        const tts_hip_orpheus_desc &ld = c->lm;
        c->H = (int) ld.hidden_size; c->L = (int) ld.n_layers; c->NH = (int) ld.n_attn_heads;
        if (c->H <= 0 || c->L <= 0 || c->NH <= 0 || ld.n_kv_heads == 0 || ld.n_ctx == 0) return set_err("plan: incomplete Orpheus hyper-parameters");
        if (ld.head_dim != 128) return set_err("plan: Orpheus head_dim %u unsupported (128, orpheus/model.h:28)", ld.head_dim);
        if (c->NH % (int) ld.n_kv_heads) return set_err("plan: attn_heads %% kv_attn_heads != 0");
        c->l_kvH = (int) (ld.n_kv_heads * ld.head_dim);
        c->l_embd = P.place_f32("orpheus.embed_tokens");
        c->l_out_norm = P.place_f32("orpheus.norm");
        c->l_ropef = P.place_f32("orpheus.rope_frequencies");
        { const Tensor *t = P.get("orpheus.lm_head"); c->l_V = t ? (int) (t->nelem() / t->ne[0]) : 0; }
        if (ld.vocab_size && (int) ld.vocab_size != c->l_V && P.err.empty()) P.err = "orpheus.vocab_size disagrees with lm_head";
        c->l_head = P.fused({"orpheus.lm_head"}, 16);
        c->l_Vpad = (int) c->l_head.N;
        c->l_layers.assign(c->L, tts_hip_ctx::LLayer{});
        for (int l = 0; l < c->L; l++) {
            const std::string p = "orpheus.layers." + std::to_string(l) + ".";
            auto &y = c->l_layers[l];
            y.in_norm = P.place_f32(p + "input_layernorm");
            y.qkv = P.fused({p + "self_attn.q_proj", p + "self_attn.k_proj", p + "self_attn.v_proj"});
            y.o = P.mat(p + "self_attn.o_proj");
            y.post_norm = P.place_f32(p + "post_attention_layernorm");
            y.gu = P.fused({p + "mlp.gate_proj", p + "mlp.up_proj"});
            y.down = P.mat(p + "mlp.down_proj");
        }
        c->F = c->l_layers.empty() ? 0 : (int) (c->l_layers[0].gu.N / 2);
        if (!c->l_layers.empty() && (int) c->l_layers[0].qkv.N != (c->NH + 2 * (int) ld.n_kv_heads) * (int) ld.head_dim && P.err.empty())
            P.err = "orpheus q/k/v projection shapes disagree with attn_heads / kv_attn_heads / head_dim";
    }
    if (c->has_dia) {
        // dia_model (dia/model.h:16-84, assign_weight dia/model.cpp:3-139)
        const tts_hip_dia_desc &dd = c->dia;
        const int HD = (int) dd.head_dim;
        c->H = (int) dd.dec_hidden_size; c->L = (int) dd.dec_layers; c->NH = (int) dd.dec_attn_heads; c->NO = (int) dd.n_output_heads;
        c->di_EH = (int) dd.enc_hidden_size; c->di_A = c->NH * HD; c->di_kvH = (int) dd.dec_kv_heads * HD;
        if (c->H <= 0 || c->L <= 0 || c->NH <= 0 || dd.dec_kv_heads == 0 || dd.enc_layers == 0 || dd.enc_attn_heads == 0 || dd.max_ctx == 0 || dd.max_gen == 0 ||
            c->NO <= 0 || c->NO > 16 || dd.output_vocab_size == 0)
            return set_err("plan: incomplete Dia hyper-parameters");
        if (HD != 128) return set_err("plan: Dia head size %d unsupported (128, dia/model.h:74)", HD);
        if (c->NH % (int) dd.dec_kv_heads) return set_err("plan: Dia attn_heads %% k/v groups != 0");
        if ((int) dd.enc_attn_heads * HD != c->H || c->di_A != c->H)
            return set_err("plan: Dia attention width (heads x head size) must equal the decoder hidden size (dia/model.cpp:410,593)");
        c->di_enc_embd = P.place_f32("dia.encoder.embedding");
        { const Tensor *t = P.get("dia.encoder.embedding"); c->di_evocab = t ? (int) (t->nelem() / t->ne[0]) : 0;
          if (t && (int) t->ne[0] != c->di_EH && P.err.empty()) P.err = "dia.encoder.embedding width != enc_hidden_size"; }
        c->di_enc_norm = P.place_f32("dia.encoder.norm");
        c->di_dec_norm = P.place_f32("dia.decoder.norm");
        std::vector<std::string> hn;
        for (int i = 0; i < c->NO; i++) {
            c->di_embd[i] = P.place_f32("dia.decoder.embeddings." + std::to_string(i));
            hn.push_back("dia.decoder.heads." + std::to_string(i));
        }
        { const Tensor *t = P.get(hn[0]); c->di_V = t ? (int) (t->nelem() / t->ne[0]) : 0; }
        if (c->di_V != (int) dd.output_vocab_size && P.err.empty()) P.err = "dia.decoder.output_vocab_size disagrees with dia.decoder.heads.0";
        c->di_heads = P.fused(hn, 16);
        c->di_Vpad = (int) c->di_heads.N;
        c->di_enc.assign(dd.enc_layers, tts_hip_ctx::DiaEnc{});
        for (uint32_t l = 0; l < dd.enc_layers; l++) {
            const std::string p = "dia.encoder.layers." + std::to_string(l) + ".";
            auto &y = c->di_enc[l];
            y.sa_norm = P.place_f32(p + "pre_sa_norm");
            y.qkv = P.fused({p + "q_proj", p + "k_proj", p + "v_proj"});
            y.o = P.mat(p + "o_proj");
            y.mlp_norm = P.place_f32(p + "post_sa_norm");
            y.gu = P.fused({p + "gate", p + "up"});
            y.out = P.mat(p + "wo");
        }
        c->di_dec.assign(dd.dec_layers, tts_hip_ctx::DiaDec{});
        for (uint32_t l = 0; l < dd.dec_layers; l++) {
            const std::string p = "dia.decoder.layers." + std::to_string(l) + ".";
            auto &y = c->di_dec[l];
            y.sa_norm = P.place_f32(p + "pre_sa_norm");
            y.sqkv = P.fused({p + "self_q_proj", p + "self_k_proj", p + "self_v_proj"});
            y.so = P.mat(p + "self_o_proj");
            y.ca_norm = P.place_f32(p + "pre_ca_norm");
            y.cq = P.mat(p + "cross_q_proj");
            y.ckv = P.fused({p + "cross_k_proj", p + "cross_v_proj"});
            y.co = P.mat(p + "cross_o_proj");
            y.mlp_norm = P.place_f32(p + "pre_mlp_norm");
            y.gu = P.fused({p + "gate", p + "up"});
            y.out = P.mat(p + "wo");
        }
        c->di_EF = c->di_enc.empty() ? 0 : (int) (c->di_enc[0].gu.N / 2);
        c->di_DF = c->di_dec.empty() ? 0 : (int) (c->di_dec[0].gu.N / 2);
        c->F = c->di_DF;
        if (P.err.empty() && !c->di_enc.empty() && !c->di_dec.empty()) {
            if ((int) c->di_enc[0].qkv.N != 3 * c->di_A || (int) c->di_enc[0].qkv.K != c->di_EH) P.err = "dia encoder q/k/v projection shapes disagree with the hyper-parameters";
            else if ((int) c->di_dec[0].sqkv.N != c->di_A + 2 * c->di_kvH) P.err = "dia decoder self q/k/v projection shapes disagree with attn_heads / query_heads / head size";
            else if ((int) c->di_dec[0].ckv.N != 2 * c->di_A || (int) c->di_dec[0].ckv.K != c->di_EH) P.err = "dia decoder cross k/v projection shapes disagree with the hyper-parameters";
            else if (c->di_EF > 4096 && !(c->d.flags & TTS_HIP_FLAG_VALU_GEMM)) P.err = "dia encoder feed-forward width above 4096 is not supported";
        }
    }
    if (c->has_kokoro) {
        // every tensor by name, fp32 (kokoro_model::assign_weight kokoro/model.cpp:413-428 walks the same names)
        c->k_tensors.clear();
        std::vector<std::string> names;
        for (auto &kv : c->tensors)
            if (starts_with(kv.first, "kokoro.")) names.push_back(kv.first);
        std::sort(names.begin(), names.end());
        for (auto &nm : names) {
            tts_hip_ctx::KTensor kt;
            kt.off = P.place_f32(nm);
            const Tensor *t = P.get(nm);
            for (int d = 0; d < 4; d++) kt.ne[d] = t->ne[d];
            c->k_tensors[nm] = kt;
        }
        if (names.empty() && P.err.empty()) P.err = "no kokoro.* tensors were uploaded";
    }
    if (c->has_snac) {
        // snac_model (snac_model.h:10-40, assign_weight snac_model.cpp:50-84, layer tensors gnac.cpp:9-34)
        const tts_hip_snac_desc &sd = c->snac;
        std::vector<std::string> cb, pw, pb;
        for (uint32_t i = 0; i < sd.n_codebooks; i++) {
            const std::string p = "snac.quantizers." + std::to_string(i) + ".";
            cb.push_back(p + "codebook.weight"); pw.push_back(p + "out_proj.weight"); pb.push_back(p + "out_proj.bias");
        }
        const Tensor *t0 = P.get(cb[0]);
        const Tensor *w0 = P.get(pw[0]);
        if (t0 && w0) {
            c->s_cbdim = (int) t0->ne[0]; c->s_cbsize = (int) t0->ne[1];
            c->s_latent = (int) (w0->nelem() / c->s_cbdim);
        }
        c->s_codebook = P.fused(cb).off; c->s_projw = P.fused(pw).off; c->s_projb = P.fused(pb).off;
        c->s_inw = P.place_f32("snac.in.weight"); c->s_inb = P.place_f32("snac.in.bias");
        { const Tensor *t = P.get("snac.up.weight"); c->s_c0 = t ? (int) t->ne[2] : 0; }
        c->s_upw = P.place_f32("snac.up.weight"); c->s_upb = P.place_f32("snac.up.bias");
        c->sblocks.assign(sd.n_blocks, tts_hip_ctx::SBlock{});
        c->s_up = 1;
        int C = c->s_c0;
        for (uint32_t i = 0; i < sd.n_blocks; i++) {
            const std::string p = "snac.layers." + std::to_string(i) + ".";
            auto &b = c->sblocks[i];
            b.stride = (int) sd.stride[i]; b.padding = (int) sd.padding[i];
            const Tensor *t = P.get(p + "weight");  // ne = [K, Cout, Cin]
            if (t) {
                b.cin = (int) t->ne[2]; b.cout = (int) t->ne[1];
                if ((int) t->ne[0] != 2 * b.stride && P.err.empty()) P.err = "SNAC layer kernel size != 2*stride: " + p;
                if (b.cin != C && P.err.empty()) P.err = "SNAC layer channel mismatch: " + p;
                if (((int) t->ne[0] - 2 * b.padding) != b.stride && P.err.empty()) P.err = "SNAC layer does not upsample by exactly its stride: " + p;
                if ((int) sd.groups[i] != b.cout && P.err.empty()) P.err = "SNAC layer grouping != channels (only depthwise residual units are supported): " + p;
            }
            b.alpha = P.place_f32(p + "alpha"); b.w = P.place_f32(p + "weight"); b.b = P.place_f32(p + "bias");
            b.noise_w = P.place_f32(p + "noise_weight");
            for (int r = 0; r < 3; r++) {
                const std::string q = p + "residual_unit." + std::to_string(r) + ".res.";
                b.res[r].in_alpha = P.place_f32(q + "initial.alpha"); b.res[r].in_w = P.place_f32(q + "initial.weight");
                b.res[r].in_b = P.place_f32(q + "initial.bias"); b.res[r].out_alpha = P.place_f32(q + "final.alpha");
                b.res[r].out_w = P.place_f32(q + "final.weight"); b.res[r].out_b = P.place_f32(q + "final.bias");
            }
            C = b.cout;
            c->s_up *= b.stride;
        }
        c->s_clast = C;
        c->s_falpha = P.place_f32("snac.alpha_out"); c->s_fw = P.place_f32("snac.final.weight"); c->s_fb = P.place_f32("snac.final.bias");
    }
    if (c->has_t5) {
        // t5_encoder (t5/model.h:39-60, tensor names t5/model.cpp:3-18, py-gguf t5_encoder_gguf_encoder.py:73-90)
        c->H = (int) c->t5.hidden_size; c->L = (int) c->t5.n_layers; c->NH = (int) c->t5.n_attn_heads;
        if (c->H <= 0 || c->L <= 0 || c->NH <= 0 || c->t5.max_ctx_length == 0) return set_err("plan: incomplete T5 hyper-parameters");
        if (c->H / c->NH != 64 || c->H % c->NH) return set_err("plan: T5 head size %d unsupported (64, t5/model.h:46)", c->H / c->NH);
        const Tensor *te = P.get("t5encoder.token_embd");
        if (!te) return set_err("plan: t5encoder.token_embd missing");
        c->t5_vocab = (int) (te->nelem() / te->ne[0]);
        c->t5_embd = P.place_f32("t5encoder.token_embd");
        c->t5_out_norm = P.place_f32("t5encoder.enc.final_layer_norm");
        c->t5_relb = P.place_f32("t5encoder.enc.blk.0.attn_rel_b");
        { const Tensor *rb = P.get("t5encoder.enc.blk.0.attn_rel_b");
          if (rb && ((int) rb->ne[0] != c->NH || (uint32_t) (rb->nelem() / rb->ne[0]) != c->t5.n_buckets) && P.err.empty())
              P.err = "t5 relative attention bias is not [n_buckets][n_heads]"; }
        c->t5_has_down = c->tensors.count("t5encoder.down_proj") != 0;
        c->t5_has_down_b = c->tensors.count("t5encoder.down_proj_bias") != 0;
        if (c->t5_has_down) c->t5_down = P.mat("t5encoder.down_proj");
        if (c->t5_has_down_b) c->t5_down_b = P.place_f32("t5encoder.down_proj_bias");
        c->t5_out = c->t5_has_down ? (int) c->t5_down.N : c->H;
        if (c->t5.output_size && (int) c->t5.output_size != c->t5_out && P.err.empty()) P.err = "t5encoder.output_size disagrees with the tensors";
        c->t5_layers.assign(c->L, tts_hip_ctx::T5Layer{});
        for (int l = 0; l < c->L; l++) {
            const std::string p = "t5encoder.enc.blk." + std::to_string(l) + ".";
            auto &y = c->t5_layers[l];
            y.attn_norm = P.place_f32(p + "attn_norm");
            y.qkv = P.fused({p + "attn_q", p + "attn_k", p + "attn_v"});
            y.o = P.mat(p + "attn_o");
            y.mlp_norm = P.place_f32(p + "ffn_norm");
            y.wi = P.fused({p + "ffn_up", p + "ffn_gate"});   // wi_0 | wi_1
            y.wo = P.mat(p + "ffn_down");
        }
        c->F = c->t5_layers.empty() ? 0 : (int) (c->t5_layers[0].wi.N / 2);
    }
    if (c->has_dac) {
        std::vector<std::string> cb, pw, pb;
        int ncb = 0;
        while (c->tensors.count("audio_encoder.quantizers." + std::to_string(ncb) + ".codebook.weight")) ncb++;
        if (ncb == 0) return set_err("plan: no audio_encoder.quantizers.*.codebook.weight tensors");
        if (c->has_parler && ncb < c->NO) return set_err("plan: %d DAC codebooks < %d output heads", ncb, c->NO);
        if (c->has_parler) ncb = c->NO;  // dac_model::prep_constants: n_heads = output_heads (dac_model.cpp:16-19)
        for (int i = 0; i < ncb; i++) {
            const std::string p = "audio_encoder.quantizers." + std::to_string(i) + ".";
            cb.push_back(p + "codebook.weight"); pw.push_back(p + "out_proj.weight"); pb.push_back(p + "out_proj.bias");
        }
        c->d_ncb = ncb;
        const Tensor *t0 = P.get(cb[0]);
        const Tensor *w0 = P.get(pw[0]);
        if (t0 && w0) {
            c->d_cbdim = (int) t0->ne[0]; c->d_cbsize = (int) t0->ne[1];
            c->d_latent = (int) (w0->nelem() / c->d_cbdim);
        }
        c->d_codebook = P.fused(cb).off;
        c->d_projw = P.fused(pw).off;
        c->d_projb = P.fused(pb).off;
        { const Tensor *t = P.get("audio_encoder.initial.weight"); c->d_c0 = t ? (int) t->ne[2] : 0; }
        c->d_initw = P.place_f32("audio_encoder.initial.weight");
        c->d_initb = P.place_f32("audio_encoder.initial.bias");
        c->dblocks.assign(d.dac_n_blocks, DBlock{});
        c->d_up = 1;
        int C = c->d_c0;
        for (uint32_t i = 0; i < d.dac_n_blocks; i++) {
            const std::string p = "audio_encoder.decoder_block." + std::to_string(i + 1) + ".";
            DBlock &b = c->dblocks[i];
            b.stride = (int) d.dac_stride[i]; b.padding = (int) d.dac_padding[i];
            const Tensor *t = P.get(p + "final.weight");  // ne = [K, Cout, Cin]
            if (t) {
                b.cin = (int) t->ne[2]; b.cout = (int) t->ne[1];
                if ((int) t->ne[0] != 2 * b.stride && P.err.empty()) P.err = "DAC block kernel size != 2*stride: " + p;
                if (b.cin != C && P.err.empty()) P.err = "DAC block channel mismatch: " + p;
                if (((int) t->ne[0] - 2 * b.padding) != b.stride && P.err.empty()) P.err = "DAC block does not upsample by exactly its stride: " + p;
            }
            b.alpha = P.place_f32(p + "final.alpha");
            b.w = P.place_f32(p + "final.weight");
            b.b = P.place_f32(p + "final.bias");
            for (int r = 0; r < 3; r++) {
                const std::string q = p + "residual_unit." + std::to_string(r) + ".res.";
                b.res[r].in_alpha = P.place_f32(q + "initial.alpha");
                b.res[r].in_w = P.place_f32(q + "initial.weight");
                b.res[r].in_b = P.place_f32(q + "initial.bias");
                b.res[r].out_alpha = P.place_f32(q + "final.alpha");
                b.res[r].out_w = P.place_f32(q + "final.weight");
                b.res[r].out_b = P.place_f32(q + "final.bias");
            }
            C = b.cout;
            c->d_up *= b.stride;
        }
        c->d_clast = C;
        c->d_falpha = P.place_f32("audio_encoder.final.alpha");
        c->d_fw = P.place_f32("audio_encoder.final.weight");
        c->d_fb = P.place_f32("audio_encoder.final.bias");
        // --convert-dac-to-f16 turns every audio_encoder tensor except the snake alphas into F16 (quantize_impl.cpp:264-266)
        c->dac_f16 = !(d.flags & TTS_HIP_FLAG_DAC_F32);
        for (auto &kv : c->tensors) {
            const std::string &nm = kv.first;
            if (starts_with(nm, "audio_encoder.") && ends_with(nm, ".weight") && kv.second.n_dims == 3 && nm.find(".in_proj") == std::string::npos)
                c->dac_f16 = c->dac_f16 && kv.second.src_type == TTS_HIP_F16;
        }
    }
    if (!P.err.empty()) return set_err("plan: %s", P.err.c_str());
    c->arena_bytes = (P.cur + 255) & ~(size_t) 255;
    c->planned = true;
    return 0;
}

extern "C" size_t tts_hip_arena_bytes(tts_hip_ctx *c) {
    if (!c) return 0;
    if (plan(c) != 0) return 0;
    return c->arena_bytes;
}
extern "C" void *tts_hip_arena_ptr(tts_hip_ctx *c) { return c ? c->arena : nullptr; }
extern "C" void *tts_hip_stream(tts_hip_ctx *c) { return c ? (void *) c->stream : nullptr; }
extern "C" int tts_hip_synchronize(tts_hip_ctx *c) {
    if (!c) return set_err("null ctx");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// kernel launch plumbing (+ optional per-class event timing)
// ------------------------------------------------------------------------------------------------
// hipFuncSetAttribute applies to the current device: remember per device (a host may drive several GPUs from one
// process, e.g. device_pool), not per process
static bool attr_needed(std::atomic<uint64_t> &done, int device) {
    const uint64_t bit = 1ull << (device & 63);
    if (done.load(std::memory_order_relaxed) & bit) return false;
    done.fetch_or(bit, std::memory_order_relaxed);
    return true;
}

static bool prof_on(const tts_hip_ctx *c, int kclass) {
    return c->prof || (c->prof_light && kclass >= TTS_HIP_K_DAC_EMBED);
}
static int prof_begin(tts_hip_ctx *c, int kclass, double bytes, double flops) {
    c->prof_cur = prof_on(c, kclass);
    if (!c->prof_cur) return 0;
    ProfEv e;
    HIPCHK(hipEventCreate(&e.a));
    HIPCHK(hipEventCreate(&e.b));
    e.kclass = kclass;
    HIPCHK(hipEventRecord(e.a, c->stream));
    c->prof_events.push_back(e);
    c->kstat[kclass].launches++;
    c->kstat[kclass].bytes_total += bytes;
    c->kstat[kclass].flops_total += flops;
    return 0;
}
static int prof_end(tts_hip_ctx *c) {
    if (!c->prof_cur) return 0;
    c->prof_cur = false;
    HIPCHK(hipEventRecord(c->prof_events.back().b, c->stream));
    return 0;
}
static int prof_collect(tts_hip_ctx *c) {
    if (c->prof_events.empty()) return 0;
    HIPCHK(hipStreamSynchronize(c->stream));
    for (auto &e : c->prof_events) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, e.a, e.b));
        c->kstat[e.kclass].ms_total += ms;
        (void) hipEventDestroy(e.a);
        (void) hipEventDestroy(e.b);
    }
    c->prof_events.clear();
    return 0;
}

extern "C" int tts_hip_dac_arith(tts_hip_ctx *c) {
    if (!c || !c->has_dac) return 0;
    if (c->dac_f16) return 8;
    if (c->d.flags & TTS_HIP_FLAG_VALU_GEMM) return 16;
    return (c->dac_b3 ? 1 : 0) | (c->dac_fuse ? 2 : 0) | (c->dac_convt_b3 ? 4 : 0) | (c->dac_planes && c->dac_b3 ? 32 : 0);
}
extern "C" int tts_hip_profile(tts_hip_ctx *c, int enable) {
    if (!c) return set_err("null ctx");
    HIPCHK(hipSetDevice(c->device));
    CHK(prof_collect(c));
    c->prof = enable == 1;
    c->prof_light = enable == 2;
    if (enable) memset(c->kstat, 0, sizeof(c->kstat));
    return 0;
}
extern "C" int tts_hip_profile_get(tts_hip_ctx *c, int k, tts_hip_kstat *out) {
    if (!c || !out || k < 0 || k >= TTS_HIP_K_COUNT) return set_err("tts_hip_profile_get: bad argument");
    HIPCHK(hipSetDevice(c->device));
    CHK(prof_collect(c));
    *out = c->kstat[k];
    return 0;
}

template <int WT, int PRO, int EPI, int RB>
static int launch_gemm16(tts_hip_ctx *c, const GemmArgs &a) {
    const int ksplit = a.kchunk ? a.K / a.kchunk : 1;
    const int nw = (a.kchunk ? a.kchunk : a.K) / 256;
    // wave sets working on different row groups in parallel (up to 16 waves per workgroup)
    const int rows_wg = a.rows_per_z ? std::min(a.rows_per_z, a.R) : a.R;
    const int n_groups = (rows_wg + 16 * RB - 1) / (16 * RB);
    const int ngs = PRO == PRO_LN || PRO == PRO_ATTN ? 1 : std::max(1, std::min(std::min(n_groups, 16 / nw), c->gemm_ngs_max));
    size_t lds = 0;
    if (PRO == PRO_LN || PRO == PRO_ATTN) {
        lds = (size_t) RB * 16 * (a.K + (WT == 1 ? 8 : 4)) * (WT == 1 ? 2 : 4);
        lds = (lds + 15) & ~(size_t) 15;
    }
    if (nw > 1) lds += (size_t) ngs * nw * RB * 4 * 64 * 4;
    static std::atomic<uint64_t> attr{0};
    if (lds > 48 * 1024 && attr_needed(attr, c->device)) {
        HIPCHK(hipFuncSetAttribute((const void *) gemm16_kernel<WT, PRO, EPI, RB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    if (lds > 160 * 1024) return set_err("gemm16: LDS request %zu exceeds 160 KiB", lds);
    if (PRO == PRO_LN && a.K > 2048) return set_err("gemm16: LayerNorm prologue supports hidden sizes up to 2048 (got %d)", a.K);
    if ((PRO == PRO_LN || PRO == PRO_ATTN) && ngs * nw > 8) return set_err("gemm16: fused-prologue launch wants %d waves (> 8)", ngs * nw);
    const int nz = a.rows_per_z ? (a.R + a.rows_per_z - 1) / a.rows_per_z : 1;
    hipLaunchKernelGGL((gemm16_kernel<WT, PRO, EPI, RB>), dim3(a.N / 16, ksplit, nz), dim3(ngs * nw * 64), lds, c->stream, a);
    HIPCHK(hipGetLastError());
    return 0;
}

template <int WT, int PRO, int EPI>
static int launch_gemm16_rb(tts_hip_ctx *c, const GemmArgs &a_in) {
    GemmArgs a = a_in;
    if (PRO != PRO_LN && PRO != PRO_ATTN && c->gemm_rows_per_wg > 0 && a.R > c->gemm_rows_per_wg) {
        a.rows_per_z = c->gemm_rows_per_wg;
        if (a.rows_per_z <= 16) return launch_gemm16<WT, PRO, EPI, 1>(c, a);
        if (a.rows_per_z <= 32) return launch_gemm16<WT, PRO, EPI, 2>(c, a);
        return launch_gemm16<WT, PRO, EPI, 4>(c, a);
    }
    if (a.R <= 16) return launch_gemm16<WT, PRO, EPI, 1>(c, a);
    if (a.R <= 32) return launch_gemm16<WT, PRO, EPI, 2>(c, a);
    if (PRO == PRO_LN && (WT == 0 || a.R > 64)) return set_err("gemm16: %d rows with a fused LayerNorm prologue do not fit LDS", a.R);
    return launch_gemm16<WT, PRO, EPI, 4>(c, a);  // loops over groups of 64 rows, weights stay in registers
}

// rows one forward can carry: 256 through the 16-feature workgroups, 512 when every decoder matrix is fp16 (LDS-tiled GEMM)
static int max_rows_for(const tts_hip_ctx *c) {
    if (c->tile_min_rows <= 0) return 256;
    for (const PLayer &y : c->layers)
        for (const W *w : {&y.qkv, &y.o, &y.cq, &y.co, &y.fc1, &y.fc2})
            if (w->type != TTS_HIP_F16 && w->N) return 256;
    if (c->heads.type != TTS_HIP_F16) return 256;
    // 1024 rows per forward: every GEMM of a Parler-Mini layer is a whole number of rounds of 128 x 128 (N = 4096, 3072) or 64 x 64 (N = 1024)
    // tiles over the 256 CUs; at 1152 rows the ninth row tile costs a second, nearly empty round (146 vs 101 us of GEMMs per layer,
    // profiles/r03/rows1152_classes.txt / rows1024_classes.txt).  TTS_HIP_MAX_ROWS raises or lowers the cap.
    if (const char *e = getenv("TTS_HIP_MAX_ROWS")) return std::min(GRAPH_KEY_ROWS - 1, std::max(256, atoi(e)));
    return 1024;
}

template <int EPI, int RB, int QPRO>
static int launch_qgemm16(tts_hip_ctx *c, const QGemmArgs &qa) {
    const int kc = qa.g.kchunk ? qa.g.kchunk : qa.g.K;
    const int nw = kc / 256;
    size_t lds = nw > 1 ? (size_t) nw * RB * 4 * 64 * 4 : 0;
    if (QPRO >= 1) lds += (((size_t) qa.g.R * kc + 15) & ~(size_t) 15) + (((size_t) qa.g.R * (kc / 32) * 4 + 15) & ~(size_t) 15);
    static std::atomic<uint64_t> attr{0};
    if (attr_needed(attr, c->device)) {
        HIPCHK(hipFuncSetAttribute((const void *) qgemm16_kernel<EPI, RB, QPRO>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    hipLaunchKernelGGL((qgemm16_kernel<EPI, RB, QPRO>), dim3(qa.g.N / 16, qa.g.K / kc), dim3(nw * 64), lds, c->stream, qa);
    HIPCHK(hipGetLastError());
    return 0;
}
template <int EPI>
static int launch_qgemm16_rb(tts_hip_ctx *c, const QGemmArgs &qa, bool fused_quant, bool fused_ln) {
    if (fused_ln) return launch_qgemm16<EPI, 1, 2>(c, qa);
    if (fused_quant) return launch_qgemm16<EPI, 1, 1>(c, qa);
    if (qa.g.R <= 16) return launch_qgemm16<EPI, 1, 0>(c, qa);
    if (qa.g.R <= 32) return launch_qgemm16<EPI, 2, 0>(c, qa);
    return launch_qgemm16<EPI, 4, 0>(c, qa);
}

// ------------------------------------------------------------------------------------------------
// LDS-tiled GEMM for many rows (gemm_tile_kernels.h): shape choice + launch
// ------------------------------------------------------------------------------------------------
struct TileShape { int BM, BN, threads; };
static const TileShape TILE_SHAPES[] = {{32, 32, 128}, {32, 64, 256}, {64, 32, 256}, {64, 64, 512}, {128, 64, 512}, {128, 128, 512}};
enum { N_TILE_SHAPES = 6 };

template <int BM, int BN, int WM, int WN, int EPI>
static int launch_tile_shape(tts_hip_ctx *c, const GemmArgs &a, const TileMap &tm, bool deep) {
    // deep: 128-wide k-tiles, 3 LDS buffers — fewer barriers for a single wave of workgroups; otherwise 64-wide k-tiles,
    // 4 buffers (half the LDS: two workgroups per CU when the grid exceeds the CU count)
    constexpr bool can_deep = (BM + BN) * 256 * 3 <= 160 * 1024;
    const int total = tm.m_tiles * tm.n_tiles * tm.k_slices;
    const int grid = (total + 7) / 8 * 8;
    const int kc = a.kchunk ? a.kchunk : a.K;
    if (can_deep && deep && kc % 128 == 0) {
        const size_t lds = (size_t) 3 * (BM + BN) * 256;
        static std::atomic<uint64_t> attr{0};
        if (attr_needed(attr, c->device))
            HIPCHK(hipFuncSetAttribute((const void *) gemm_tile_kernel<BM, BN, WM, WN, 128, 3, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL((gemm_tile_kernel<BM, BN, WM, WN, 128, 3, EPI>), dim3(grid), dim3(WM * WN * 64), lds, c->stream, a, tm);
    } else {
        const size_t lds = (size_t) 4 * (BM + BN) * 128;
        static std::atomic<uint64_t> attr{0};
        if (attr_needed(attr, c->device))
            HIPCHK(hipFuncSetAttribute((const void *) gemm_tile_kernel<BM, BN, WM, WN, 64, 4, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL((gemm_tile_kernel<BM, BN, WM, WN, 64, 4, EPI>), dim3(grid), dim3(WM * WN * 64), lds, c->stream, a, tm);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

template <int EPI>
static int launch_tile(tts_hip_ctx *c, const GemmArgs &a, int shape, int ks) {
    const TileShape &t = TILE_SHAPES[shape];
    TileMap tm{(a.R + t.BM - 1) / t.BM, (a.N + t.BN - 1) / t.BN, ks};
    const bool deep = c->tile_deep && tm.m_tiles * tm.n_tiles * tm.k_slices <= 320;
    switch (shape) {
        case 0: return launch_tile_shape<32, 32, 1, 2, EPI>(c, a, tm, deep);
        case 1: return launch_tile_shape<32, 64, 1, 4, EPI>(c, a, tm, deep);
        case 2: return launch_tile_shape<64, 32, 2, 2, EPI>(c, a, tm, deep);
        case 3: return launch_tile_shape<64, 64, 2, 4, EPI>(c, a, tm, deep);
        case 4: return launch_tile_shape<128, 64, 4, 2, EPI>(c, a, tm, deep);
        default: return launch_tile_shape<128, 128, 2, 4, EPI>(c, a, tm, deep);
    }
}

// Cost model fitted to profiles/r02/gemm_tile_sweep.log: a launch costs ~ (fixed + bytes one workgroup stages) per wave of
// workgroups (a CU pulls ~28 B/clk of mixed L2 / HBM traffic whatever the tile), so the best shape is the largest tile that
// still gives about one workgroup per CU; residual GEMMs (N = hidden size) may split K to get there.
static void choose_tile(const tts_hip_ctx *c, int R, int N, int K, bool may_split, int *shape_out, int *ks_out) {
    double best = 1e30;
    int bs = 0, bk = 1;
    for (int s = 0; s < N_TILE_SHAPES; s++) {
        const TileShape &t = TILE_SHAPES[s];
        if (t.BM >= 2 * R && s > 0) continue;   // mostly padding rows
        for (int ks = 1; ks <= (may_split ? 8 : 1); ks *= 2) {
            if (K % (ks * 128) || K / ks < 256) continue;
            const double blocks = (double) ((R + t.BM - 1) / t.BM) * ((N + t.BN - 1) / t.BN) * ks;
            const double bytes = (double) (t.BM + t.BN) * (K / ks) * 2.0;
            double cost = std::max(1.0, blocks / 256.0) * (96.0 * 1024 + bytes);
            if (ks > 1) cost += 0.18 * ks * (double) R * N;   // slab write by this launch + read by the folding LayerNorm (~1 us per 1.5 MB slab)
            if (cost < best) { best = cost; bs = s; bk = ks; }
        }
    }
    if (c->tile_force >= 0 && c->tile_force < N_TILE_SHAPES) bs = c->tile_force;
    if (c->tile_force_ks > 0 && may_split && K % (c->tile_force_ks * 128) == 0) bk = c->tile_force_ks;
    *shape_out = bs;
    *ks_out = bk;
}

// GGUF-quantised matrix: LayerNorm (if any) -> Q8_0-quantise the activation rows -> integer block GEMM
// LayerNorm of R rows (+ slab fold): the kernel instantiated for this row width and slab count (ln_rows_t_kernel), so that a launch
// carries the registers of its own variant only
static void launch_ln_rows(tts_hip_ctx *c, int rows_per_wg, float *x, int H, const float *lw, const float *lb, float *y32, _Float16 *y16, int R, const float *parts, int n_parts,
                           int64_t slab_stride) {
    const dim3 grid((unsigned) ((R + rows_per_wg - 1) / rows_per_wg)), block((unsigned) (64 * rows_per_wg));
    const int np = parts ? n_parts : 0;
#define LN_CASE(NIv, NPv) hipLaunchKernelGGL((ln_rows_t_kernel<NIv, NPv>), grid, block, 0, c->stream, x, H, lw, lb, y32, y16, R, parts, n_parts, slab_stride)
    if (H <= 2048 && (H & 3) == 0) {
        if (H <= 1024) {
            if (np == 0) LN_CASE(4, 0);
            else if (np == 4) LN_CASE(4, 4);
            else if (np == 2) LN_CASE(4, 2);
            else if (np == 8) LN_CASE(4, 8);
            else LN_CASE(4, -1);
        } else {
            if (np == 0) LN_CASE(8, 0);
            else LN_CASE(8, -1);
        }
    } else {
        hipLaunchKernelGGL(ln_rows_kernel, grid, block, 0, c->stream, x, H, lw, lb, y32, y16, R, parts, n_parts, slab_stride);
    }
#undef LN_CASE
}

// ------------------------------------------------------------------------------------------------
// weight-streaming GEMM for <= 16 rows (gemv_stream_kernels.h): K slices -> fp32 slabs the consumer folds
// ------------------------------------------------------------------------------------------------
// K slices for an [N][K] fp16 matrix: as many 256-column chunks as give every wave one (feature tile, slice) item, up to ~4096
// items (one per wave slot of the chip) and the consumer's slab budget; 0 = the shape does not go through gemv_stream_kernel.
// Measured on MI355X at 8 rows (profiles/r02/gemv_bench_r8.log): Dia gate|up 24.9 -> 14.2 us, wo 15.2 -> 9.6, self qkv 8.4 -> 6.0,
// o / cross q / cross o 8.1 -> 4.3.
enum { DIA_STREAM_SLABS = 8 };   // slab budget of the Dia step buffers (di_qkv, di_q, di_gu, di_parts)
static int stream_slices(const tts_hip_ctx *c, const W &w, int R, int max_slabs) {
    if (!c->gemv_stream || w.type != TTS_HIP_F16 || R > 16 || w.K % 256 || w.N % 16 || (c->d.flags & TTS_HIP_FLAG_VALU_GEMM)) return 0;
    const int tiles = (int) w.N / 16;
    int ks = 1;
    while (ks * 2 <= max_slabs && (int) w.K % (ks * 2 * 256) == 0 && tiles * ks * 2 <= 4096) ks *= 2;
    if ((size_t) 16 * (w.K / ks + 32) * 2 > 96 * 1024) return 0;   // the slice of the rows must fit LDS
    return ks;
}

template <int NWV, int PRO, int EPI>
static int launch_stream_one(tts_hip_ctx *c, const GemmArgs &a, StreamMap sm, int grid, size_t lds) {
    static std::atomic<uint64_t> attr{0};
    if (lds > 48 * 1024 && attr_needed(attr, c->device))
        HIPCHK(hipFuncSetAttribute((const void *) gemv_stream_kernel<NWV, PRO, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((gemv_stream_kernel<NWV, PRO, EPI>), dim3(grid), dim3(NWV * 64), lds, c->stream, a, sm);
    HIPCHK(hipGetLastError());
    return 0;
}
static int launch_stream(tts_hip_ctx *c, const GemmArgs &a, int pro, int epi) {
    const int ks = a.kchunk ? a.K / a.kchunk : 1;
    if (ks > 1 && epi != EPI_STORE) return set_err("gemv_stream: K slices need the slab epilogue");
    if (epi != EPI_STORE && epi != EPI_RESID) return set_err("gemv_stream: no kernel for epilogue %d", epi);
    const StreamMap sm{ks, a.K / ks};
    const int items = a.N / 16 * ks;
    const int nwv = items >= 4096 ? 16 : 4;
    const int grid = ((items + nwv - 1) / nwv + ks - 1) / ks * ks;
    const size_t lds = (size_t) (a.R <= 8 ? 8 : 16) * (sm.kslice + 32) * 2;
#define STREAM_CASE(NWVv, PROv, EPIv) if (nwv == NWVv && pro == PROv && epi == EPIv) return launch_stream_one<NWVv, PROv, EPIv>(c, a, sm, grid, lds);
    STREAM_CASE(4, PRO_F32, EPI_STORE) STREAM_CASE(16, PRO_F32, EPI_STORE) STREAM_CASE(4, PRO_F32, EPI_RESID) STREAM_CASE(16, PRO_F32, EPI_RESID)
    STREAM_CASE(4, PRO_F16, EPI_STORE) STREAM_CASE(16, PRO_F16, EPI_STORE)
#undef STREAM_CASE
    return set_err("gemv_stream: no kernel for pro=%d epi=%d", pro, epi);
}

static int run_qgemm(tts_hip_ctx *c, int kclass, const W &w, GemmArgs a, int pro, int epi) {
    if (pro == PRO_F16) return set_err("run_qgemm: fp16 activations are never produced for a quantised consumer");
    const bool have_q = c->aq_src != nullptr && c->aq_src == a.A && a.lda == a.K;
    c->aq_src = nullptr;
    if (c->gemv_rows && a.R <= 4 && pro == PRO_F32 && (epi == EPI_STORE || epi == EPI_RESID) && !a.kchunk) {
        CHK(prof_begin(c, kclass, (double) w.K * w.N * (1.0 + 2.0 / 32) + (double) a.R * a.K * 5 + (double) a.R * a.N * 4, 2.0 * a.R * (double) w.K * w.N));
        if (!have_q) {   // otherwise the producing kernel (rms norm, silu*up, attention combine) left the Q8_0 blocks in aq / ad
            hipLaunchKernelGGL(quant_rows_q8_kernel, dim3((a.K / 32 + 7) / 8, a.R), dim3(256), 0, c->stream, (const float *) a.A, a.lda, a.K, c->aq, c->ad, a.R);
            HIPCHK(hipGetLastError());
        }
        QGemmArgs qa{};
        qa.g = a;
        qa.wd = (const _Float16 *) (c->arena + w.soff);
        qa.aq = c->aq;
        qa.ad = c->ad;
        const size_t q4_lds = (size_t) a.R * a.K + (size_t) a.R * (a.K / 32) * 4;
        if (w.q4 && c->q4_lds && q4_lds <= 64 * 1024 && a.K % 512 == 0) {
            // activations in LDS, 2 or 4 features per wave (gemv_q4_rows_lds_kernel): fewer load instructions per weight byte
            if (a.N >= 8192) hipLaunchKernelGGL((gemv_q4_rows_lds_kernel<4, 4>), dim3((a.N + 15) / 16), dim3(256), q4_lds, c->stream, qa, w.q4, epi);
            else hipLaunchKernelGGL((gemv_q4_rows_lds_kernel<4, 2>), dim3((a.N + 7) / 8), dim3(256), q4_lds, c->stream, qa, w.q4, epi);
        } else if (w.q4) hipLaunchKernelGGL(gemv_q4_rows_kernel<4>, dim3((a.N + 3) / 4), dim3(256), 0, c->stream, qa, w.q4, epi);
        else hipLaunchKernelGGL(gemv_q8_rows_kernel<4>, dim3((a.N + 3) / 4), dim3(256), 0, c->stream, qa, epi);
        HIPCHK(hipGetLastError());
        return prof_end(c);
    }
    // few rows: LayerNorm + quantisation inside every GEMM workgroup (one launch instead of three)
    const bool fused_ln = pro == PRO_LN && a.R <= std::min(c->ln_fuse_max, 8) && a.K <= 2048 && !c->pending_parts && c->q_fuse_max > 0;
    if (pro == PRO_LN && !fused_ln) {
        CHK(prof_begin(c, TTS_HIP_K_LN, (double) a.R * a.K * 8, 0));
        launch_ln_rows(c, c->ln_waves, (float *) a.A, a.K, a.ln_w, a.ln_b, c->dbg, (_Float16 *) nullptr, a.R,
                       c->pending_parts ? (const float *) c->partials : (const float *) nullptr, c->pending_parts, (int64_t) c->RMAX * c->H);
        HIPCHK(hipGetLastError());
        c->pending_parts = 0;
        CHK(prof_end(c));
        a.A = c->dbg;
        a.lda = a.K;
    }
    const bool fused_quant = fused_ln || a.R <= c->q_fuse_max;  // up to 16 rows: each workgroup quantises them itself
    if (epi == EPI_RESID && a.R > c->ln_fuse_max && a.H <= 2048 && a.N == a.H && a.out == c->x) {
        // N = H only gives H/16 workgroups: spread K over 4x more; the slabs are folded into x by the next LayerNorm
        const int ks = a.K >= 4096 ? c->ksplit_big : 4;
        if (a.K % (ks * 256) == 0) {
            a.kchunk = a.K / ks;
            a.slab_stride = (int64_t) c->RMAX * c->H;
            a.out = c->partials;
            epi = EPI_STORE;
            c->pending_parts = ks;
        }
    }
    const double wbytes = (double) w.K * w.N * (1.0 + 2.0 / 32);
    CHK(prof_begin(c, kclass, wbytes + (double) a.R * a.K * 5 + (double) a.R * a.N * 4, 2.0 * a.R * (double) w.K * w.N));
    if (!fused_quant) {
        hipLaunchKernelGGL(quant_rows_q8_kernel, dim3((a.K / 32 + 7) / 8, a.R), dim3(256), 0, c->stream, (const float *) a.A, a.lda, a.K, c->aq, c->ad, a.R);
        HIPCHK(hipGetLastError());
    }
    QGemmArgs qa{};
    qa.g = a;
    qa.wd = (const _Float16 *) (c->arena + w.soff);
    qa.aq = c->aq;
    qa.ad = c->ad;
    int rc;
    if (epi == EPI_STORE) rc = launch_qgemm16_rb<EPI_STORE>(c, qa, fused_quant, fused_ln);
    else if (epi == EPI_QKV) rc = launch_qgemm16_rb<EPI_QKV>(c, qa, fused_quant, fused_ln);
    else if (epi == EPI_RESID) rc = launch_qgemm16_rb<EPI_RESID>(c, qa, fused_quant, fused_ln);
    else rc = launch_qgemm16_rb<EPI_GELU>(c, qa, fused_quant, fused_ln);
    CHK(rc);
    return prof_end(c);
}

// one GEMM of the forward: picks MFMA or the scalar reference path
// every matrix the <= 4-row forward touches goes through gemm16_kernel<1, ...> (the only consumer that folds fc2's K-slice slabs)
static bool chain_all_f16(const tts_hip_ctx *c) {
    if ((c->d.flags & TTS_HIP_FLAG_VALU_GEMM) || c->gemv_rows || c->H % 256) return false;
    for (const PLayer &y : c->layers)
        for (const W *m : {&y.qkv, &y.o, &y.cq, &y.co, &y.fc1, &y.fc2})
            if (m->N && m->type != TTS_HIP_F16) return false;
    return c->heads.type == TTS_HIP_F16;
}

static int run_gemm(tts_hip_ctx *c, int kclass, const W &w, GemmArgs a, int pro, int epi) {
    a.W = c->arena + w.off;
    a.K = (int) w.K;
    a.N = (int) w.N;
    if (w.type == TTS_HIP_Q8I) return run_qgemm(c, kclass, w, a, pro, epi);
    const double wbytes = (double) w.K * w.N * (w.type == TTS_HIP_F16 ? 2 : 4);
    const double bytes = wbytes + (double) a.R * a.K * 4 + (double) a.R * a.N * 4;
    const double flops = 2.0 * a.R * (double) w.K * w.N;
    const bool valu = (c->d.flags & TTS_HIP_FLAG_VALU_GEMM) || (a.K % 256) || (a.N % 16);
    if (valu) {
        // scalar path: LayerNorm materialised first
        GemmArgs b = a;
        if (pro == PRO_LN) {
            CHK(prof_begin(c, TTS_HIP_K_LN, (double) a.R * a.K * 8, 0));
            launch_ln_rows(c, 4, (float *) a.A, a.K, a.ln_w, a.ln_b, c->dbg, (_Float16 *) nullptr, a.R, (const float *) nullptr, 0, (int64_t) 0);
            HIPCHK(hipGetLastError());
            CHK(prof_end(c));
            b.A = c->dbg;
            b.lda = a.K;
        }
        CHK(prof_begin(c, kclass, bytes, flops));
        const int wpb = 4;
        if (w.type == TTS_HIP_F16) hipLaunchKernelGGL(gemv_valu_kernel<1>, dim3((a.N + wpb - 1) / wpb), dim3(wpb * 64), 0, c->stream, b, epi, pro == PRO_F16 ? 1 : 0);
        else hipLaunchKernelGGL(gemv_valu_kernel<0>, dim3((a.N + wpb - 1) / wpb), dim3(wpb * 64), 0, c->stream, b, epi, pro == PRO_F16 ? 1 : 0);
        HIPCHK(hipGetLastError());
        return prof_end(c);
    }
    if (a.stream && w.type == TTS_HIP_F16 && a.R <= 16 && pro != PRO_LN) {
        CHK(prof_begin(c, kclass, bytes, flops));
        CHK(launch_stream(c, a, pro, epi));
        return prof_end(c);
    }
    if (c->gemv_rows && a.R <= 4 && pro == PRO_F32 && (epi == EPI_STORE || epi == EPI_RESID) && !a.kchunk) {
        CHK(prof_begin(c, kclass, bytes, flops));
        if (w.type == TTS_HIP_F16) hipLaunchKernelGGL((gemv_rows_kernel<1, 4>), dim3((a.N + 3) / 4), dim3(256), 0, c->stream, a, epi);
        else hipLaunchKernelGGL((gemv_rows_kernel<0, 4>), dim3((a.N + 3) / 4), dim3(256), 0, c->stream, a, epi);
        HIPCHK(hipGetLastError());
        return prof_end(c);
    }
    if (pro == PRO_LN && (a.R > c->ln_fuse_max || (w.type != TTS_HIP_F16 && a.R > 32) || a.R > 64)) {
        // many rows: normalise once (one wave per row) instead of once per GEMM workgroup
        const bool h16 = w.type == TTS_HIP_F16;
        CHK(prof_begin(c, TTS_HIP_K_LN, (double) a.R * a.K * (h16 ? 6 : 8), 0));
        launch_ln_rows(c, c->ln_waves, (float *) a.A, a.K, a.ln_w, a.ln_b, h16 ? (float *) nullptr : c->dbg, h16 ? c->xn16 : (_Float16 *) nullptr, a.R,
                       c->pending_parts ? (const float *) c->partials : (const float *) nullptr, c->pending_parts, (int64_t) c->RMAX * c->H);
        c->pending_parts = 0;
        HIPCHK(hipGetLastError());
        CHK(prof_end(c));
        a.A = h16 ? (const void *) c->xn16 : (const void *) c->dbg;
        a.lda = a.K;
        pro = h16 ? PRO_F16 : PRO_F32;
    }
    if (w.type == TTS_HIP_F16 && pro == PRO_F16 && c->tile_min_rows > 0 && a.R >= c->tile_min_rows && a.K % 128 == 0 && a.N % 16 == 0) {
        // many rows: LDS-tiled MFMA GEMM; a residual GEMM may split K into fp32 slabs that the next LayerNorm folds into x
        const bool may_split = epi == EPI_RESID && a.H <= 2048 && a.N == a.H && a.out == c->x;
        int shape = 0, ks = 1;
        choose_tile(c, a.R, a.N, a.K, may_split, &shape, &ks);
        if (ks > 1) {
            a.kchunk = a.K / ks;
            a.slab_stride = (int64_t) c->RMAX * c->H;
            a.out = c->partials;
            epi = EPI_STORE;
            c->pending_parts = ks;
        }
        CHK(prof_begin(c, kclass, bytes, flops));
        int rc;
        if (epi == EPI_STORE) rc = launch_tile<EPI_STORE>(c, a, shape, ks);
        else if (epi == EPI_QKV) rc = launch_tile<EPI_QKV>(c, a, shape, ks);
        else if (epi == EPI_RESID) rc = launch_tile<EPI_RESID>(c, a, shape, ks);
        else rc = launch_tile<EPI_GELU>(c, a, shape, ks);
        CHK(rc);
        return prof_end(c);
    }
    if (epi == EPI_RESID && a.R > c->ln_fuse_max && a.H <= 2048 && a.N == a.H && a.out == c->x) {
        // many rows: spread K over 4-8x more workgroups; the partial slabs are folded into x by the next LayerNorm
        const int ks = a.K >= 4096 ? c->ksplit_big : 4;
        if (a.K % (ks * 256) == 0) {
            a.kchunk = a.K / ks;
            a.slab_stride = (int64_t) c->RMAX * c->H;
            a.out = c->partials;
            epi = EPI_STORE;
            c->pending_parts = ks;
        }
    }
    if (epi == EPI_RESID && pro == PRO_F16 && w.type == TTS_HIP_F16 && a.R <= 4 && c->b1_fc2_split && a.K >= 4096 && a.K % 1024 == 0 && a.N == a.H && a.H <= 1024 &&
        a.out == c->x && !a.n_parts && chain_all_f16(c)) {
        // batch-1 chain: 64 workgroups streaming 128 KB of fc2 each and reducing 16 K slices through LDS take 7.4 us; 256 workgroups of 32 KB take what
        // out_proj takes (3 us).  The four K-slice slabs are folded by the consumers: the next LayerNorm prologue and the next residual epilogue.
        a.kchunk = a.K / 4;
        a.slab_stride = (int64_t) c->RMAX * c->H;
        a.out = c->partials;
        epi = EPI_STORE;
        c->pending_parts = 4;
    }
    CHK(prof_begin(c, kclass, bytes, flops));
    int rc = -1;
#define GEMM_CASE(WTv, PROv, EPIv) \
    if ((w.type == TTS_HIP_F16 ? 1 : 0) == WTv && pro == PROv && epi == EPIv) rc = launch_gemm16_rb<WTv, PROv, EPIv>(c, a); else
    GEMM_CASE(1, PRO_LN, EPI_QKV) GEMM_CASE(0, PRO_LN, EPI_QKV)
    GEMM_CASE(1, PRO_LN, EPI_STORE) GEMM_CASE(0, PRO_LN, EPI_STORE)
    GEMM_CASE(1, PRO_LN, EPI_GELU) GEMM_CASE(0, PRO_LN, EPI_GELU)
    GEMM_CASE(1, PRO_F32, EPI_RESID) GEMM_CASE(0, PRO_F32, EPI_RESID)
    GEMM_CASE(1, PRO_F32, EPI_STORE) GEMM_CASE(0, PRO_F32, EPI_STORE)
    GEMM_CASE(1, PRO_F16, EPI_RESID) GEMM_CASE(1, PRO_ATTN, EPI_RESID)
    GEMM_CASE(1, PRO_F16, EPI_QKV) GEMM_CASE(1, PRO_F16, EPI_STORE) GEMM_CASE(1, PRO_F16, EPI_GELU)
    GEMM_CASE(0, PRO_F32, EPI_QKV) GEMM_CASE(0, PRO_F32, EPI_GELU)
    { rc = set_err("run_gemm: no kernel for type=%d pro=%d epi=%d", w.type, pro, epi); }
#undef GEMM_CASE
    CHK(rc);
    return prof_end(c);
}

static int run_attn(tts_hip_ctx *c, int kclass, AttnArgs a, int R, int nsplit, double kv_bytes, bool defer_combine = false) {
    a.max_T = (nsplit > 1) ? (c->NCTX + nsplit - 1) / nsplit + 1 : std::max(c->NCTX, c->ECAP);
    // few (head,row) pairs: 1024-thread workgroups (64 key groups) instead of a split-T pass + combine launch
    const bool wide = nsplit == 1 && c->NH * R < 128 && a.row_pos != nullptr;
    const int threads = wide ? 1024 : 256;
    const size_t lds = ((size_t) (threads / 16) * 66 + 16) * 4;
    static std::atomic<uint64_t> attr{0};
    if (attr_needed(attr, c->device)) {
        HIPCHK(hipFuncSetAttribute((const void *) attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    CHK(prof_begin(c, kclass, kv_bytes + 2.0 * R * c->H * 4, 0));
    if (!a.row_pos && a.T_fixed <= 32 && a.T_fixed >= 1 && nsplit == 1 && c->attn_short && !a.kv_f16) {
        // cross-attention over a short voice prompt: one wave per (row, head), no merges
        hipLaunchKernelGGL(attn_short_kernel, dim3((c->NH + 3) / 4, R), dim3(256), 0, c->stream, a);
        HIPCHK(hipGetLastError());
        return prof_end(c);
    }
    const bool fused = nsplit > 1 && c->attn_fused && c->attn_cnt != nullptr && !defer_combine;
    a.counters = fused ? c->attn_cnt : nullptr;
    hipLaunchKernelGGL(attn_kernel, dim3(c->NH, R, nsplit), dim3(threads), lds, c->stream, a);
    HIPCHK(hipGetLastError());
    if (nsplit > 1 && !fused && !defer_combine) {
        hipLaunchKernelGGL(attn_combine_kernel, dim3(c->NH, R), dim3(64), 0, c->stream, (const float *) a.part, nsplit, c->H, c->NH, a.out, a.out16);
        HIPCHK(hipGetLastError());
    }
    return prof_end(c);
}

static int attn_nsplit(const tts_hip_ctx *c, int R, bool same_seq) {
    if (c->attn_nsplit_override > 0) return std::min(c->attn_nsplit_override, 16);
    (void) same_seq;
    // up to 4 rows (64 (head, row) pairs on 256 CUs): 8 key slices per pair, folded by the workgroup that finishes last (attn_kernel);
    // batch 1: 1.55 -> 1.33 ms/step over a 2564-step utterance, 1.71 -> 1.38 at T > 1024.  Larger batches fill the chip with one workgroup per pair.
    if (c->attn_fused && c->attn_cnt && R * c->NH <= 64) return 8;
    return 1;  // 1024-thread workgroups for few pairs (run_attn); split-T stays available via TTS_HIP_ATTN_NSPLIT
}

// ------------------------------------------------------------------------------------------------
// the decoder forward over R rows (ids / positions / cache slots already in d_ids / d_pos / d_seq)
// ------------------------------------------------------------------------------------------------
static int enqueue_forward(tts_hip_ctx *c, int R, bool audio, bool want_logits, bool same_seq) {
    const int H = c->H;
    const int64_t seq_stride = (int64_t) c->KVPOS * H;
    const size_t kv_esz = c->d.kv_type == TTS_HIP_F16 ? 2 : 4;
    const size_t layer_kv_bytes = (size_t) c->d.max_seqs * seq_stride * kv_esz;

    c->pending_parts = 0;
    EmbedArgs ea{};
    const W &tab = audio ? c->embed_tokens : c->embed_prompts;
    ea.tab = c->arena + tab.off;
    ea.tab_f16 = tab.type == TTS_HIP_F16;
    ea.tab_stride = audio ? (int64_t) c->EROWS * H : 0;
    ea.n_tabs = audio ? c->NO : 1;
    ea.ids = c->d_ids;
    ea.pos_embed = (const float *) (c->arena + c->pos_embed);
    ea.row_pos = c->d_pos;
    ea.x = c->x;
    ea.H = H;
    CHK(prof_begin(c, TTS_HIP_K_EMBED, (double) R * (ea.n_tabs + 2) * H * 4, 0));
    hipLaunchKernelGGL(embed_rows_kernel, dim3(R, R <= 64 ? (H + 255) / 256 : 1), dim3(256), 0, c->stream, ea);
    HIPCHK(hipGetLastError());
    CHK(prof_end(c));

    double self_kv_bytes = 0;
    for (int r = 0; r < R && r < (int) c->host_pos.size(); r++) self_kv_bytes += 2.0 * (c->host_pos[r] + 1) * H * kv_esz;
    const int nsplit = attn_nsplit(c, R, same_seq);

    // debug timeline of the batch-1 chain: every launch of a <= 4-row forward gets its own 16-stamp record (graph replays rewrite it)
    const bool stamped = c->b1_stamps != nullptr && R <= 4;
    if (stamped) c->b1_stamp_slot = 0;
    auto stamp_slot = [&]() -> long long * { return stamped ? c->b1_stamps + 16 * (size_t) (c->b1_stamp_slot++) : nullptr; };
    for (int l = 0; l < c->L; l++) {
        const PLayer &y = c->layers[l];
        GemmArgs g{};
        g.R = R; g.H = H; g.gelu_mode = (int) c->d.gelu_mode;
        g.stamps = stamp_slot();
        if (c->pending_parts && R <= 4) { g.parts = c->partials; g.n_parts = c->pending_parts; g.parts_stride = (int64_t) c->RMAX * H; }   // the previous layer's fc2 slabs
        // self attention -------------------------------------------------------------------
        g.A = c->x; g.lda = H;
        g.ln_w = (const float *) (c->arena + y.sa_w); g.ln_b = (const float *) (c->arena + y.sa_b);
        g.q = c->q;
        g.kc = (char *) c->kcache + (size_t) l * layer_kv_bytes;
        g.vc = (char *) c->vcache + (size_t) l * layer_kv_bytes;
        g.kv_f16 = c->d.kv_type == TTS_HIP_F16;
        g.seq_stride = seq_stride; g.row_seq = c->d_seq; g.row_pos = c->d_pos;
        CHK(run_gemm(c, TTS_HIP_K_GEMM_QKV, y.qkv, g, PRO_LN, EPI_QKV));

        AttnArgs at{};
        at.q = c->q; at.kc = g.kc; at.vc = g.vc; at.kv_f16 = g.kv_f16; at.seq_stride = seq_stride;
        at.row_seq = c->d_seq; at.row_pos = c->d_pos; at.H = H; at.n_heads = c->NH;
        // an fp16-weight out_proj rounds its input to fp16 anyway: let the attention kernel store fp16
        const bool valu_mode = (c->d.flags & TTS_HIP_FLAG_VALU_GEMM) != 0;
        const bool o_half = y.o.type == TTS_HIP_F16 && !valu_mode && (H % 256 == 0);
        at.scale = 1.0f / sqrtf(64.0f); at.out = c->att; at.out16 = o_half ? c->att16 : nullptr; at.part = c->part;
        at.stamps = stamp_slot();
        // <= 4 rows: the key-split partials are folded by out_proj's workgroups as they load them (one kernel boundary instead of an arrival
        // counter + a dependent read-back inside the attention launch: 9.6 -> 6.3 us per layer at T ~ 1000)
        const bool defer = R <= 4 && nsplit > 1 && o_half && c->b1_defer_combine && H <= 2048 && H == c->NH * 64 && (int) y.o.K == H;
        CHK(run_attn(c, TTS_HIP_K_ATTN_SELF, at, R, nsplit, self_kv_bytes, defer));

        GemmArgs go{};
        go.R = R; go.H = H; go.A = o_half ? (const void *) c->att16 : (const void *) c->att; go.lda = H; go.out = c->x; go.ldo = H;
        go.stamps = stamp_slot();
        if (defer) { go.att_part = c->part; go.att_nz = nsplit; go.att_heads = c->NH; }
        if (c->pending_parts && R <= 4) { go.parts = c->partials; go.n_parts = c->pending_parts; go.parts_stride = (int64_t) c->RMAX * H; }
        CHK(run_gemm(c, TTS_HIP_K_GEMM_ATTN_OUT, y.o, go, defer ? PRO_ATTN : (o_half ? PRO_F16 : PRO_F32), EPI_RESID));
        if (go.n_parts) { c->pending_parts = 0; go.parts = nullptr; go.n_parts = 0; }   // x is whole again

        // cross attention ------------------------------------------------------------------
        if (c->d.use_cross_attn) {
            GemmArgs gq{};
            gq.R = R; gq.H = H; gq.A = c->x; gq.lda = H;
            gq.ln_w = (const float *) (c->arena + y.ca_w); gq.ln_b = (const float *) (c->arena + y.ca_b);
            gq.out = c->q; gq.ldo = H;
            gq.stamps = stamp_slot();
            CHK(run_gemm(c, TTS_HIP_K_GEMM_CROSS_Q, y.cq, gq, PRO_LN, EPI_STORE));
            AttnArgs ac{};
            ac.q = c->q;
            ac.kc = c->cross_kv_ptr() + ((size_t) l * 2 + 0) * c->ECAP * H * 4;
            ac.vc = c->cross_kv_ptr() + ((size_t) l * 2 + 1) * c->ECAP * H * 4;
            ac.kv_f16 = 0; ac.seq_stride = 0; ac.row_seq = nullptr; ac.row_pos = nullptr; ac.T_fixed = c->E;
            const bool co_half = y.co.type == TTS_HIP_F16 && !valu_mode && (H % 256 == 0);
            ac.H = H; ac.n_heads = c->NH; ac.scale = at.scale; ac.out = c->att; ac.out16 = co_half ? c->att16 : nullptr; ac.part = c->part;
            ac.stamps = stamp_slot();
            CHK(run_attn(c, TTS_HIP_K_ATTN_CROSS, ac, R, 1, 2.0 * c->E * H * 4));
            GemmArgs gc = go;
            gc.stamps = stamp_slot();
            gc.A = co_half ? (const void *) c->att16 : (const void *) c->att;
            CHK(run_gemm(c, TTS_HIP_K_GEMM_CROSS_OUT, y.co, gc, co_half ? PRO_F16 : PRO_F32, EPI_RESID));
        }

        // FFN ------------------------------------------------------------------------------
        GemmArgs g1{};
        g1.R = R; g1.H = H; g1.gelu_mode = (int) c->d.gelu_mode; g1.A = c->x; g1.lda = H;
        g1.ln_w = (const float *) (c->arena + y.f_w); g1.ln_b = (const float *) (c->arena + y.f_b);
        const bool u_half = (y.fc2.type == TTS_HIP_F16) && !(c->d.flags & TTS_HIP_FLAG_VALU_GEMM) && (c->F % 256 == 0);
        g1.out = c->u32; g1.out16 = u_half ? c->u16 : nullptr; g1.ldo = c->F;
        g1.stamps = stamp_slot();
        CHK(run_gemm(c, TTS_HIP_K_GEMM_FC1, y.fc1, g1, PRO_LN, EPI_GELU));
        GemmArgs g2{};
        g2.R = R; g2.H = H; g2.A = u_half ? (const void *) c->u16 : (const void *) c->u32; g2.lda = c->F;
        g2.out = c->x; g2.ldo = H;
        g2.stamps = stamp_slot();
        CHK(run_gemm(c, TTS_HIP_K_GEMM_FC2, y.fc2, g2, u_half ? PRO_F16 : PRO_F32, EPI_RESID));
    }

    if (want_logits) {
        GemmArgs gh{};
        gh.R = R; gh.H = H; gh.A = c->x; gh.lda = H;
        gh.ln_w = (const float *) (c->arena + c->ln_w); gh.ln_b = (const float *) (c->arena + c->ln_b);
        gh.out = c->logits; gh.ldo = c->NO * c->V;
        if (c->pending_parts && R <= 4) { gh.parts = c->partials; gh.n_parts = c->pending_parts; gh.parts_stride = (int64_t) c->RMAX * H; }
        CHK(run_gemm(c, TTS_HIP_K_GEMM_HEADS, c->heads, gh, PRO_LN, EPI_STORE));
    }
    return 0;
}

// prep_cross_key_values (model.cpp:110-173): K_c / V_c = W_k / W_v · text_encoding, per layer
static int compute_cross_kv(tts_hip_ctx *c) {
    if (!c->has_parler || !c->d.use_cross_attn) return 0;
    const int H = c->H, step = 32;
    for (int l = 0; l < c->L; l++) {
        for (int kv = 0; kv < 2; kv++) {
            for (int e0 = 0; e0 < c->E; e0 += step) {
                GemmArgs g{};
                g.R = std::min(step, c->E - e0); g.H = H;
                g.A = (const float *) c->text_enc_ptr() + (size_t) e0 * H; g.lda = H;
                g.out = (float *) (c->cross_kv_ptr() + ((size_t) l * 2 + kv) * c->ECAP * H * 4) + (size_t) e0 * H;
                g.ldo = H;
                CHK(run_gemm(c, TTS_HIP_K_GEMM_OTHER, kv == 0 ? c->layers[l].ck : c->layers[l].cv, g, PRO_F32, EPI_STORE));
            }
        }
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// finalize
// ------------------------------------------------------------------------------------------------
template <typename T>
static int dmalloc(T **p, size_t n) {
    HIPCHK(hipMalloc((void **) p, n * sizeof(T)));
    HIPCHK(hipMemset(*p, 0, n * sizeof(T)));
    return 0;
}

extern "C" int tts_hip_finalize(tts_hip_ctx *c, void *external_arena) {
    if (!c) return set_err("null ctx");
    if (c->finalized) return set_err("tts_hip_finalize: already finalized");
    HIPCHK(hipSetDevice(c->device));
    CHK(plan(c));
    if (external_arena) { c->arena = (char *) external_arena; c->arena_external = true; }
    else HIPCHK(hipMalloc((void **) &c->arena, c->arena_bytes));
    bool all = true, any = false;
    for (auto &ci : c->copies) {
        Tensor &t = c->tensors[ci.src];
        if (t.has_data) {
            any = true;
            HIPCHK(hipMemcpy(c->arena + ci.dst, (const char *) t.tmp + ci.src_off, ci.bytes ? ci.bytes : t.nbytes, hipMemcpyDeviceToDevice));
        }
        else all = false;
    }
    if (any && !all) return set_err("tts_hip_finalize: some tensors were uploaded with data and some without");
    for (auto &t : c->tensors) { free_dev(t.second.tmp); t.second.tmp = nullptr; }
    c->weights_present = all;

    if (c->has_parler) {
        const int H = c->H;
        c->RMAX = std::max(max_rows_for(c), 1);
        if ((int) c->d.max_seqs > c->RMAX) return set_err("max_seqs=%u exceeds the %d rows one forward can carry with these weight types", c->d.max_seqs, c->RMAX);
        const size_t kv_esz = c->d.kv_type == TTS_HIP_F16 ? 2 : 4;
        const size_t kvb = (size_t) c->L * c->d.max_seqs * c->KVPOS * H * kv_esz;
        HIPCHK(hipMalloc(&c->kcache, kvb));
        HIPCHK(hipMalloc(&c->vcache, kvb));
        HIPCHK(hipMemset(c->kcache, 0, kvb));  // ggml_backend_buffer_clear(buf, 0), model.cpp:381
        HIPCHK(hipMemset(c->vcache, 0, kvb));
        const int R = c->RMAX;
        CHK(dmalloc(&c->x, (size_t) R * H));
        CHK(dmalloc(&c->q, (size_t) R * H));
        CHK(dmalloc(&c->att, (size_t) R * H));
        CHK(dmalloc(&c->dbg, (size_t) R * std::max(H, c->F)));
        CHK(dmalloc(&c->u32, (size_t) R * c->F));
        CHK(dmalloc(&c->u16, (size_t) R * c->F));
        CHK(dmalloc(&c->xn16, (size_t) R * H));
        CHK(dmalloc(&c->att16, (size_t) R * H));
        CHK(dmalloc(&c->partials, (size_t) 8 * R * H));
        CHK(dmalloc(&c->aq, (size_t) R * std::max(H, c->F)));
        CHK(dmalloc(&c->ad, (size_t) R * std::max(H, c->F) / 32));
        CHK(dmalloc(&c->logits, (size_t) R * c->NO * c->V));
        CHK(dmalloc(&c->part, (size_t) R * c->NH * 16 * ATT_PS));
        CHK(dmalloc(&c->attn_cnt, (size_t) R * c->NH));
        if (getenv("TTS_HIP_B1_STAMPS") && atoi(getenv("TTS_HIP_B1_STAMPS"))) CHK(dmalloc(&c->b1_stamps, (size_t) 16 * (c->L * 8 + 8)));
        CHK(dmalloc(&c->d_ids, (size_t) R * c->NO));
        CHK(dmalloc(&c->d_pos, (size_t) R));
        CHK(dmalloc(&c->d_seq, (size_t) R));
        CHK(dmalloc(&c->d_gather, (size_t) R * (c->NO + 4)));
        CHK(dmalloc(&c->d_tok, (size_t) R * c->NO));
        CHK(dmalloc(&c->d_step, (size_t) R));
        CHK(dmalloc(&c->d_steps_done, (size_t) R));
        CHK(dmalloc(&c->d_eos, (size_t) R * c->NO));
        CHK(dmalloc(&c->d_last, (size_t) R * c->NO));
        CHK(dmalloc(&c->d_repc, (size_t) R * c->NO));
        HIPCHK(hipHostMalloc((void **) &c->h_ids, (size_t) R * c->NO * 4));
        HIPCHK(hipHostMalloc((void **) &c->h_pos, (size_t) R * 4));
        HIPCHK(hipHostMalloc((void **) &c->h_seq, (size_t) R * 4));
        HIPCHK(hipHostMalloc((void **) &c->h_tok, (size_t) R * c->NO * 4));
        HIPCHK(hipHostMalloc((void **) &c->h_logits, (size_t) R * c->NO * c->V * 4));
    }
    if (c->has_llama && c->q4_native && c->gemv_rows && c->weights_present) {
        // the 4-bit codes of the Q4_0 matrices, repacked from the int8 expansion that the many-row MFMA path keeps using
        auto repack = [&](W &w) -> int {
            if (w.type != TTS_HIP_Q8I || !w.src_q4) return 0;
            const int64_t nbytes = w.N * w.K / 2;
            uint8_t *buf = nullptr;
            HIPCHK(hipMalloc((void **) &buf, (size_t) nbytes));
            c->q4_bufs.push_back(buf);
            hipLaunchKernelGGL(repack_i8_to_q4_kernel, dim3((unsigned) ((nbytes + 255) / 256)), dim3(256), 0, c->stream, (const int8_t *) (c->arena + w.off), buf, nbytes);
            HIPCHK(hipGetLastError());
            w.q4 = buf;
            return 0;
        };
        for (auto &y : c->l_layers) { CHK(repack(y.qkv)); CHK(repack(y.o)); CHK(repack(y.gu)); CHK(repack(y.down)); }
        CHK(repack(c->l_head));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    if (c->has_llama) {
        const int H = c->H, F = c->F, NCTX = (int) c->lm.n_ctx, QKV = (c->NH + 2 * (int) c->lm.n_kv_heads) * (int) c->lm.head_dim;
        c->RMAX = 256;
        // down_proj: K = F columns in slices of at most 4096 (16 waves x 256) per workgroup
        c->l_ksplit = 1;
        while (F / c->l_ksplit > 4096 || (F % c->l_ksplit)) c->l_ksplit++;
        if ((F / c->l_ksplit) % 256) c->l_ksplit = 1;
        const size_t kvb = (size_t) c->L * NCTX * c->l_kvH;
        c->attn_part_cap = (size_t) 4 * c->NH * 16;   // up to 4 rows x heads x 16 splits (more rows take the unsplit kernel)
        CHK(dmalloc(&c->attn_part, c->attn_part_cap * ATTN_PART));
        CHK(dmalloc(&c->l_kc, kvb));   // ggml_backend_buffer_clear(buf, 0), orpheus/model.cpp:181
        CHK(dmalloc(&c->l_vc, kvb));
        const int R = c->RMAX;
        CHK(dmalloc(&c->l_x, (size_t) R * H));
        CHK(dmalloc(&c->l_xn, (size_t) R * H));
        CHK(dmalloc(&c->l_qkv, (size_t) R * QKV));
        CHK(dmalloc(&c->l_att, (size_t) R * c->NH * c->lm.head_dim));
        CHK(dmalloc(&c->l_gu, (size_t) R * 2 * F));
        CHK(dmalloc(&c->l_g, (size_t) R * F));
        CHK(dmalloc(&c->l_parts, (size_t) 8 * R * H));
        CHK(dmalloc(&c->l_logits, (size_t) c->l_Vpad));
        CHK(dmalloc(&c->dbg, (size_t) R * std::max(H, F)));
        CHK(dmalloc(&c->aq, (size_t) R * std::max(std::max(H, F), c->NH * (int) c->lm.head_dim)));
        CHK(dmalloc(&c->ad, (size_t) R * std::max(std::max(H, F), c->NH * (int) c->lm.head_dim) / 32 + 1));
        CHK(dmalloc(&c->l_ids, (size_t) R));
        CHK(dmalloc(&c->l_pos, (size_t) R));
        CHK(dmalloc(&c->l_tok, (size_t) 1 + 2 * ARGMAX_PARTS + LLAMA_GREEDY_CHUNK + 1));
        HIPCHK(hipMalloc((void **) &c->l_cand, (size_t) TOPK_PARTS * TOPK_MAXK * 8));
        CHK(dmalloc(&c->l_smp, (size_t) 4));
    }
    if (c->has_dia) {
        const tts_hip_dia_desc &dd = c->dia;
        const int S = (int) dd.max_ctx, G = (int) dd.max_gen, EH = c->di_EH, EF = c->di_EF, DH = c->H, DF = c->di_DF, A = c->di_A, kvH = c->di_kvH;
        const size_t n = (size_t) 2 * S;
        c->RMAX = 256;
        // decoder wo: K = DF columns in slices of at most 4096 (16 waves x 256) per workgroup, folded by the next rms norm
        c->di_ksplit = 1;
        while (DF / c->di_ksplit > 4096 || (DF % c->di_ksplit)) c->di_ksplit++;
        if ((DF / c->di_ksplit) % 256 || (c->d.flags & TTS_HIP_FLAG_VALU_GEMM) || c->gemv_rows) c->di_ksplit = 1;
        CHK(dmalloc(&c->di_ex, n * EH)); CHK(dmalloc(&c->di_exn, n * EH)); CHK(dmalloc(&c->di_eqkv, n * 3 * A)); CHK(dmalloc(&c->di_eatt, n * A));
        CHK(dmalloc(&c->di_egu, n * 2 * EF)); CHK(dmalloc(&c->di_eg, n * EF)); CHK(dmalloc(&c->di_ek, n * A)); CHK(dmalloc(&c->di_ev, n * A));
        CHK(dmalloc(&c->di_ckv, n * 2 * A));
        HIPCHK(hipMalloc((void **) &c->di_e16, n * (size_t) std::max(std::max(EH, EF), A) * 2));
        const int U = std::max(1, std::min((int) dd.max_utterances, 64)), R = 2 * U;
        c->di_U = U;
        c->di_slot_encoded.assign((size_t) U, 0);
        c->attn_part_cap = (size_t) R * c->NH * 16;
        CHK(dmalloc(&c->attn_part, c->attn_part_cap * ATTN_PART));
        CHK(dmalloc(&c->di_ck, (size_t) c->L * U * n * A));   // [L][2U][S][A], zero like the reference's cleared cache (dia/model.cpp:329)
        CHK(dmalloc(&c->di_cv, (size_t) c->L * U * n * A));
        CHK(dmalloc(&c->di_k, (size_t) c->L * R * G * kvH));  // [L][2U][G][kvH]
        CHK(dmalloc(&c->di_v, (size_t) c->L * R * G * kvH));
        CHK(dmalloc(&c->di_x, (size_t) R * DH)); CHK(dmalloc(&c->di_xn, (size_t) R * DH)); 
        // the projections of a step with <= 16 rows may arrive as up to DIA_STREAM_SLABS K-slice slabs of 16 rows (gemv_stream_kernels.h)
        const size_t RSL = std::max((size_t) R, (size_t) DIA_STREAM_SLABS * 16);
        CHK(dmalloc(&c->di_qkv, RSL * (A + 2 * kvH)));
        CHK(dmalloc(&c->di_q, RSL * A)); CHK(dmalloc(&c->di_att, (size_t) R * A)); CHK(dmalloc(&c->di_gu, RSL * 2 * DF)); CHK(dmalloc(&c->di_g, (size_t) R * DF));
        CHK(dmalloc(&c->di_parts, (size_t) 8 * c->RMAX * DH));
        CHK(dmalloc(&c->di_logits, (size_t) R * c->di_Vpad)); CHK(dmalloc(&c->di_guided, (size_t) U * c->NO * c->di_V));
        const int maxK = std::max(std::max(EH, EF), std::max(std::max(DH, DF), A));
        CHK(dmalloc(&c->dbg, (size_t) c->RMAX * maxK));
        CHK(dmalloc(&c->aq, (size_t) c->RMAX * maxK));
        CHK(dmalloc(&c->ad, (size_t) c->RMAX * maxK / 32 + 1));
        CHK(dmalloc(&c->di_tok, n)); CHK(dmalloc(&c->di_epos, n)); CHK(dmalloc(&c->di_eseq, n)); CHK(dmalloc(&c->di_kbeg, n)); CHK(dmalloc(&c->di_kend, n));
        CHK(dmalloc(&c->di_ids, (size_t) U * 16)); CHK(dmalloc(&c->di_pos, (size_t) R)); CHK(dmalloc(&c->di_seq, (size_t) R)); CHK(dmalloc(&c->di_cend, (size_t) R));
        CHK(dmalloc(&c->di_stok, (size_t) U * 16)); CHK(dmalloc(&c->di_loop, (size_t) 3 * U)); CHK(dmalloc(&c->di_hist, (size_t) U * G * c->NO));
        CHK(dmalloc(&c->d_last, (size_t) U * c->NO)); CHK(dmalloc(&c->d_repc, (size_t) U * c->NO));
        HIPCHK(hipHostMalloc((void **) &c->h_di, ((size_t) U * 16 + 2 * (size_t) R) * 4));
        std::vector<uint32_t> cend((size_t) R, (uint32_t) S);
        HIPCHK(hipMemcpy(c->di_cend, cend.data(), (size_t) R * 4, hipMemcpyHostToDevice));
    }
    if (c->has_t5) {
        const int H = c->H, F = c->F, S = (int) c->t5.max_ctx_length;
        c->RMAX = 256;
        CHK(dmalloc(&c->t5_x, (size_t) S * H));
        CHK(dmalloc(&c->t5_qkv, (size_t) S * 3 * H));
        CHK(dmalloc(&c->t5_att, (size_t) S * H));
        CHK(dmalloc(&c->t5_ug, (size_t) S * 2 * F));
        CHK(dmalloc(&c->t5_g, (size_t) S * F));
        CHK(dmalloc(&c->t5_y, (size_t) S * std::max(H, c->t5_out)));
        CHK(dmalloc(&c->dbg, (size_t) S * std::max(H, F)));
        CHK(dmalloc(&c->aq, (size_t) c->RMAX * std::max(H, F)));
        CHK(dmalloc(&c->ad, (size_t) c->RMAX * std::max(H, F) / 32 + 1));
        CHK(dmalloc(&c->t5_ids, (size_t) S));
        // relative position buckets, t5_runner::set_inputs (t5/model.cpp:303-316) — a function of key - query only;
        // evaluated here with the reference's arithmetic (float denominator, integer division inside the log, double log)
        std::vector<int> tab((size_t) 2 * S - 1);
        const int n_buckets = (int) c->t5.n_buckets / 2, max_exact = n_buckets / 2;
        const float logarithmic_denominator = (float) log(128.0 / max_exact);
        for (int delta = -(S - 1); delta <= S - 1; delta++) {
            const int ab_rpos = abs(delta);
            int v = ab_rpos;
            if (ab_rpos >= max_exact) v = std::min(n_buckets - 1, max_exact + (int) ((log((double) (ab_rpos / max_exact)) / logarithmic_denominator) * max_exact));
            tab[(size_t) (delta + S - 1)] = (delta > 0 ? n_buckets : 0) + v;
        }
        HIPCHK(hipMalloc((void **) &c->t5_bucket, tab.size() * 4));
        HIPCHK(hipMemcpy(c->t5_bucket, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    }
    if (c->has_dac) {
        // largest activation per frame over all stages (C * L / frames)
        size_t mx = (size_t) std::max(c->d_latent, c->d_c0), up = 1;
        for (auto &b : c->dblocks) { mx = std::max(mx, (size_t) b.cin * up); up *= b.stride; mx = std::max(mx, (size_t) b.cout * up); }
        c->dac_frame_elems = mx;
    }
    c->finalized = true;
    if (c->weights_present) CHK(compute_cross_kv(c));
    return 0;
}

extern "C" int tts_hip_arena_filled(tts_hip_ctx *c) {
    if (!c || !c->finalized) return set_err("tts_hip_arena_filled: context not finalized");
    c->weights_present = true;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// RCCL weight broadcast (the path's only collective).  librccl is opened lazily: 570 MB that a single-GPU user never maps.
// ------------------------------------------------------------------------------------------------
#include <dlfcn.h>
#include <rccl/rccl.h>
namespace {
struct RcclApi {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi *rccl() {
    static RcclApi api;
    static std::atomic<int> state{0};   // 0 untried, 1 ready, -1 failed
    static std::mutex m;
    if (state.load() == 1) return &api;
    std::lock_guard<std::mutex> lock(m);
    if (state.load() == 1) return &api;
    if (state.load() == -1) return nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        api.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (api.h) break;
    }
    if (!api.h) { state = -1; return nullptr; }
#define RCCL_SYM(field, sym) api.field = (decltype(api.field)) dlsym(api.h, sym); if (!api.field) { state = -1; return nullptr; }
    RCCL_SYM(GetUniqueId, "ncclGetUniqueId") RCCL_SYM(CommInitRank, "ncclCommInitRank") RCCL_SYM(CommInitAll, "ncclCommInitAll")
    RCCL_SYM(CommDestroy, "ncclCommDestroy") RCCL_SYM(Broadcast, "ncclBroadcast") RCCL_SYM(GroupStart, "ncclGroupStart")
    RCCL_SYM(GroupEnd, "ncclGroupEnd") RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef RCCL_SYM
    state = 1;
    return &api;
}
#define NCCLCHK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return set_err("%s failed: %s", #x, api->GetErrorString(r_)); } while (0)

int check_broadcast_ctx(tts_hip_ctx *c, int i, size_t bytes) {
    if (!c) return set_err("tts_hip_broadcast_weights: context %d is NULL", i);
    if (!c->finalized || !c->arena) return set_err("tts_hip_broadcast_weights: context %d is not finalized", i);
    if (c->arena_bytes != bytes) return set_err("tts_hip_broadcast_weights: context %d has an arena of %zu bytes, the root %zu (different models?)", i, c->arena_bytes, bytes);
    return 0;
}
}  // namespace

extern "C" int tts_hip_broadcast_weights(tts_hip_ctx **ctxs, int n, int root) {
    if (!ctxs || n < 1) return set_err("tts_hip_broadcast_weights: no contexts");
    if (root < 0 || root >= n) return set_err("tts_hip_broadcast_weights: root %d outside 0..%d", root, n - 1);
    if (!ctxs[root]) return set_err("tts_hip_broadcast_weights: root context is NULL");
    const size_t bytes = ctxs[root]->arena_bytes;
    for (int i = 0; i < n; i++) CHK(check_broadcast_ctx(ctxs[i], i, bytes));
    if (!ctxs[root]->weights_present) return set_err("tts_hip_broadcast_weights: the root context holds no weights (declare-only)");
    for (int i = 0; i < n; i++)
        for (int j = i + 1; j < n; j++)
            if (ctxs[i]->device == ctxs[j]->device)
                return set_err("tts_hip_broadcast_weights: contexts %d and %d share device %d (contexts of one device share the arena: tts_hip_finalize(ctx, tts_hip_arena_ptr(other)))", i, j, ctxs[i]->device);
    if (n == 1) return 0;
    RcclApi *api = rccl();
    if (!api) return set_err("tts_hip_broadcast_weights: librccl.so could not be opened (%s)", dlerror() ? dlerror() : "symbols missing");
    std::vector<int> devs((size_t) n);
    for (int i = 0; i < n; i++) devs[(size_t) i] = ctxs[i]->device;
    std::vector<ncclComm_t> comms((size_t) n, nullptr);
    NCCLCHK(api->CommInitAll(comms.data(), n, devs.data()));
    int rc = 0;
    // <= 1 GiB pieces: one launch per piece and device, all devices of a piece inside one group
    const size_t piece = (size_t) 1 << 30;
    for (size_t off = 0; off < bytes && rc == 0; off += piece) {
        const size_t cnt = std::min(piece, bytes - off);
        ncclResult_t r = api->GroupStart();
        for (int i = 0; i < n && r == ncclSuccess; i++) {
            (void) hipSetDevice(ctxs[i]->device);
            r = api->Broadcast(ctxs[root]->arena + off, ctxs[i]->arena + off, cnt, ncclUint8, root, comms[(size_t) i], ctxs[i]->stream);
        }
        const ncclResult_t e = api->GroupEnd();
        if (r == ncclSuccess) r = e;
        if (r != ncclSuccess) rc = set_err("ncclBroadcast failed: %s", api->GetErrorString(r));
    }
    for (int i = 0; i < n; i++) {
        (void) hipSetDevice(ctxs[i]->device);
        if (hipStreamSynchronize(ctxs[i]->stream) != hipSuccess && rc == 0) rc = set_err("tts_hip_broadcast_weights: stream sync failed on device %d", ctxs[i]->device);
    }
    for (ncclComm_t cm : comms) if (cm) (void) api->CommDestroy(cm);
    if (rc) return rc;
    for (int i = 0; i < n; i++) if (i != root) ctxs[i]->weights_present = true;
    return 0;
}

extern "C" int tts_hip_comm_unique_id(void *id128) {
    if (!id128) return set_err("tts_hip_comm_unique_id: null buffer");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    RcclApi *api = rccl();
    if (!api) return set_err("tts_hip_comm_unique_id: librccl.so could not be opened");
    ncclUniqueId id;
    NCCLCHK(api->GetUniqueId(&id));
    memcpy(id128, &id, sizeof id);
    return 0;
}

extern "C" int tts_hip_broadcast_weights_rank(tts_hip_ctx *c, const void *id128, int rank, int world, int root) {
    if (!c || !id128) return set_err("tts_hip_broadcast_weights_rank: null argument");
    if (world < 1 || rank < 0 || rank >= world || root < 0 || root >= world) return set_err("tts_hip_broadcast_weights_rank: rank %d / root %d outside world %d", rank, root, world);
    if (!c->finalized || !c->arena) return set_err("tts_hip_broadcast_weights_rank: context is not finalized");
    if (rank == root && !c->weights_present) return set_err("tts_hip_broadcast_weights_rank: the root rank holds no weights");
    if (world == 1) return 0;
    RcclApi *api = rccl();
    if (!api) return set_err("tts_hip_broadcast_weights_rank: librccl.so could not be opened");
    HIPCHK(hipSetDevice(c->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t comm = nullptr;
    NCCLCHK(api->CommInitRank(&comm, world, id, rank));
    int rc = 0;
    const size_t piece = (size_t) 1 << 30;
    for (size_t off = 0; off < c->arena_bytes && rc == 0; off += piece) {
        const ncclResult_t r = api->Broadcast(c->arena + off, c->arena + off, std::min(piece, c->arena_bytes - off), ncclUint8, root, comm, c->stream);
        if (r != ncclSuccess) rc = set_err("ncclBroadcast failed: %s", api->GetErrorString(r));
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess && rc == 0) rc = set_err("tts_hip_broadcast_weights_rank: stream sync failed");
    (void) api->CommDestroy(comm);
    if (rc) return rc;
    if (rank != root) c->weights_present = true;
    return 0;
}

extern "C" int tts_hip_parler_set_text_encoding(tts_hip_ctx *c, const float *enc, uint32_t n_tokens) {
    if (!c || !c->finalized || !c->has_parler) return set_err("set_text_encoding: context not ready");
    if (!c->d.use_cross_attn) return set_err("set_text_encoding: cross attention disabled");
    if ((int) n_tokens > c->ECAP || n_tokens == 0) return set_err("set_text_encoding: %u tokens outside 1..%d", n_tokens, c->ECAP);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    // (stream copies, not hipMemcpy: the legacy stream may not be used while another context captures a graph)
    if (!c->cond_text_enc) {   // first new prompt of this context: its own copies from here on, the (possibly shared) arena is never written
        HIPCHK(hipMalloc((void **) &c->cond_text_enc, (size_t) c->ECAP * c->H * 4));
        HIPCHK(hipMalloc((void **) &c->cond_cross_kv, (size_t) c->L * 2 * c->ECAP * c->H * 4));
        HIPCHK(hipMemsetAsync(c->cond_text_enc, 0, (size_t) c->ECAP * c->H * 4, c->stream));
        HIPCHK(hipMemsetAsync(c->cond_cross_kv, 0, (size_t) c->L * 2 * c->ECAP * c->H * 4, c->stream));
    }
    HIPCHK(hipMemcpyAsync(c->text_enc_ptr(), enc, (size_t) n_tokens * c->H * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->E = (int) n_tokens;  // n_encode_length = conditional_prompt->n_outputs (model.cpp:135)
    for (auto &g : c->graphs) (void) hipGraphExecDestroy(g.second);  // E is baked into captured launches
    c->graphs.clear();
    return compute_cross_kv(c);
}

// ------------------------------------------------------------------------------------------------
// Parler entry points
// ------------------------------------------------------------------------------------------------
static int ready(tts_hip_ctx *c, const char *who) {
    if (!c) return set_err("%s: null ctx", who);
    if (!c->finalized) return set_err("%s: context not finalized", who);
    if (!c->weights_present) return set_err("%s: weights not present (declare-only context: fill the arena, then tts_hip_arena_filled)", who);
    if (!c->has_parler) return set_err("%s: context has no Parler decoder", who);
    HIPCHK(hipSetDevice(c->device));
    return 0;
}

extern "C" int tts_hip_parler_reset(tts_hip_ctx *c) {
    CHK(ready(c, "tts_hip_parler_reset"));
    return 0;  // positions are caller-supplied; the cache is overwritten position by position like the reference's
}

extern "C" int tts_hip_parler_prefill(tts_hip_ctx *c, uint32_t seq, const uint32_t *ids, uint32_t n, uint32_t pos0) {
    CHK(ready(c, "tts_hip_parler_prefill"));
    if (seq >= c->d.max_seqs) return set_err("prefill: seq %u >= max_seqs %u", seq, c->d.max_seqs);
    if (pos0 + n > (uint32_t) c->KVPOS || pos0 + n > (uint32_t) c->NPOS) return set_err("prefill: positions %u..%u exceed the %d cached positions", pos0, pos0 + n, c->KVPOS);
    for (uint32_t i = 0; i < n; i++) if (ids[i] >= (uint32_t) c->PV) return set_err("prefill: text id %u >= prompt vocab %d", ids[i], c->PV);
    for (uint32_t o = 0; o < n; o += c->RMAX) {
        const int R = (int) std::min<uint32_t>(c->RMAX, n - o);
        c->host_pos.resize(R);
        for (int r = 0; r < R; r++) {
            c->h_ids[r] = ids[o + r];
            c->h_pos[r] = pos0 + o + r;
            c->h_seq[r] = seq;
            c->host_pos[r] = pos0 + o + r;
        }
        HIPCHK(hipMemcpyAsync(c->d_ids, c->h_ids, (size_t) R * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_pos, c->h_pos, (size_t) R * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_seq, c->h_seq, (size_t) R * 4, hipMemcpyHostToDevice, c->stream));
        CHK(enqueue_forward(c, R, /*audio=*/false, /*logits=*/false, /*same_seq=*/true));
        HIPCHK(hipStreamSynchronize(c->stream));  // staging buffers are reused by the next chunk
    }
    return 0;
}

extern "C" int tts_hip_parler_prefill_batch(tts_hip_ctx *c, uint32_t n, const uint32_t *seqs, const uint32_t *ids,
                                            const uint32_t *lens, const uint32_t *pos0) {
    CHK(ready(c, "tts_hip_parler_prefill_batch"));
    if (!ids || !lens) return set_err("prefill_batch: null argument");
    // flatten to rows (seq, position, id); rows of one sequence stay in order, a forward carries up to RMAX rows
    std::vector<uint32_t> rs, rp, ri;
    size_t off = 0;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t s = seqs ? seqs[i] : i, p0 = pos0 ? pos0[i] : 0;
        if (s >= c->d.max_seqs) return set_err("prefill_batch: seq %u >= max_seqs %u", s, c->d.max_seqs);
        if (p0 + lens[i] > (uint32_t) c->KVPOS || p0 + lens[i] > (uint32_t) c->NPOS)
            return set_err("prefill_batch: positions %u..%u exceed the %d cached positions", p0, p0 + lens[i], c->KVPOS);
        for (uint32_t j = 0; j < lens[i]; j++) {
            if (ids[off + j] >= (uint32_t) c->PV) return set_err("prefill_batch: text id %u >= prompt vocab %d", ids[off + j], c->PV);
            rs.push_back(s); rp.push_back(p0 + j); ri.push_back(ids[off + j]);
        }
        off += lens[i];
    }
    for (size_t o = 0; o < rs.size(); o += (size_t) c->RMAX) {
        const int R = (int) std::min<size_t>((size_t) c->RMAX, rs.size() - o);
        c->host_pos.resize(R);
        for (int r = 0; r < R; r++) {
            c->h_ids[r] = ri[o + r]; c->h_pos[r] = rp[o + r]; c->h_seq[r] = rs[o + r]; c->host_pos[r] = rp[o + r];
        }
        HIPCHK(hipMemcpyAsync(c->d_ids, c->h_ids, (size_t) R * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_pos, c->h_pos, (size_t) R * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_seq, c->h_seq, (size_t) R * 4, hipMemcpyHostToDevice, c->stream));
        CHK(enqueue_forward(c, R, /*audio=*/false, /*logits=*/false, /*same_seq=*/false));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return 0;
}

enum { MODE_LOGITS = 0, MODE_GREEDY = 1, MODE_GEN = 2, MODE_GEN_SAMPLE = 3 };
// captured steps are keyed mode * GRAPH_KEY_ROWS + rows (rows <= TTS_HIP_MAX_ROWS, 1024 by default); drop_gen_graphs() recovers the mode from the
// key — with the two sites out of step (keys in units of 8192, the drop in units of 1000) the generation graphs were never dropped and a
// replay used whatever sampling parameters / tokens_out pointer its capture had baked in


static int stage_step_inputs(tts_hip_ctx *c, uint32_t n, const uint32_t *ids, const uint32_t *pos, const uint32_t *seqs) {
    if (n == 0 || (int) n > c->RMAX || n > c->d.max_seqs) return set_err("step: n_seqs=%u outside 1..%u", n, std::min<uint32_t>(c->RMAX, c->d.max_seqs));
    c->host_pos.resize(n);
    for (uint32_t r = 0; r < n; r++) {
        const uint32_t s = seqs ? seqs[r] : r;
        if (s >= c->d.max_seqs) return set_err("step: seq %u >= max_seqs %u", s, c->d.max_seqs);
        if (pos[r] >= (uint32_t) c->KVPOS || pos[r] >= (uint32_t) c->NPOS) return set_err("step: position %u exceeds the %d cached positions", pos[r], c->KVPOS);
        for (int i = 0; i < c->NO; i++) {
            const uint32_t id = ids[r * c->NO + i];
            if (id >= (uint32_t) c->EROWS) return set_err("step: audio id %u >= embedding rows %d", id, c->EROWS);
            c->h_ids[r * c->NO + i] = id;
        }
        c->h_pos[r] = pos[r];
        c->h_seq[r] = s;
        c->host_pos[r] = pos[r];
    }
    return 0;
}

// enqueue (or replay) one audio step for R rows in the given mode
static int enqueue_step_body(tts_hip_ctx *c, int R, int mode, uint32_t bos, uint32_t eos) {
    if (mode != MODE_GEN && mode != MODE_GEN_SAMPLE) {
        HIPCHK(hipMemcpyAsync(c->d_ids, c->h_ids, (size_t) R * c->NO * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_pos, c->h_pos, (size_t) R * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_seq, c->h_seq, (size_t) R * 4, hipMemcpyHostToDevice, c->stream));
    }
    CHK(enqueue_forward(c, R, true, true, false));
    if (mode == MODE_LOGITS) {
        HIPCHK(hipMemcpyAsync(c->h_logits, c->logits, (size_t) R * c->NO * c->V * 4, hipMemcpyDeviceToHost, c->stream));
    } else {
        CHK(prof_begin(c, TTS_HIP_K_SAMPLE, (double) R * c->NO * c->V * 4, 0));
        if (mode == MODE_GEN_SAMPLE) {
            SampleArgs sa{};
            sa.logits = c->logits; sa.V = c->V; sa.n_out = c->NO; sa.R = R;
            sa.top_k = c->smp.top_k; sa.top_p = c->smp.top_p; sa.temperature = c->smp.temperature;
            sa.uniforms = c->d_uniforms; sa.row_step = c->d_step; sa.out = c->d_tok;
            sa.pen_table = c->smp.repetition_penalty != 1.0f ? c->d_pen : nullptr; sa.pen_len = c->pen_len;
            sa.last_ids = c->d_last; sa.rep_counts = c->d_repc;
            sa.orig = c->d_seq; sa.R_total = c->gen_total;   // the loop's rows sit in cache slot = utterance index
            hipLaunchKernelGGL(sample_kernel, dim3(c->NO, R), dim3(256), 0, c->stream, sa);
        } else {
            hipLaunchKernelGGL(argmax_kernel, dim3(R * c->NO), dim3(256), 0, c->stream, (const float *) c->logits, c->V, c->d_tok);
        }
        HIPCHK(hipGetLastError());
        if (mode == MODE_GEN || mode == MODE_GEN_SAMPLE) {
            FeedArgs f{};
            f.tokens = c->d_tok; f.ids = c->d_ids; f.row_pos = c->d_pos; f.row_step = c->d_step; f.eos_seen = c->d_eos;
            f.steps_done = c->d_steps_done; f.tokens_out = c->d_tokens_out; f.R = R; f.n_out = c->NO; f.bos = bos; f.eos = eos;
            f.max_pos = (uint32_t) std::min(c->KVPOS, c->NPOS);
            f.orig = c->d_seq; f.R_total = c->gen_total;
            hipLaunchKernelGGL(feed_kernel, dim3(R), dim3(64), 0, c->stream, f);
            HIPCHK(hipGetLastError());
        }
        CHK(prof_end(c));
        if (mode == MODE_GREEDY) HIPCHK(hipMemcpyAsync(c->h_tok, c->d_tok, (size_t) R * c->NO * 4, hipMemcpyDeviceToHost, c->stream));
    }
    return 0;
}

static int run_step(tts_hip_ctx *c, int R, int mode, uint32_t bos, uint32_t eos) {
    const bool use_graph = !(c->d.flags & TTS_HIP_FLAG_NO_GRAPH) && !c->prof;
    if (!use_graph) return enqueue_step_body(c, R, mode, bos, eos);
    const int key = mode * GRAPH_KEY_ROWS + R;
    auto it = c->graphs.find(key);
    if (it == c->graphs.end()) {
        hipGraph_t graph = nullptr;
        HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        const int rc = enqueue_step_body(c, R, mode, bos, eos);
        const hipError_t e = hipStreamEndCapture(c->stream, &graph);
        if (rc != 0) { if (graph) (void) hipGraphDestroy(graph); return rc; }
        if (e != hipSuccess) return set_err("hipStreamEndCapture: %s", hipGetErrorString(e));
        hipGraphExec_t exec = nullptr;
        HIPCHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        (void) hipGraphDestroy(graph);
        it = c->graphs.emplace(key, exec).first;
    }
    HIPCHK(hipGraphLaunch(it->second, c->stream));
    return 0;
}

extern "C" int tts_hip_parler_step(tts_hip_ctx *c, uint32_t n, const uint32_t *ids, const uint32_t *pos,
                                   const uint32_t *seqs, float *logits_out) {
    CHK(ready(c, "tts_hip_parler_step"));
    if (!ids || !pos || !logits_out) return set_err("step: null argument");
    CHK(stage_step_inputs(c, n, ids, pos, seqs));
    CHK(run_step(c, (int) n, MODE_LOGITS, 0, 0));
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(logits_out, c->h_logits, (size_t) n * c->NO * c->V * 4);
    return 0;
}

extern "C" int tts_hip_parler_step_greedy(tts_hip_ctx *c, uint32_t n, const uint32_t *ids, const uint32_t *pos,
                                          const uint32_t *seqs, uint32_t *tokens_out) {
    CHK(ready(c, "tts_hip_parler_step_greedy"));
    if (!ids || !pos || !tokens_out) return set_err("step_greedy: null argument");
    CHK(stage_step_inputs(c, n, ids, pos, seqs));
    CHK(run_step(c, (int) n, MODE_GREEDY, 0, 0));
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(tokens_out, c->h_tok, (size_t) n * c->NO * 4);
    return 0;
}

static void drop_gen_graphs(tts_hip_ctx *c) {
    for (auto g = c->graphs.begin(); g != c->graphs.end();) {
        const int mode = g->first / GRAPH_KEY_ROWS;   // run_step's key
        if (mode == MODE_GEN || mode == MODE_GEN_SAMPLE) { (void) hipGraphExecDestroy(g->second); g = c->graphs.erase(g); } else ++g;
    }
}

// the device-resident generation loop; mode MODE_GEN (sampler::max) or MODE_GEN_SAMPLE (sample_kernel, c->smp / c->d_uniforms)
static int generate_loop(tts_hip_ctx *c, int mode, uint32_t n, const uint32_t *start_pos, uint32_t n_steps,
                         uint32_t bos, uint32_t eos, uint32_t *tokens_out, uint32_t *steps_done) {
    if (!start_pos || !tokens_out) return set_err("generate_greedy: null argument");
    if (n == 0 || (int) n > c->RMAX || n > c->d.max_seqs) return set_err("generate_greedy: n_seqs=%u out of range", n);
    if (bos >= (uint32_t) c->EROWS || eos >= (uint32_t) c->EROWS) return set_err("generate_greedy: bos/eos outside the embedding table");
    c->host_pos.resize(n);
    for (uint32_t r = 0; r < n; r++) {
        // a row may be asked for more steps than its cache holds: it finishes when its position reaches the end of the cache
        // (steps_done says after how many steps) and idles there while the other rows go on
        if (start_pos[r] >= (uint32_t) c->KVPOS || start_pos[r] >= (uint32_t) c->NPOS)
            return set_err("generate_greedy: sequence %u starts outside the cached positions (%u >= %d)", r, start_pos[r], c->KVPOS);
        for (int i = 0; i < c->NO; i++) c->h_ids[r * c->NO + i] = bos;  // model.cpp:781 with current_step == 0
        c->h_pos[r] = start_pos[r];
        c->h_seq[r] = r;
        c->host_pos[r] = start_pos[r];
    }
    const size_t need = (size_t) n_steps * n * c->NO;
    if (need > c->tokens_out_cap) {
        free_dev(c->d_tokens_out);
        c->d_tokens_out = nullptr;
        HIPCHK(hipMalloc((void **) &c->d_tokens_out, need * 4));
        c->tokens_out_cap = need;
        drop_gen_graphs(c);  // the captured graphs baked the old pointer in
    }
    for (uint32_t r = 0; r < n; r++) c->h_tok[r] = 1;  // current_step of the first audio decode (model.cpp:783-785)
    HIPCHK(hipMemcpyAsync(c->d_ids, c->h_ids, (size_t) n * c->NO * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_pos, c->h_pos, (size_t) n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_seq, c->h_seq, (size_t) n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_step, c->h_tok, (size_t) n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemsetAsync(c->d_eos, 0, (size_t) n * c->NO, c->stream));
    HIPCHK(hipMemsetAsync(c->d_steps_done, 0, (size_t) n * 4, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    // bos/eos are baked into the captured feed kernel: key the graph on them too
    if (c->g_bos != bos || c->g_eos != eos) {
        drop_gen_graphs(c);
        c->g_bos = bos; c->g_eos = eos;
    }
    if (c->gen_total != (int) n) {   // the utterance count is baked into the captured sampler / feed launches (tokens_out stride)
        drop_gen_graphs(c);
        c->gen_total = (int) n;
    }
    // Row compaction.  Every 32 steps the host looks at steps_done (one small D2H + sync) to see whether check_stopping() has fired for every
    // utterance; utterances that have finished (EOS on every head, or their position reached max_generation) used to idle in the lock-step
    // forward until the last one was done — a ragged batch paid for its longest row (165 against 323 audio-s/s at 1024 steps).  Now the
    // finished rows are dropped from the forward: the live rows are gathered to the front (ids, position, cache slot, step counter; everything
    // else is indexed by utterance) and the loop goes on with R' = the live count rounded up to a multiple of 128 (64 below 256) — finished
    // rows fill the remainder, so that a forward keeps whole row tiles and only a handful of row counts are ever captured as graphs.
    std::vector<uint32_t> row_utt(n);            // utterance of row r
    for (uint32_t r = 0; r < n; r++) row_utt[r] = r;
    uint32_t R = n, ran = 0;
    for (uint32_t s = 0; s < n_steps; s++) {
        for (uint32_t r = 0; r < R; r++) c->host_pos[r] = std::min<uint32_t>(start_pos[row_utt[r]] + s, (uint32_t) std::min(c->KVPOS, c->NPOS) - 1);
        CHK(run_step(c, (int) R, mode, bos, eos));
        ran = s + 1;
        if ((ran % 32) == 0 && ran < n_steps) {
            HIPCHK(hipMemcpyAsync(c->h_tok, c->d_steps_done, (size_t) n * 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            uint32_t live = 0;
            for (uint32_t r = 0; r < R; r++) live += c->h_tok[row_utt[r]] == 0;
            if (live == 0) break;
            const uint32_t q = live >= 256 ? 128 : 64;
            const uint32_t R2 = std::min(R, (live + q - 1) / q * q);
            if (c->gen_compact && R2 < R) {
                std::vector<uint32_t> map, fill;
                for (uint32_t r = 0; r < R; r++) (c->h_tok[row_utt[r]] == 0 ? map : fill).push_back(r);
                for (uint32_t i = 0; map.size() < R2; i++) map.push_back(fill[i]);
                std::sort(map.begin(), map.end());   // keep the row order: rows only move towards lower indices
                std::vector<uint32_t> utt2(R2);
                for (uint32_t r = 0; r < R2; r++) utt2[r] = row_utt[map[r]];
                GatherArgs ga{};
                ga.map = c->d_gather; ga.R2 = (int) R2; ga.n_out = c->NO;
                ga.ids = c->d_ids; ga.pos = c->d_pos; ga.seq = c->d_seq; ga.step = c->d_step;
                ga.s_ids = c->d_gather + c->RMAX; ga.s_pos = ga.s_ids + (size_t) c->RMAX * c->NO; ga.s_seq = ga.s_pos + c->RMAX; ga.s_step = ga.s_seq + c->RMAX;
                HIPCHK(hipMemcpyAsync(c->d_gather, map.data(), (size_t) R2 * 4, hipMemcpyHostToDevice, c->stream));
                hipLaunchKernelGGL(gather_rows_kernel, dim3(R2), dim3(64), 0, c->stream, ga, 0);
                hipLaunchKernelGGL(gather_rows_kernel, dim3(R2), dim3(64), 0, c->stream, ga, 1);
                HIPCHK(hipGetLastError());
                HIPCHK(hipStreamSynchronize(c->stream));   // map lives on the host stack
                row_utt.swap(utt2);
                R = R2;
            }
        }
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpyAsync(tokens_out, c->d_tokens_out, (size_t) ran * n * c->NO * 4, hipMemcpyDeviceToHost, c->stream));
    if (ran < n_steps) memset(tokens_out + (size_t) ran * n * c->NO, 0, (size_t) (n_steps - ran) * n * c->NO * 4);
    if (steps_done) HIPCHK(hipMemcpyAsync(steps_done, c->d_steps_done, (size_t) n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int tts_hip_parler_generate_greedy(tts_hip_ctx *c, uint32_t n, const uint32_t *start_pos, uint32_t n_steps,
                                              uint32_t bos, uint32_t eos, uint32_t *tokens_out, uint32_t *steps_done) {
    CHK(ready(c, "tts_hip_parler_generate_greedy"));
    return generate_loop(c, MODE_GEN, n, start_pos, n_steps, bos, eos, tokens_out, steps_done);
}

static int check_sampling(const tts_hip_ctx *c, const tts_hip_sampling *sp, const char *what) {
    if (!sp) return set_err("%s: null sampling parameters", what);
    if (c->V > SMP_VMAX) return set_err("%s: output vocabulary %d > %d (sample on the host from tts_hip_parler_step)", what, c->V, SMP_VMAX);
    if (!(sp->temperature > 0.0f)) return set_err("%s: temperature must be > 0", what);
    if (!(sp->top_p > 0.0f)) return set_err("%s: top_p must be > 0", what);
    if (!(sp->repetition_penalty > 0.0f)) return set_err("%s: repetition_penalty must be > 0 (1 = off)", what);
    return 0;
}

static int stage_uniforms(tts_hip_ctx *c, const float *uniforms, size_t count) {
    if (count > c->uniforms_cap) {
        free_dev(c->d_uniforms);
        c->d_uniforms = nullptr;
        HIPCHK(hipMalloc((void **) &c->d_uniforms, count * 4));
        c->uniforms_cap = count;
        drop_gen_graphs(c);
    }
    HIPCHK(hipMemcpyAsync(c->d_uniforms, uniforms, count * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// pow(penalty, count) for count = 0..n: evaluated here with the host libm, the arithmetic sampler.cpp:90 performs
static int stage_penalty(tts_hip_ctx *c, float penalty, int n) {
    if (penalty == 1.0f) return 0;
    if (n + 1 > c->pen_len) {
        free_dev(c->d_pen);
        c->d_pen = nullptr;
        HIPCHK(hipMalloc((void **) &c->d_pen, (size_t) (n + 1) * 8));
        c->pen_len = n + 1;
        drop_gen_graphs(c);
    }
    std::vector<double> t((size_t) c->pen_len);
    for (int i = 0; i < c->pen_len; i++) t[(size_t) i] = pow((double) penalty, (double) i);
    HIPCHK(hipMemcpyAsync(c->d_pen, t.data(), t.size() * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int tts_hip_parler_generate_sampled(tts_hip_ctx *c, uint32_t n, const uint32_t *start_pos, uint32_t n_steps,
                                               uint32_t bos, uint32_t eos, const tts_hip_sampling *sp, const float *uniforms,
                                               uint32_t *tokens_out, uint32_t *steps_done) {
    CHK(ready(c, "tts_hip_parler_generate_sampled"));
    CHK(check_sampling(c, sp, "tts_hip_parler_generate_sampled"));
    if (!uniforms) return set_err("tts_hip_parler_generate_sampled: null uniforms");
    if (n == 0 || (int) n > c->RMAX || n > c->d.max_seqs) return set_err("generate_sampled: n_seqs=%u out of range", n);
    if (sp->top_k != c->smp.top_k || sp->top_p != c->smp.top_p || sp->temperature != c->smp.temperature ||
        (sp->repetition_penalty != 1.0f) != (c->smp.repetition_penalty != 1.0f)) {
        drop_gen_graphs(c);  // parameters are baked into the captured sample_kernel launch
    }
    c->smp = *sp;
    CHK(stage_uniforms(c, uniforms, (size_t) n_steps * n * c->NO));
    CHK(stage_penalty(c, sp->repetition_penalty, (int) n_steps));
    if (sp->repetition_penalty != 1.0f) {  // sampler::reset (sampler.cpp:71-80)
        HIPCHK(hipMemsetAsync(c->d_last, 0xFF, (size_t) n * c->NO * 4, c->stream));
        HIPCHK(hipMemsetAsync(c->d_repc, 0, (size_t) n * c->NO * 4, c->stream));
    }
    return generate_loop(c, MODE_GEN_SAMPLE, n, start_pos, n_steps, bos, eos, tokens_out, steps_done);
}

extern "C" int tts_hip_sample_logits(tts_hip_ctx *c, uint32_t n_rows, const float *logits, const tts_hip_sampling *sp,
                                     const float *uniforms, int32_t *last_ids, uint32_t *rep_counts, uint32_t *tokens_out) {
    CHK(ready(c, "tts_hip_sample_logits"));
    CHK(check_sampling(c, sp, "tts_hip_sample_logits"));
    if (!logits || !uniforms || !tokens_out) return set_err("tts_hip_sample_logits: null argument");
    if (n_rows == 0 || (int) n_rows > c->RMAX) return set_err("tts_hip_sample_logits: n_rows=%u outside 1..%d", n_rows, c->RMAX);
    CHK(stage_uniforms(c, uniforms, (size_t) n_rows * c->NO));
    HIPCHK(hipMemcpyAsync(c->logits, logits, (size_t) n_rows * c->NO * c->V * 4, hipMemcpyHostToDevice, c->stream));
    SampleArgs sa{};
    sa.logits = c->logits; sa.V = c->V; sa.n_out = c->NO; sa.R = (int) n_rows;
    sa.top_k = sp->top_k; sa.top_p = sp->top_p; sa.temperature = sp->temperature;
    sa.uniforms = c->d_uniforms; sa.row_step = nullptr; sa.out = c->d_tok;
    const bool rep = sp->repetition_penalty != 1.0f;
    if (rep) {
        if (!last_ids || !rep_counts) return set_err("tts_hip_sample_logits: repetition penalty needs last_ids and rep_counts");
        uint32_t mx = 0;
        for (size_t i = 0; i < (size_t) n_rows * c->NO; i++) mx = std::max(mx, rep_counts[i]);
        CHK(stage_penalty(c, sp->repetition_penalty, (int) std::min<uint32_t>(mx + 2, 1u << 20)));
        HIPCHK(hipMemcpyAsync(c->d_last, last_ids, (size_t) n_rows * c->NO * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_repc, rep_counts, (size_t) n_rows * c->NO * 4, hipMemcpyHostToDevice, c->stream));
        sa.pen_table = c->d_pen; sa.pen_len = c->pen_len; sa.last_ids = c->d_last; sa.rep_counts = c->d_repc;
    }
    hipLaunchKernelGGL(sample_kernel, dim3(c->NO, n_rows), dim3(256), 0, c->stream, sa);
    HIPCHK(hipGetLastError());
    if (rep) {
        HIPCHK(hipMemcpyAsync(last_ids, c->d_last, (size_t) n_rows * c->NO * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(rep_counts, c->d_repc, (size_t) n_rows * c->NO * 4, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(hipMemcpyAsync(tokens_out, c->d_tok, (size_t) n_rows * c->NO * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}


// ------------------------------------------------------------------------------------------------
// T5 voice-prompt encoder (src/models/parler/t5/model.cpp:216-357)
// ------------------------------------------------------------------------------------------------
extern "C" tts_hip_ctx *tts_hip_t5_create(int device, const tts_hip_t5_desc *td) {
    if (!td || td->struct_size != sizeof(tts_hip_t5_desc)) { set_err("tts_hip_t5_create: bad desc (struct_size mismatch)"); return nullptr; }
    tts_hip_desc d{};
    d.struct_size = sizeof(d);
    d.hidden_size = td->hidden_size; d.n_layers = td->n_layers; d.n_attn_heads = td->n_attn_heads; d.max_ctx_length = td->max_ctx_length;
    d.max_seqs = 1; d.gelu_mode = td->gelu_mode;
    d.flags = (td->flags & (TTS_HIP_FLAG_VALU_GEMM | TTS_HIP_FLAG_DEQUANT_Q | TTS_HIP_FLAG_NO_GRAPH)) | TTS_HIP_FLAG_NO_PARLER | TTS_HIP_FLAG_NO_DAC;
    tts_hip_ctx *c = tts_hip_create(device, &d);
    if (!c) return nullptr;
    c->has_t5 = true;
    c->t5 = *td;
    if (c->t5.n_buckets == 0) c->t5.n_buckets = 32;  // t5/model.h:48
    return c;
}

static int t5_gemm(tts_hip_ctx *c, const W &w, const float *A, int lda, float *out, int ldo, int n, int epi) {
    for (int r0 = 0; r0 < n; r0 += c->RMAX) {  // the activation-quantisation scratch holds RMAX rows
        GemmArgs g{};
        g.R = std::min(c->RMAX, n - r0); g.H = c->H; g.gelu_mode = (int) c->d.gelu_mode;
        g.A = A + (size_t) r0 * lda; g.lda = lda;
        g.out = out + (size_t) r0 * ldo; g.ldo = ldo;
        CHK(run_gemm(c, TTS_HIP_K_GEMM_OTHER, w, g, PRO_F32, epi));
    }
    return 0;
}

extern "C" int tts_hip_t5_encode(tts_hip_ctx *c, const uint32_t *ids, uint32_t n_tokens, float *out) {
    if (!c || !c->has_t5) return set_err("tts_hip_t5_encode: not a T5 context (tts_hip_t5_create)");
    if (!c->finalized || !c->weights_present) return set_err("tts_hip_t5_encode: context not finalized");
    if (!ids || !out) return set_err("tts_hip_t5_encode: null argument");
    if (n_tokens == 0 || n_tokens > c->t5.max_ctx_length) return set_err("tts_hip_t5_encode: %u tokens outside 1..%u (t5encoder.context_length)", n_tokens, c->t5.max_ctx_length);
    for (uint32_t i = 0; i < n_tokens; i++)
        if (ids[i] >= (uint32_t) c->t5_vocab) return set_err("tts_hip_t5_encode: token id %u >= vocabulary %d", ids[i], c->t5_vocab);
    HIPCHK(hipSetDevice(c->device));
    const int n = (int) n_tokens, H = c->H, F = c->F, S = (int) c->t5.max_ctx_length;
    auto f32 = [&](size_t off) { return (const float *) (c->arena + off); };
    HIPCHK(hipMemcpyAsync(c->t5_ids, ids, (size_t) n * 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(t5_embed_kernel, dim3(n), dim3(256), 0, c->stream, f32(c->t5_embd), (const uint32_t *) c->t5_ids, H, c->t5_x);
    HIPCHK(hipGetLastError());
    auto rms = [&](const float *x, size_t w_off, float *y) {
        hipLaunchKernelGGL(t5_rms_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, c->stream, x, H, f32(w_off), y, n);
        return hipGetLastError() == hipSuccess ? 0 : set_err("t5_rms_rows_kernel launch failed");
    };
    for (int l = 0; l < c->L; l++) {
        const auto &y = c->t5_layers[l];
        CHK(rms(c->t5_x, y.attn_norm, c->dbg));
        CHK(t5_gemm(c, y.qkv, c->dbg, H, c->t5_qkv, 3 * H, n, EPI_STORE));
        hipLaunchKernelGGL(t5_attn_kernel, dim3(c->NH, n), dim3(64), (size_t) (64 + n) * 4, c->stream, (const float *) c->t5_qkv, n, H, c->NH,
                           (const int *) c->t5_bucket, S, f32(c->t5_relb), c->t5_att);
        HIPCHK(hipGetLastError());
        CHK(t5_gemm(c, y.o, c->t5_att, H, c->t5_x, H, n, EPI_RESID));          // ggml_add(attn_out, residual) :267
        CHK(rms(c->t5_x, y.mlp_norm, c->dbg));
        CHK(t5_gemm(c, y.wi, c->dbg, H, c->t5_ug, 2 * F, n, EPI_STORE));
        hipLaunchKernelGGL(t5_gated_gelu_kernel, dim3((unsigned) (((size_t) n * F + 255) / 256)), dim3(256), 0, c->stream, (const float *) c->t5_ug, F, n,
                           (int) c->d.gelu_mode, c->t5_g);
        HIPCHK(hipGetLastError());
        CHK(t5_gemm(c, y.wo, c->t5_g, F, c->t5_x, H, n, EPI_RESID));           // :278
    }
    float *result = c->dbg;
    CHK(rms(c->t5_x, c->t5_out_norm, c->dbg));
    if (c->t5_has_down) {
        CHK(t5_gemm(c, c->t5_down, c->dbg, H, c->t5_y, c->t5_out, n, EPI_STORE));
        if (c->t5_has_down_b) {
            hipLaunchKernelGGL(t5_add_bias_kernel, dim3((unsigned) (((size_t) n * c->t5_out + 255) / 256)), dim3(256), 0, c->stream, c->t5_y, f32(c->t5_down_b),
                               c->t5_out, n);
            HIPCHK(hipGetLastError());
        }
        result = c->t5_y;
    }
    HIPCHK(hipMemcpyAsync(out, result, (size_t) n * c->t5_out * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int tts_hip_t5_output_size(tts_hip_ctx *c) {
    if (!c || !c->has_t5 || !c->planned) { set_err("tts_hip_t5_output_size: not a planned T5 context"); return -1; }
    return c->t5_out;
}

// ------------------------------------------------------------------------------------------------
// DAC
// ------------------------------------------------------------------------------------------------
// [C][L] out of a device tensor whose rows are LS apart
static int dac_snapshot(tts_hip_ctx *c, int stage, const float *dev, size_t C, size_t L, size_t LS) {
    if (!c->debug) return 0;
    HIPCHK(hipStreamSynchronize(c->stream));
    std::vector<float> &v = c->dac_dbg[stage];
    v.resize(C * L);
    HIPCHK(hipMemcpy2DAsync(v.data(), L * 4, dev, LS * 4, L * 4, C, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// ---- MFMA tile selection (shared by the packer and the launchers) ---------------------------------
// input channels per LDS chunk of the fp32 k=1 kernels: 16 keeps the footprint at 40-45 KB (3-4 workgroups per CU);
// with 32 the 96- and 64-channel tiles needed 80-90 KB = ONE workgroup per CU, and a k=1 conv has only C/32 chunks to
// pipeline over, so its residual loads and stores ran with nothing to overlap (0.88 TB/s)
#define CI32_K1 16
// k=7: 4 channels per chunk (40 KB, 4 workgroups per CU) measured 3.7 % faster than 8 (75 KB, 2 per CU)
#ifndef CI32_T
#define CI32_T 8
#endif
#ifndef CI32_K7
#define CI32_K7 4
#endif
static int conv_tile(int cout, int K, int *CO_T, int *CI_T) {
    if (K != 7 && K != 1) return -1;
    *CI_T = K == 7 ? CI32_K7 : CI32_K1;
    if (cout % 128 == 0) { *CO_T = 128; return 0; }
    if (cout % 96 == 0 && cout % 64 != 0) { *CO_T = 96; return 1; }
    if (cout % 64 == 0) { *CO_T = 64; return 2; }
    return -1;
}
// the k = 1 conv of a residual unit at <= 192 channels goes through conv1x1_direct_kernel (96-channel tiles; fp32 tensors only)
static bool conv1_direct(const tts_hip_ctx *c, int cout, int cin) {
    return c->dac_conv1_direct && !c->dac_f16 && cout == cin && (cin == 96 || cin == 192);
}
static int convt_tile(int cout, int s, int *CO_T) {
    if (s == 8 && cout % 64 == 0) { *CO_T = 64; return 0; }
    if (s == 4 && cout % 64 == 0) { *CO_T = 64; return 1; }
    if (s == 2 && cout % 96 == 0) { *CO_T = 96; return 2; }
    if (s == 2 && cout % 64 == 0) { *CO_T = 64; return 3; }
    return -1;
}

static int pack_one(tts_hip_ctx *c, size_t w_off, int cout, int cin, int KT, int CO_T, int CI_T, bool transposed) {
    const int n_chunks = (cin + CI_T - 1) / CI_T;
    const size_t n = (size_t) ((cout + CO_T - 1) / CO_T) * n_chunks * KT * CI_T * CO_T;
    float *dst = nullptr;
    HIPCHK(hipMalloc((void **) &dst, n * 4));
    hipLaunchKernelGGL(pack_conv_w_kernel, dim3(1024), dim3(256), 0, c->stream, (const float *) (c->arena + w_off), dst, cout, cin, KT,
                       CO_T, CI_T, n_chunks, transposed ? 1 : 0);
    HIPCHK(hipGetLastError());
    c->packed[w_off] = dst;
    return 0;
}

// one-time re-layout of the DAC conv weights into MFMA LDS images (after the arena holds the weights,
// i.e. also after an RCCL broadcast filled it)
static int pack_one16(tts_hip_ctx *c, size_t w_off, int cout, int cin, int KT, int CO_T, int CI_T, bool transposed) {
    const int n_chunks = (cin + CI_T - 1) / CI_T;
    const size_t n = (size_t) ((cout + CO_T - 1) / CO_T) * n_chunks * KT * CI_T * CO_T;
    _Float16 *dst = nullptr;
    HIPCHK(hipMalloc((void **) &dst, n * 2));
    hipLaunchKernelGGL(pack_conv_w16_kernel, dim3(1024), dim3(256), 0, c->stream, (const float *) (c->arena + w_off), dst, cout, cin, KT,
                       CO_T, CI_T, n_chunks, transposed ? 1 : 0);
    HIPCHK(hipGetLastError());
    c->packed16[w_off] = dst;
    return 0;
}
static int pack_one_b3(tts_hip_ctx *c, size_t w_off, int cout, int cin, int CO_T) {   // k = 7, 64- or 96-channel tiles, 8 input channels per chunk
    const int n_chunks = (cin + 7) / 8;
    const size_t n = (size_t) ((cout + CO_T - 1) / CO_T) * n_chunks * 3 * 8 * CO_T * 8;
    __bf16 *dst = nullptr;
    HIPCHK(hipMalloc((void **) &dst, n * 2));
    hipLaunchKernelGGL(pack_conv_w_b3_kernel, dim3(1024), dim3(256), 0, c->stream, (const float *) (c->arena + w_off), dst, cout, cin, CO_T, n_chunks);
    HIPCHK(hipGetLastError());
    c->packed_b3[w_off] = dst;
    return 0;
}
// residual units of 96 / 192 channels as one launch (resunit_b3_kernel): k-steps per stage of the k = 7 / k = 1 part
static bool resunit_shape(int C, int *KS, int *KS2) {
    if (C == 96) { *KS = 4; *KS2 = 3; return true; }
    if (C == 192) { *KS = 2; *KS2 = 4; return true; }
    return false;
}
static int pack_resunit(tts_hip_ctx *c, const DRes &r, int C) {
    int KS = 0, KS2 = 0;
    if (!resunit_shape(C, &KS, &KS2)) return 0;
    __bf16 *dst = nullptr;
    if (c->dac_tap7) {   // resunit_t7_kernel: one tap per k-step, stages of {4, 3} / {2, 2, 2, 1} k-steps per 16-channel chunk
        const int MI = C / 32, SPC = MI == 3 ? 2 : 4, MAXCNT = MI == 3 ? 4 : 2;
        const size_t WST = (size_t) 3 * MAXCNT * 2 * C * 8;
        const size_t n = (size_t) ((C / 16) * SPC + (C / 96) * ((C / 16) / KS2) + 1) * WST;
        HIPCHK(hipMalloc((void **) &dst, n * 2));
        HIPCHK(hipMemsetAsync(dst, 0, n * 2, c->stream));
        hipLaunchKernelGGL(pack_resunit_t7_kernel, dim3(1024), dim3(256), 0, c->stream, (const float *) (c->arena + r.in_w), (const float *) (c->arena + r.out_w), dst, C, KS2);
    } else {
        const ResUnitGeom g = resunit_geom(C, KS, KS2);
        const size_t n = (size_t) (g.n7 + g.n1 + 1) * g.WST;   // + 1: the prefetch of the stage after the last one stays inside the buffer
        HIPCHK(hipMalloc((void **) &dst, n * 2));
        HIPCHK(hipMemsetAsync(dst, 0, n * 2, c->stream));
        hipLaunchKernelGGL(pack_resunit_b3_kernel, dim3(1024), dim3(256), 0, c->stream, (const float *) (c->arena + r.in_w), (const float *) (c->arena + r.out_w), dst, C, KS, KS2);
    }
    HIPCHK(hipGetLastError());
    c->packed_ru[r.in_w] = dst;
    return 0;
}
// transposed convs as bf16 x 3 products (convt_b3_kernel): 32 MI output channels per workgroup
static int convt_b3_tile(int cout, int cin, int s) {
    if (cin % 16) return 0;
    if (s == 8 && cout % 32 == 0) return 32;
    if (s == 4 && cout % 64 == 0) return 64;
    if (s == 2 && cout % 96 == 0) return 96;
    return 0;
}
static int pack_convt_b3(tts_hip_ctx *c, const DBlock &b) {
    const int CO_T = convt_b3_tile(b.cout, b.cin, b.stride);
    if (!CO_T) return 0;
    const int n_chunks = b.cin / 16;
    const size_t n = (size_t) (b.cout / CO_T) * n_chunks * 3 * 2 * b.stride * 2 * CO_T * 8;
    __bf16 *dst = nullptr;
    HIPCHK(hipMalloc((void **) &dst, n * 2));
    hipLaunchKernelGGL(pack_convt_w_b3_kernel, dim3(1024), dim3(256), 0, c->stream, (const float *) (c->arena + b.w), dst, b.cout, b.cin, b.stride, CO_T, n_chunks);
    HIPCHK(hipGetLastError());
    c->packed_ct[b.w] = dst;
    return 0;
}
// wide classes on split planes (conv_b3p_kernel): k = 7 in 64-channel tiles (four k-steps per 8-channel chunk), k = 1 in 128-channel
// tiles (one k-step = 16 channels per chunk)
static bool planes_class(const tts_hip_ctx *c, int ch) {
    int ks = 0, ks2 = 0;
    return c->dac_planes && c->dac_b3 && !c->dac_f16 && ch % 128 == 0 && !(c->dac_fuse && resunit_shape(ch, &ks, &ks2));
}
static int pack_planes(tts_hip_ctx *c, size_t w_off, int cout, int cin, int KT) {
    const bool tapk = KT == 7 && c->dac_tap7 && cin % 16 == 0;
    const int CO_T = KT == 7 ? 64 : ((c->dac_k1_variant == 1 && cout % 256 == 0) ? 256 : 128), NS = KT == 7 ? (tapk ? 7 : 4) : 1;
    const int n_chunks = KT == 7 && !tapk ? cin / 8 : cin / 16;
    const size_t n = (size_t) (cout / CO_T) * n_chunks * 3 * NS * 2 * CO_T * 8;
    __bf16 *dst = nullptr;
    HIPCHK(hipMalloc((void **) &dst, n * 2));
    hipLaunchKernelGGL(pack_conv_w_b3p_kernel, dim3(1024), dim3(256), 0, c->stream, (const float *) (c->arena + w_off), dst, cout, cin, KT, CO_T, NS, n_chunks);
    HIPCHK(hipGetLastError());
    c->packed_p[w_off] = dst;
    return 0;
}
#define CI16_K7 16
#define CI16_K1 32
#define CI16_T  16

static int ensure_packed(tts_hip_ctx *c) {
    if (c->dac_packed || (c->d.flags & TTS_HIP_FLAG_VALU_GEMM)) return 0;
    int CO_T = 0, CI_T = 0;
    if (c->dac_f16) {
        if (conv_tile(c->d_c0, 7, &CO_T, &CI_T) >= 0) CHK(pack_one16(c, c->d_initw, c->d_c0, c->d_latent, 7, CO_T, CI16_K7, false));
        for (auto &b : c->dblocks) {
            if (convt_tile(b.cout, b.stride, &CO_T) >= 0) CHK(pack_one16(c, b.w, b.cout, b.cin, 2 * b.stride, CO_T, CI16_T, true));
            for (int r = 0; r < 3; r++) {
                if (conv_tile(b.cout, 7, &CO_T, &CI_T) >= 0) CHK(pack_one16(c, b.res[r].in_w, b.cout, b.cout, 7, CO_T, CI16_K7, false));
                if (conv_tile(b.cout, 1, &CO_T, &CI_T) >= 0) CHK(pack_one16(c, b.res[r].out_w, b.cout, b.cout, 1, CO_T, CI16_K1, false));
            }
        }
        HIPCHK(hipStreamSynchronize(c->stream));
        c->dac_packed = true;
        return 0;
    }
    if (conv_tile(c->d_c0, 7, &CO_T, &CI_T) >= 0) CHK(pack_one(c, c->d_initw, c->d_c0, c->d_latent, 7, CO_T, CI_T, false));
    if (c->dac_b3 && c->d_c0 % 64 == 0) CHK(pack_one_b3(c, c->d_initw, c->d_c0, c->d_latent, 64));
    if (planes_class(c, c->d_c0) && c->d_latent % 8 == 0) CHK(pack_planes(c, c->d_initw, c->d_c0, c->d_latent, 7));
    for (auto &b : c->dblocks) {
        if (convt_tile(b.cout, b.stride, &CO_T) >= 0) CHK(pack_one(c, b.w, b.cout, b.cin, 2 * b.stride, CO_T, CI32_T, true));
        if (c->dac_convt_b3) CHK(pack_convt_b3(c, b));
        for (int r = 0; r < 3; r++) {
            if (planes_class(c, b.cout)) {
                CHK(pack_planes(c, b.res[r].in_w, b.cout, b.cout, 7));
                CHK(pack_planes(c, b.res[r].out_w, b.cout, b.cout, 1));
            }
            if (c->dac_fuse) CHK(pack_resunit(c, b.res[r], b.cout));
            if (c->dac_b3 && b.cout % 64 == 0) CHK(pack_one_b3(c, b.res[r].in_w, b.cout, b.cout, 64));
            else if (c->dac_b3 >= 2 && b.cout % 96 == 0) CHK(pack_one_b3(c, b.res[r].in_w, b.cout, b.cout, 96));   // not run on a GPU yet
            if (c->dac_c192 && b.cout == 192) {
                CHK(pack_one(c, b.res[r].in_w, b.cout, b.cout, 7, 192, CI32_K7, false));
                c->packed_c192.insert(b.res[r].in_w);
            } else if (conv_tile(b.cout, 7, &CO_T, &CI_T) >= 0) CHK(pack_one(c, b.res[r].in_w, b.cout, b.cout, 7, CO_T, CI_T, false));
            if (conv1_direct(c, b.cout, b.cout)) {   // [cin][cout] for conv1x1_direct_kernel
                CHK(pack_one(c, b.res[r].out_w, b.cout, b.cout, 1, b.cout, CI32_K1, false));
                c->packed_direct.insert(b.res[r].out_w);
            } else if (conv_tile(b.cout, 1, &CO_T, &CI_T) >= 0) CHK(pack_one(c, b.res[r].out_w, b.cout, b.cout, 1, CO_T, CI_T, false));
        }
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    c->dac_packed = true;
    return 0;
}

// LDS request of a codec kernel.  With `dac_lds_reserve` > 0 (contexts sharing a GPU) the codec leaves that much LDS
// free on every CU so that another context's decoder workgroups (<= 32 KB) can be resident next to the codec's:
// the alpha table is dropped when that is what it takes, and the request is padded so that one workgroup fewer fits
// when the natural size would fill the CU.
static size_t dac_lds_request(const tts_hip_ctx *c, size_t base, size_t table, int *use_table, bool keep_residency = false) {
    const size_t CU = 160 * 1024, reserve = (size_t) c->dac_lds_reserve_kb * 1024;
    *use_table = table ? 1 : 0;
    size_t natural = base + table;
    if (table && !c->dac_alpha_tab) { *use_table = 0; natural = base; }
    if (table && *use_table && (keep_residency || c->dac_alpha_tab == 2) && CU / (base + table) < CU / base) { *use_table = 0; natural = base; }  // the table would cost a resident workgroup
    if (!reserve) return natural;
    auto leaves = [&](size_t req) { const size_t n = CU / req; return CU - n * req; };
    if (leaves(natural) >= reserve) return natural;
    if (table && leaves(base) >= reserve) { *use_table = 0; return base; }
    if (table) { *use_table = 0; natural = base; }
    for (size_t n = CU / natural; n >= 1; n--) {       // pad so that exactly n fit and `reserve` stays free
        const size_t req = std::max(natural, CU / (n + 1) + 16);
        if (n * req + reserve <= CU && CU / req == n) return req;
    }
    return natural;
}

template <int KT, int MI, int NI, int WM, int WN, int CI_T>
static int launch_conv_mfma(tts_hip_ctx *c, const ConvArgs &a_in, int nz) {
    constexpr int CO_T = 32 * MI * WM, T_T = 32 * NI * WN, WCH = KT * CI_T * CO_T;
    const int xw = T_T + (KT - 1) * a_in.dil;
    const int cin_pad = (a_in.cin + CI_T - 1) / CI_T * CI_T;
    ConvArgs a = a_in;
    const size_t lds = dac_lds_request(c, ((size_t) 2 * WCH + 2 * (size_t) ((CI_T * xw + 3) & ~3)) * 4, (a.alpha ? 2 * (size_t) cin_pad : 0) * 4, &a.alpha_tab);
    if (a.dil > 9) return set_err("conv1d_mfma: dilation %d > 9 unsupported", a.dil);
    static std::atomic<uint64_t> attr{0};
    if (attr_needed(attr, c->device)) {
        HIPCHK(hipFuncSetAttribute((const void *) conv1d_mfma_kernel<KT, MI, NI, WM, WN, CI_T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    const dim3 grid((a.L + T_T - 1) / T_T, (a.cout + CO_T - 1) / CO_T, nz);
    hipLaunchKernelGGL((conv1d_mfma_kernel<KT, MI, NI, WM, WN, CI_T>), grid, dim3(64 * WM * WN), lds, c->stream, a);
    HIPCHK(hipGetLastError());
    return 0;
}

// k = 7 conv as six bf16 MFMAs per product (experiment): 64 channels x 256 positions per workgroup of 4 waves,
// or 96 channels x 256 positions per workgroup of 8 waves
template <int MI, int NI, int WM, int WN>
static int launch_conv_b3(tts_hip_ctx *c, const ConvArgs &a_in, int nz) {
    constexpr int CO_T = 32 * MI * WM, T_T = 32 * NI * WN, WPL = 8 * CO_T * 8;
    const int xw = T_T + 6 * a_in.dil;
    const int cin_pad = (a_in.cin + 7) / 8 * 8;
    ConvArgs a = a_in;
    const size_t lds = dac_lds_request(c, ((size_t) 6 * WPL + 6 * (size_t) xw * 8) * 2, (a.alpha ? 2 * (size_t) cin_pad : 0) * 4, &a.alpha_tab, true);
    static std::atomic<uint64_t> attr{0};
    if (attr_needed(attr, c->device)) {
        HIPCHK(hipFuncSetAttribute((const void *) conv1d_mfma_b3_kernel<MI, NI, WM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    const dim3 grid((a.L + T_T - 1) / T_T, (a.cout + CO_T - 1) / CO_T, nz);
    hipLaunchKernelGGL((conv1d_mfma_b3_kernel<MI, NI, WM, WN>), grid, dim3(64 * WM * WN), lds, c->stream, a);
    HIPCHK(hipGetLastError());
    return 0;
}

template <int KT, int MI, int NI, int WM, int WN, int CI_T>
static int launch_conv_mfma16(tts_hip_ctx *c, const ConvArgs &a_in, int nz) {
    constexpr int CO_T = 32 * MI * WM, T_T = 32 * NI * WN, WCH = KT * CI_T * CO_T, XS = CI_T + 8;
    const int xw = T_T + (KT - 1) * a_in.dil;
    const int cin_pad = (a_in.cin + CI_T - 1) / CI_T * CI_T;
    ConvArgs a = a_in;
    const size_t lds = dac_lds_request(c, ((size_t) 2 * WCH + 2 * (size_t) xw * XS) * 2, (a.alpha ? 2 * (size_t) cin_pad : 0) * 4, &a.alpha_tab);
    if (a.dil > 9) return set_err("conv1d_mfma16: dilation %d > 9 unsupported", a.dil);
    static std::atomic<uint64_t> attr{0};
    if (attr_needed(attr, c->device)) {
        HIPCHK(hipFuncSetAttribute((const void *) conv1d_mfma16_kernel<KT, MI, NI, WM, WN, CI_T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    const dim3 grid((a.L + T_T - 1) / T_T, (a.cout + CO_T - 1) / CO_T, nz);
    hipLaunchKernelGGL((conv1d_mfma16_kernel<KT, MI, NI, WM, WN, CI_T>), grid, dim3(64 * WM * WN), lds, c->stream, a);
    HIPCHK(hipGetLastError());
    return 0;
}

template <int S, int MI, int WM, int WN, int CI_T>
static int launch_convt_mfma16(tts_hip_ctx *c, const ConvTArgs &a_in, int nz) {
    constexpr int CO_T = 32 * MI * WM, TI_T = 32 * WN, WCH = CI_T * 2 * S * CO_T, XS = CI_T + 8;
    const int cin_pad = (a_in.cin + CI_T - 1) / CI_T * CI_T;
    ConvTArgs a = a_in;
    const size_t lds = dac_lds_request(c, ((size_t) 2 * WCH + 2 * (size_t) (TI_T + 1) * XS) * 2, (a.alpha ? 2 * (size_t) cin_pad : 0) * 4, &a.alpha_tab, true);
    static std::atomic<uint64_t> attr{0};
    if (attr_needed(attr, c->device)) {
        HIPCHK(hipFuncSetAttribute((const void *) convt1d_mfma16_kernel<S, MI, WM, WN, CI_T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    const dim3 grid((a.L + 1 + TI_T - 1) / TI_T, (a.cout + CO_T - 1) / CO_T, nz);  // ti runs 0..L inclusive
    hipLaunchKernelGGL((convt1d_mfma16_kernel<S, MI, WM, WN, CI_T>), grid, dim3(64 * WM * WN), lds, c->stream, a);
    HIPCHK(hipGetLastError());
    return 0;
}

// Row stride of a codec activation [C][L]: L is frames x 2^k, so rows packed back to back start a multiple of 2 KB (up to 496 KB) apart and the
// C rows a tile touches at the same position fall on few HBM channels.  An odd multiple of 256 B makes consecutive rows walk through
// all of them, whatever power of two the interleave is; the kernels take the stride (a.L / a.Lout) apart from the valid length
// (frames x mult) already.
static int dac_row_stride(const tts_hip_ctx *c, int L) {
    if (!c->dac_pad) return L;
    int LS = (L + 63) / 64 * 64;
    if (((LS / 64) & 1) == 0) LS += 64;
    return LS;
}

struct DacBatch {
    int n = 1;                  // utterances (grid.z)
    const uint32_t *frames = nullptr;  // device [n]
    int mult = 1;               // valid length at this stage = frames[z] * mult
    double tot_frames = 0;      // sum of frames (for flop/byte accounting)
};

static int launch_conv(tts_hip_ctx *c, const DacBatch &bt, const float *x, int cin, int L, size_t w, size_t b, size_t alpha, bool has_alpha,
                       int cout, int K, int pad, int dil, const float *resid, bool do_tanh, float *y, size_t alpha_out = 0,
                       bool has_alpha_out = false, bool has_bias = true) {
    ConvArgs a{};
    a.x = x; a.w = (const float *) (c->arena + w); a.b = has_bias ? (const float *) (c->arena + b) : nullptr;
    a.alpha = has_alpha ? (const float *) (c->arena + alpha) : nullptr;
    a.alpha_out = has_alpha_out ? (const float *) (c->arena + alpha_out) : nullptr;
    a.resid = resid; a.y = y; a.cin = cin; a.cout = cout; a.L = L; a.dil = dil; a.pad = pad; a.do_tanh = do_tanh;
    a.frames = bt.frames; a.mult = bt.mult;
    a.x_f16 = c->dac_f16 ? 1 : 0;
    a.prio = K == 7 ? c->dac_prio : 0;
    const double Lv = bt.tot_frames * bt.mult;  // valid positions over the batch
    const double bytes = ((double) cin * Lv + (double) cout * Lv * (resid ? 2 : 1) + (double) cout * cin * K) * 4;
    CHK(prof_begin(c, cout == 1 ? TTS_HIP_K_DAC_FINAL : (K == 7 ? TTS_HIP_K_DAC_CONV7 : TTS_HIP_K_DAC_CONV1), bytes, 2.0 * cout * (double) cin * K * Lv));
    const bool valu = (c->d.flags & TTS_HIP_FLAG_VALU_GEMM) != 0;
    int CO_T = 0, CI_T = 0;
    const int cfg = valu ? -1 : conv_tile(cout, K, &CO_T, &CI_T);
    auto pk = c->packed.find(w);
    auto pk16 = c->packed16.find(w);
    if (!valu && cout == 1 && K == 7) {
        hipLaunchKernelGGL(conv1d_cout1_kernel, dim3((L + C1_T - 1) / C1_T, 1, bt.n), dim3(256), 0, c->stream, a);
        HIPCHK(hipGetLastError());
    } else if (!valu && K == 7 && c->dac_b3 && !c->dac_f16 && c->packed_b3.count(w) && dil <= 9) {
        a.w = (const float *) c->packed_b3[w];   // three bf16 planes (experiment)
        // tile variants of the 64-channel class for the sweep of round 3 (only 0 has run on a GPU)
        if (cout % 64 == 0 && c->dac_b3_variant == 1) CHK((launch_conv_b3<2, 1, 1, 4>(c, a, bt.n)));        // 64 ch x 128 pos, 4 waves
        else if (cout % 64 == 0 && c->dac_b3_variant == 2) CHK((launch_conv_b3<2, 1, 1, 8>(c, a, bt.n)));   // 64 ch x 256 pos, 8 waves
        else if (cout % 64 == 0 && c->dac_b3_variant == 3) CHK((launch_conv_b3<1, 2, 2, 2>(c, a, bt.n)));   // 64 ch x 128 pos, waves 32 x 64
        else if (cout % 64 == 0) CHK((launch_conv_b3<2, 2, 1, 4>(c, a, bt.n)));                             // 64 ch x 256 pos, 4 waves
        else CHK((launch_conv_b3<3, 1, 1, 8>(c, a, bt.n)));                                                  // 96 ch x 256 pos, 8 waves
    } else if (cfg >= 0 && c->dac_f16 && pk16 != c->packed16.end()) {
        a.w = (const float *) pk16->second;  // fp16 LDS images
        if (K == 7 && cfg == 0) CHK((launch_conv_mfma16<7, 2, 2, 2, 2, CI16_K7>(c, a, bt.n)));
        else if (K == 7 && cfg == 1) CHK((launch_conv_mfma16<7, 3, 2, 1, 4, CI16_K7>(c, a, bt.n)));
        else if (K == 7 && cfg == 2) CHK((launch_conv_mfma16<7, 2, 2, 1, 4, CI16_K7>(c, a, bt.n)));
        else if (K == 1 && cfg == 0) CHK((launch_conv_mfma16<1, 2, 2, 2, 2, CI16_K1>(c, a, bt.n)));
        else if (K == 1 && cfg == 1) CHK((launch_conv_mfma16<1, 3, 2, 1, 4, CI16_K1>(c, a, bt.n)));
        else CHK((launch_conv_mfma16<1, 2, 2, 1, 4, CI16_K1>(c, a, bt.n)));
    } else if (K == 1 && pk != c->packed.end() && c->packed_direct.count(w) && !a.alpha && !a.alpha_out && !do_tanh) {
        a.w = pk->second;
        static std::atomic<uint64_t> attr{0};
        if (attr_needed(attr, c->device)) {
            HIPCHK(hipFuncSetAttribute((const void *) conv1x1_direct_kernel<3, 2, 96, 96>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIPCHK(hipFuncSetAttribute((const void *) conv1x1_direct_kernel<6, 1, 192, 48>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
        if (cin == 96) hipLaunchKernelGGL((conv1x1_direct_kernel<3, 2, 96, 96>), dim3((L + 255) / 256, 1, bt.n), dim3(256), (size_t) 96 * 96 * 4, c->stream, a);
        else hipLaunchKernelGGL((conv1x1_direct_kernel<6, 1, 192, 48>), dim3((L + 127) / 128, 1, bt.n), dim3(256), (size_t) 48 * 192 * 4, c->stream, a);
        HIPCHK(hipGetLastError());
    } else if (cfg >= 0 && pk != c->packed.end()) {
        a.w = pk->second;
        // position-tile variants of the k = 7 kernel (same packed weights: the image depends on CO_T and CI_T only); TTS_HIP_DAC_VARIANT picks
        // per channel-tile class (decimal digits: class 0 / 1 / 2), measured in profiles/r02/dac_variants.log
        const int v0 = c->dac_variant % 10, v1 = (c->dac_variant / 10) % 10, v2 = (c->dac_variant / 100) % 10;
        if (K == 7 && c->packed_c192.count(w) && c->dac_c192 == 1) CHK((launch_conv_mfma<7, 6, 1, 1, 4, CI32_K7>(c, a, bt.n)));        // 192 ch x 128 pos: input staged once
        else if (K == 7 && c->packed_c192.count(w)) CHK((launch_conv_mfma<7, 3, 1, 2, 2, CI32_K7>(c, a, bt.n)));                        // 192 ch x 64 pos, 2 x 2 waves
        else if (K == 7 && cfg == 0 && v0 == 1) CHK((launch_conv_mfma<7, 2, 4, 2, 2, CI32_K7>(c, a, bt.n)));        // 128 ch x 256 pos, wave 64 x 128
        else if (K == 7 && cfg == 0 && v0 == 2) CHK((launch_conv_mfma<7, 4, 2, 1, 4, CI32_K7>(c, a, bt.n)));   // 128 ch x 256 pos, wave 128 x 64
        else if (K == 7 && cfg == 0 && v0 == 3) CHK((launch_conv_mfma<7, 2, 1, 2, 2, CI32_K7>(c, a, bt.n)));   // 128 ch x 64 pos
        else if (K == 7 && cfg == 0 && v0 == 4) CHK((launch_conv_mfma<7, 1, 2, 4, 1, CI32_K7>(c, a, bt.n)));   // 128 ch x 64 pos, wave 32 x 64
        else if (K == 7 && cfg == 0 && v0 == 5) CHK((launch_conv_mfma<7, 2, 2, 2, 4, CI32_K7>(c, a, bt.n)));   // 128 ch x 256 pos, 8 waves
        else if (K == 7 && cfg == 0) CHK((launch_conv_mfma<7, 2, 2, 2, 2, CI32_K7>(c, a, bt.n)));
        else if (K == 7 && cfg == 1 && v1 == 1) CHK((launch_conv_mfma<7, 3, 4, 1, 4, CI32_K7>(c, a, bt.n)));   // 96 ch x 512 pos
        else if (K == 7 && cfg == 1 && v1 == 2) CHK((launch_conv_mfma<7, 3, 1, 1, 4, CI32_K7>(c, a, bt.n)));   // 96 ch x 128 pos
        else if (K == 7 && cfg == 1 && v1 == 3) CHK((launch_conv_mfma<7, 3, 2, 1, 2, CI32_K7>(c, a, bt.n)));   // 96 ch x 128 pos, 2 waves
        else if (K == 7 && cfg == 1) CHK((launch_conv_mfma<7, 3, 2, 1, 4, CI32_K7>(c, a, bt.n)));
        else if (K == 7 && cfg == 2 && v2 == 1) CHK((launch_conv_mfma<7, 2, 4, 1, 4, CI32_K7>(c, a, bt.n)));   // 64 ch x 512 pos
        else if (K == 7 && cfg == 2 && v2 == 2) CHK((launch_conv_mfma<7, 2, 2, 1, 8, CI32_K7>(c, a, bt.n)));   // 64 ch x 512 pos, 8 waves
        else if (K == 7 && cfg == 2 && v2 == 3) CHK((launch_conv_mfma<7, 2, 1, 1, 4, CI32_K7>(c, a, bt.n)));   // 64 ch x 128 pos
        else if (K == 7 && cfg == 2 && v2 == 4) CHK((launch_conv_mfma<7, 2, 2, 1, 2, CI32_K7>(c, a, bt.n)));   // 64 ch x 128 pos, 2 waves
        else if (K == 7 && cfg == 2) CHK((launch_conv_mfma<7, 2, 2, 1, 4, CI32_K7>(c, a, bt.n)));
        else if (K == 1 && cfg == 0) CHK((launch_conv_mfma<1, 2, 2, 2, 2, CI32_K1>(c, a, bt.n)));
        else if (K == 1 && cfg == 1) CHK((launch_conv_mfma<1, 3, 2, 1, 4, CI32_K1>(c, a, bt.n)));
        else CHK((launch_conv_mfma<1, 2, 2, 1, 4, CI32_K1>(c, a, bt.n)));
    } else {
        const dim3 grid((L + CV_T - 1) / CV_T, (cout + CV_CO - 1) / CV_CO, bt.n);
        const size_t lds = ((size_t) CV_CI * (CV_T + (K - 1) * dil) + (size_t) CV_CI * K * CV_CO) * 4;
        if (K == 7) hipLaunchKernelGGL(conv1d_kernel<7>, grid, dim3(256), lds, c->stream, a);
        else if (K == 1) hipLaunchKernelGGL(conv1d_kernel<1>, grid, dim3(256), lds, c->stream, a);
        else return set_err("conv1d: kernel size %d unsupported", K);
        HIPCHK(hipGetLastError());
    }
    return prof_end(c);
}

template <int S, int MI, int WM, int WN, int CI_T>
static int launch_convt_mfma(tts_hip_ctx *c, const ConvTArgs &a_in, int nz) {
    constexpr int CO_T = 32 * MI * WM, TI_T = 32 * WN, WCH = CI_T * 2 * S * CO_T;
    const int cin_pad = (a_in.cin + CI_T - 1) / CI_T * CI_T;
    ConvTArgs a = a_in;
    const size_t lds = dac_lds_request(c, ((size_t) 2 * WCH + 2 * (size_t) ((CI_T * (TI_T + 1) + 3) & ~3)) * 4, (a.alpha ? 2 * (size_t) cin_pad : 0) * 4, &a.alpha_tab, true);   // two resident workgroups matter more to the transposed convs than the table
    static std::atomic<uint64_t> attr{0};
    if (attr_needed(attr, c->device)) {
        HIPCHK(hipFuncSetAttribute((const void *) convt1d_mfma_kernel<S, MI, WM, WN, CI_T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    const dim3 grid((a.L + 1 + TI_T - 1) / TI_T, (a.cout + CO_T - 1) / CO_T, nz);  // ti runs 0..L inclusive
    hipLaunchKernelGGL((convt1d_mfma_kernel<S, MI, WM, WN, CI_T>), grid, dim3(64 * WM * WN), lds, c->stream, a);
    HIPCHK(hipGetLastError());
    return 0;
}

template <int S, int MI>
static int launch_convt_b3(tts_hip_ctx *c, const ConvTArgs &a, int nz) {
    constexpr int CO_T = 32 * MI, WPL = 2 * S * 2 * CO_T * 8, xpl = 2 * 257 * 8;
    const size_t lds = (size_t) 6 * WPL * 2 + (size_t) 6 * xpl * 2 + (size_t) a.cin * 8;
    if (lds > 160 * 1024) return set_err("convt_b3: %d input channels need %zu bytes of LDS", a.cin, lds);
    static std::atomic<uint64_t> attr{0};
    if (attr_needed(attr, c->device)) {
        HIPCHK(hipFuncSetAttribute((const void *) convt_b3_kernel<S, MI>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    ConvTArgs b = a;
    b.npos = (a.L + 1 + 255) / 256; b.nz = nz;   // ti runs 0..L inclusive
    b.nco = xcd_order(a.cout / CO_T, (double) a.cin * a.L * nz * 4, (double) a.cout * a.cin * 2 * S * 6);
    hipLaunchKernelGGL((convt_b3_kernel<S, MI>), dim3(xcd_grid(b.npos, a.cout / CO_T, b.nz)), dim3(512), lds, c->stream, b);
    HIPCHK(hipGetLastError());
    return 0;
}

static int launch_convt(tts_hip_ctx *c, ConvTArgs ta, size_t w_off, int nz) {
    const bool valu = (c->d.flags & TTS_HIP_FLAG_VALU_GEMM) != 0;
    const int s = ta.stride;
    auto pct = c->packed_ct.find(w_off);
    if (!valu && !c->dac_f16 && c->dac_convt_b3 && pct != c->packed_ct.end() && (size_t) 6 * (2 * s * 2 * convt_b3_tile(ta.cout, ta.cin, s) * 8) * 2 + 6 * 2 * 257 * 8 * 2 + (size_t) ta.cin * 8 <= 160 * 1024) {
        ta.w = (const float *) pct->second;
        ta.x_f16 = 0;
        if (s == 8) return launch_convt_b3<8, 1>(c, ta, nz);
        if (s == 4) return launch_convt_b3<4, 2>(c, ta, nz);
        return launch_convt_b3<2, 3>(c, ta, nz);
    }
    int CO_T = 0;
    const int cfg = valu ? -1 : convt_tile(ta.cout, s, &CO_T);
    ta.x_f16 = c->dac_f16 ? 1 : 0;
    auto pk16 = c->packed16.find(w_off);
    if (cfg >= 0 && c->dac_f16 && pk16 != c->packed16.end()) {
        ta.w = (const float *) pk16->second;
        if (cfg == 0) return launch_convt_mfma16<8, 1, 2, 2, CI16_T>(c, ta, nz);
        if (cfg == 1) return launch_convt_mfma16<4, 2, 1, 4, CI16_T>(c, ta, nz);
        if (cfg == 2) return launch_convt_mfma16<2, 3, 1, 4, CI16_T>(c, ta, nz);
        return launch_convt_mfma16<2, 2, 1, 4, CI16_T>(c, ta, nz);
    }
    auto pk = c->packed.find(w_off);
    if (cfg >= 0 && pk != c->packed.end()) {
        ta.w = pk->second;
        if (cfg == 0) return launch_convt_mfma<8, 1, 2, 2, CI32_T>(c, ta, nz);
        if (cfg == 1) return launch_convt_mfma<4, 2, 1, 4, CI32_T>(c, ta, nz);
        if (cfg == 2) return launch_convt_mfma<2, 3, 1, 4, CI32_T>(c, ta, nz);
        return launch_convt_mfma<2, 2, 1, 4, CI32_T>(c, ta, nz);
    }
    const dim3 grid((ta.Lout + CV_T - 1) / CV_T, (ta.cout + CV_CO - 1) / CV_CO, nz);
    const size_t lds = ((size_t) CT_CI * ((CV_T + s - 1) / s + 2) + (size_t) CT_CI * 2 * s * CV_CO) * 4;
    hipLaunchKernelGGL(convt1d_kernel, grid, dim3(256), lds, c->stream, ta);
    HIPCHK(hipGetLastError());
    return 0;
}

// ---- wide classes on split planes --------------------------------------------------------------------------------------------------
static int launch_split(tts_hip_ctx *c, const DacBatch &bt, const float *x, int C, int LS, size_t alpha, bool has_alpha, __bf16 *yp) {
    SplitArgs a{};
    a.x = x; a.alpha = has_alpha ? (const float *) (c->arena + alpha) : nullptr; a.yp = yp; a.C = C; a.L = LS; a.frames = bt.frames; a.mult = bt.mult;
    const double Lv = bt.tot_frames * bt.mult;
    CHK(prof_begin(c, TTS_HIP_K_DAC_CONV1, (double) C * Lv * 10, 0));
    hipLaunchKernelGGL(snake_split_kernel, dim3((LS + 255) / 256, C / 8, bt.n), dim3(256), 0, c->stream, a);
    HIPCHK(hipGetLastError());
    return prof_end(c);
}
template <int KT, int MI, int NI, int WM, int WN, int NS, int MINW, int NB = 2>
static int launch_conv_b3p_t(tts_hip_ctx *c, const PConvArgs &a, int nz) {
    constexpr int CO_T = 32 * MI * WM, T_T = 32 * NI * WN, WPL = NS * 2 * CO_T * 8, NCG = KT == 7 ? (NS == 7 ? 2 : 1) : 2 * NS;
    const int xw = T_T + (KT - 1) * a.dil;
    const size_t lds = (size_t) NB * (3 * WPL * 2 + (size_t) 3 * NCG * xw * 8 * 2);
    static std::atomic<uint64_t> attr{0};
    if (attr_needed(attr, c->device)) {
        HIPCHK(hipFuncSetAttribute((const void *) conv_b3p_kernel<KT, MI, NI, WM, WN, NS, MINW, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    PConvArgs b = a;
    b.npos = (a.L + T_T - 1) / T_T; b.nz = nz;
    b.nco = xcd_order(a.cout / CO_T, (double) a.cin * a.L * nz * 6, (double) a.cout * a.cin * KT * 6);
    hipLaunchKernelGGL((conv_b3p_kernel<KT, MI, NI, WM, WN, NS, MINW, NB>), dim3(xcd_grid(b.npos, a.cout / CO_T, b.nz)), dim3(64 * WM * WN), lds, c->stream, b);
    HIPCHK(hipGetLastError());
    return 0;
}
// conv on planes: y (fp32, may be NULL) and / or yp (planes with the consumer's snake, may be NULL)
static int launch_conv_planes(tts_hip_ctx *c, const DacBatch &bt, const __bf16 *xp, int cin, int LS, size_t w, size_t b, int cout, int K, int dil, const float *resid,
                              float *y, __bf16 *yp, size_t alpha_out, bool has_alpha_out) {
    PConvArgs a{};
    a.xp = xp; a.w = c->packed_p.at(w); a.b = (const float *) (c->arena + b); a.resid = resid; a.y = y; a.yp = yp;
    a.alpha_out = has_alpha_out ? (const float *) (c->arena + alpha_out) : nullptr;
    a.cin = cin; a.cout = cout; a.L = LS; a.dil = dil; a.pad = K == 7 ? 3 * dil : 0; a.frames = bt.frames; a.mult = bt.mult;
    const double Lv = bt.tot_frames * bt.mult;
    const double bytes = ((double) cin * Lv * 6 + (double) cout * Lv * ((resid ? 4 : 0) + (y ? 4 : 0) + (yp ? 6 : 0)) + (double) cout * cin * K * 6);
    CHK(prof_begin(c, K == 7 ? TTS_HIP_K_DAC_CONV7 : TTS_HIP_K_DAC_CONV1, bytes, 2.0 * cout * (double) cin * K * Lv));
    if (K == 7) {
        const bool tapk = c->dac_tap7 && cin % 16 == 0;
        // TTS_HIP_DAC_P_VARIANT: 0 / 2 = 4 / 8 waves; with one tap per k-step: one LDS buffer (73 KB, two workgroups per CU), +10 = two buffers.
        // 64-utterance pass, k = 7 family: 45.7 (0) / 49.9 (2: 128 registers, spills) / 53.5 (10) / 48.6 (12) ms; tap pairs 50.3 ms
        const int pv = c->dac_p_variant;
        if (tapk && pv == 0) CHK((launch_conv_b3p_t<7, 2, 2, 1, 4, 7, 2, 1>(c, a, bt.n)));
        else if (tapk && pv == 10) CHK((launch_conv_b3p_t<7, 2, 2, 1, 4, 7, 2, 2>(c, a, bt.n)));
        else if (tapk && pv == 12) CHK((launch_conv_b3p_t<7, 2, 1, 1, 8, 7, 2, 2>(c, a, bt.n)));
        else if (tapk) CHK((launch_conv_b3p_t<7, 2, 1, 1, 8, 7, 4, 1>(c, a, bt.n)));
        else if (pv == 0) CHK((launch_conv_b3p_t<7, 2, 2, 1, 4, 4, 2>(c, a, bt.n)));      // 64 ch x 256 pos, 4 waves
        else CHK((launch_conv_b3p_t<7, 2, 1, 1, 8, 4, 4>(c, a, bt.n)));                             // 64 ch x 256 pos, 8 waves, two workgroups per CU
    } else if (c->dac_k1_variant == 1 && cout % 256 == 0) {
        CHK((launch_conv_b3p_t<1, 4, 2, 2, 4, 1, 2>(c, a, bt.n)));                                  // 256 ch x 256 pos, 8 waves
    } else if (c->dac_k1_variant == 1) {
        CHK((launch_conv_b3p_t<1, 2, 4, 2, 4, 1, 2>(c, a, bt.n)));                                  // 128 ch x 512 pos, 8 waves
    } else {
        CHK((launch_conv_b3p_t<1, 2, 2, 2, 4, 1, 2>(c, a, bt.n)));                                  // 128 ch x 256 pos, 8 waves
    }
    return prof_end(c);
}

// one residual unit (gnac.cpp:133-149) as one launch
template <int MI, int KS, int KS2>
static int launch_resunit_t(tts_hip_ctx *c, const ResUnitArgs &a, int nz) {
    constexpr int C = 32 * MI;
    const ResUnitGeom g = resunit_geom(C, KS, KS2);
    const int xw = 256 + 6 * a.dil;
    const size_t lds = (size_t) 2 * g.WST * 2 + (size_t) 6 * xw * 8 * 2 + (size_t) C * 24;
    static std::atomic<uint64_t> attr{0};
    if (attr_needed(attr, c->device)) {
        HIPCHK(hipFuncSetAttribute((const void *) resunit_b3_kernel<MI, KS, KS2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    hipLaunchKernelGGL((resunit_b3_kernel<MI, KS, KS2>), dim3((a.L + 255) / 256, 1, nz), dim3(512), lds, c->stream, a);
    HIPCHK(hipGetLastError());
    return 0;
}
template <int MI, int KS2>
static int launch_resunit_t7(tts_hip_ctx *c, const ResUnitArgs &a, int nz) {
    constexpr int C = 32 * MI;
    const int xw = 256 + 6 * a.dil;
    const size_t WST = (size_t) 3 * ResT7<MI>::MAXCNT * 2 * C * 8;
    const size_t lds = 2 * WST * 2 + (size_t) 6 * 2 * xw * 8 * 2 + (size_t) C * 24;
    static std::atomic<uint64_t> attr{0};
    if (attr_needed(attr, c->device)) {
        HIPCHK(hipFuncSetAttribute((const void *) resunit_t7_kernel<MI, KS2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    hipLaunchKernelGGL((resunit_t7_kernel<MI, KS2>), dim3((a.L + 255) / 256, 1, nz), dim3(512), lds, c->stream, a);
    HIPCHK(hipGetLastError());
    return 0;
}
static bool resunit_fused(const tts_hip_ctx *c, const DRes &r, int dil) {
    return c->dac_fuse && !c->dac_f16 && !(c->d.flags & TTS_HIP_FLAG_VALU_GEMM) && dil <= 9 && c->packed_ru.count(r.in_w);
}
static int launch_resunit(tts_hip_ctx *c, const DacBatch &bt, const DRes &r, int C, int LS, int dil, const float *x, float *y) {
    ResUnitArgs a{};
    a.x = x; a.y = y; a.w = c->packed_ru.at(r.in_w);
    a.b7 = (const float *) (c->arena + r.in_b); a.b1 = (const float *) (c->arena + r.out_b);
    a.alpha_in = (const float *) (c->arena + r.in_alpha); a.alpha_mid = (const float *) (c->arena + r.out_alpha);
    a.L = LS; a.dil = dil; a.pad = 3 * dil; a.frames = bt.frames; a.mult = bt.mult;
    const double Lv = bt.tot_frames * bt.mult;
    CHK(prof_begin(c, TTS_HIP_K_DAC_RESUNIT, (2.0 * C * Lv + 8.0 * C * C) * 4, 2.0 * C * (double) C * 8 * Lv));
    if (c->dac_tap7) {
        if (C == 96) CHK((launch_resunit_t7<3, 3>(c, a, bt.n)));
        else CHK((launch_resunit_t7<6, 4>(c, a, bt.n)));
    } else if (C == 96) CHK((launch_resunit_t<3, 4, 3>(c, a, bt.n)));
    else CHK((launch_resunit_t<6, 2, 4>(c, a, bt.n)));
    return prof_end(c);
}

// dac_runner::run for n utterances at once (grid.z = utterance, per-utterance lengths): the early blocks have
// few positions per utterance, so batching is what fills the 256 CUs there.
static int dac_decode_batch_on(tts_hip_ctx *c, const uint32_t *codes, const uint32_t *frames, uint32_t n, float *pcm_out);
// One codec pass carries at most `dac_group` utterances (TTS_HIP_DAC_GROUP, default 64): the activation buffers are sized for a group
// (3 x 197 KB per frame: 9.4 GB for 64 x 248 frames instead of 56 GB for a 384-utterance batch), and passes of different contexts on
// one device take turns (a per-device mutex): a pass fills the chip with compute-bound convolutions, two of them interleaved only
// stretch each other, while another context's latency-bound decoder loop does fit next to one.
static int dac_decode_batch(tts_hip_ctx *c, const uint32_t *codes, const uint32_t *frames, uint32_t n, float *pcm_out) {
    if (!c || !c->finalized || !c->has_dac) return set_err("tts_hip_dac_decode: context has no finalized DAC");
    if (!frames || !codes || !pcm_out) return set_err("tts_hip_dac_decode: null argument");
    const uint32_t G = (uint32_t) std::max(1, c->dac_group);
    size_t code_off = 0, pcm_off = 0;
    for (uint32_t g0 = 0; g0 < n; g0 += G) {
        const uint32_t m = std::min(G, n - g0);
        size_t fr = 0;
        for (uint32_t i = 0; i < m; i++) fr += frames[g0 + i];
        int rc;
        {
            std::lock_guard<std::mutex> lock(g_dac_pass_mutex[(unsigned) c->device % 64]);
            if (!c->dac_stream) {
                rc = dac_decode_batch_on(c, codes + code_off * c->d_ncb, frames + g0, m, pcm_out + pcm_off);
            } else {
                // the decoder stream is idle here (every decoder entry point synchronises before it returns)
                hipStream_t ar = c->stream;
                c->stream = c->dac_stream;
                rc = dac_decode_batch_on(c, codes + code_off * c->d_ncb, frames + g0, m, pcm_out + pcm_off);
                c->stream = ar;
            }
        }
        if (rc) return rc;
        code_off += fr;
        pcm_off += fr * (size_t) c->d_up;
    }
    return 0;
}

static int dac_decode_batch_on(tts_hip_ctx *c, const uint32_t *codes, const uint32_t *frames, uint32_t n, float *pcm_out) {
    if (!c->weights_present) return set_err("tts_hip_dac_decode: weights not present");
    if (!codes || !pcm_out || !frames) return set_err("tts_hip_dac_decode: null argument");
    HIPCHK(hipSetDevice(c->device));
    uint32_t Fmax = 0;
    size_t tot = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (frames[i] > c->d.dac_max_frames) return set_err("tts_hip_dac_decode: %u frames > max %u", frames[i], c->d.dac_max_frames);
        Fmax = std::max(Fmax, frames[i]);
        tot += frames[i];
    }
    if (Fmax == 0) return 0;  // empty response (cli.cpp:87-90 treats n_outputs==0 as the soft failure)
    for (size_t i = 0; i < tot * c->d_ncb; i++)
        if (codes[i] >= (uint32_t) c->d_cbsize) return set_err("tts_hip_dac_decode: code %u >= codebook size %d", codes[i], c->d_cbsize);
    c->dac_dbg.clear();
    CHK(ensure_packed(c));
    // buffers: n utterances, rows at the padded stride of the stage
    const size_t need_frames = (size_t) n * Fmax;
    size_t need_elems = 0, pcm_elems = 0;
    {
        int Lq = (int) Fmax;
        need_elems = (size_t) std::max(c->d_latent, c->d_c0) * dac_row_stride(c, Lq);
        // split planes (6 bytes per element) of the wide classes live in these buffers too
        if (c->packed_p.count(c->d_initw)) need_elems = std::max(need_elems, ((size_t) c->d_latent * dac_row_stride(c, Lq) * 3 + 1) / 2);
        for (auto &b : c->dblocks) {
            need_elems = std::max(need_elems, (size_t) b.cin * dac_row_stride(c, Lq));
            Lq = (Lq - 1) * b.stride - 2 * b.padding + 2 * b.stride;
            need_elems = std::max(need_elems, (size_t) b.cout * dac_row_stride(c, Lq));
            if (c->packed_p.count(b.res[0].in_w)) need_elems = std::max(need_elems, ((size_t) b.cout * dac_row_stride(c, Lq) * 3 + 1) / 2);
        }
        need_elems *= n;
        pcm_elems = (size_t) n * dac_row_stride(c, Lq);
    }
    // the device's codec buffers (the caller holds the device's pass lock)
    DacBuffers &B = g_dac_buffers[(unsigned) c->device % 64];
    if (!c->dac_buf_user) { c->dac_buf_user = true; B.users++; }
    if (need_frames * c->d_ncb > B.cap_codes || need_elems > B.dbuf_elems || pcm_elems > B.h_pcm_elems || (!c->packed_p.empty() && !B.dplanes)) {
        HIPCHK(hipStreamSynchronize(c->stream));
        for (int i = 0; i < 3; i++) { free_dev(B.dbuf[i]); B.dbuf[i] = nullptr; }
        free_dev(B.dplanes); B.dplanes = nullptr;
        free_dev(B.d_codes); B.d_codes = nullptr;
        if (B.h_pcm) { (void) hipHostFree(B.h_pcm); B.h_pcm = nullptr; }
        B.dbuf_elems = std::max(need_elems, B.dbuf_elems);
        const size_t cap_codes = std::max(need_frames * c->d_ncb, B.cap_codes);
        for (int i = 0; i < 3; i++) HIPCHK(hipMalloc((void **) &B.dbuf[i], B.dbuf_elems * 4));
        if (!c->packed_p.empty()) HIPCHK(hipMalloc((void **) &B.dplanes, B.dbuf_elems * 4));   // 6 bytes per element of the widest planes class <= a dbuf
        HIPCHK(hipMalloc((void **) &B.d_codes, cap_codes * 4));
        B.h_pcm_elems = std::max(pcm_elems, B.h_pcm_elems);
        HIPCHK(hipHostMalloc((void **) &B.h_pcm, B.h_pcm_elems * 4));
        B.cap_codes = cap_codes;
    }
    if (n > c->d_frames_cap) {
        free_dev(c->d_frames);
        HIPCHK(hipMalloc((void **) &c->d_frames, (size_t) n * 4));
        c->d_frames_cap = n;
    }
    HIPCHK(hipMemcpyAsync(c->d_frames, frames, (size_t) n * 4, hipMemcpyHostToDevice, c->stream));
    {   // codes padded to [n][Fmax][n_cb]
        size_t off = 0;
        for (uint32_t i = 0; i < n; i++) {
            if (frames[i]) HIPCHK(hipMemcpyAsync(B.d_codes + (size_t) i * Fmax * c->d_ncb, codes + off * c->d_ncb, (size_t) frames[i] * c->d_ncb * 4,
                                                 hipMemcpyHostToDevice, c->stream));
            off += frames[i];
        }
    }
    DacBatch bt;
    bt.n = (int) n; bt.frames = c->d_frames; bt.mult = 1; bt.tot_frames = (double) tot;
    int L = (int) Fmax;                      // longest utterance at this stage
    int LS = dac_row_stride(c, L);           // row stride of this stage's activations
    float *cur = B.dbuf[0], *t1 = B.dbuf[1], *t2 = B.dbuf[2];

    DacEmbedArgs ea{};
    ea.frames = c->d_frames;
    ea.codes = B.d_codes; ea.codebook = (const float *) (c->arena + c->d_codebook); ea.proj_w = (const float *) (c->arena + c->d_projw);
    ea.proj_b = (const float *) (c->arena + c->d_projb); ea.n_cb = c->d_ncb; ea.cb_size = c->d_cbsize; ea.cb_dim = c->d_cbdim;
    ea.latent = c->d_latent; ea.T = L; ea.Tout = LS; ea.out = cur; ea.x_f16 = c->dac_f16 ? 1 : 0;
    CHK(prof_begin(c, TTS_HIP_K_DAC_EMBED, (double) c->d_latent * tot * 4, 2.0 * c->d_latent * tot * c->d_ncb * c->d_cbdim));
    if (c->d_ncb == 9 && c->d_cbdim == 8 && !getenv("TTS_HIP_DAC_EMBED_SIMPLE"))
        hipLaunchKernelGGL((dac_embed_tile_kernel<9, 8>), dim3((L + 63) / 64, (c->d_latent + 4 * EMB_CH - 1) / (4 * EMB_CH), n), dim3(256), 0, c->stream, ea);
    else
        hipLaunchKernelGGL(dac_embed_kernel, dim3((L + 63) / 64, c->d_latent, n), dim3(64), 0, c->stream, ea);
    HIPCHK(hipGetLastError());
    CHK(prof_end(c));
    if (n == 1) CHK(dac_snapshot(c, 0, cur, (size_t) c->d_latent, (size_t) L, (size_t) LS));

    if (c->packed_p.count(c->d_initw) && !(c->d.flags & TTS_HIP_FLAG_VALU_GEMM)) {
        // quantizer output -> split planes (no snake in front of the first conv) -> k = 7 conv on planes -> fp32 for the first transposed conv
        CHK(launch_split(c, bt, cur, c->d_latent, LS, 0, false, (__bf16 *) t2));
        CHK(launch_conv_planes(c, bt, (const __bf16 *) t2, c->d_latent, LS, c->d_initw, c->d_initb, c->d_c0, 7, 1, nullptr, t1, nullptr, 0, false));
    } else
    CHK(launch_conv(c, bt, cur, c->d_latent, LS, c->d_initw, c->d_initb, 0, false, c->d_c0, 7, 3, 1, nullptr, false, t1));
    std::swap(cur, t1);
    if (n == 1) CHK(dac_snapshot(c, 1, cur, (size_t) c->d_c0, (size_t) L, (size_t) LS));

    int C = c->d_c0;
    for (size_t bi = 0; bi < c->dblocks.size(); bi++) {
        const DBlock &b = c->dblocks[bi];
        const int Lout = (L - 1) * b.stride - 2 * b.padding + 2 * b.stride, LSout = dac_row_stride(c, Lout);
        ConvTArgs ta{};
        ta.x = cur; ta.w = (const float *) (c->arena + b.w); ta.b = (const float *) (c->arena + b.b);
        ta.alpha = (const float *) (c->arena + b.alpha); ta.y = t1; ta.cin = b.cin; ta.cout = b.cout; ta.L = LS;
        ta.Lout = LSout; ta.stride = b.stride; ta.pad = b.padding;
        ta.frames = c->d_frames; ta.mult = bt.mult;
        const double Lov = bt.tot_frames * bt.mult * b.stride;
        CHK(prof_begin(c, TTS_HIP_K_DAC_CONVT, ((double) b.cin * bt.tot_frames * bt.mult + (double) b.cout * Lov + (double) b.cin * b.cout * 2 * b.stride) * 4,
                       2.0 * b.cin * (double) b.cout * 2 * Lov));
        CHK(launch_convt(c, ta, b.w, (int) n));
        CHK(prof_end(c));
        std::swap(cur, t1);
        L = Lout; LS = LSout; C = b.cout;
        bt.mult *= b.stride;
        if (c->packed_p.count(b.res[0].in_w) && !(c->d.flags & TTS_HIP_FLAG_VALU_GEMM)) {
            // a wide class on split planes: the activation a conv consumes is written by its producer already snaked (the consumer's alpha)
            // and split; the fp32 tensor exists only where the residual add and the next transposed conv need it.
            //   X (cur, fp32) --split(snake in_alpha 0)--> PA ;  k7(PA) -> PB = split(snake out_alpha) ;  k1(PB) + X -> X' (fp32) [+ PA for the next unit]
            __bf16 *PA = (__bf16 *) t2, *PB = (__bf16 *) B.dplanes;
            CHK(launch_split(c, bt, cur, C, LS, b.res[0].in_alpha, true, PA));
            for (int r = 0; r < 3; r++) {
                int dil = 1;
                for (int e = 0; e < r; e++) dil *= 3;
                CHK(launch_conv_planes(c, bt, PA, C, LS, b.res[r].in_w, b.res[r].in_b, C, 7, dil, nullptr, nullptr, PB, b.res[r].out_alpha, true));
                CHK(launch_conv_planes(c, bt, PB, C, LS, b.res[r].out_w, b.res[r].out_b, C, 1, 1, cur, t1, r < 2 ? PA : nullptr,
                                       r < 2 ? b.res[r + 1].in_alpha : 0, r < 2));
                std::swap(cur, t1);
            }
            if (n == 1) CHK(dac_snapshot(c, 2 + (int) bi, cur, (size_t) C, (size_t) L, (size_t) LS));
            continue;
        }
        for (int r = 0; r < 3; r++) {  // build_residual_unit: dilation 3^r, padding 3^(r+1) (gnac.h:44-48)
            int dil = 1;
            for (int e = 0; e < r; e++) dil *= 3;
            // snake(out_alpha) of the k=1 conv's input is applied in the k=7 conv's epilogue (same arithmetic, once
            // per element instead of once per output-channel tile)
            if (resunit_fused(c, b.res[r], dil)) {
                CHK(launch_resunit(c, bt, b.res[r], C, LS, dil, cur, t1));
                std::swap(cur, t1);
                continue;
            }
            CHK(launch_conv(c, bt, cur, C, LS, b.res[r].in_w, b.res[r].in_b, b.res[r].in_alpha, true, C, 7, 3 * dil, dil, nullptr, false, t1,
                            b.res[r].out_alpha, true));
            CHK(launch_conv(c, bt, t1, C, LS, b.res[r].out_w, b.res[r].out_b, 0, false, C, 1, 0, 1, cur, false, t2));
            std::swap(cur, t2);
        }
        if (n == 1) CHK(dac_snapshot(c, 2 + (int) bi, cur, (size_t) C, (size_t) L, (size_t) LS));
    }
    CHK(launch_conv(c, bt, cur, C, LS, c->d_fw, c->d_fb, c->d_falpha, true, 1, 7, 3, 1, nullptr, true, t1));
    HIPCHK(hipMemcpyAsync(B.h_pcm, t1, (size_t) n * LS * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    size_t off = 0;
    for (uint32_t i = 0; i < n; i++) {
        memcpy(pcm_out + off, B.h_pcm + (size_t) i * LS, (size_t) frames[i] * c->d_up * 4);
        off += (size_t) frames[i] * c->d_up;
    }
    return 0;
}

extern "C" int tts_hip_dac_decode(tts_hip_ctx *c, const uint32_t *codes, uint32_t frames, float *pcm_out) {
    return dac_decode_batch(c, codes, &frames, 1, pcm_out);
}

extern "C" int tts_hip_dac_decode_batch(tts_hip_ctx *c, const uint32_t *codes, const uint32_t *frames, uint32_t n, float *pcm_out) {
    if (n == 0) return 0;
    return dac_decode_batch(c, codes, frames, n, pcm_out);
}

// ------------------------------------------------------------------------------------------------
// Orpheus decoder (src/models/orpheus/model.cpp:186-325)
// ------------------------------------------------------------------------------------------------
extern "C" tts_hip_ctx *tts_hip_orpheus_create(int device, const tts_hip_orpheus_desc *ld) {
    if (!ld || ld->struct_size != sizeof(tts_hip_orpheus_desc)) { set_err("tts_hip_orpheus_create: bad desc (struct_size mismatch)"); return nullptr; }
    tts_hip_desc d{};
    d.struct_size = sizeof(d);
    d.hidden_size = ld->hidden_size; d.n_layers = ld->n_layers; d.n_attn_heads = ld->n_attn_heads; d.max_ctx_length = ld->n_ctx;
    d.max_seqs = 1;
    d.flags = (ld->flags & (TTS_HIP_FLAG_VALU_GEMM | TTS_HIP_FLAG_DEQUANT_Q)) | TTS_HIP_FLAG_NO_PARLER | TTS_HIP_FLAG_NO_DAC;
    tts_hip_ctx *c = tts_hip_create(device, &d);
    if (!c) return nullptr;
    c->has_llama = true;
    c->lm = *ld;
    // measured on MI355X at the orpheus-3b Q4_0 shapes (profiles/r02/first_call_orpheus_*.log): 3.84 ms/step through the
    // lock-step workgroups, 2.98 with the streaming 1-4 row kernels, 2.84 reading the Q4_0 codes themselves, 2.80 with the
    // step captured in one hipGraph -> all three are the default here; TTS_HIP_GEMV_ROWS / _Q4_NATIVE / _LLAMA_GRAPH=0 turn them off
    if (!getenv("TTS_HIP_GEMV_ROWS")) c->gemv_rows = true;
    if (!getenv("TTS_HIP_Q4_NATIVE")) c->q4_native = true;
    if (!getenv("TTS_HIP_LLAMA_GRAPH")) c->llama_graph = true;
    if (c->lm.rope_base == 0.0f) c->lm.rope_base = 500000.0f;
    return c;
}

// attention of the Llama / Dia steps: one workgroup per (head, row), or — few rows, many keys — the keys split over `nz` workgroups
// plus a combine launch (attn_gqa_split_kernel).  max_keys bounds the LDS score buffer.
static int launch_attn_gqa(tts_hip_ctx *c, int NHq, int rows, int max_keys, const float *qkv, int ld, const uint32_t *pos, const float *kc, const float *vc, int NKV,
                           float scale, float *out, const uint32_t *kbeg, const uint32_t *kend, const uint32_t *row_seq, int64_t seq_stride, bool fixed_split, bool q_out = false,
                           QPre qp = QPre{}) {
    int nz = 1;
    if (c->attn_split_max > 1 && NHq * rows <= 256) {
        // a graph captured once replays for every position: the split count must not depend on the position then
        nz = fixed_split ? c->attn_split_max : std::min(c->attn_split_max, std::max(1, max_keys / 128));
        while (nz > 1 && (size_t) rows * NHq * nz > c->attn_part_cap) nz--;
    }
    if (nz <= 1) {
        hipLaunchKernelGGL(attn_gqa_kernel<128>, dim3(NHq, rows), dim3(256), (size_t) (128 + max_keys) * 4, c->stream, qkv, ld, pos, kc, vc, NHq, NKV, scale, out, kbeg, kend,
                           row_seq, seq_stride, qp);
        HIPCHK(hipGetLastError());
        return 0;
    }
    const int chunk = (max_keys + nz - 1) / nz;
    hipLaunchKernelGGL(attn_gqa_split_kernel<128>, dim3(NHq, rows, nz), dim3(256), (size_t) (128 + chunk + 1) * 4, c->stream, qkv, ld, pos, kc, vc, NHq, NKV, scale, c->attn_part,
                       kbeg, kend, row_seq, seq_stride, qp);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(attn_gqa_combine_kernel, dim3(NHq, rows), dim3(128), 0, c->stream, (const float *) c->attn_part, nz, NHq, out, q_out ? c->aq : (int8_t *) nullptr, q_out ? c->ad : (float *) nullptr);
    if (q_out) c->aq_src = out;
    HIPCHK(hipGetLastError());
    return 0;
}

static int llama_gemm(tts_hip_ctx *c, const W &w, const float *A, int lda, float *out, int ldo, int n, int epi, int ksplit = 1) {
    for (int r0 = 0; r0 < n; r0 += c->RMAX) {
        GemmArgs g{};
        g.R = std::min(c->RMAX, n - r0); g.H = c->H;
        g.A = A + (size_t) r0 * lda; g.lda = lda;
        g.out = out + (size_t) r0 * ldo; g.ldo = ldo;
        if (ksplit > 1) {  // slabs [ksplit][RMAX][ldo], folded into the residual stream by the next rms_fold_rows_kernel
            g.kchunk = (int) w.K / ksplit;
            g.slab_stride = (int64_t) c->RMAX * ldo;
        }
        CHK(run_gemm(c, TTS_HIP_K_GEMM_OTHER, w, g, PRO_F32, epi));
    }
    return 0;
}

// one call of orpheus_runner::decode: n rows (<= RMAX) at pos0..; leaves the final-normed last row's logits in l_logits
// ids == nullptr: one row whose token id and position are already in l_ids[0] / l_pos[0] (the device-resident greedy loop)
// attn_positions != 0: size the attention scratch for that many cached positions instead of pos0 + n (a captured step is replayed
// at every position)
static int llama_forward(tts_hip_ctx *c, const uint32_t *ids, int n, uint32_t pos0, int attn_positions = 0) {
    const int H = c->H, F = c->F, NH = c->NH, NKV = (int) c->lm.n_kv_heads, HD = (int) c->lm.head_dim;
    const int QKV = (NH + 2 * NKV) * HD, NCTX = (int) c->lm.n_ctx;
    if (n < 1 || n > c->RMAX) return set_err("tts_hip_orpheus_decode: %d tokens per call outside 1..%d", n, c->RMAX);
    if (pos0 + (uint32_t) n > (uint32_t) NCTX) return set_err("tts_hip_orpheus_decode: positions up to %u exceed the %d cached positions", pos0 + n, NCTX);
    auto f32 = [&](size_t off) { return (const float *) (c->arena + off); };
    if (ids) {
        std::vector<uint32_t> hp((size_t) n);
        for (int i = 0; i < n; i++) {
            if (ids[i] >= (uint32_t) c->l_V) return set_err("tts_hip_orpheus_decode: token id %u >= vocabulary %d", ids[i], c->l_V);
            hp[(size_t) i] = pos0 + (uint32_t) i;
        }
        HIPCHK(hipMemcpyAsync(c->l_ids, ids, (size_t) n * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->l_pos, hp.data(), (size_t) n * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));  // hp is a local
    } else if (n != 1) {
        return set_err("llama_forward: device-resident inputs carry one row");
    }
    hipLaunchKernelGGL(t5_embed_kernel, dim3(n), dim3(256), 0, c->stream, f32(c->l_embd), (const uint32_t *) c->l_ids, H, c->l_x);
    HIPCHK(hipGetLastError());
    const float theta_scale = powf(c->lm.rope_base, -2.0f / (float) HD);
    c->l_pending = 0;
    // the streaming integer GEMV of 1..4 rows takes its Q8_0 activation blocks from the producing kernel where there is one
    auto q_for = [&](const W &w, int rows) {
        return c->gemv_rows && rows <= 4 && w.type == TTS_HIP_Q8I && !(c->d.flags & (TTS_HIP_FLAG_VALU_GEMM | TTS_HIP_FLAG_DEQUANT_Q)) && w.K % 32 == 0;
    };
    auto rms = [&](size_t w_off, int rows, float *x, float *y, const W *next) {
        const bool q = next && q_for(*next, rows);
        hipLaunchKernelGGL(rms_fold_rows_kernel, dim3(rows), dim3(256), 0, c->stream, x, H, f32(w_off), y, rows, 1e-5f,
                           c->l_pending ? (const float *) c->l_parts : (const float *) nullptr, c->l_pending, (int64_t) c->RMAX * H,
                           q ? c->aq : (int8_t *) nullptr, q ? c->ad : (float *) nullptr);
        c->l_pending = 0;
        c->aq_src = q ? y : nullptr;
        return hipGetLastError() == hipSuccess ? 0 : set_err("rms_fold_rows_kernel launch failed");
    };
    for (int l = 0; l < c->L; l++) {
        const auto &y = c->l_layers[l];
        float *kc = c->l_kc + (size_t) l * NCTX * c->l_kvH, *vc = c->l_vc + (size_t) l * NCTX * c->l_kvH;
        const size_t qkv_lds = (size_t) n * H + (size_t) n * (H / 32) * 4;
        const bool qkv_fused = c->q4_rope && c->q4_lds && y.qkv.q4 && q_for(y.qkv, n) && HD == 128 && H % 512 == 0 && qkv_lds <= 64 * 1024 && !c->prof;
        // the rms norm inside the consuming projection's staging (stage_rms_q8): no slabs may be pending, the row is held in registers
        const bool rms_fused = c->q4_rms && !c->l_pending && H <= 4096;
        if (qkv_fused && rms_fused) {
            QGemmArgs qa{};
            qa.g.W = c->arena + y.qkv.off; qa.g.K = H; qa.g.N = QKV; qa.g.R = n; qa.g.out = c->l_qkv; qa.g.ldo = QKV;
            qa.wd = (const _Float16 *) (c->arena + y.qkv.soff);
            RopeEpi re{(const uint32_t *) c->l_pos, f32(c->l_ropef), theta_scale, NH, NKV, kc, vc};
            RmsSrc rs{c->l_x, f32(y.in_norm), 1e-5f};
            hipLaunchKernelGGL((gemv_q4_qkv_rope_kernel<4, 2>), dim3((QKV / 2 + 3) / 4), dim3(256), qkv_lds + 16, c->stream, qa, y.qkv.q4, re, rs);
            HIPCHK(hipGetLastError());
            c->aq_src = nullptr;
        } else if (qkv_fused) {
            CHK(rms(y.in_norm, n, c->l_x, c->l_xn, &y.qkv));
            // the projection, the rope of q and k and the cache append in one launch (gemv_q4_qkv_rope_kernel)
            QGemmArgs qa{};
            qa.g.W = c->arena + y.qkv.off; qa.g.K = H; qa.g.N = QKV; qa.g.R = n; qa.g.out = c->l_qkv; qa.g.ldo = QKV;
            qa.wd = (const _Float16 *) (c->arena + y.qkv.soff); qa.aq = c->aq; qa.ad = c->ad;
            RopeEpi re{(const uint32_t *) c->l_pos, f32(c->l_ropef), theta_scale, NH, NKV, kc, vc};
            hipLaunchKernelGGL(gemv_q4_qkv_rope_kernel<4>, dim3((QKV / 2 + 3) / 4), dim3(256), qkv_lds, c->stream, qa, y.qkv.q4, re);
            HIPCHK(hipGetLastError());
            c->aq_src = nullptr;
        } else {
            CHK(rms(y.in_norm, n, c->l_x, c->l_xn, &y.qkv));
            CHK(llama_gemm(c, y.qkv, c->l_xn, H, c->l_qkv, QKV, n, EPI_STORE));
            hipLaunchKernelGGL(llama_rope_kv_kernel, dim3(n, NH + NKV), dim3(64), 0, c->stream, c->l_qkv, (const uint32_t *) c->l_pos, f32(c->l_ropef), theta_scale, NH, NKV, HD, kc, vc,
                               (const uint32_t *) nullptr, (int64_t) 0);
            HIPCHK(hipGetLastError());
        }
        CHK(launch_attn_gqa(c, NH, n, (int) (attn_positions ? (uint32_t) attn_positions : pos0 + n), (const float *) c->l_qkv, QKV, (const uint32_t *) c->l_pos,
                            (const float *) kc, (const float *) vc, NKV, 1.0f / sqrtf((float) HD), c->l_att, nullptr, nullptr, nullptr, (int64_t) 0, attn_positions != 0,
                            q_for(y.o, n)));
        CHK(llama_gemm(c, y.o, c->l_att, NH * HD, c->l_x, H, n, EPI_RESID));
        const size_t gu_lds = (size_t) n * H + (size_t) n * (H / 32) * 4, dn_lds = (size_t) n * F + (size_t) n * (F / 32) * 4;
        const bool gu_fused = c->q4_silu && c->q4_lds && y.gu.q4 && y.down.q4 && q_for(y.gu, n) && q_for(y.down, n) && H % 512 == 0 && F % 512 == 0 &&
                              gu_lds <= 64 * 1024 && dn_lds <= 64 * 1024 && (int) y.gu.N == 2 * F && !c->prof;
        if (!(gu_fused && rms_fused)) CHK(rms(y.post_norm, n, c->l_x, c->l_xn, &y.gu));
        if (gu_fused) {
            // gate | up with silu * up in the epilogue, then the down projection quantising that product while it stages it: two launches
            // instead of three (gemv_q4_gateup_silu_kernel, gemv_q4_rows_lds_kernel<.., QSRC 1>)
            QGemmArgs qa{};
            qa.g.W = c->arena + y.gu.off; qa.g.K = H; qa.g.N = 2 * F; qa.g.R = n;
            qa.wd = (const _Float16 *) (c->arena + y.gu.soff); qa.aq = c->aq; qa.ad = c->ad;
            if (rms_fused) {
                RmsSrc rs{c->l_x, f32(y.post_norm), 1e-5f};
                hipLaunchKernelGGL((gemv_q4_gateup_silu_kernel<4, 2>), dim3((F / 2 + 3) / 4), dim3(256), gu_lds + 16, c->stream, qa, y.gu.q4, F, c->l_g, rs);
            } else {
                hipLaunchKernelGGL(gemv_q4_gateup_silu_kernel<4>, dim3((F / 2 + 3) / 4), dim3(256), gu_lds, c->stream, qa, y.gu.q4, F, c->l_g);
            }
            HIPCHK(hipGetLastError());
            c->aq_src = nullptr;
            QGemmArgs qd{};
            qd.g.W = c->arena + y.down.off; qd.g.K = F; qd.g.N = H; qd.g.R = n; qd.g.A = c->l_g; qd.g.lda = F; qd.g.out = c->l_x; qd.g.ldo = H;
            qd.wd = (const _Float16 *) (c->arena + y.down.soff);
            static std::atomic<uint64_t> attr{0};
            if (attr_needed(attr, c->device))
                HIPCHK(hipFuncSetAttribute((const void *) gemv_q4_rows_lds_kernel<4, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            hipLaunchKernelGGL((gemv_q4_rows_lds_kernel<4, 2, 1>), dim3((H + 7) / 8), dim3(256), dn_lds, c->stream, qd, y.down.q4, (int) EPI_RESID);
            HIPCHK(hipGetLastError());
            continue;
        }
        CHK(llama_gemm(c, y.gu, c->l_xn, H, c->l_gu, 2 * F, n, EPI_STORE));
        const int ks = (c->gemv_rows && n <= 4) ? 1 : c->l_ksplit;   // the streaming kernels walk all of K themselves
        const bool qd = ks == 1 && q_for(y.down, n);
        hipLaunchKernelGGL(silu_mul_kernel, dim3((unsigned) (((size_t) n * F + 255) / 256)), dim3(256), 0, c->stream, (const float *) c->l_gu, F, n, c->l_g,
                           qd ? c->aq : (int8_t *) nullptr, qd ? c->ad : (float *) nullptr);
        HIPCHK(hipGetLastError());
        c->aq_src = qd ? c->l_g : nullptr;
        if (ks > 1) {
            CHK(llama_gemm(c, y.down, c->l_g, F, c->l_parts, H, n, EPI_STORE, ks));
            c->l_pending = ks;
        } else {
            CHK(llama_gemm(c, y.down, c->l_g, F, c->l_x, H, n, EPI_RESID));
        }
    }
    // lm_head on the last token only (:287-290)
    CHK(rms(c->l_out_norm, n, c->l_x, c->l_xn, nullptr));
    GemmArgs g{};
    g.R = 1; g.H = H; g.A = c->l_xn + (size_t) (n - 1) * H; g.lda = H; g.out = c->l_logits; g.ldo = c->l_Vpad;
    CHK(run_gemm(c, TTS_HIP_K_GEMM_HEADS, c->l_head, g, PRO_F32, EPI_STORE));
    return 0;
}

extern "C" int tts_hip_orpheus_decode(tts_hip_ctx *c, const uint32_t *ids, uint32_t n, uint32_t pos0, float *logits_out, uint32_t *token_out) {
    if (!c || !c->has_llama) return set_err("tts_hip_orpheus_decode: not an Orpheus context (tts_hip_orpheus_create)");
    if (!c->finalized || !c->weights_present) return set_err("tts_hip_orpheus_decode: context not finalized");
    if (!ids || n == 0) return set_err("tts_hip_orpheus_decode: no tokens");
    HIPCHK(hipSetDevice(c->device));
    uint32_t done = 0;
    while (done < n) {   // a long prompt goes through in pieces of RMAX rows (same cache semantics as one call)
        const uint32_t m = std::min<uint32_t>((uint32_t) c->RMAX, n - done);
        CHK(llama_forward(c, ids + done, (int) m, pos0 + done));
        done += m;
    }
    if (token_out) {
        // l_tok: [0] the token, [1..] stage-1 indices, then stage-1 maxima
        uint32_t *pi = c->l_tok + 1;
        float *pv = (float *) (c->l_tok + 1 + ARGMAX_PARTS);
        hipLaunchKernelGGL(argmax_parts_kernel, dim3(ARGMAX_PARTS), dim3(256), 0, c->stream, (const float *) c->l_logits, c->l_V, pv, pi);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(argmax_fold_kernel, dim3(1), dim3(64), 0, c->stream, (const float *) pv, (const uint32_t *) pi, c->l_tok, (uint32_t *) nullptr,
                           (uint32_t *) nullptr, (uint32_t *) nullptr);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(token_out, c->l_tok, 4, hipMemcpyDeviceToHost, c->stream));
    }
    if (logits_out) HIPCHK(hipMemcpyAsync(logits_out, c->l_logits, (size_t) c->l_V * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// sampler::max or sampler::sample of l_logits -> l_tok[0]; captured: the history slot and the uniform come from device counters and the
// token is fed back; eager: hist_slot (may be NULL) receives the token, feed says whether it goes back as the next input
static int llama_select(tts_hip_ctx *c, const tts_hip_sampling *sp, bool captured, uint32_t *hist_slot, bool feed) {
    uint32_t *pi = c->l_tok + 1, *hist = c->l_tok + 1 + 2 * ARGMAX_PARTS, *hist_idx = hist + LLAMA_GREEDY_CHUNK;
    float *pv = (float *) (c->l_tok + 1 + ARGMAX_PARTS);
    if (!sp) {
        hipLaunchKernelGGL(argmax_parts_kernel, dim3(ARGMAX_PARTS), dim3(256), 0, c->stream, (const float *) c->l_logits, c->l_V, pv, pi);
        HIPCHK(hipGetLastError());
        if (captured)
            hipLaunchKernelGGL(argmax_fold_graph_kernel, dim3(1), dim3(64), 0, c->stream, (const float *) pv, (const uint32_t *) pi, c->l_tok, hist, hist_idx, c->l_ids, c->l_pos);
        else
            hipLaunchKernelGGL(argmax_fold_kernel, dim3(1), dim3(64), 0, c->stream, (const float *) pv, (const uint32_t *) pi, c->l_tok, hist_slot,
                               feed ? c->l_ids : (uint32_t *) nullptr, feed ? c->l_pos : (uint32_t *) nullptr);
        HIPCHK(hipGetLastError());
        return 0;
    }
    const double *pen = sp->repetition_penalty != 1.0f ? c->d_pen : nullptr;
    int32_t *last = (int32_t *) c->l_smp;
    uint32_t *repc = c->l_smp + 1, *call = c->l_smp + 2;
    hipLaunchKernelGGL(topk_parts_kernel, dim3(TOPK_PARTS), dim3(512), 0, c->stream, (const float *) c->l_logits, c->l_V, (int) sp->top_k, pen, c->pen_len, (const int32_t *) last,
                       (const uint32_t *) repc, c->l_cand);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(topk_sample_kernel, dim3(1), dim3(1024), 0, c->stream, (const unsigned long long *) c->l_cand, (int) sp->top_k, sp->temperature, (const float *) c->d_uniforms,
                       call, pen, last, repc, c->l_tok, captured ? hist : hist_slot, captured ? hist_idx : (uint32_t *) nullptr,
                       (captured || feed) ? c->l_ids : (uint32_t *) nullptr, (captured || feed) ? c->l_pos : (uint32_t *) nullptr);
    HIPCHK(hipGetLastError());
    return 0;
}

// what the two-stage device sampler covers (topk_parts_kernel / topk_sample_kernel, llama_kernels.h); anything else: sample on the host
// from tts_hip_orpheus_decode's logits
static int check_llama_sampling(const tts_hip_ctx *c, const tts_hip_sampling *sp, const char *what) {
    if (!sp) return set_err("%s: null sampling parameters", what);
    if (!(sp->temperature > 0.0f)) return set_err("%s: temperature must be > 0", what);
    if (!(sp->repetition_penalty > 0.0f)) return set_err("%s: repetition_penalty must be > 0 (1 = off)", what);
    if (sp->top_p < 1.0f) return set_err("%s: top_p < 1 needs the softmax over the whole vocabulary in index order: sample on the host", what);
    if (sp->top_k == 0 || sp->top_k > TOPK_MAXK || (int) sp->top_k >= c->l_V)
        return set_err("%s: the device sampler takes top_k in 1..%d (got %u): sample on the host", what, TOPK_MAXK, sp->top_k);
    if (c->l_V > TOPK_PARTS * TOPK_SLICE) return set_err("%s: vocabulary %d > %d", what, c->l_V, TOPK_PARTS * TOPK_SLICE);
    return 0;
}

// generate_from_batch (:378-392) with sampler::max (sp == NULL) or sampler::sample (sp, uniforms[max_new])
static int orpheus_generate(tts_hip_ctx *c, const char *what, const uint32_t *prompt, uint32_t n_prompt, uint32_t max_new, uint32_t stop_id, const tts_hip_sampling *sp,
                            const float *uniforms, uint32_t *tokens_out, uint32_t *n_out) {
    if (!c || !c->has_llama) return set_err("%s: not an Orpheus context (tts_hip_orpheus_create)", what);
    if (!c->finalized || !c->weights_present) return set_err("%s: context not finalized", what);
    if (!prompt || n_prompt == 0 || !tokens_out || !n_out) return set_err("%s: null argument", what);
    *n_out = 0;
    HIPCHK(hipSetDevice(c->device));
    if (sp) {
        CHK(check_llama_sampling(c, sp, what));
        if (!uniforms) return set_err("%s: null uniforms", what);
        if (max_new == 0) return 0;
        CHK(stage_uniforms(c, uniforms, (size_t) max_new));   // one sampler call per token, at most max_new tokens
        CHK(stage_penalty(c, sp->repetition_penalty, (int) max_new));
        const uint32_t init[3] = {0xFFFFFFFFu, 0u, 0u};   // sampler::reset (sampler.cpp:71-80): last token -1, count 0; call index 0
        HIPCHK(hipMemcpyAsync(c->l_smp, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        const void *pen = sp->repetition_penalty != 1.0f ? (const void *) c->d_pen : nullptr;
        if (c->l_smp_baked.uni != c->d_uniforms || c->l_smp_baked.pen != pen || c->l_smp_baked.k != sp->top_k || c->l_smp_baked.temp != sp->temperature) {
            auto it = c->graphs.find(9000002);
            if (it != c->graphs.end()) { (void) hipGraphExecDestroy(it->second); c->graphs.erase(it); }
            c->l_smp_baked.uni = c->d_uniforms; c->l_smp_baked.pen = pen; c->l_smp_baked.k = sp->top_k; c->l_smp_baked.temp = sp->temperature;
        }
    }
    uint32_t tok = 0, pos = n_prompt;
    {   // the prompt (pieces of RMAX rows), then the first selection
        uint32_t done = 0;
        while (done < n_prompt) {
            const uint32_t m = std::min<uint32_t>((uint32_t) c->RMAX, n_prompt - done);
            CHK(llama_forward(c, prompt + done, (int) m, done));
            done += m;
        }
        CHK(llama_select(c, sp, false, nullptr, false));
        HIPCHK(hipMemcpyAsync(&tok, c->l_tok, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    // stop once the last token is the stopping token or max_generation_size ids exist.  The token never leaves the device inside a
    // chunk of LLAMA_GREEDY_CHUNK steps (the selection writes it back as the next input and bumps the position); the host looks at
    // a chunk's tokens at once, so at most CHUNK-1 steps run past the stopping token (their cache rows are never read: the next
    // call starts at position 0; the sampler draws they consume belong to no token).
    uint32_t *hist = c->l_tok + 1 + 2 * ARGMAX_PARTS;
    uint32_t host_hist[LLAMA_GREEDY_CHUNK];
    while (*n_out < max_new) {
        tokens_out[(*n_out)++] = tok;
        if (tok == stop_id || *n_out >= max_new) break;
        if (pos >= c->lm.n_ctx) break;
        const uint32_t chunk = std::min<uint32_t>(std::min<uint32_t>(LLAMA_GREEDY_CHUNK, max_new - *n_out), c->lm.n_ctx - pos);
        HIPCHK(hipMemcpyAsync(c->l_ids, &tok, 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->l_pos, &pos, 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));  // tok / pos are reused below
        if (c->llama_graph && !c->prof) {
            // one captured step (forward + selection + feedback) replayed `chunk` times; the history slot is a device counter
            uint32_t *hist_idx = hist + LLAMA_GREEDY_CHUNK;
            HIPCHK(hipMemsetAsync(hist_idx, 0, 4, c->stream));
            if (pos + chunk > c->lm.n_ctx) return set_err("%s: positions exceed the cache", what);
            const int key = sp ? 9000002 : 9000001;
            auto it = c->graphs.find(key);
            if (it == c->graphs.end()) {
                // one eager pass first: per-kernel attributes are set outside the capture (it rewrites the cache row of `pos`
                // with the values the first replay writes again, nothing else)
                CHK(llama_forward(c, nullptr, 1, pos, (int) c->lm.n_ctx));
                HIPCHK(hipStreamSynchronize(c->stream));
                hipGraph_t graph = nullptr;
                HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
                int rc = llama_forward(c, nullptr, 1, 0, (int) c->lm.n_ctx);
                if (rc == 0) rc = llama_select(c, sp, true, nullptr, true);
                const hipError_t e = hipStreamEndCapture(c->stream, &graph);
                if (rc != 0) { if (graph) (void) hipGraphDestroy(graph); return rc; }
                if (e != hipSuccess) return set_err("hipStreamEndCapture: %s", hipGetErrorString(e));
                hipGraphExec_t exec = nullptr;
                HIPCHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
                (void) hipGraphDestroy(graph);
                it = c->graphs.emplace(key, exec).first;
            }
            for (uint32_t s = 0; s < chunk; s++) HIPCHK(hipGraphLaunch(it->second, c->stream));
        } else {
            for (uint32_t s = 0; s < chunk; s++) {
                CHK(llama_forward(c, nullptr, 1, pos + s));
                CHK(llama_select(c, sp, false, hist + s, true));
            }
        }
        HIPCHK(hipMemcpyAsync(host_hist, hist, (size_t) chunk * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        pos += chunk;
        // all but the chunk's last token are final here; the last one goes through the loop head like any other
        uint32_t s = 0;
        for (; s + 1 < chunk; s++) {
            tokens_out[(*n_out)++] = host_hist[s];
            if (host_hist[s] == stop_id || *n_out >= max_new) return 0;
        }
        tok = host_hist[s];
    }
    return 0;
}

extern "C" int tts_hip_orpheus_generate_greedy(tts_hip_ctx *c, const uint32_t *prompt, uint32_t n_prompt, uint32_t max_new, uint32_t stop_id,
                                               uint32_t *tokens_out, uint32_t *n_out) {
    return orpheus_generate(c, "tts_hip_orpheus_generate_greedy", prompt, n_prompt, max_new, stop_id, nullptr, nullptr, tokens_out, n_out);
}

extern "C" int tts_hip_orpheus_generate_sampled(tts_hip_ctx *c, const uint32_t *prompt, uint32_t n_prompt, uint32_t max_new, uint32_t stop_id,
                                                const tts_hip_sampling *sampling, const float *uniforms, uint32_t *tokens_out, uint32_t *n_out) {
    if (!sampling) return set_err("tts_hip_orpheus_generate_sampled: null sampling parameters");
    return orpheus_generate(c, "tts_hip_orpheus_generate_sampled", prompt, n_prompt, max_new, stop_id, sampling, uniforms, tokens_out, n_out);
}

extern "C" int tts_hip_orpheus_sample_logits(tts_hip_ctx *c, const float *logits, const tts_hip_sampling *sp, float uniform, int32_t *last_id, uint32_t *rep_count,
                                             uint32_t *token_out) {
    if (!c || !c->has_llama) return set_err("tts_hip_orpheus_sample_logits: not an Orpheus context (tts_hip_orpheus_create)");
    if (!c->finalized) return set_err("tts_hip_orpheus_sample_logits: context not finalized");
    if (!logits || !token_out) return set_err("tts_hip_orpheus_sample_logits: null argument");
    CHK(check_llama_sampling(c, sp, "tts_hip_orpheus_sample_logits"));
    HIPCHK(hipSetDevice(c->device));
    const bool rep = sp->repetition_penalty != 1.0f;
    if (rep && (!last_id || !rep_count)) return set_err("tts_hip_orpheus_sample_logits: repetition penalty needs last_id and rep_count");
    CHK(stage_uniforms(c, &uniform, 1));
    if (rep) CHK(stage_penalty(c, sp->repetition_penalty, (int) std::min<uint32_t>(*rep_count + 2, 1u << 20)));
    const uint32_t init[3] = {rep ? (uint32_t) *last_id : 0xFFFFFFFFu, rep ? *rep_count : 0u, 0u};
    HIPCHK(hipMemcpyAsync(c->l_smp, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->l_logits, logits, (size_t) c->l_V * 4, hipMemcpyHostToDevice, c->stream));
    CHK(llama_select(c, sp, false, nullptr, false));
    uint32_t back[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(token_out, c->l_tok, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(back, c->l_smp, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (rep) { *last_id = (int32_t) back[0]; *rep_count = back[1]; }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Dia (src/models/dia/model.cpp:383-659)
// ------------------------------------------------------------------------------------------------
extern "C" tts_hip_ctx *tts_hip_dia_create(int device, const tts_hip_dia_desc *dd) {
    if (!dd || dd->struct_size != sizeof(tts_hip_dia_desc)) { set_err("tts_hip_dia_create: bad desc (struct_size mismatch)"); return nullptr; }
    tts_hip_desc d{};
    d.struct_size = sizeof(d);
    d.hidden_size = dd->dec_hidden_size; d.n_layers = dd->dec_layers; d.n_attn_heads = dd->dec_attn_heads; d.max_ctx_length = dd->max_gen;
    d.max_seqs = 1;
    d.flags = (dd->flags & (TTS_HIP_FLAG_VALU_GEMM | TTS_HIP_FLAG_DEQUANT_Q)) | TTS_HIP_FLAG_NO_PARLER | TTS_HIP_FLAG_NO_DAC;
    tts_hip_ctx *c = tts_hip_create(device, &d);
    if (!c) return nullptr;
    c->has_dia = true;
    c->dia = *dd;
    if (c->dia.cfg_scale == 0.0f) c->dia.cfg_scale = 3.0f;
    return c;
}

// rows in pieces of RMAX (the activation-quantisation scratch holds RMAX rows); ksplit > 1 only with n <= RMAX
static int dia_gemm(tts_hip_ctx *c, const W &w, const float *A, int lda, float *out, int ldo, int n, int epi, int ksplit = 1) {
    if (ksplit > 1 && n > c->RMAX) return set_err("dia_gemm: split-K needs all rows in one piece");
    for (int r0 = 0; r0 < n; r0 += c->RMAX) {
        GemmArgs g{};
        g.R = std::min(c->RMAX, n - r0); g.H = c->H;
        g.A = A + (size_t) r0 * lda; g.lda = lda;
        g.out = out + (size_t) r0 * ldo; g.ldo = ldo;
        if (ksplit > 1) {
            g.kchunk = (int) w.K / ksplit;
            g.slab_stride = (int64_t) c->RMAX * ldo;
        }
        CHK(run_gemm(c, TTS_HIP_K_GEMM_OTHER, w, g, PRO_F32, epi));
    }
    return 0;
}

// the encoder's GEMMs (2 x max_ctx rows): fp16 matrices take gemm_tile_kernel with all rows in one launch — the activations are rounded
// to fp16 once (what ggml_mul_mat does with them for an F16 weight), every matrix leaves HBM once instead of once per RMAX rows
static int dia_gemm_rows(tts_hip_ctx *c, const W &w, const float *A, int lda, float *out, int ldo, int n, int epi) {
    if (w.type != TTS_HIP_F16 || c->tile_min_rows <= 0 || n < c->tile_min_rows || w.K % 128 || w.N % 16 || (c->d.flags & TTS_HIP_FLAG_VALU_GEMM) || c->prof)
        return dia_gemm(c, w, A, lda, out, ldo, n, epi);
    const int64_t n8 = (int64_t) n * (int64_t) (w.K / 8);
    hipLaunchKernelGGL(rows_to_f16_kernel, dim3((unsigned) ((n8 + 255) / 256)), dim3(256), 0, c->stream, A, lda, (int) w.K, n8, c->di_e16);
    HIPCHK(hipGetLastError());
    GemmArgs g{};
    g.R = n; g.H = c->H;
    g.A = c->di_e16; g.lda = (int) w.K;
    g.out = out; g.ldo = ldo;
    return run_gemm(c, TTS_HIP_K_GEMM_OTHER, w, g, PRO_F16, epi);
}

// <= 16 rows through gemv_stream_kernel: `out` receives *slabs K-slice slabs 16 * ldo floats apart (the consumer folds them);
// *slabs = 0: the shape does not qualify and nothing was launched
static int dia_gemm_stream(tts_hip_ctx *c, const W &w, const float *A, int lda, float *out, int ldo, int n, int max_slabs, int64_t slab_stride, int *slabs) {
    const int ks = stream_slices(c, w, n, max_slabs);
    *slabs = ks;
    if (!ks) return 0;
    GemmArgs g{};
    g.R = n; g.H = c->H;
    g.A = A; g.lda = lda;
    g.out = out; g.ldo = ldo;
    g.stream = 1;
    g.kchunk = ks > 1 ? (int) w.K / ks : 0;
    g.slab_stride = slab_stride;
    return run_gemm(c, TTS_HIP_K_GEMM_OTHER, w, g, PRO_F32, EPI_STORE);
}

static int dia_rms(tts_hip_ctx *c, size_t w_off, int rows, int H, float *x, float *y, bool fold) {
    const int pend = fold ? c->di_pending : 0;
    hipLaunchKernelGGL(rms_fold_rows_kernel, dim3(rows), dim3(256), 0, c->stream, x, H, (const float *) (c->arena + w_off), y, rows, 1e-5f,
                       pend ? (const float *) c->di_parts : (const float *) nullptr, pend, (int64_t) c->RMAX * H, (int8_t *) nullptr, (float *) nullptr);
    if (fold) c->di_pending = 0;
    return hipGetLastError() == hipSuccess ? 0 : set_err("rms_fold_rows_kernel launch failed");
}

extern "C" int tts_hip_dia_encode_slot(tts_hip_ctx *c, uint32_t slot, const uint32_t *tokens, uint32_t sentence_len, float *enc_out) {
    if (!c || !c->has_dia) return set_err("tts_hip_dia_encode: not a Dia context (tts_hip_dia_create)");
    if (slot >= (uint32_t) c->di_U) return set_err("tts_hip_dia_encode_slot: slot %u outside the %d utterance slots of this context (max_utterances)", slot, c->di_U);
    if (!c->finalized || !c->weights_present) return set_err("tts_hip_dia_encode: context not finalized");
    if (!tokens) return set_err("tts_hip_dia_encode: null argument");
    const int S = (int) c->dia.max_ctx, EH = c->di_EH, EF = c->di_EF, A = c->di_A, HD = (int) c->dia.head_dim, ENH = (int) c->dia.enc_attn_heads;
    const int NH = c->NH, n = 2 * S;
    if (sentence_len == 0 || sentence_len > (uint32_t) S) return set_err("tts_hip_dia_encode: sentence length %u outside 1..%d", sentence_len, S);
    std::vector<uint32_t> tok((size_t) n, 0u), epos((size_t) n), eseq((size_t) n), kbeg((size_t) n), kend((size_t) n);
    for (int t = 0; t < S; t++) {
        if (tokens[t] >= (uint32_t) c->di_evocab) return set_err("tts_hip_dia_encode: token %u >= encoder vocabulary %d", tokens[t], c->di_evocab);
        tok[(size_t) t] = tokens[t];
    }
    for (int t = 0; t < n; t++) {   // set_inputs :712-721: real positions see real positions, pad positions see pad positions
        const uint32_t p = (uint32_t) (t % S);
        epos[(size_t) t] = p; eseq[(size_t) t] = (uint32_t) (t / S);
        kbeg[(size_t) t] = p < sentence_len ? 0u : sentence_len;
        kend[(size_t) t] = p < sentence_len ? sentence_len : (uint32_t) S;
    }
    HIPCHK(hipSetDevice(c->device));
    const size_t nb = (size_t) n * 4;
    HIPCHK(hipMemcpyAsync(c->di_tok, tok.data(), nb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->di_epos, epos.data(), nb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->di_eseq, eseq.data(), nb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->di_kbeg, kbeg.data(), nb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->di_kend, kend.data(), nb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));  // the vectors are locals
    auto f32 = [&](size_t off) { return (const float *) (c->arena + off); };
    const float theta_scale = powf(10000.0f, -2.0f / (float) HD);   // ggml_rope(..., head_size, 2): default base
    const size_t attn_lds = (size_t) (128 + S) * 4;
    static std::atomic<uint64_t> attr{0};
    if (attn_lds > 48 * 1024 && attr_needed(attr, c->device))
        HIPCHK(hipFuncSetAttribute((const void *) attn_gqa_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8192));
    hipLaunchKernelGGL(t5_embed_kernel, dim3(n), dim3(256), 0, c->stream, f32(c->di_enc_embd), (const uint32_t *) c->di_tok, EH, c->di_ex);
    HIPCHK(hipGetLastError());
    for (const auto &y : c->di_enc) {
        CHK(dia_rms(c, y.sa_norm, n, EH, c->di_ex, c->di_exn, false));
        CHK(dia_gemm_rows(c, y.qkv, c->di_exn, EH, c->di_eqkv, 3 * A, n, EPI_STORE));
        hipLaunchKernelGGL(llama_rope_kv_kernel, dim3(n, 2 * ENH), dim3(64), 0, c->stream, c->di_eqkv, (const uint32_t *) c->di_epos, (const float *) nullptr, theta_scale,
                           ENH, ENH, HD, c->di_ek, c->di_ev, (const uint32_t *) c->di_eseq, (int64_t) S * A);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(attn_gqa_kernel<128>, dim3(ENH, n), dim3(256), attn_lds, c->stream, (const float *) c->di_eqkv, 3 * A, (const uint32_t *) c->di_epos,
                           (const float *) c->di_ek, (const float *) c->di_ev, ENH, ENH, 1.0f, c->di_eatt, (const uint32_t *) c->di_kbeg, (const uint32_t *) c->di_kend,
                           (const uint32_t *) c->di_eseq, (int64_t) S * A);
        HIPCHK(hipGetLastError());
        CHK(dia_gemm_rows(c, y.o, c->di_eatt, A, c->di_ex, EH, n, EPI_RESID));
        CHK(dia_rms(c, y.mlp_norm, n, EH, c->di_ex, c->di_exn, false));
        CHK(dia_gemm_rows(c, y.gu, c->di_exn, EH, c->di_egu, 2 * EF, n, EPI_STORE));
        hipLaunchKernelGGL(silu_mul_kernel, dim3((unsigned) (((size_t) n * EF + 255) / 256)), dim3(256), 0, c->stream, (const float *) c->di_egu, EF, n, c->di_eg, (int8_t *) nullptr, (float *) nullptr);
        HIPCHK(hipGetLastError());
        CHK(dia_gemm_rows(c, y.out, c->di_eg, EF, c->di_ex, EH, n, EPI_RESID));
    }
    CHK(dia_rms(c, c->di_enc_norm, n, EH, c->di_ex, c->di_exn, false));
    // cross K/V of every decoder layer (build_dia_cross_kv_store :505-541): V for all positions, K (rope'd with the encoder
    // positions) only for the sentence; the other K rows are zero as in the freshly cleared cache
    for (int l = 0; l < c->L; l++) {
        const auto &y = c->di_dec[(size_t) l];
        float *ck = c->di_ck + ((size_t) l * c->di_U + slot) * n * A, *cv = c->di_cv + ((size_t) l * c->di_U + slot) * n * A;   // rows 2*slot, 2*slot+1
        CHK(dia_gemm_rows(c, y.ckv, c->di_exn, EH, c->di_ckv, 2 * A, n, EPI_STORE));
        hipLaunchKernelGGL(llama_rope_kv_kernel, dim3(n, NH), dim3(64), 0, c->stream, c->di_ckv, (const uint32_t *) c->di_epos, (const float *) nullptr, theta_scale, 0, NH,
                           HD, ck, cv, (const uint32_t *) c->di_eseq, (int64_t) S * A);
        HIPCHK(hipGetLastError());
        if ((int) sentence_len < S)
            for (int b = 0; b < 2; b++)
                HIPCHK(hipMemsetAsync(ck + ((size_t) b * S + sentence_len) * A, 0, (size_t) (S - (int) sentence_len) * A * 4, c->stream));
    }
    if (enc_out) HIPCHK(hipMemcpyAsync(enc_out, c->di_exn, (size_t) n * EH * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->di_slot_encoded[slot] = 1;
    return 0;
}

extern "C" int tts_hip_dia_encode(tts_hip_ctx *c, const uint32_t *tokens, uint32_t sentence_len, float *enc_out) {
    return tts_hip_dia_encode_slot(c, 0, tokens, sentence_len, enc_out);
}

// the decoder step for the U utterances whose input ids / positions / cache rows are in di_ids / di_pos / di_seq; leaves the guided
// logits in di_guided.  self_keys sizes the self-attention scratch; fixed_split: a captured step is replayed at every position, so the
// key-split count must not depend on it (the kernels read the true extent from di_pos)
static int dia_forward(tts_hip_ctx *c, int U, int self_keys, bool fixed_split) {
    const int S = (int) c->dia.max_ctx, G = (int) c->dia.max_gen, DH = c->H, DF = c->di_DF, A = c->di_A, kvH = c->di_kvH, HD = (int) c->dia.head_dim;
    const int NH = c->NH, NKV = (int) c->dia.dec_kv_heads, NO = c->NO, V = c->di_V, QKV = A + 2 * kvH;
    const int R = 2 * U, RS = 2 * c->di_U;
    auto f32 = [&](size_t off) { return (const float *) (c->arena + off); };
    const float theta_scale = powf(10000.0f, -2.0f / (float) HD);
    static std::atomic<uint64_t> attr{0};
    if ((size_t) (128 + std::max(S, G)) * 4 > 48 * 1024 && attr_needed(attr, c->device))
        HIPCHK(hipFuncSetAttribute((const void *) attn_gqa_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8192));
    DiaEmbedArgs ea{};
    for (int i = 0; i < NO; i++) ea.table[i] = f32(c->di_embd[i]);
    ea.ids = c->di_ids; ea.n_out = NO; ea.H = DH; ea.x = c->di_x;
    hipLaunchKernelGGL(dia_embed_kernel, dim3((DH + 255) / 256, U), dim3(256), 0, c->stream, ea);
    HIPCHK(hipGetLastError());
    c->di_pending = 0;
    const uint32_t *nul = nullptr;
    for (int l = 0; l < c->L; l++) {
        const auto &y = c->di_dec[(size_t) l];
        float *kc = c->di_k + (size_t) l * RS * G * kvH, *vc = c->di_v + (size_t) l * RS * G * kvH;
        const float *ck = c->di_ck + (size_t) l * RS * S * A, *cv = c->di_cv + (size_t) l * RS * S * A;
        // every projection: gemv_stream_kernel slabs folded by its consumer when the step has <= 16 rows and fp16 matrices
        // (sl = slabs written, 0 = shape does not qualify -> gemm16_kernel as before)
        int sl = 0;
        const int64_t st16 = 16;   // slab stride in rows
        CHK(dia_rms(c, y.sa_norm, R, DH, c->di_x, c->di_xn, true));
        CHK(dia_gemm_stream(c, y.sqkv, c->di_xn, DH, c->di_qkv, QKV, R, DIA_STREAM_SLABS, st16 * QKV, &sl));
        if (!sl) CHK(dia_gemm(c, y.sqkv, c->di_xn, DH, c->di_qkv, QKV, R, EPI_STORE));
        hipLaunchKernelGGL(llama_rope_kv_kernel, dim3(R, NH + NKV), dim3(64), 0, c->stream, c->di_qkv, (const uint32_t *) c->di_pos, (const float *) nullptr, theta_scale, NH,
                           NKV, HD, kc, vc, (const uint32_t *) c->di_seq, (int64_t) G * kvH, std::max(sl, 1), st16 * QKV);
        HIPCHK(hipGetLastError());
        CHK(launch_attn_gqa(c, NH, R, self_keys, (const float *) c->di_qkv, QKV, (const uint32_t *) c->di_pos, (const float *) kc, (const float *) vc, NKV, 1.0f,
                            c->di_att, nul, nul, (const uint32_t *) c->di_seq, (int64_t) G * kvH, fixed_split));
        CHK(dia_gemm_stream(c, y.so, c->di_att, A, c->di_parts, DH, R, DIA_STREAM_SLABS, (int64_t) c->RMAX * DH, &sl));
        if (sl) c->di_pending = sl;
        else CHK(dia_gemm(c, y.so, c->di_att, A, c->di_x, DH, R, EPI_RESID));
        CHK(dia_rms(c, y.ca_norm, R, DH, c->di_x, c->di_xn, true));
        CHK(dia_gemm_stream(c, y.cq, c->di_xn, DH, c->di_q, A, R, DIA_STREAM_SLABS, st16 * A, &sl));
        if (!sl) CHK(dia_gemm(c, y.cq, c->di_xn, DH, c->di_q, A, R, EPI_STORE));
        QPre qp;   // slab fold + rope of the cross-attention query happen as the attention workgroups load it
        qp.n_parts = std::max(sl, 1); qp.part_stride = st16 * A; qp.rope_pos = c->di_pos; qp.theta_scale = theta_scale;
        CHK(launch_attn_gqa(c, NH, R, S, (const float *) c->di_q, A, (const uint32_t *) c->di_pos, ck, cv, NH, 1.0f, c->di_att, nul, (const uint32_t *) c->di_cend,
                            (const uint32_t *) c->di_seq, (int64_t) S * A, false, false, qp));
        CHK(dia_gemm_stream(c, y.co, c->di_att, A, c->di_parts, DH, R, DIA_STREAM_SLABS, (int64_t) c->RMAX * DH, &sl));
        if (sl) c->di_pending = sl;
        else CHK(dia_gemm(c, y.co, c->di_att, A, c->di_x, DH, R, EPI_RESID));
        CHK(dia_rms(c, y.mlp_norm, R, DH, c->di_x, c->di_xn, true));
        CHK(dia_gemm_stream(c, y.gu, c->di_xn, DH, c->di_gu, 2 * DF, R, DIA_STREAM_SLABS, st16 * 2 * DF, &sl));
        if (!sl) CHK(dia_gemm(c, y.gu, c->di_xn, DH, c->di_gu, 2 * DF, R, EPI_STORE));
        hipLaunchKernelGGL(silu_mul_kernel, dim3((unsigned) (((size_t) R * DF + 255) / 256)), dim3(256), 0, c->stream, (const float *) c->di_gu, DF, R, c->di_g, (int8_t *) nullptr,
                           (float *) nullptr, std::max(sl, 1), st16 * 2 * DF);
        HIPCHK(hipGetLastError());
        CHK(dia_gemm_stream(c, y.out, c->di_g, DF, c->di_parts, DH, R, DIA_STREAM_SLABS, (int64_t) c->RMAX * DH, &sl));
        if (sl) {
            c->di_pending = sl;
        } else if (c->di_ksplit > 1) {
            CHK(dia_gemm(c, y.out, c->di_g, DF, c->di_parts, DH, R, EPI_STORE, c->di_ksplit));
            c->di_pending = c->di_ksplit;
        } else {
            CHK(dia_gemm(c, y.out, c->di_g, DF, c->di_x, DH, R, EPI_RESID));
        }
    }
    CHK(dia_rms(c, c->di_dec_norm, R, DH, c->di_x, c->di_xn, true));
    GemmArgs g{};
    g.R = R; g.H = DH; g.A = c->di_xn; g.lda = DH; g.out = c->di_logits; g.ldo = c->di_Vpad;
    CHK(run_gemm(c, TTS_HIP_K_GEMM_HEADS, c->di_heads, g, PRO_F32, EPI_STORE));
    hipLaunchKernelGGL(dia_cfg_kernel, dim3((NO * V + 255) / 256, U), dim3(256), 0, c->stream, (const float *) c->di_logits, c->di_Vpad, NO * V, c->dia.cfg_scale, c->di_guided);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int tts_hip_dia_step_batch(tts_hip_ctx *c, uint32_t n_utt, const uint32_t *slots, const uint32_t *ids, const uint32_t *pos, float *logits_out,
                                      float *raw_out) {
    if (!c || !c->has_dia) return set_err("tts_hip_dia_step: not a Dia context (tts_hip_dia_create)");
    if (!c->finalized || !c->weights_present) return set_err("tts_hip_dia_step: context not finalized");
    if (!ids || !pos || !logits_out) return set_err("tts_hip_dia_step: null argument");
    if (n_utt == 0 || n_utt > (uint32_t) c->di_U) return set_err("tts_hip_dia_step_batch: %u utterances outside 1..%d (max_utterances)", n_utt, c->di_U);
    const int G = (int) c->dia.max_gen, NO = c->NO, V = c->di_V;
    const int U = (int) n_utt, R = 2 * U, RS = 2 * c->di_U;   // rows of this step, row slots of the caches
    uint32_t max_pos = 0;
    uint32_t *h_ids = c->h_di, *h_pos = c->h_di + (size_t) c->di_U * 16, *h_seq = h_pos + RS;
    for (int u = 0; u < U; u++) {
        const uint32_t slot = slots ? slots[u] : (uint32_t) u;
        if (slot >= (uint32_t) c->di_U) return set_err("tts_hip_dia_step_batch: slot %u outside the %d utterance slots", slot, c->di_U);
        if (!c->di_slot_encoded[slot]) return set_err("tts_hip_dia_step: tts_hip_dia_encode has not run%s", c->di_U > 1 ? " for this slot" : "");
        if (pos[u] >= (uint32_t) G) return set_err("tts_hip_dia_step: position %u outside the %d cached positions", pos[u], G);
        for (int i = 0; i < NO; i++) {
            if (ids[u * NO + i] >= (uint32_t) V) return set_err("tts_hip_dia_step: id %u >= output vocabulary %d", ids[u * NO + i], V);
            h_ids[u * NO + i] = ids[u * NO + i];
        }
        h_pos[2 * u] = h_pos[2 * u + 1] = pos[u];
        h_seq[2 * u] = 2 * slot; h_seq[2 * u + 1] = 2 * slot + 1;
        max_pos = std::max(max_pos, pos[u]);
    }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(c->di_ids, h_ids, (size_t) U * NO * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->di_pos, h_pos, (size_t) R * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->di_seq, h_seq, (size_t) R * 4, hipMemcpyHostToDevice, c->stream));
    CHK(dia_forward(c, U, (int) max_pos + 1, false));
    HIPCHK(hipMemcpyAsync(logits_out, c->di_guided, (size_t) U * NO * V * 4, hipMemcpyDeviceToHost, c->stream));
    if (raw_out)
        for (int b = 0; b < R; b++)
            HIPCHK(hipMemcpyAsync(raw_out + (size_t) b * NO * V, c->di_logits + (size_t) b * c->di_Vpad, (size_t) NO * V * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

#define DIA_LOOP_CHUNK 16
extern "C" int tts_hip_dia_generate(tts_hip_ctx *c, uint32_t n_utt, uint32_t max_gen, const tts_hip_dia_codes *codes, const tts_hip_sampling *sp, const float *uniforms,
                                    uint32_t *tokens_out, uint32_t *steps_out) {
    if (!c || !c->has_dia) return set_err("tts_hip_dia_generate: not a Dia context (tts_hip_dia_create)");
    if (!c->finalized || !c->weights_present) return set_err("tts_hip_dia_generate: context not finalized");
    if (!codes || !tokens_out || !steps_out) return set_err("tts_hip_dia_generate: null argument");
    if (n_utt == 0 || n_utt > (uint32_t) c->di_U) return set_err("tts_hip_dia_generate: %u utterances outside 1..%d (max_utterances)", n_utt, c->di_U);
    const int G = (int) c->dia.max_gen, NO = c->NO, V = c->di_V, U = (int) n_utt;
    if (max_gen == 0 || max_gen > (uint32_t) G) return set_err("tts_hip_dia_generate: max_gen %u outside 1..%d cached positions", max_gen, G);
    if (codes->max_delay >= max_gen) return set_err("tts_hip_dia_generate: max_gen %u must exceed max_delay %u", max_gen, codes->max_delay);
    if (codes->bos >= (uint32_t) V || codes->eos >= (uint32_t) V || codes->pad >= (uint32_t) V) return set_err("tts_hip_dia_generate: special ids outside the vocabulary %d", V);
    for (int u = 0; u < U; u++)
        if (!c->di_slot_encoded[(size_t) u]) return set_err("tts_hip_dia_generate: slot %d has not been encoded (tts_hip_dia_encode_slot)", u);
    if (sp) {
        if (V > SMP_VMAX) return set_err("tts_hip_dia_generate: output vocabulary %d > %d", V, SMP_VMAX);
        if (!(sp->temperature > 0.0f) || !(sp->top_p > 0.0f) || !(sp->repetition_penalty > 0.0f)) return set_err("tts_hip_dia_generate: temperature, top_p, repetition_penalty must be > 0");
        if (!uniforms) return set_err("tts_hip_dia_generate: null uniforms");
    }
    HIPCHK(hipSetDevice(c->device));
    const bool rep = sp && sp->repetition_penalty != 1.0f;
    if (sp) {
        CHK(stage_uniforms(c, uniforms, (size_t) max_gen * U * NO));
        CHK(stage_penalty(c, sp->repetition_penalty, (int) max_gen));
    }
    // loop state: ids = BOS everywhere, positions 0, countdown -1, nothing done; sampler::reset (sampler.cpp:71-80)
    {
        std::vector<uint32_t> ids((size_t) U * NO, codes->bos), zero((size_t) 3 * c->di_U, 0u), seq((size_t) 2 * U);
        for (int u = 0; u < U; u++) { zero[(size_t) u] = 0xFFFFFFFFu; seq[(size_t) 2 * u] = 2 * u; seq[(size_t) 2 * u + 1] = 2 * u + 1; }
        HIPCHK(hipMemcpyAsync(c->di_ids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemsetAsync(c->di_pos, 0, (size_t) 2 * U * 4, c->stream));
        HIPCHK(hipMemcpyAsync(c->di_seq, seq.data(), seq.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->di_loop, zero.data(), zero.size() * 4, hipMemcpyHostToDevice, c->stream));
        if (rep) {
            HIPCHK(hipMemsetAsync(c->d_last, 0xFF, (size_t) U * NO * 4, c->stream));
            HIPCHK(hipMemsetAsync(c->d_repc, 0, (size_t) U * NO * 4, c->stream));
        }
        HIPCHK(hipStreamSynchronize(c->stream));   // the vectors are locals
    }
    DiaLoopArgs la{};
    la.n_utt = U; la.n_out = NO;
    la.bos = codes->bos; la.eos = codes->eos; la.pad = codes->pad; la.max_delay = codes->max_delay; la.max_gen = max_gen;
    for (int i = 0; i < 16; i++) la.delay_pattern[i] = codes->delay_pattern[i];
    la.ids = c->di_ids; la.pos = c->di_pos;
    la.delay = (int32_t *) c->di_loop; la.done = c->di_loop + c->di_U; la.call = c->di_loop + 2 * c->di_U;
    la.tok = c->di_stok; la.hist = c->di_hist;
    auto one_step = [&](bool captured) -> int {
        hipLaunchKernelGGL(dia_prestep_kernel, dim3((U + 63) / 64), dim3(64), 0, c->stream, la);
        HIPCHK(hipGetLastError());
        CHK(dia_forward(c, U, G, captured));
        if (sp) {
            SampleArgs sa{};
            sa.logits = c->di_guided; sa.V = V; sa.n_out = NO; sa.R = U;
            sa.top_k = sp->top_k; sa.top_p = sp->top_p; sa.temperature = sp->temperature;
            sa.uniforms = c->d_uniforms; sa.row_step = la.call; sa.out = c->di_stok;
            if (rep) { sa.pen_table = c->d_pen; sa.pen_len = c->pen_len; sa.last_ids = c->d_last; sa.rep_counts = c->d_repc; }
            hipLaunchKernelGGL(sample_kernel, dim3(NO, U), dim3(256), 0, c->stream, sa);
        } else {
            hipLaunchKernelGGL(argmax_kernel, dim3(U * NO), dim3(256), 0, c->stream, (const float *) c->di_guided, V, c->di_stok);
        }
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(dia_poststep_kernel, dim3((U + 63) / 64), dim3(64), 0, c->stream, la);
        HIPCHK(hipGetLastError());
        return 0;
    };
    // everything the captured launches hold by value: a change drops the graph
    const int mode = sp ? 1 : 0;
    const void *pen = rep ? (const void *) c->d_pen : nullptr;
    const tts_hip_sampling spv = sp ? *sp : tts_hip_sampling{};
    auto &bk = c->di_baked;
    const bool same = bk.mode == mode && bk.U == n_utt && bk.max_gen == max_gen && memcmp(&bk.codes, codes, sizeof(*codes)) == 0 &&
                      (!sp || (bk.uni == c->d_uniforms && bk.pen == pen && memcmp(&bk.sp, &spv, sizeof(spv)) == 0));
    const int key = 9100001;
    if (!same) {
        auto it = c->graphs.find(key);
        if (it != c->graphs.end()) { (void) hipGraphExecDestroy(it->second); c->graphs.erase(it); }
        bk.mode = mode; bk.U = n_utt; bk.max_gen = max_gen; bk.codes = *codes; bk.uni = c->d_uniforms; bk.pen = pen; bk.sp = spv;
    }
    const bool use_graph = !(c->d.flags & TTS_HIP_FLAG_NO_GRAPH) && !c->prof;
    std::vector<uint32_t> done((size_t) U);
    uint32_t ran = 0;
    while (ran < max_gen + 1) {   // at most max_gen sampler calls, then one pre-step that ends the countdown
        const uint32_t chunk = std::min<uint32_t>(DIA_LOOP_CHUNK, max_gen + 1 - ran);
        if (use_graph) {
            auto it = c->graphs.find(key);
            if (it == c->graphs.end()) {
                // the first step runs eagerly (per-kernel attributes are set outside a capture), the capture follows
                CHK(one_step(false));
                HIPCHK(hipStreamSynchronize(c->stream));
                hipGraph_t graph = nullptr;
                HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
                const int rc = one_step(true);
                const hipError_t e = hipStreamEndCapture(c->stream, &graph);
                if (rc != 0) { if (graph) (void) hipGraphDestroy(graph); return rc; }
                if (e != hipSuccess) return set_err("hipStreamEndCapture: %s", hipGetErrorString(e));
                hipGraphExec_t exec = nullptr;
                HIPCHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
                (void) hipGraphDestroy(graph);
                it = c->graphs.emplace(key, exec).first;
                for (uint32_t s = 1; s < chunk; s++) HIPCHK(hipGraphLaunch(it->second, c->stream));
            } else {
                for (uint32_t s = 0; s < chunk; s++) HIPCHK(hipGraphLaunch(it->second, c->stream));
            }
        } else {
            for (uint32_t s = 0; s < chunk; s++) CHK(one_step(false));
        }
        ran += chunk;
        HIPCHK(hipMemcpyAsync(done.data(), la.done, (size_t) U * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        bool all = true;
        for (int u = 0; u < U; u++) all = all && done[(size_t) u] != 0;
        if (all) break;
    }
    std::vector<uint32_t> pos((size_t) 2 * U);
    HIPCHK(hipMemcpyAsync(pos.data(), c->di_pos, pos.size() * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(tokens_out, c->di_hist, (size_t) U * max_gen * NO * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int u = 0; u < U; u++) steps_out[u] = pos[(size_t) 2 * u];
    return 0;
}

extern "C" int tts_hip_dia_step(tts_hip_ctx *c, const uint32_t *ids, uint32_t pos, float *logits_out, float *raw_out) {
    return tts_hip_dia_step_batch(c, 1, nullptr, ids, &pos, logits_out, raw_out);
}

// ------------------------------------------------------------------------------------------------
// Kokoro (src/models/kokoro/model.cpp:938-1047, 1141-1242, 195-244); first version, see kokoro_kernels.h
// ------------------------------------------------------------------------------------------------
extern "C" tts_hip_ctx *tts_hip_kokoro_create(int device, const tts_hip_kokoro_desc *kd) {
    if (!kd || kd->struct_size != sizeof(tts_hip_kokoro_desc)) { set_err("tts_hip_kokoro_create: bad desc (struct_size mismatch)"); return nullptr; }
    if (kd->n_upsamples == 0 || kd->n_upsamples > 4 || kd->n_kernels == 0 || kd->n_upsamples * kd->n_kernels > 16 || kd->n_fft < 2 || kd->hop == 0 || kd->max_ctx < 3) {
        set_err("tts_hip_kokoro_create: generator geometry out of range");
        return nullptr;
    }
    tts_hip_desc d{};
    d.struct_size = sizeof(d);
    d.max_seqs = 1;
    d.flags = TTS_HIP_FLAG_NO_PARLER | TTS_HIP_FLAG_NO_DAC;
    tts_hip_ctx *c = tts_hip_create(device, &d);
    if (!c) return nullptr;
    c->has_kokoro = true;
    c->ko = *kd;
    if (hipMalloc((void **) &c->kk_stuck, 4) != hipSuccess || hipMemset(c->kk_stuck, 0, 4) != hipSuccess) { set_err("tts_hip_kokoro_create: hipMalloc failed"); tts_hip_destroy(c); return nullptr; }
    return c;
}

namespace {
// scratch for one call: pieces of a context-owned pool that grows to the largest call seen (a synthesis asks for ~100 buffers: one
// hipMalloc / hipFree pair each cost more host time than the kernels they fed); what does not fit yet is allocated for this call alone
struct KScratch {
    tts_hip_ctx *c;
    size_t off = 0, want = 0;
    std::vector<void *> extra;
    bool failed = false;
    explicit KScratch(tts_hip_ctx *c_) : c(c_) {
        if (c->kk_pool_next > c->kk_pool_cap) {
            (void) hipStreamSynchronize(c->stream);
            if (c->kk_pool) (void) hipFree(c->kk_pool);
            c->kk_pool = nullptr; c->kk_pool_cap = 0;
            const size_t cap = c->kk_pool_next + c->kk_pool_next / 4;
            if (hipMalloc((void **) &c->kk_pool, cap) == hipSuccess) c->kk_pool_cap = cap;
        }
    }
    ~KScratch() {
        if (!extra.empty()) (void) hipStreamSynchronize(c->stream);
        for (void *p : extra) (void) hipFree(p);
        if (want > c->kk_pool_cap) c->kk_pool_next = std::max(c->kk_pool_next, want);
    }
    float *f(size_t n) {
        const size_t bytes = ((n ? n : 1) * sizeof(float) + 255) & ~(size_t) 255;
        want += bytes;
        if (c->kk_pool && off + bytes <= c->kk_pool_cap) {
            float *p = (float *) (c->kk_pool + off);
            off += bytes;
            return p;
        }
        void *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) { failed = true; return nullptr; }
        extra.push_back(p);
        return (float *) p;
    }
};
inline dim3 kgrid(int64_t n, int bs = 256) { return dim3((unsigned) ((n + bs - 1) / bs)); }

struct KRun {
    tts_hip_ctx *c;
    KScratch &s;
    std::string err;
    hipStream_t st;
    KRun(tts_hip_ctx *c_, KScratch &s_) : c(c_), s(s_), st(c_->stream) {}
    bool has(const std::string &n) const { return c->k_tensors.count("kokoro." + n) != 0; }
    const float *w(const std::string &n, int64_t *ne = nullptr) {
        auto it = c->k_tensors.find("kokoro." + n);
        if (it == c->k_tensors.end()) { if (err.empty()) err = "missing tensor 'kokoro." + n + "'"; return nullptr; }
        if (ne) memcpy(ne, it->second.ne, sizeof(int64_t) * 4);
        return (const float *) (c->arena + it->second.off);
    }
    bool ok() {
        if (!err.empty()) return false;
        if (s.failed) { err = "device scratch allocation failed"; return false; }
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) { err = std::string("kernel launch failed: ") + hipGetErrorString(e); return false; }
        return true;
    }
    void linear(const float *W, const float *b, const float *x, int ldx, int R, int K, int N, float *y, int ldy, int acc = 0) {
        if (!W || !x || !y) return;
        if (c->kk_mfma && R >= 32 && K % 16 == 0 && ldx % 4 == 0 && ((uintptr_t) W & 15) == 0 && ((uintptr_t) x & 15) == 0) {
            // many rows (ALBERT and the predictor over the whole sequence): 64 x 64 tiles on the exact-fp32 matrix pipe
            hipLaunchKernelGGL(kk_linear_mfma_kernel, dim3((unsigned) ((N + 63) / 64), (unsigned) ((R + 63) / 64)), dim3(256), 0, st, W, b, x, ldx, R, K, N, y, ldy, acc);
            return;
        }
        hipLaunchKernelGGL(kk_linear_kernel, dim3((unsigned) (((int64_t) R * N + 3) / 4)), dim3(256), 0, st, W, b, x, ldx, R, K, N, y, ldy, acc);
    }
    void norm_rows(const float *x, int ldx, int R, int H, float eps, const float *w_, const float *b, int mode, float *y, int ldy) {
        if (!x || !y) return;
        hipLaunchKernelGGL(kk_norm_rows_kernel, dim3(R), dim3(64), 0, st, x, ldx, H, eps, w_, b, mode, y, ldy);
    }
    void copy(float *dst, const float *src, size_t n) { if (dst && src) (void) hipMemcpyAsync(dst, src, n * 4, hipMemcpyDeviceToDevice, st); }
    // build_lstm (:35-51): one bidirectional cell; x [L][in] -> out [L][2 hid]
    void bilstm(const std::string &base, const float *x, int L, int in, int hid, float *out) {
        const int cp = hid / 4;
        const bool split = c->kk_lstm_split && hid % 16 == 0 && (cp == 4 || cp == 8 || cp == 16 || cp == 32 || cp == 64 || cp == 128) && L > 1;
        float *pre = s.f((size_t) 8 * L * hid);   // both directions' input pre-activations
        if (split) {
            // the recurrence of both directions in one launch over hid/16 workgroups each (kk_lstm_split_kernel)
            LstmArgs la{};
            for (int dir = 0; dir < 2; dir++) {
                const std::string wn = dir ? ".0.reverse_weights." : ".0.weights.", bn = dir ? ".0.reverse_biases." : ".0.biases.";
                float *pd = pre ? pre + (size_t) dir * 4 * L * hid : nullptr;
                for (int g = 0; g < 4; g++) {
                    linear(w(base + wn + std::to_string(2 * g)), w(base + bn + std::to_string(2 * g)), x, in, L, in, hid, pd ? pd + (size_t) g * L * hid : nullptr, hid);
                    la.whh[dir][g] = w(base + wn + std::to_string(2 * g + 1));
                    la.bhh[dir][g] = w(base + bn + std::to_string(2 * g + 1));
                }
                la.pre[dir] = pd;
            }
            float *xch = s.f((size_t) 2 * 2 * hid * 2);   // 8-byte granules
            if (!err.empty() || !pre || !xch) return;
            (void) hipMemsetAsync(xch, 0, (size_t) 2 * 2 * hid * 8, st);
            (void) hipMemsetAsync(c->kk_stuck, 0, 4, st);
            la.xch = (unsigned long long *) xch; la.out = out; la.L = L; la.hid = hid; la.out_stride = 2 * hid; la.stuck = c->kk_stuck;
            const dim3 grid(hid / 16, 2);
            const size_t lds = (size_t) hid * 4;
            switch (cp) {
                case 4: hipLaunchKernelGGL(kk_lstm_split_kernel<4>, grid, dim3(256), lds, st, la); break;
                case 8: hipLaunchKernelGGL(kk_lstm_split_kernel<8>, grid, dim3(256), lds, st, la); break;
                case 16: hipLaunchKernelGGL(kk_lstm_split_kernel<16>, grid, dim3(256), lds, st, la); break;
                case 32: hipLaunchKernelGGL(kk_lstm_split_kernel<32>, grid, dim3(256), lds, st, la); break;
                case 64: hipLaunchKernelGGL(kk_lstm_split_kernel<64>, grid, dim3(256), lds, st, la); break;
                default: hipLaunchKernelGGL(kk_lstm_split_kernel<128>, grid, dim3(256), lds, st, la); break;
            }
            return;
        }
        for (int dir = 0; dir < 2; dir++) {
            const std::string wn = dir ? ".0.reverse_weights." : ".0.weights.", bn = dir ? ".0.reverse_biases." : ".0.biases.";
            const float *whh[4], *bhh[4];
            for (int g = 0; g < 4; g++) {
                linear(w(base + wn + std::to_string(2 * g)), w(base + bn + std::to_string(2 * g)), x, in, L, in, hid, pre + (size_t) g * L * hid, hid);
                whh[g] = w(base + wn + std::to_string(2 * g + 1));
                bhh[g] = w(base + bn + std::to_string(2 * g + 1));
            }
            if (!err.empty() || !pre) return;
            const int threads = std::max(64, (hid + 63) / 64 * 64);
            hipLaunchKernelGGL(kk_lstm_kernel, dim3(1), dim3(threads), (size_t) hid * 4, st, (const float *) pre, whh[0], whh[1], whh[2], whh[3], bhh[0], bhh[1], bhh[2], bhh[3],
                               L, hid, dir, out, 2 * hid, dir * hid);
        }
    }
    // gamma / beta = W style + b, then the fused instance norm (:93-101)
    void adain(float *x, int C, int64_t L, const float *style, int S, const std::string &gw, const std::string &gb, const std::string &bw, const std::string &bb, int act,
               float slope, const float *alpha) {
        float *gamma = s.f(C), *beta = s.f(C);
        linear(w(gw), w(gb), style, S, 1, S, C, gamma, C);
        linear(w(bw), w(bb), style, S, 1, S, C, beta, C);
        if (!err.empty() || !gamma || !beta) return;
        if (c->kk_mfma && L >= 8192) {   // long rows: slices over workgroups, three phases (kk_adain_split_kernel)
            const int S = (int) std::min<int64_t>(32, std::max<int64_t>(2, (int64_t) 2048 / C));
            float *part = s.f((size_t) 2 * C * S);
            if (!part) return;
            for (int phase = 0; phase < 3; phase++)
                hipLaunchKernelGGL(kk_adain_split_kernel, dim3(C, S), dim3(256), 0, st, x, L, (const float *) gamma, (const float *) beta, act, slope, alpha, part, phase);
            return;
        }
        hipLaunchKernelGGL(kk_adain_kernel, dim3(C), dim3(256), 0, st, x, L, (const float *) gamma, (const float *) beta, act, slope, alpha);
    }
    // Stride-1 "same" convolutions with enough channels (the generator's residual blocks k = 3 / 7 / 11 with dilations 1 / 3 / 5, the
    // AdaIN residual blocks k = 3, the text encoder k = 5, conv_post k = 7: 59 % of the model's kernel time through the one-thread-per-
    // output kernel, profiles/r02/kernel_stats_kokoro_82m.csv) go through the codec's exact-fp32 MFMA conv kernel (conv1d_mfma_kernel: input
    // channels staged through LDS, weights pre-packed once per tensor into its LDS image); `acc` becomes its residual input (y += conv).
    // Everything else (stride 2, nearest-2x input, one output channel, the scaled shortcut) stays on kk_conv1d_kernel.
    template <int KT, int CI_T>
    bool conv_mfma(const float *x, int cin, int64_t L, const float *wt, const float *b, int cout, int pad, int dil, float *y, int acc) {
        const size_t w_off = (size_t) ((const char *) wt - c->arena);
        const int CO_T = cout % 128 == 0 ? 128 : 64;   // other widths (conv_post: 22 channels): 64-channel tiles, zero-padded weights, stores masked
        if (c->packed.find(w_off) == c->packed.end() && pack_one(c, w_off, cout, cin, KT, CO_T, CI_T, false) != 0) { err = tts_hip_last_error(); return true; }
        ConvArgs a{};
        a.x = x; a.w = c->packed[w_off]; a.b = b; a.alpha = nullptr; a.alpha_out = nullptr; a.resid = acc ? y : nullptr; a.y = y;
        a.cin = cin; a.cout = cout; a.L = (int) L; a.dil = dil; a.pad = pad; a.do_tanh = 0; a.frames = nullptr; a.mult = 1; a.x_f16 = 0;
        if (prof_begin(c, TTS_HIP_K_KOKORO_CONV, ((double) cin * L + (double) cout * L * (acc ? 2 : 1) + (double) cout * cin * KT) * 4, 2.0 * cout * (double) cin * KT * L) != 0) { err = tts_hip_last_error(); return true; }
        const int rc = CO_T == 128 ? launch_conv_mfma<KT, 2, 2, 2, 2, CI_T>(c, a, 1) : launch_conv_mfma<KT, 2, 2, 1, 4, CI_T>(c, a, 1);
        if (rc != 0 || prof_end(c) != 0) err = tts_hip_last_error();
        return true;
    }
    void conv1d(const float *x, int cin, int64_t L, const float *wt, const float *b, int cout, int K, int stride, int pad, int dil, int in_shift, float *y, int64_t Lout,
                int acc, float post) {
        if (!x || !wt || !y) return;
        const bool same = stride == 1 && !in_shift && Lout == L && pad * 2 == dil * (K - 1) && post == 1.0f && dil <= 9;
        if (same && c->kk_mfma && cout >= 16 && cin >= 16 && L < (1 << 30) && (const char *) wt >= c->arena && (const char *) wt < c->arena + c->arena_bytes) {
            if (K == 1 && (L & 3) == 0 && (((uintptr_t) x | (uintptr_t) y) & 15) == 0 && conv_mfma<1, 16>(x, cin, L, wt, b, cout, pad, dil, y, acc)) return;   // k = 1 stages 16-byte pieces
            if (K == 3 && conv_mfma<3, 8>(x, cin, L, wt, b, cout, pad, dil, y, acc)) return;
            if (K == 5 && conv_mfma<5, 4>(x, cin, L, wt, b, cout, pad, dil, y, acc)) return;
            if (K == 7 && conv_mfma<7, 4>(x, cin, L, wt, b, cout, pad, dil, y, acc)) return;
            if (K == 11 && conv_mfma<11, 4>(x, cin, L, wt, b, cout, pad, dil, y, acc)) return;
        }
        if (c->kk_mfma && K == 1 && stride == 1 && pad == 0 && dil == 1 && cin >= 16 && cout >= 16 && Lout == (in_shift ? 2 * L : L)) {
            // the k = 1 shortcuts (any channel count, nearest-2x input, accumulate + 1/sqrt 2): 64 x 64 tiles on the exact-fp32 matrix pipe
            hipLaunchKernelGGL(kk_conv1x1_mfma_kernel, dim3((unsigned) ((Lout + 63) / 64), (unsigned) ((cout + 63) / 64)), dim3(256), 0, st, x, cin, L, wt, b, cout, in_shift, y, Lout, acc, post);
            return;
        }
        hipLaunchKernelGGL(kk_conv1d_kernel, kgrid((int64_t) cout * Lout), dim3(256), 0, st, x, cin, L, wt, b, cout, K, stride, pad, dil, in_shift, y, Lout, acc, post);
    }
    // build_ada_residual_conv (:88-134): x [cin][L] -> [cout][L or 2L]
    float *ada_block(const std::string &base, const float *x, int64_t L, const float *style, int S, int &C, int64_t &Lout) {
        int64_t ne[4];
        const float *conv1 = w(base + ".conv1_weight", ne);
        if (!conv1) return nullptr;
        const int cin = (int) ne[1], cout = (int) ne[2];
        if (cin != C) { err = base + ": channel count mismatch"; return nullptr; }
        float *cur = s.f((size_t) cin * L);
        copy(cur, x, (size_t) cin * L);
        adain(cur, cin, L, style, S, base + ".norm1_gamma_weight", base + ".norm1_gamma_bias", base + ".norm1_beta_weight", base + ".norm1_beta_bias", 1, 0.2f, nullptr);
        const bool pool = has(base + ".pool_weight");
        int64_t Lc = L;
        if (pool) {
            float *up = s.f((size_t) cin * 2 * L);
            if (up) hipLaunchKernelGGL(kk_pool_convt_kernel, kgrid((int64_t) cin * 2 * L), dim3(256), 0, st, (const float *) cur, cin, L, w(base + ".pool_weight"), w(base + ".pool_bias"), up);
            cur = up;
            Lc = 2 * L;
        }
        float *y = s.f((size_t) cout * Lc);
        conv1d(cur, cin, Lc, conv1, w(base + ".conv1_bias"), cout, 3, 1, 1, 1, 0, y, Lc, 0, 1.0f);
        adain(y, cout, Lc, style, S, base + ".norm2_gamma_weight", base + ".norm2_gamma_bias", base + ".norm2_beta_weight", base + ".norm2_beta_bias", 1, 0.2f, nullptr);
        float *res = s.f((size_t) cout * Lc);
        conv1d(y, cout, Lc, w(base + ".conv2_weight"), w(base + ".conv2_bias"), cout, 3, 1, 1, 1, 0, res, Lc, 0, 1.0f);
        const float inv = 1.0f / sqrtf(2.0f);
        if (has(base + ".conv1x1_weight")) {
            conv1d(x, cin, L, w(base + ".conv1x1_weight"), nullptr, cout, 1, 1, 0, 1, pool ? 1 : 0, res, Lc, 1, inv);
        } else if (res) {
            hipLaunchKernelGGL(kk_add_kernel, kgrid((int64_t) cout * Lc), dim3(256), 0, st, (const float *) res, x, res, (int64_t) cout * Lc, inv);
        }
        C = cout;
        Lout = Lc;
        return res;
    }
    // build_kokoro_generator_res_block (:136-165), in place on x [C][L]
    void gen_res(const std::string &base, float *x, int C, int64_t L, const float *style, int S, const uint32_t *pads, const uint32_t *dils) {
        for (int i = 0; i < 3; i++) {
            const std::string b = base + "." + std::to_string(i) + ".";
            float *cur = s.f((size_t) C * L), *y = s.f((size_t) C * L);
            copy(cur, x, (size_t) C * L);
            adain(cur, C, L, style, S, b + "gamma1_weight", b + "gamma1_bias", b + "beta1_weight", b + "beta1_bias", 2, 0.0f, w(b + "alpha1"));
            int64_t ne[4];
            const float *w1 = w(b + "convs1_weight", ne);
            if (!w1) return;
            conv1d(cur, C, L, w1, w(b + "convs1_bias"), C, (int) ne[0], 1, (int) pads[i], (int) dils[i], 0, y, L, 0, 1.0f);
            adain(y, C, L, style, S, b + "gamma2_weight", b + "gamma2_bias", b + "beta2_weight", b + "beta2_bias", 2, 0.0f, w(b + "alpha2"));
            const float *w2 = w(b + "convs2_weight", ne);
            if (!w2) return;
            conv1d(y, C, L, w2, w(b + "convs2_bias"), C, (int) ne[0], 1, (int) pads[0], 1, 0, x, L, 1, 1.0f);   // x += conv (:160-161)
        }
    }
};

int kokoro_dims(tts_hip_ctx *c, KRun &k, int &D, int &S) {
    int64_t ne[4];
    if (!k.w("duration_predictor.encode", ne)) return -1;
    D = (int) ne[1];
    if (!k.w("duration_predictor.layers.1.gamma_weight", ne)) return -1;
    S = (int) ne[0];
    return 0;
}
const float *kokoro_voice(tts_hip_ctx *c, KRun &k, const char *voice, uint32_t n, int S, bool second_half) {
    int64_t ne[4];
    const float *v = k.w(std::string("voice_tensors.") + (voice ? voice : ""), ne);
    if (!v) return nullptr;
    if ((int) ne[0] != 2 * S || (int64_t) n - 3 >= ne[1]) { k.err = "voice tensor shape does not cover this token count"; return nullptr; }
    return v + (size_t) (n - 3) * 2 * S + (second_half ? S : 0);   // row n_tokens - 3 (:1012, :1149, :1220)
}
}  // namespace

// after a synchronised Kokoro call: did a bounded spin of kk_lstm_split_kernel give up?
static int kokoro_check_stuck(tts_hip_ctx *c, const char *who) {
    int stuck = 0;
    HIPCHK(hipMemcpy(&stuck, c->kk_stuck, 4, hipMemcpyDeviceToHost));
    if (stuck) {
        (void) hipMemset(c->kk_stuck, 0, 4);
        return set_err("%s: the workgroups of a split LSTM recurrence never saw each other's hidden state (device oversubscribed?); "
                       "TTS_HIP_KOKORO_LSTM_SPLIT=0 selects the single-workgroup kernel", who);
    }
    return 0;
}

extern "C" int tts_hip_kokoro_durations(tts_hip_ctx *c, const uint32_t *tokens, uint32_t n, const char *voice, float *lens_out, float *hidden_out) {
    if (!c || !c->has_kokoro) return set_err("tts_hip_kokoro_durations: not a Kokoro context (tts_hip_kokoro_create)");
    if (!c->finalized || !c->weights_present) return set_err("tts_hip_kokoro_durations: context not finalized");
    if (!tokens || !lens_out) return set_err("tts_hip_kokoro_durations: null argument");
    if (n < 3 || n > c->ko.max_ctx) return set_err("tts_hip_kokoro_durations: %u tokens outside 3..%u", n, c->ko.max_ctx);
    HIPCHK(hipSetDevice(c->device));
    KScratch s(c);
    KRun k(c, s);
    int64_t ne[4];
    const float *tok_embd = k.w("albert.token_embd", ne);
    if (!tok_embd) return set_err("tts_hip_kokoro_durations: %s", k.err.c_str());
    const int E = (int) ne[0], vocab = (int) ne[1];
    for (uint32_t i = 0; i < n; i++)
        if (tokens[i] >= (uint32_t) vocab) return set_err("tts_hip_kokoro_durations: token %u >= vocabulary %d", tokens[i], vocab);
    const float *embd = k.w("albert.embd", ne);
    const int H = embd ? (int) ne[1] : 0, NH = (int) c->ko.n_attn_heads, hs = NH ? H / NH : 0;
    const float *ffn_w = k.w("albert.layer.0.ffn", ne);
    const int F = ffn_w ? (int) ne[1] : 0;
    int D = 0, S = 0;
    if (kokoro_dims(c, k, D, S) != 0 || !embd || !ffn_w || NH == 0 || H % NH) return set_err("tts_hip_kokoro_durations: %s", k.err.empty() ? "bad ALBERT shapes" : k.err.c_str());
    const float *style = kokoro_voice(c, k, voice, n, S, true);
    if (!style) return set_err("tts_hip_kokoro_durations: %s", k.err.c_str());
    const int N = (int) n, Wd = D + S;
    uint32_t *d_tok = (uint32_t *) s.f(n);
    float *x0 = s.f((size_t) N * E), *x = s.f((size_t) N * H), *q = s.f((size_t) N * H), *kk = s.f((size_t) N * H), *v = s.f((size_t) N * H), *att = s.f((size_t) N * H);
    float *o = s.f((size_t) N * H), *ff = s.f((size_t) N * F), *cur = s.f((size_t) N * Wd), *ls = s.f((size_t) N * D), *gamma = s.f(D), *beta = s.f(D);
    if (s.failed) return set_err("tts_hip_kokoro_durations: device scratch allocation failed");
    HIPCHK(hipMemcpyAsync(d_tok, tokens, (size_t) n * 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(kk_albert_embed_kernel, kgrid((int64_t) N * E), dim3(256), 0, c->stream, tok_embd, k.w("albert.position_embd"), k.w("albert.token_type_embd"),
                       (const uint32_t *) d_tok, N, E, x0);
    k.norm_rows(x0, E, N, E, 1e-12f, k.w("albert.norm"), k.w("albert.norm_bias"), 0, x0, E);
    k.linear(embd, k.w("albert.embd_bias"), x0, E, N, E, H, x, H);
    const std::string L0 = "albert.layer.0.";
    for (uint32_t r = 0; r < c->ko.n_recurrence; r++) {
        k.linear(k.w(L0 + "q"), k.w(L0 + "q_bias"), x, H, N, H, H, q, H);
        k.linear(k.w(L0 + "k"), k.w(L0 + "k_bias"), x, H, N, H, H, kk, H);
        k.linear(k.w(L0 + "v"), k.w(L0 + "v_bias"), x, H, N, H, H, v, H);
        hipLaunchKernelGGL(kk_albert_attn_kernel, dim3(NH, N), dim3(64), (size_t) N * 4, c->stream, (const float *) q, (const float *) kk, (const float *) v, N, H, hs,
                           c->ko.attn_scale, att);
        k.linear(k.w(L0 + "o"), k.w(L0 + "o_bias"), att, H, N, H, H, o, H);
        hipLaunchKernelGGL(kk_add_kernel, kgrid((int64_t) N * H), dim3(256), 0, c->stream, (const float *) o, (const float *) x, o, (int64_t) N * H, 1.0f);
        k.norm_rows(o, H, N, H, 1e-12f, k.w(L0 + "ffn_norm"), k.w(L0 + "ffn_norm_bias"), 0, x, H);
        k.linear(ffn_w, k.w(L0 + "ffn_bias"), x, H, N, H, F, ff, F);
        hipLaunchKernelGGL(kk_gelu_kernel, kgrid((int64_t) N * F), dim3(256), 0, c->stream, ff, (int64_t) N * F);
        k.linear(k.w(L0 + "ffn_out"), k.w(L0 + "ffn_out_bias"), ff, F, N, F, H, o, H);
        hipLaunchKernelGGL(kk_add_kernel, kgrid((int64_t) N * H), dim3(256), 0, c->stream, (const float *) o, (const float *) x, o, (int64_t) N * H, 1.0f);
        k.norm_rows(o, H, N, H, 1e-12f, k.w(L0 + "attn_norm"), k.w(L0 + "attn_norm_bias"), 0, x, H);
        if (!k.ok()) return set_err("tts_hip_kokoro_durations: %s", k.err.c_str());
    }
    const std::string dp = "duration_predictor.";
    k.linear(k.w(dp + "encode"), k.w(dp + "encode_bias"), x, H, N, H, D, cur, Wd);
    hipLaunchKernelGGL(kk_fill_cols_kernel, kgrid((int64_t) N * S), dim3(256), 0, c->stream, cur, N, Wd, D, style, S);
    for (uint32_t l = 0; l < c->ko.n_dp_layers; l++) {
        const std::string lb = dp + "layers." + std::to_string(2 * l + 1) + ".";
        k.bilstm(dp + "layers." + std::to_string(2 * l) + ".lstm", cur, N, Wd, D / 2, ls);
        k.linear(k.w(lb + "gamma_weight"), k.w(lb + "gamma_bias"), style, S, 1, S, D, gamma, D);
        k.linear(k.w(lb + "beta_weight"), k.w(lb + "beta_bias"), style, S, 1, S, D, beta, D);
        k.norm_rows(ls, D, N, D, 1e-5f, gamma, beta, 1, cur, Wd);   // the style columns of cur stay in place
        if (!k.ok()) return set_err("tts_hip_kokoro_durations: %s", k.err.c_str());
    }
    if (hidden_out) HIPCHK(hipMemcpyAsync(hidden_out, cur, (size_t) N * Wd * 4, hipMemcpyDeviceToHost, c->stream));
    k.bilstm(dp + "duration_lstm", cur, N, Wd, D / 2, ls);
    const float *dpw = k.w(dp + "duration_proj", ne);
    const int ND = dpw ? (int) ne[1] : 0;
    float *dur = s.f((size_t) N * ND), *lens = s.f(n);
    k.linear(dpw, k.w(dp + "duration_proj_bias"), ls, D, N, D, ND, dur, ND);
    if (dur && lens) hipLaunchKernelGGL(kk_duration_kernel, kgrid(N, 64), dim3(64), 0, c->stream, (const float *) dur, N, ND, lens);
    if (!k.ok()) return set_err("tts_hip_kokoro_durations: %s", k.err.c_str());
    HIPCHK(hipMemcpyAsync(lens_out, lens, (size_t) n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return kokoro_check_stuck(c, "tts_hip_kokoro_durations");
}

extern "C" int tts_hip_kokoro_generate(tts_hip_ctx *c, const uint32_t *tokens, uint32_t n, const float *lens, const float *hidden, const char *voice, const float *noise,
                                       float *pcm_out, float *hsrc_out, const float *hsrc_in) {
    if (!c || !c->has_kokoro) return set_err("tts_hip_kokoro_generate: not a Kokoro context (tts_hip_kokoro_create)");
    if (!c->finalized || !c->weights_present) return set_err("tts_hip_kokoro_generate: context not finalized");
    if (!tokens || !lens || !hidden || !noise || !pcm_out) return set_err("tts_hip_kokoro_generate: null argument");
    if (n < 3 || n > c->ko.max_ctx) return set_err("tts_hip_kokoro_generate: %u tokens outside 3..%u", n, c->ko.max_ctx);
    HIPCHK(hipSetDevice(c->device));
    const tts_hip_kokoro_desc &kd = c->ko;
    KScratch s(c);
    KRun k(c, s);
    int D = 0, S = 0;
    if (kokoro_dims(c, k, D, S) != 0) return set_err("tts_hip_kokoro_generate: %s", k.err.c_str());
    const int N = (int) n, Wd = D + S;
    std::vector<int> tok_of;
    for (int i = 0; i < N; i++) {
        if (!(lens[i] >= 1.0f) || lens[i] > 50.0f || lens[i] != floorf(lens[i])) return set_err("tts_hip_kokoro_generate: length %g of token %d is not a whole number in 1..50", lens[i], i);
        for (int r = 0; r < (int) lens[i]; r++) tok_of.push_back(i);
    }
    const int64_t T = (int64_t) tok_of.size();
    const float *style_p = kokoro_voice(c, k, voice, n, S, true), *style_d = kokoro_voice(c, k, voice, n, S, false);
    if (!style_p || !style_d) return set_err("tts_hip_kokoro_generate: %s", k.err.c_str());
    int64_t ne[4];
    const float *te = k.w("text_encoder.embedding_weight", ne);
    if (!te) return set_err("tts_hip_kokoro_generate: %s", k.err.c_str());
    const int C = (int) ne[0], vocab = (int) ne[1];
    for (uint32_t i = 0; i < n; i++)
        if (tokens[i] >= (uint32_t) vocab) return set_err("tts_hip_kokoro_generate: token %u >= vocabulary %d", tokens[i], vocab);
    const int NHm = (int) kd.harmonic_num + 1, up = (int) kd.upsample_scale;
    const int64_t L2 = 2 * T, LS = L2 * up, out_len = T * kd.up_sampling_factor;
    const int Nf = (int) kd.n_fft, hop = (int) kd.hop, nbins = Nf / 2 + 1;
    const int64_t F = LS / hop + 1;

    uint32_t *d_tok = (uint32_t *) s.f(n);
    int *d_idx = (int *) s.f((size_t) T);
    float *d_hidden = s.f((size_t) N * Wd), *d_noise = s.f((size_t) NHm * LS);
    if (s.failed) return set_err("tts_hip_kokoro_generate: device scratch allocation failed");
    HIPCHK(hipMemcpyAsync(d_tok, tokens, (size_t) n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_idx, tok_of.data(), (size_t) T * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_hidden, hidden, (size_t) N * Wd * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_noise, noise, (size_t) NHm * LS * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));   // tok_of is a local

    // alignment + shared LSTM (:1157-1166)
    float *en = s.f((size_t) T * Wd), *sh = s.f((size_t) T * D), *shc = s.f((size_t) D * T);
    if (s.failed) return set_err("tts_hip_kokoro_generate: device scratch allocation failed");
    hipLaunchKernelGGL(kk_gather_rows_kernel, kgrid(T * Wd), dim3(256), 0, c->stream, (const float *) d_hidden, (const int *) d_idx, (int) T, Wd, en);
    k.bilstm("duration_predictor.shared_lstm", en, (int) T, Wd, D / 2, sh);
    hipLaunchKernelGGL(kk_transpose_kernel, kgrid(T * D), dim3(256), 0, c->stream, (const float *) sh, (int) T, D, shc);
    // F0 / N branches (:1169-1192)
    float *curves[2] = {nullptr, nullptr};
    const char *branch[2] = {"f0", "n"};
    for (int b = 0; b < 2; b++) {
        float *cur = shc;
        int Cb = D;
        int64_t L = T;
        for (uint32_t i = 0; i < kd.f0_n_blocks; i++) {
            cur = k.ada_block(std::string("duration_predictor.") + branch[b] + "_blocks." + std::to_string(i), cur, L, style_p, S, Cb, L);
            if (!cur || !k.ok()) return set_err("tts_hip_kokoro_generate: %s", k.err.c_str());
        }
        if (L != L2) return set_err("tts_hip_kokoro_generate: the %s branch does not double the frame count", branch[b]);
        curves[b] = s.f((size_t) L2);
        k.conv1d(cur, Cb, L, k.w(std::string("duration_predictor.") + branch[b] + "_proj_kernel"), k.w(std::string("duration_predictor.") + branch[b] + "_proj_bias"), 1, 1, 1,
                 0, 1, 0, curves[b], L2, 0, 1.0f);
    }
    // text encoder (:1196-1210)
    float *tx = s.f((size_t) C * N);
    hipLaunchKernelGGL(kk_embed_cols_kernel, kgrid((int64_t) N * C), dim3(256), 0, c->stream, te, (const uint32_t *) d_tok, N, C, tx);
    for (uint32_t l = 0; l < kd.n_conv_layers; l++) {
        const std::string lb = "text_encoder.layers." + std::to_string(l) + ".";
        const float *cw = k.w(lb + "weight", ne);
        if (!cw) return set_err("tts_hip_kokoro_generate: %s", k.err.c_str());
        float *y = s.f((size_t) C * N);
        k.conv1d(tx, C, N, cw, k.w(lb + "bias"), C, (int) ne[0], 1, 2, 1, 0, y, N, 0, 1.0f);
        if (y) hipLaunchKernelGGL(kk_chan_norm_kernel, dim3(N), dim3(64), 0, c->stream, y, C, (int64_t) N, k.w(lb + "gamma"), k.w(lb + "beta"), 0.2f);
        tx = y;
    }
    float *txr = s.f((size_t) N * C), *tl = s.f((size_t) N * C), *asr = s.f((size_t) C * T);
    if (s.failed) return set_err("tts_hip_kokoro_generate: device scratch allocation failed");
    hipLaunchKernelGGL(kk_transpose_kernel, kgrid((int64_t) C * N), dim3(256), 0, c->stream, (const float *) tx, C, N, txr);
    k.bilstm("text_encoder.lstm", txr, N, C, C / 2, tl);
    hipLaunchKernelGGL(kk_gather_cols_kernel, kgrid((int64_t) C * T), dim3(256), 0, c->stream, (const float *) tl, (const int *) d_idx, (int) T, C, asr);
    if (!k.ok()) return set_err("tts_hip_kokoro_generate: %s", k.err.c_str());
    // decoder (:1222-1241)
    float *f0d = s.f((size_t) T), *nd = s.f((size_t) T);
    k.conv1d(curves[0], 1, L2, k.w("decoder.f0_conv_weight"), k.w("decoder.f0_conv_bias"), 1, 3, 2, 1, 1, 0, f0d, T, 0, 1.0f);
    k.conv1d(curves[1], 1, L2, k.w("decoder.n_conv_weight"), k.w("decoder.n_conv_bias"), 1, 3, 2, 1, 1, 0, nd, T, 0, 1.0f);
    int Cc = C + 2;
    float *cat0 = s.f((size_t) Cc * T);
    k.copy(cat0, asr, (size_t) C * T);
    k.copy(cat0 + (size_t) C * T, f0d, (size_t) T);
    k.copy(cat0 + (size_t) (C + 1) * T, nd, (size_t) T);
    int64_t Lc = T;
    float *cur = k.ada_block("decoder.encoder_block", cat0, T, style_d, S, Cc, Lc);
    const float *aw = k.w("decoder.asr_conv_weight", ne);
    if (!cur || !aw) return set_err("tts_hip_kokoro_generate: %s", k.err.c_str());
    const int CA = (int) ne[2];
    float *asr_res = s.f((size_t) CA * T);
    k.conv1d(asr, C, T, aw, k.w("decoder.asr_conv_bias"), CA, 1, 1, 0, 1, 0, asr_res, T, 0, 1.0f);
    for (uint32_t i = 0; i < kd.n_decoder_blocks; i++) {
        int Cin = Cc + CA + 2;
        float *cat = s.f((size_t) Cin * T);
        k.copy(cat, cur, (size_t) Cc * T);
        k.copy(cat + (size_t) Cc * T, asr_res, (size_t) CA * T);
        k.copy(cat + (size_t) (Cc + CA) * T, f0d, (size_t) T);
        k.copy(cat + (size_t) (Cc + CA + 1) * T, nd, (size_t) T);
        cur = k.ada_block("decoder.decoder_blocks." + std::to_string(i), cat, T, style_d, S, Cin, Lc);
        if (!cur || !k.ok()) return set_err("tts_hip_kokoro_generate: %s", k.err.c_str());
        Cc = Cin;
    }
    if (Lc != L2) return set_err("tts_hip_kokoro_generate: the decoder does not end at twice the frame count");
    // harmonic source + STFT conditioning (:173-206)
    float *phase = s.f((size_t) NHm * L2), *sine = s.f((size_t) NHm * LS), *har = s.f((size_t) LS), *win = s.f(Nf), *hs = s.f((size_t) 2 * nbins * F);
    if (s.failed) return set_err("tts_hip_kokoro_generate: device scratch allocation failed");
    {
        std::vector<float> hw((size_t) Nf);
        for (int i = 0; i < Nf; i++) hw[(size_t) i] = (float) pow(sin(M_PI * (double) i / (double) Nf), 2.0);   // hann_window, util.cpp:134-139
        HIPCHK(hipMemcpyAsync(win, hw.data(), (size_t) Nf * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    hipLaunchKernelGGL(kk_sine_phase_kernel, kgrid(NHm, 64), dim3(64), 0, c->stream, (const float *) curves[0], L2, NHm, kd.sample_rate, kd.upsample_scale * 2.0f * (float) M_PI, phase);
    hipLaunchKernelGGL(kk_sine_source_kernel, kgrid((int64_t) NHm * LS), dim3(256), 0, c->stream, (const float *) phase, (const float *) curves[0], L2, NHm, up, kd.voice_threshold,
                       kd.sin_amp, kd.noise_std, (const float *) d_noise, sine);
    hipLaunchKernelGGL(kk_source_merge_kernel, kgrid(LS), dim3(256), 0, c->stream, (const float *) sine, NHm, LS, k.w("decoder.generator.m_source_weight"),
                       k.w("decoder.generator.m_source_bias"), har);
    hipLaunchKernelGGL(kk_stft_kernel, kgrid((int64_t) nbins * F), dim3(256), 0, c->stream, (const float *) har, LS, (const float *) win, Nf, hop, F, hs);
    if (!k.ok()) return set_err("tts_hip_kokoro_generate: %s", k.err.c_str());
    if (hsrc_out) HIPCHK(hipMemcpyAsync(hsrc_out, hs, (size_t) 2 * nbins * F * 4, hipMemcpyDeviceToHost, c->stream));
    if (hsrc_in) {
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipMemcpyAsync(hs, hsrc_in, (size_t) 2 * nbins * F * 4, hipMemcpyHostToDevice, c->stream));
    }
    // generator (:208-241)
    float *g = cur;
    int Cg = Cc;
    int64_t Lg = Lc;
    const std::string gb = "decoder.generator.";
    for (uint32_t i = 0; i < kd.n_upsamples; i++) {
        hipLaunchKernelGGL(kk_leaky_kernel, kgrid((int64_t) Cg * Lg), dim3(256), 0, c->stream, g, (int64_t) Cg * Lg, 0.1f);
        const float *uw = k.w(gb + "ups." + std::to_string(i) + ".weight", ne);
        if (!uw) return set_err("tts_hip_kokoro_generate: %s", k.err.c_str());
        const int K = (int) ne[0], Co = (int) ne[1];
        const int64_t Lo = (Lg - 1) * kd.up_stride[i] - 2 * (int64_t) kd.up_padding[i] + K;
        float *y = s.f((size_t) Co * Lo);
        const int S_ = (int) kd.up_stride[i];
        const bool mfma_up = c->kk_mfma && K == 2 * S_ && (S_ == 10 || S_ == 6) && Co % 64 == 0 && 2 * (int) kd.up_padding[i] == K - S_ && Lo == Lg * S_;
        if (y && mfma_up) {
            // the generator's ConvTranspose1d (stride 10 / 6, kernel = 2 x stride: every output touches two taps) on the codec's
            // phase-decomposed MFMA kernel; the one-thread-per-output kernel spent 18 ms per launch here
            const size_t w_off = (size_t) ((const char *) uw - c->arena);
            if (c->packed.find(w_off) == c->packed.end()) CHK(pack_one(c, w_off, Co, Cg, K, 64, CI32_T, true));
            ConvTArgs ta{};
            ta.x = g; ta.w = c->packed[w_off]; ta.b = k.w(gb + "ups." + std::to_string(i) + ".bias"); ta.alpha = nullptr; ta.y = y;
            ta.cin = Cg; ta.cout = Co; ta.L = (int) Lg; ta.Lout = (int) Lo; ta.stride = S_; ta.pad = (int) kd.up_padding[i]; ta.frames = nullptr; ta.mult = 1;
            if (S_ == 10) CHK((launch_convt_mfma<10, 1, 2, 2, CI32_T>(c, ta, 1)));
            else CHK((launch_convt_mfma<6, 1, 2, 2, CI32_T>(c, ta, 1)));
        } else
        if (y) hipLaunchKernelGGL(kk_convt1d_kernel, kgrid((int64_t) Co * Lo), dim3(256), 0, c->stream, (const float *) g, Cg, Lg, uw, k.w(gb + "ups." + std::to_string(i) + ".bias"), Co, K,
                                  (int) kd.up_stride[i], (int) kd.up_padding[i], y, Lo);
        g = y; Cg = Co; Lg = Lo;
        if (i == kd.n_upsamples - 1) {
            float *p = s.f((size_t) Cg * (Lg + 1));
            if (p) hipLaunchKernelGGL(kk_pad_front_kernel, kgrid((int64_t) Cg * (Lg + 1)), dim3(256), 0, c->stream, (const float *) g, Cg, Lg, p);
            g = p; Lg += 1;
        }
        const std::string nbk = gb + "noise_blocks." + std::to_string(i) + ".";
        const float *nw = k.w(nbk + "conv_weight", ne);
        if (!nw) return set_err("tts_hip_kokoro_generate: %s", k.err.c_str());
        const int NK = (int) ne[0];
        const int64_t Ls = (F + 2 * (int64_t) kd.noise_padding[i] - (NK - 1) - 1) / kd.noise_stride[i] + 1;
        if (Ls != Lg) return set_err("tts_hip_kokoro_generate: source length %lld != %lld at stage %u", (long long) Ls, (long long) Lg, i);
        float *xs = s.f((size_t) Cg * Lg);
        k.conv1d(hs, 2 * nbins, F, nw, k.w(nbk + "conv_bias"), Cg, NK, (int) kd.noise_stride[i], (int) kd.noise_padding[i], 1, 0, xs, Lg, 0, 1.0f);
        k.gen_res(nbk + "resblock", xs, Cg, Lg, style_d, S, kd.noise_res_padding[i], kd.noise_res_dilation[i]);
        if (g && xs) hipLaunchKernelGGL(kk_add_kernel, kgrid((int64_t) Cg * Lg), dim3(256), 0, c->stream, (const float *) g, (const float *) xs, g, (int64_t) Cg * Lg, 1.0f);
        float *sum = s.f((size_t) Cg * Lg), *br = s.f((size_t) Cg * Lg);
        for (uint32_t ii = 0; ii < kd.n_kernels; ii++) {
            float *dst = ii == 0 ? sum : br;
            k.copy(dst, g, (size_t) Cg * Lg);
            const uint32_t ri = i * kd.n_kernels + ii;
            k.gen_res(gb + "resblocks." + std::to_string(ri), dst, Cg, Lg, style_d, S, kd.res_padding[ri], kd.res_dilation[ri]);
            if (ii > 0 && sum && br)
                hipLaunchKernelGGL(kk_add_kernel, kgrid((int64_t) Cg * Lg), dim3(256), 0, c->stream, (const float *) sum, (const float *) br, sum, (int64_t) Cg * Lg,
                                   ii == kd.n_kernels - 1 ? 1.0f / (float) kd.n_kernels : 1.0f);
        }
        g = sum;
        if (!k.ok()) return set_err("tts_hip_kokoro_generate: %s", k.err.c_str());
    }
    hipLaunchKernelGGL(kk_leaky_kernel, kgrid((int64_t) Cg * Lg), dim3(256), 0, c->stream, g, (int64_t) Cg * Lg, 0.01f);
    const float *pw = k.w(gb + "conv_post_weight", ne);
    if (!pw) return set_err("tts_hip_kokoro_generate: %s", k.err.c_str());
    if (Lg != F) return set_err("tts_hip_kokoro_generate: generator length %lld != STFT frames %lld", (long long) Lg, (long long) F);
    float *post = s.f((size_t) 2 * nbins * Lg), *pcm = s.f((size_t) out_len);
    k.conv1d(g, Cg, Lg, pw, k.w(gb + "conv_post_bias"), 2 * nbins, (int) ne[0], 1, (int) kd.out_conv_padding, 1, 0, post, Lg, 0, 1.0f);
    if (post && pcm) {
        hipLaunchKernelGGL(kk_spec_phase_kernel, kgrid((int64_t) 2 * nbins * Lg), dim3(256), 0, c->stream, post, nbins, Lg);
        hipLaunchKernelGGL(kk_istft_kernel, kgrid(out_len), dim3(256), 0, c->stream, (const float *) post, Lg, (const float *) win, Nf, hop, pcm, out_len);
    }
    if (!k.ok()) return set_err("tts_hip_kokoro_generate: %s", k.err.c_str());
    HIPCHK(hipMemcpyAsync(pcm_out, pcm, (size_t) out_len * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return kokoro_check_stuck(c, "tts_hip_kokoro_generate");
}

// ------------------------------------------------------------------------------------------------
// SNAC codec (src/decoder/snac_model.cpp:110-208)
// ------------------------------------------------------------------------------------------------
extern "C" tts_hip_ctx *tts_hip_snac_create(int device, const tts_hip_snac_desc *sd) {
    if (!sd || sd->struct_size != sizeof(tts_hip_snac_desc)) { set_err("tts_hip_snac_create: bad desc (struct_size mismatch)"); return nullptr; }
    if (sd->n_blocks == 0 || sd->n_blocks > TTS_HIP_MAX_DAC_BLOCKS || sd->n_codebooks == 0 || sd->n_codebooks > 4) { set_err("tts_hip_snac_create: n_blocks / n_codebooks out of range"); return nullptr; }
    tts_hip_desc d{};
    d.struct_size = sizeof(d);
    d.max_seqs = 1;
    d.flags = (sd->flags & TTS_HIP_FLAG_VALU_GEMM) | TTS_HIP_FLAG_NO_PARLER | TTS_HIP_FLAG_NO_DAC;
    tts_hip_ctx *c = tts_hip_create(device, &d);
    if (!c) return nullptr;
    c->has_snac = true;
    c->snac = *sd;
    for (uint32_t i = 0; i < sd->n_codebooks; i++)
        if (c->snac.repeats[i] == 0) c->snac.repeats[i] = 1;
    return c;
}

static int ensure_packed_snac(tts_hip_ctx *c) {
    if (c->snac_packed || (c->d.flags & TTS_HIP_FLAG_VALU_GEMM)) return 0;
    int CO_T = 0, CI_T = 0;
    if (conv_tile(c->s_c0, 1, &CO_T, &CI_T) >= 0) CHK(pack_one(c, c->s_upw, c->s_c0, c->s_latent, 1, CO_T, CI_T, false));
    for (auto &b : c->sblocks) {
        if (convt_tile(b.cout, b.stride, &CO_T) >= 0) CHK(pack_one(c, b.w, b.cout, b.cin, 2 * b.stride, CO_T, CI32_T, true));
        if (conv_tile(b.cout, 1, &CO_T, &CI_T) >= 0) {
            CHK(pack_one(c, b.noise_w, b.cout, b.cout, 1, CO_T, CI_T, false));
            for (int r = 0; r < 3; r++) CHK(pack_one(c, b.res[r].out_w, b.cout, b.cout, 1, CO_T, CI_T, false));
        }
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    c->snac_packed = true;
    return 0;
}

extern "C" int tts_hip_snac_decode(tts_hip_ctx *c, const uint32_t *codes, uint32_t T_, const float *noise, float *pcm_out) {
    if (!c || !c->has_snac) return set_err("tts_hip_snac_decode: not a SNAC context (tts_hip_snac_create)");
    if (!c->finalized || !c->weights_present) return set_err("tts_hip_snac_decode: context not finalized");
    if (!codes || !pcm_out) return set_err("tts_hip_snac_decode: null argument");
    if (T_ == 0) return 0;
    const tts_hip_snac_desc &sd = c->snac;
    if (sd.max_frames && T_ > sd.max_frames) return set_err("tts_hip_snac_decode: %u tokens exceed snac.max_generation_size %u", T_, sd.max_frames);
    size_t n_codes = 0;
    for (uint32_t i = 0; i < sd.n_codebooks; i++) {
        if (T_ % sd.repeats[i]) return set_err("tts_hip_snac_decode: T=%u is not a multiple of the level-%u repeat %u", T_, i, sd.repeats[i]);
        n_codes += T_ / sd.repeats[i];
    }
    for (size_t i = 0; i < n_codes; i++)
        if (codes[i] >= (uint32_t) c->s_cbsize) return set_err("tts_hip_snac_decode: code %u >= codebook size %d", codes[i], c->s_cbsize);
    HIPCHK(hipSetDevice(c->device));
    CHK(ensure_packed_snac(c));
    const int T = (int) T_;
    // buffers sized for max_frames (or this call): largest activation = max over stages of C * L
    const size_t Tcap = std::max<size_t>(sd.max_frames, T_);
    if (!c->sbuf[0] || Tcap > c->dac_cap_frames) {
        size_t mx = (size_t) std::max(c->s_latent, c->s_c0), up = 1, noise_len = 0;
        for (auto &b : c->sblocks) { mx = std::max(mx, (size_t) b.cin * up); up *= b.stride; mx = std::max(mx, (size_t) b.cout * up); noise_len += up; }
        HIPCHK(hipStreamSynchronize(c->stream));
        for (int i = 0; i < 3; i++) { free_dev(c->sbuf[i]); c->sbuf[i] = nullptr; HIPCHK(hipMalloc((void **) &c->sbuf[i], mx * Tcap * 4)); }
        free_dev(c->s_noise); c->s_noise = nullptr;
        HIPCHK(hipMalloc((void **) &c->s_noise, noise_len * Tcap * 4));
        free_dev(c->s_codes); c->s_codes = nullptr;
        HIPCHK(hipMalloc((void **) &c->s_codes, Tcap * sd.n_codebooks * 4));
        c->dac_cap_frames = Tcap;
    }
    auto f32 = [&](size_t off) { return (const float *) (c->arena + off); };
    HIPCHK(hipMemcpyAsync(c->s_codes, codes, n_codes * 4, hipMemcpyHostToDevice, c->stream));
    size_t noise_len = 0;
    { size_t up = 1; for (auto &b : c->sblocks) { up *= b.stride; noise_len += up * (size_t) T; } }
    if (noise) HIPCHK(hipMemcpyAsync(c->s_noise, noise, noise_len * 4, hipMemcpyHostToDevice, c->stream));

    float *cur = c->sbuf[0], *t1 = c->sbuf[1], *t2 = c->sbuf[2];
    int L = T;
    SnacEmbedArgs ea{};
    ea.codes = c->s_codes; ea.codebook = f32(c->s_codebook); ea.proj_w = f32(c->s_projw); ea.proj_b = f32(c->s_projb);
    ea.n_cb = (int) sd.n_codebooks; ea.cb_size = c->s_cbsize; ea.cb_dim = c->s_cbdim; ea.latent = c->s_latent; ea.T = T; ea.out = cur;
    for (uint32_t i = 0; i < 4; i++) ea.rep[i] = i < sd.n_codebooks ? (int) sd.repeats[i] : 1;
    hipLaunchKernelGGL(snac_embed_kernel, dim3((T + 63) / 64, c->s_latent), dim3(64), 0, c->stream, ea);
    HIPCHK(hipGetLastError());
    auto dw = [&](const float *x, size_t w, size_t b, const float *ain, const float *aout, float *y, int C, int Ln, int pad, int dil) {
        hipLaunchKernelGGL(dwconv7_kernel, dim3((Ln + 255) / 256, C), dim3(256), 0, c->stream, x, f32(w), f32(b), ain, aout, y, C, Ln, pad, dil);
        return hipGetLastError() == hipSuccess ? 0 : set_err("dwconv7_kernel launch failed");
    };
    DacBatch bt;
    bt.n = 1; bt.frames = nullptr; bt.mult = 1; bt.tot_frames = (double) T;
    CHK(dw(cur, c->s_inw, c->s_inb, nullptr, nullptr, t1, c->s_latent, L, 3, 1));                                 // :141-142
    CHK(launch_conv(c, bt, t1, c->s_latent, L, c->s_upw, c->s_upb, 0, false, c->s_c0, 1, 0, 1, nullptr, false, cur));   // :143-144
    int C = c->s_c0;
    size_t noise_off = 0;
    for (auto &b : c->sblocks) {                                                                                  // build_layer, gnac.cpp:151-164
        ConvTArgs ta{};
        ta.x = cur; ta.w = f32(b.w); ta.b = f32(b.b); ta.alpha = f32(b.alpha); ta.y = t1; ta.cin = b.cin; ta.cout = b.cout; ta.L = L;
        ta.Lout = (L - 1) * b.stride - 2 * b.padding + 2 * b.stride; ta.stride = b.stride; ta.pad = b.padding;
        ta.frames = nullptr; ta.mult = 1;
        CHK(prof_begin(c, TTS_HIP_K_DAC_CONVT, 0, 2.0 * b.cin * (double) b.cout * 2 * ta.Lout));
        CHK(launch_convt(c, ta, b.w, 1));
        CHK(prof_end(c));
        std::swap(cur, t1);
        L = ta.Lout; C = b.cout;
        bt.tot_frames = (double) L;   // launch_conv's accounting: valid positions = tot_frames * mult
        if (noise) {                                                                                              // gnac.cpp:155-159
            CHK(launch_conv(c, bt, cur, C, L, b.noise_w, 0, 0, false, C, 1, 0, 1, nullptr, false, t1, 0, false, false));
            hipLaunchKernelGGL(noise_fma_kernel, dim3((unsigned) (((size_t) C * L + 255) / 256)), dim3(256), 0, c->stream, cur, (const float *) t1,
                               (const float *) (c->s_noise + noise_off), C, L);
            HIPCHK(hipGetLastError());
        }
        noise_off += (size_t) L;
        for (int r = 0; r < 3; r++) {                                                                             // build_residual_unit, groups > 1
            int dil = 1;
            for (int e = 0; e < r; e++) dil *= 3;
            // snake(in_alpha) on the way in, depthwise k7, bias, and the pointwise conv's snake(out_alpha) on the way out
            CHK(dw(cur, b.res[r].in_w, b.res[r].in_b, f32(b.res[r].in_alpha), f32(b.res[r].out_alpha), t1, C, L, 3 * dil, dil));
            CHK(launch_conv(c, bt, t1, C, L, b.res[r].out_w, b.res[r].out_b, 0, false, C, 1, 0, 1, cur, false, t2));
            std::swap(cur, t2);
        }
    }
    CHK(launch_conv(c, bt, cur, C, L, c->s_fw, c->s_fb, c->s_falpha, true, 1, 7, 3, 1, nullptr, true, t1));        // :152-155
    HIPCHK(hipMemcpyAsync(pcm_out, t1, (size_t) L * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// introspection
// ------------------------------------------------------------------------------------------------
extern "C" int tts_hip_set_debug(tts_hip_ctx *c, int on) {
    if (!c) return set_err("null ctx");
    c->debug = on != 0;
    return 0;
}

__global__ void half_to_float_kernel(const _Float16 *in, float *out, size_t n) {
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float) in[i];
}

extern "C" int64_t tts_hip_debug_read(tts_hip_ctx *c, const char *what, float *out, size_t max_floats) {
    if (!c || !what || !out || !c->finalized) { set_err("tts_hip_debug_read: bad argument"); return -1; }
    if (hipSetDevice(c->device) != hipSuccess) { set_err("hipSetDevice failed"); return -1; }
    if (hipStreamSynchronize(c->stream) != hipSuccess) { set_err("sync failed"); return -1; }
    std::string w(what);
    if (w == "hidden") {
        const size_t R = c->host_pos.size();
        if (R == 0 || R * c->H > max_floats) { set_err("debug_read(hidden): no forward yet or buffer too small"); return -1; }
        launch_ln_rows(c, 4, c->x, c->H, (const float *) (c->arena + c->ln_w), (const float *) (c->arena + c->ln_b), c->dbg, (_Float16 *) nullptr, (int) R,
                       c->pending_parts ? (const float *) c->partials : (const float *) nullptr, c->pending_parts, (int64_t) c->RMAX * c->H);
        c->pending_parts = 0;
        if (hipMemcpyAsync(out, c->dbg, R * c->H * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess) { set_err("debug_read(hidden): copy failed"); return -1; }
        return (int64_t) (R * c->H);
    }
    if (w == "x") {
        const size_t R = c->host_pos.size();
        if (R == 0 || R * c->H > max_floats) { set_err("debug_read(x): no forward yet or buffer too small"); return -1; }
        if (c->pending_parts) {   // the last fc2 left K-slice slabs: fold them into x (the LayerNorm output goes to scratch)
            launch_ln_rows(c, 4, c->x, c->H, (const float *) (c->arena + c->ln_w), (const float *) (c->arena + c->ln_b), c->dbg, (_Float16 *) nullptr, (int) R,
                           (const float *) c->partials, c->pending_parts, (int64_t) c->RMAX * c->H);
            c->pending_parts = 0;
            if (hipStreamSynchronize(c->stream) != hipSuccess) { set_err("debug_read(x): fold failed"); return -1; }
        }
        if (hipMemcpy(out, c->x, R * c->H * 4, hipMemcpyDeviceToHost) != hipSuccess) { set_err("copy failed"); return -1; }
        return (int64_t) (R * c->H);
    }
    if (w.size() > 2 && (w[0] == 'k' || w[0] == 'v') && w[1] == ':') {
        int layer = 0, seq = 0;
        if (sscanf(w.c_str() + 2, "%d:%d", &layer, &seq) != 2 || layer < 0 || layer >= c->L || seq < 0 || seq >= (int) c->d.max_seqs) {
            set_err("debug_read(%s): bad layer/seq", what);
            return -1;
        }
        const size_t n = std::min(max_floats / c->H, (size_t) c->KVPOS) * c->H;
        const size_t kv_esz = c->d.kv_type == TTS_HIP_F16 ? 2 : 4;
        const char *base = (const char *) (w[0] == 'k' ? c->kcache : c->vcache) +
                           ((size_t) layer * c->d.max_seqs + seq) * (size_t) c->KVPOS * c->H * kv_esz;
        if (kv_esz == 4) {
            if (hipMemcpy(out, base, n * 4, hipMemcpyDeviceToHost) != hipSuccess) { set_err("copy failed"); return -1; }
        } else {
            float *tmp = nullptr;
            if (hipMalloc((void **) &tmp, n * 4) != hipSuccess) { set_err("alloc failed"); return -1; }
            hipLaunchKernelGGL(half_to_float_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, c->stream, (const _Float16 *) base, tmp, n);
            (void) hipStreamSynchronize(c->stream);
            const hipError_t e = hipMemcpy(out, tmp, n * 4, hipMemcpyDeviceToHost);
            (void) hipFree(tmp);
            if (e != hipSuccess) { set_err("copy failed"); return -1; }
        }
        return (int64_t) n;
    }
    if (w == "stamps") {  // TTS_HIP_B1_STAMPS=1: [launch][16] int64 s_memrealtime stamps of the last <= 4-row forward, two floats per stamp
        if (!c->b1_stamps) { set_err("debug_read(stamps): set TTS_HIP_B1_STAMPS=1 before tts_hip_finalize"); return -1; }
        const size_t n = (size_t) 16 * c->b1_stamp_slot * 2;
        if (n > max_floats) { set_err("buffer too small"); return -1; }
        if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(out, c->b1_stamps, n * 4, hipMemcpyDeviceToHost) != hipSuccess) { set_err("copy failed"); return -1; }
        return (int64_t) n;
    }
    if (starts_with(w, "cross:")) {  // cross:<layer>:<0|1>  -> [E][H]
        int layer = 0, kv = 0;
        if (sscanf(w.c_str() + 6, "%d:%d", &layer, &kv) != 2 || layer < 0 || layer >= c->L || kv < 0 || kv > 1) { set_err("bad cross spec"); return -1; }
        const size_t n = (size_t) c->E * c->H;
        if (n > max_floats) { set_err("buffer too small"); return -1; }
        if (hipMemcpy(out, c->cross_kv_ptr() + ((size_t) layer * 2 + kv) * c->ECAP * c->H * 4, n * 4, hipMemcpyDeviceToHost) != hipSuccess) { set_err("copy failed"); return -1; }
        return (int64_t) n;
    }
    if (starts_with(w, "dac:")) {
        const int stage = atoi(w.c_str() + 4);
        auto it = c->dac_dbg.find(stage);
        if (it == c->dac_dbg.end()) { set_err("debug_read(%s): no snapshot (enable tts_hip_set_debug before decode)", what); return -1; }
        if (it->second.size() > max_floats) { set_err("buffer too small"); return -1; }
        memcpy(out, it->second.data(), it->second.size() * 4);
        return (int64_t) it->second.size();
    }
    set_err("tts_hip_debug_read: unknown item '%s'", what);
    return -1;
}

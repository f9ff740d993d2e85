// shim_internal.h — what the translation units of libtts_hip.so share: error plumbing, the arena / tensor records, the device
// context (tts_hip_ctx) and the few host functions one unit calls in another.  Kernel headers are included by the unit that launches them.
#pragma once
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

#include <unordered_map>
#define LLAMA_GREEDY_CHUNK 8

int set_err(const char *fmt, ...);   // shim_core.hip: thread-local message behind tts_hip_last_error()
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
    size_t stoff = 0;               // TTS_HIP_Q8I matrices of the Parler decoder: the block scales once more, transposed (float [K/32][ldw]) for qgemm_tile_kernel
    int ldw = 0;
};

// where a tiled integer GEMM's epilogue leaves its result rows as Q8_0 blocks for the next quantised matrix (set by the caller right before run_gemm)
struct QTileOut { int8_t *q = nullptr; float *dT = nullptr; int ldq = 0; const void *stands_for = nullptr; };   // stands_for: the fp32 buffer the consumer names as its input

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
extern DacBuffers g_dac_buffers[64];
extern std::mutex g_dac_pass_mutex[64];

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
    bool arena_counted = false;   // the external arena is another context's allocation and this context holds a reference (shim_core.hip: g_arena_refs)
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
    bool q4_rms = true;         // tune("q4_rms")=0: the rms norms in front of the q/k/v and gate|up projections keep their own launches
    bool q4_silu = true;        // tune("q4_silu")=0: gate|up, silu * up and the down projection stay three launches
    bool q4_rope = true;        // tune("q4_rope")=0: the Llama q/k/v projection keeps its separate rope + cache-append launch
    bool q4_lds = true;         // tune("q4_lds")=0: Q4_0 row products stay on gemv_q4_rows_kernel (one feature per wave, activations from L2)
    int q_stream = 31;          // tune("q_stream"): which quantised projections of a 5 .. 64 row Llama step take qgemv_stream_kernel instead of qgemm16_kernel
                                // (bits: 1 qkv, 2 o, 4 gate|up, 8 down, 16 head; 0 none, 1 = all)
    bool gemv_stream = true;    // tune("gemv_stream")=0: <= 16-row F16 GEMMs of the Dia step stay on gemm16_kernel (gemv_stream_kernels.h otherwise)
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
    uint32_t *l_seq = nullptr, *l_btok = nullptr, *l_bpi = nullptr, *l_bsmp = nullptr;   // lock-step utterances: row -> cache slot, selected tokens [rows], arg-max partials, sampler state [utterance][3]
    float *l_bpv = nullptr;
    int l_pending = 0;
    int64_t l_pstride = 0;   // floats between the pending slabs (0: RMAX * H)
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
    struct { const void *uni = nullptr, *pen = nullptr; uint32_t k = 0; float temp = 0, top_p = 1.0f; } l_smp_baked;   // what the captured sampled step holds
    double *d_pen = nullptr;      // pow(repetition_penalty, count) table (host-evaluated)
    int pen_len = 0;
    int32_t *d_last = nullptr;    // [RMAX][n_out] sampler::last_token_ids
    uint32_t *d_repc = nullptr;   // [RMAX][n_out] sampler::repetition_counts
    size_t uniforms_cap = 0;
    uint32_t g_bos = 0xFFFFFFFFu, g_eos = 0xFFFFFFFFu;  // ids baked into the captured feed kernel
    int pending_parts = 0;      // slabs waiting to be folded into x by the next LayerNorm launch
    uint32_t *d_ids = nullptr, *d_pos = nullptr, *d_seq = nullptr, *d_tok = nullptr, *d_step = nullptr, *d_steps_done = nullptr;
    uint32_t *d_gather = nullptr;   // scratch of the row compaction: map [R] + ids [R][n_out] + pos / seq / step [R] each
    // continuous batching (tts_hip_parler_stream_*): the rows' state between two stream_run calls lives on the host
    struct GenStream {
        bool active = false, sampled = false;
        uint32_t n_slots = 0, max_steps = 0, bos = 0, eos = 0;
        std::vector<uint8_t> slot_live;
        std::vector<uint32_t> row_slot, pos, step, ids;   // live rows: cache slot, next position, step counter, next input ids [rows][heads]
    } gs;
    uint32_t gs_baked_steps = 0;    // ... and the step budget they carry
    bool gs_graphs = false;         // the captured generation graphs were made for a stream (feed_kernel's padding slot baked in)
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
    std::set<size_t> packed_direct;    // ... of those, the k = 1 weights packed as [cin][cout] for conv1x1_direct_kernel
    std::map<size_t, _Float16 *> packed16;  // same, fp16 images (dac_f16)
    std::map<size_t, __bf16 *> packed_b3;   // k = 7 conv weights as three bf16 planes (dac_b3, experiment)
    std::map<size_t, __bf16 *> packed_ru;   // residual unit (keyed by its k = 7 weight) -> stage stream of resunit_b3_kernel
    std::map<size_t, __bf16 *> packed_p;    // conv weight -> bf16 planes in stage order for conv_b3p_kernel (k = 7: 64-channel tiles, k = 1: 128-channel tiles)
    int dac_split = 1;          // tune("dac_split") / TTS_HIP_DAC_SPLIT: operand split of the codec's plane kernels for F32 tensors: 1 (default since round 6) = fp16 hi + lo, three
                                // products; 0 = bf16 x 3, six products (rounds 3 - 5).  64 x 248-frame pass 132.4 -> 84.6 ms, every DAC bar held (profiles/r06/dac_split_call1.txt)
    int dac_f16_planes = 1;     // tune("dac_f16_planes")=0: F16 codec tensors stay on round 2's fp16 tile kernels (conv1d_mfma16_kernel / convt1d_mfma16_kernel) instead of the plane kernels with one fp16 plane
    int dac_tap7 = 1;           // tune("dac_tap7")=0: the k = 7 convs on planes keep the tap-pair k-steps (8 slots for 7 taps) instead of one tap per k-step
    int dac_wdma = 1;           // tune("dac_wdma")=0: the fused units as in round 5 (weight stages through registers + ds_write_b128, the k = 1 operand rebuilt per pass)
    int dac_planes = 1;         // tune("dac_planes")=0: the wide classes (channels % 128 == 0, no fused unit) keep fp32 activations and stage snake + split per tile
    bool dac_buf_user = false;  // counted in g_dac_buffers[device].users
    std::map<size_t, __bf16 *> packed_ct;   // transposed conv weight -> bf16 planes of convt_b3_kernel
    int dac_convt_planes = 1;   // tune("dac_convt_planes") = 0: the bf16 x 3 transposed convs always stage fp32 input (snake + split per workgroup) instead of the producer's planes
    int dac_convt_b3 = 1;       // tune("dac_convt_b3")=0: the transposed convs stay on the exact-fp32 MFMA kernel
    int dac_fuse = 1;           // tune("dac_fuse")=0: residual units at 96 / 192 channels stay two launches (k = 7 conv, k = 1 conv + residual)
    int dac_b3 = 2;             // TTS_HIP_DAC_BF16X3 / tune("dac_exact_fp32") (default 2 since round 3; 0 = exact-fp32 MFMA convs): k = 7 convs of F32 tensors as six bf16 MFMAs per product (conv1d_mfma_b3_kernel); 1 = only the layers with 64-channel tiles, 2 = also the 96-channel tile (both measured and in the parity tests since round 3)
    bool kk_lstm_split = true;  // tune("kokoro_lstm_split")=0: the bidirectional LSTMs through the one-workgroup-per-direction kernel
    char *kk_pool = nullptr;    // Kokoro scratch pool (KScratch): grows to the largest call
    size_t kk_pool_cap = 0, kk_pool_next = 0;
    int *kk_stuck = nullptr;    // set by kk_lstm_split_kernel when a granule never arrives (bounded spin)
    bool kk_b3 = true;          // tune("kokoro_b3") = 0: Kokoro's k = 3 / 5 / 7 / 11 same-convolutions stay on the exact-fp32 MFMA kernel instead of bf16 x 3 split products
    bool kk_attn_lds = true;    // tune("kokoro_attn_lds") = 0: ALBERT's attention as one wave per (head, row) walking the K rows from memory (kk_albert_attn_kernel)
    int kk_split = 1;           // tune("kokoro_split"): the operand split of those convolutions — 1 = fp16 hi + lo, three products (round 6), 0 = three bf16 planes, six products
    bool kk_mfma = true;        // tune("kokoro_mfma")=0: every Kokoro convolution through the one-thread-per-output kernel
    bool attn_short = true;     // tune("attn_short")=0: cross-attention through the general kernel
    int dac_group = 64;         // TTS_HIP_DAC_GROUP: utterances per codec pass (16: 451, 32: 458, 64: 461, 128: 460, 384: 462 audio-s/s at 3 x 384)
    bool dac_conv1_direct = true;   // tune("dac_conv1_direct")=0: the 96- / 192-channel k=1 convs stay on conv1d_mfma_kernel<1,...>
    int attn_rows_min = 256;    // tune("attn_rows_min") / TTS_HIP_ATTN_ROWS: forwards with at least this many rows run the self-attention one workgroup per ROW (attn_rows_kernel); 0 = never
    int tile_min_rows = 33;     // forwards with at least this many rows take the LDS-tiled GEMM (gemm_tile_kernels.h); 0 = never
    int tile_force = -1;        // TTS_HIP_TILE_FORCE: tile shape index for every tiled GEMM (tuning)
    int tile_force_ks = 0;      // TTS_HIP_TILE_KS: k slices for the residual GEMMs (tuning)
    const void *aq_src = nullptr;  // activation rows whose Q8_0 blocks already sit in aq / ad (written by the producing kernel)
    // many rows on quantised matrices (qgemm_tile_kernels.h): activation block scales transposed, float [max(H, F) / 32][ldr]
    float *adT = nullptr;
    int ldr = 0;
    int8_t *aq2 = nullptr;      // second set: blocks written by a GEMM epilogue (GELU output, attended cross rows) while that GEMM reads the first
    float *adT2 = nullptr;
    int aqt_set = 0;            // which set holds the blocks of aqt_src
    int qtile_min_rows = 65;    // tune("qtile_min_rows"): forwards with at least this many rows take the LDS-tiled integer GEMM; 0 = never (the 16-feature kernel, <= 256 rows)
    int qtile_shape = -1;       // tune("qtile_shape"): tile shape index for every tiled integer GEMM (tuning / tests)
    int qtile_big = 0;          // tune("qtile_big"): tile shape of the GEMMs with at least two 64 x 64 tiles per CU (0: 64 x 64; 1: 128 x 128; 2: 128 x 64; 3: 64 x 128)
    int qtile_ks = 0;           // tune("qtile_ks"): k slices of the residual GEMMs (tuning / tests)
    bool qtile_fuse = true;     // tune("qtile_fuse")=0: LayerNorm, GELU and the attentions hand fp32 rows to a quantising launch of their own instead of writing Q8_0 blocks themselves
    const void *aqt_src = nullptr; // activation rows whose Q8_0 blocks already sit in aq / adT
    QTileOut qtile_out;         // request for the next run_gemm: Q8_0 output (EPI_GELU / EPI_CROSS on the tiled integer path); cleared by it
    int attn_fold = 1;          // tune("attn_fold") = 0: Dia's step keeps attn_gqa_combine_kernel and silu_mul_kernel as launches of their own.  1 (default): the consuming
                                // projections merge the attention's key slices / apply silu * up while they stage their rows (gemv_stream_kernel<.., PRO_ATTN8 / PRO_SILU, ..>)
    bool cross_fold = true;     // tune("cross_fold")=0: the cross-attention of a many-row forward stays a launch of its own (attn_short_kernel) instead of the cross-q GEMM's epilogue
    bool cross_folded = false;  // set by run_gemm: the EPI_CROSS request was served by the 64 x 64 tile
    bool attn_wave = true;      // tune("attn_wave")=0: the split decode attention of a Llama step and the cross-attention of a Dia step through attn_gqa_split_kernel (rounds 2-4) instead of attn_gqa_wave_kernel
    int attn_split_max = 8;     // tune("attn_split"): key splits of the decode attention of the Llama / Dia steps (1 = off)
    float *attn_part = nullptr; // [rows][heads][splits][130] partial softmax results
    size_t attn_part_cap = 0;   // in (row, head, split) triples
    int tile_deep = 1;          // tune("tile_deep")=0: always the 64-wide k-tiles / 4 buffers form (half the LDS per workgroup)
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
    bool b1_fc2_split = true;        // tune("b1_fc2_split")=0: fc2 stays 64 workgroups x 128 KB of weights (7.4 us) instead of 256 x 32 KB writing four K-slice slabs
    bool b1_defer_combine = true;    // tune("b1_defer_combine")=0: the split-T self-attention folds its partials itself (last workgroup, +3.3 us) instead of out_proj's prologue
    bool b1_stamps_want = false;     // tts_hip_tune("b1_stamps") or TTS_HIP_B1_STAMPS=1 before finalize
    long long *b1_stamps = nullptr;  // TTS_HIP_B1_STAMPS=1: 16 s_memrealtime stamps per launch of a <= 4-row forward (debug_read "stamps", profiles/b1_chain.py)
    int b1_stamp_slot = 0;
    bool attn_fused = true;          // tune("attn_fused")=0: split-T self-attention keeps its separate combine launch and small batches stay unsplit
    uint32_t *attn_cnt = nullptr;    // [RMAX][heads] arrival counters of the fused combine (attn_kernel)
};

// ---- shim_core.hip
void free_dev(void *p);
void arena_own(tts_hip_ctx *c);
void arena_share(tts_hip_ctx *c);
void arena_release(tts_hip_ctx *c);
bool starts_with(const std::string &s, const char *pre);
int plan(tts_hip_ctx *c);
// ---- shim_decoder.hip (launch plumbing, the Parler forward, generation loop)
bool attr_needed(std::atomic<uint64_t> &done, int device);
int prof_begin(tts_hip_ctx *c, int kclass, double bytes, double flops);
int prof_end(tts_hip_ctx *c);
int ready(tts_hip_ctx *c, const char *who);
int stage_uniforms(tts_hip_ctx *c, const float *uniforms, size_t count);
int stage_penalty(tts_hip_ctx *c, float penalty, int n);
// ---- shim_qtile.hip (the tiled integer GEMM: its own unit, built without SLP vectorisation and with MFMA results in VGPRs)
int qtile_transpose_scales(tts_hip_ctx *c, const W &w);
// ---- shim_codec.hip
int dac_row_stride(const tts_hip_ctx *c, int L);


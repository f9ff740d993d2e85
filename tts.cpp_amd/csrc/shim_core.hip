// shim_core.hip — errors, GGUF block formats, context create / destroy, tensor upload, arena planning, the RCCL weight broadcast.
//
// Owns: the device weight arena (tts_model's backend buffer, /root/reference/src/tts_model.cpp:157-169),
// the self-attention KV cache (parler_kv_cache, src/models/parler/model.cpp:339-385), the cross K/V
// (prep_cross_key_values :110-173), one HIP stream, and the captured hipGraphs that replace the
// per-step ggml graph rebuild (build_parler_graph :520-614 + ggml_backend_sched_alloc_graph :674).
// No CPU fallback: every entry point fails if the device is unavailable.
#include "shim_internal.h"

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";
int set_err(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return -1;
}

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

DacBuffers g_dac_buffers[64];

static const char *KNAMES[TTS_HIP_K_COUNT] = {"embed", "ln", "gemm_qkv", "attn_self", "gemm_attn_out", "gemm_cross_q", "attn_cross",
                                              "gemm_cross_out", "gemm_fc1", "gemm_fc2", "gemm_heads", "sample", "gemm_other",
                                              "dac_embed", "dac_conv7", "dac_conv1", "dac_convt", "dac_final", "dac_resunit", "kokoro_conv_mfma"};
extern "C" const char *tts_hip_kclass_name(int k) { return (k >= 0 && k < TTS_HIP_K_COUNT) ? KNAMES[k] : "?"; }

extern "C" int tts_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { set_err("hipGetDeviceCount failed (no ROCm device visible)"); return 0; }
    return n;
}

extern "C" int tts_hip_tune(tts_hip_ctx *c, const char *key, int v);
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
        const bool prio = hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest;
        hipError_t rc = prio ? hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, greatest)
                             : hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (rc != hipSuccess) { set_err("hipStreamCreate failed"); delete c; return nullptr; }
        if (prio && hipStreamCreateWithPriority(&c->dac_stream, hipStreamNonBlocking, least) != hipSuccess) c->dac_stream = nullptr;
    }
    // Environment switches: only the ones a test or bench.py sets (each names its test).  Everything else that used to be read here is a
    // tts_hip_tune() key: a harness under profiles/ can still flip it, a deployment cannot flip it by accident.
    if (const char *e = getenv("TTS_HIP_ATTN_NSPLIT")) c->attn_nsplit_override = atoi(e);
    if (const char *e = getenv("TTS_HIP_ATTN_ROWS")) c->attn_rows_min = std::max(0, atoi(e));                                                 // bench.py A/B, test_gpu_parler.py: row-major self-attention from this many rows (0: never)
    if (const char *e = getenv("TTS_HIP_DAC_GROUP")) c->dac_group = std::max(1, atoi(e));                       // test_gpu_dac.py, bench.py: utterances per codec pass
    if (const char *e = getenv("TTS_HIP_DAC_BF16X3")) (void) tts_hip_tune(c, "dac_exact_fp32", atoi(e) == 0);  // test_gpu_dac.py: 0 = the exact-fp32 MFMA codec
    if (const char *e = getenv("TTS_HIP_DAC_SPLIT")) c->dac_split = atoi(e) != 0;                               // test_gpu_dac.py: 0 = bf16 x 3 products, 1 = fp16 hi + lo
    if (const char *e = getenv("TTS_HIP_GEN_COMPACT")) c->gen_compact = atoi(e) != 0;                           // test_gpu_runner.py: row compaction of the generation loop
    if (const char *e = getenv("TTS_HIP_TILE_FORCE")) c->tile_force = atoi(e);                                  // test_gpu_parler.py: every tile shape of the tiled GEMM
    if (const char *e = getenv("TTS_HIP_TILE_KS")) c->tile_force_ks = atoi(e);
    if (const char *e = getenv("TTS_HIP_GEMV_ROWS")) c->gemv_rows = atoi(e) != 0;                               // test_gpu_gemv_rows.py, test_gpu_orpheus.py
    if (const char *e = getenv("TTS_HIP_LLAMA_GRAPH")) c->llama_graph = atoi(e) != 0;
    if (const char *e = getenv("TTS_HIP_Q4_NATIVE")) c->q4_native = atoi(e) != 0;
    return c;
}

// Tuning keys (harnesses under profiles/, and the fallback parity test): set between tts_hip_create and the first launch.
extern "C" int tts_hip_tune(tts_hip_ctx *c, const char *key, int v) {
    if (!c || !key) return set_err("tts_hip_tune: null argument");
    const std::string k(key);
    if (k == "dac_exact_fp32") {   // the codec as exact-fp32 MFMA kernels throughout (round 2's pipe): the reference arithmetic the bf16 x 3 default is held against
        c->dac_b3 = v ? 0 : 2; c->dac_fuse = c->dac_convt_b3 = c->dac_planes = v ? 0 : 1;
    } else if (k == "dac_fuse") c->dac_fuse = v;                 // 0: residual units at 96 / 192 channels as two launches
    else if (k == "dac_convt_b3") c->dac_convt_b3 = v;           // 0: transposed convs on the exact-fp32 MFMA kernel
    else if (k == "dac_convt_planes") c->dac_convt_planes = v;   // 0: transposed convs stage fp32 input themselves
    else if (k == "dac_planes") c->dac_planes = v;               // 0: the wide classes keep fp32 activations
    else if (k == "dac_split") c->dac_split = v ? 1 : 0;         // 1: fp16 hi + lo split (three products) instead of bf16 x 3 (six)
    else if (k == "dac_f16_planes") c->dac_f16_planes = v;       // 0: F16 codec tensors on the fp16 tile kernels of round 2
    else if (k == "dac_tap7") c->dac_tap7 = v;                   // 0: tap-pair k-steps in the k = 7 planes convs
    else if (k == "dac_wdma") c->dac_wdma = v;                   // 0: weight stages through registers (rounds 3-6a)
    else if (k == "dac_b3") c->dac_b3 = std::max(0, v);
    else if (k == "dac_conv1_direct") c->dac_conv1_direct = v != 0;
    else if (k == "kokoro_mfma") c->kk_mfma = v != 0;
    else if (k == "kokoro_b3") c->kk_b3 = v != 0;
    else if (k == "kokoro_attn_lds") c->kk_attn_lds = v != 0;
    else if (k == "kokoro_split") {   // the packed planes belong to a scheme: drop them, the next convolution packs again
        if (c->kk_split != (v != 0)) { (void) hipDeviceSynchronize(); for (auto &pw : c->packed_b3) free_dev(pw.second); c->packed_b3.clear(); }
        c->kk_split = v != 0;
    }
    else if (k == "kokoro_lstm_split") c->kk_lstm_split = v != 0;
    else if (k == "ln_fuse_max") c->ln_fuse_max = std::max(0, std::min(32, v));
    else if (k == "attn_short") c->attn_short = v != 0;
    else if (k == "attn_rows_min") c->attn_rows_min = std::max(0, v);
    else if (k == "attn_fused") c->attn_fused = v != 0;
    else if (k == "attn_split") c->attn_split_max = std::max(1, std::min(16, v));
    else if (k == "attn_fold") c->attn_fold = v != 0;
    else if (k == "attn_wave") c->attn_wave = v != 0;
    else if (k == "cross_fold") c->cross_fold = v != 0;
    else if (k == "tile_min_rows") c->tile_min_rows = std::max(0, v);
    else if (k == "tile_deep") c->tile_deep = v;
    else if (k == "qtile_min_rows") c->qtile_min_rows = std::max(0, v);   // 0: quantised matrices stay on the 16-feature kernel (<= 256 rows per forward)
    else if (k == "qtile_shape") c->qtile_shape = v;
    else if (k == "qtile_big") c->qtile_big = std::max(0, std::min(3, v));
    else if (k == "qtile_ks") c->qtile_ks = std::max(0, v);
    else if (k == "qtile_fuse") c->qtile_fuse = v != 0;
    else if (k == "gemv_stream") c->gemv_stream = v != 0;
    else if (k == "q_stream") c->q_stream = v == 1 ? 31 : (int) v;
    else if (k == "q_fuse_max") c->q_fuse_max = std::max(0, std::min(16, v));
    else if (k == "q4_lds") c->q4_lds = v != 0;
    else if (k == "q4_rope") c->q4_rope = v != 0;
    else if (k == "q4_silu") c->q4_silu = v != 0;
    else if (k == "q4_rms") c->q4_rms = v != 0;
    else if (k == "b1_fc2_split") c->b1_fc2_split = v != 0;
    else if (k == "b1_defer_combine") c->b1_defer_combine = v != 0;
    else if (k == "b1_stamps") c->b1_stamps_want = v != 0;
    else return set_err("tts_hip_tune: unknown key '%s'", key);
    return 0;
}

void free_dev(void *p) { if (p) (void) hipFree(p); }
std::mutex g_dac_pass_mutex[64];
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

// Weight arenas are reference counted per allocation: a context that finalizes on another context's arena (tts_hip_finalize(ctx, arena),
// host: tts_load_options::share_with) takes a reference, so whichever of the sharers is destroyed last frees the memory — the owner need
// not outlive them.  An external arena this library did not allocate (a caller's buffer) is never in the table and never freed here.
static std::mutex g_arena_mutex;
static std::map<void *, int> g_arena_refs;
void arena_own(tts_hip_ctx *c) { std::lock_guard<std::mutex> l(g_arena_mutex); g_arena_refs[c->arena] = 1; }
void arena_share(tts_hip_ctx *c) {
    std::lock_guard<std::mutex> l(g_arena_mutex);
    auto it = g_arena_refs.find(c->arena);
    c->arena_counted = it != g_arena_refs.end();
    if (c->arena_counted) it->second++;
}
void arena_release(tts_hip_ctx *c) {
    if (!c->arena) return;
    if (c->arena_external && !c->arena_counted) return;
    std::lock_guard<std::mutex> l(g_arena_mutex);
    auto it = g_arena_refs.find(c->arena);
    if (it == g_arena_refs.end()) { if (!c->arena_external) free_dev(c->arena); return; }
    if (--it->second == 0) { free_dev(c->arena); g_arena_refs.erase(it); }
}

extern "C" void tts_hip_destroy(tts_hip_ctx *c) {
    if (!c) return;
    (void) hipSetDevice(c->device);
    (void) hipStreamSynchronize(c->stream);
    for (auto &g : c->graphs) (void) hipGraphExecDestroy(g.second);
    for (auto &t : c->tensors) free_dev(t.second.tmp);
    arena_release(c);
    free_dev(c->kcache); free_dev(c->vcache); free_dev(c->x); free_dev(c->q); free_dev(c->att); free_dev(c->u32);
    free_dev(c->u16); free_dev(c->xn16); free_dev(c->att16); free_dev(c->partials); free_dev(c->aq); free_dev(c->ad); free_dev(c->adT); free_dev(c->aq2); free_dev(c->adT2); free_dev(c->d_uniforms); free_dev(c->l_cand); free_dev(c->l_smp); free_dev(c->d_pen); free_dev(c->d_last); free_dev(c->d_repc);
    free_dev(c->l_x); free_dev(c->l_xn); free_dev(c->l_qkv); free_dev(c->l_att); free_dev(c->l_gu); free_dev(c->l_g); free_dev(c->l_logits); free_dev(c->l_parts);
    free_dev(c->attn_part); free_dev(c->kk_stuck); free_dev(c->kk_pool);
    free_dev(c->l_kc); free_dev(c->l_vc); free_dev(c->l_ids); free_dev(c->l_pos); free_dev(c->l_tok);
    free_dev(c->l_seq); free_dev(c->l_btok); free_dev(c->l_bpi); free_dev(c->l_bsmp); free_dev(c->l_bpv);
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
bool starts_with(const std::string &s, const char *pre) { return s.compare(0, strlen(pre), pre) == 0; }

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
        if (w.type == TTS_HIP_Q8I && c->has_parler && w.K % 128 == 0) {
            // the same scales transposed for the many-row kernel (qgemm_tile_kernels.h), written by tts_hip_finalize: part of the arena, so that
            // contexts sharing it and ranks receiving it by broadcast hold them too
            w.ldw = (int) ((w.N + 255) & ~(int64_t) 255);
            w.stoff = alloc((size_t) (w.K / 32) * w.ldw * 4);
        }
        return w;
    }
};

int plan(tts_hip_ctx *c) {
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
    if (n == 1 && !getenv("TTS_HIP_RCCL_SINGLE_RANK")) return 0;   // (the env switch: ncclCommInitAll over one device + an in-place broadcast, see _rank below)
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
    // a world of one has nothing to receive; TTS_HIP_RCCL_SINGLE_RANK=1 sends it through RCCL all the same (communicator of one rank, in-place broadcast):
    // the library path — dlopen, unique id, ncclCommInitRank, ncclBroadcast on the context's stream, destroy — runs on a one-GPU box (tests/test_gpu_runner.py)
    if (world == 1 && !getenv("TTS_HIP_RCCL_SINGLE_RANK")) return 0;
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


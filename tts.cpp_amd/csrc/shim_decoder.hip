// shim_decoder.hip — launch plumbing, the GEMM / attention launchers, the Parler decoder forward, finalize, the generation loops,
// the T5 voice-prompt encoder and the introspection entry points.  (tts_hip.hip until round 4; split so that the units build in parallel.)
#include "shim_internal.h"

#include "parler_kernels.h"
#include "gemm_tile_kernels.h"
#include "gemv_stream_kernels.h"
#include "t5_kernels.h"
#include "gemv_kernels.h"
#include "llama_kernels.h"
#include "dia_kernels.h"
#include "shim_decoder.h"
#include "shim_qtile.h"

// kernel launch plumbing (+ optional per-class event timing)
// ------------------------------------------------------------------------------------------------
// hipFuncSetAttribute applies to the current device: remember per device (a host may drive several GPUs from one
// process, e.g. device_pool), not per process
bool attr_needed(std::atomic<uint64_t> &done, int device) {
    const uint64_t bit = 1ull << (device & 63);
    if (done.load(std::memory_order_relaxed) & bit) return false;
    done.fetch_or(bit, std::memory_order_relaxed);
    return true;
}

static bool prof_on(const tts_hip_ctx *c, int kclass) {
    return c->prof || (c->prof_light && kclass >= TTS_HIP_K_DAC_EMBED);
}
int prof_begin(tts_hip_ctx *c, int kclass, double bytes, double flops) {
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
int prof_end(tts_hip_ctx *c) {
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
    if (c->dac_f16) return 8 | (c->dac_f16_planes && c->dac_b3 && c->dac_tap7 && !(c->d.flags & TTS_HIP_FLAG_VALU_GEMM) ? 128 : 0);   // 128: on the plane kernels (one fp16 plane)
    if (c->d.flags & TTS_HIP_FLAG_VALU_GEMM) return 16;
    return (c->dac_b3 ? 1 : 0) | (c->dac_fuse ? 2 : 0) | (c->dac_convt_b3 ? 4 : 0) | (c->dac_planes && c->dac_b3 ? 32 : 0) | (c->dac_split && c->dac_b3 ? 64 : 0);   // 64: fp16 hi + lo split (three products) instead of bf16 x 3 (six)
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
    const int ngs = PRO == PRO_LN || PRO == PRO_ATTN || PRO == PRO_CROSS ? 1 : std::max(1, std::min(n_groups, 16 / nw));
    size_t lds = 0;
    if (PRO == PRO_LN || PRO == PRO_ATTN || PRO == PRO_CROSS) {
        lds = (size_t) RB * 16 * (a.K + (WT == 1 ? 8 : 4)) * (WT == 1 ? 2 : 4);
        lds = (lds + 15) & ~(size_t) 15;
    }
    if (nw > 1) lds += (size_t) ngs * nw * RB * 4 * 64 * 4;
    static std::atomic<uint64_t> attr{0};
    if (lds > 48 * 1024 && attr_needed(attr, c->device)) {
        HIPCHK(hipFuncSetAttribute((const void *) gemm16_kernel<WT, PRO, EPI, RB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    // pending fc2 slabs (a.n_parts) are folded by exactly two pieces of gemm16_kernel: the LayerNorm prologue of the fp16 instance at K <= 1024 and
    // the residual epilogue — both for four slabs only.  Any other instance would silently read an x that lacks them.
    if (a.n_parts && !(a.n_parts == 4 && ((WT == 1 && PRO == PRO_LN && a.K <= 1024) || EPI == EPI_RESID)))
        return set_err("gemm16: %d pending K-slice slabs cannot be folded by this instance (WT %d, PRO %d, EPI %d, K %d)", a.n_parts, WT, PRO, EPI, a.K);
    if (lds > 160 * 1024) return set_err("gemm16: LDS request %zu exceeds 160 KiB", lds);
    if (PRO == PRO_LN && a.K > 2048) return set_err("gemm16: LayerNorm prologue supports hidden sizes up to 2048 (got %d)", a.K);
    if ((PRO == PRO_LN || PRO == PRO_ATTN || PRO == PRO_CROSS) && ngs * nw > 8) return set_err("gemm16: fused-prologue launch wants %d waves (> 8)", ngs * nw);
    const int nz = a.rows_per_z ? (a.R + a.rows_per_z - 1) / a.rows_per_z : 1;
    hipLaunchKernelGGL((gemm16_kernel<WT, PRO, EPI, RB>), dim3(a.N / 16, ksplit, nz), dim3(ngs * nw * 64), lds, c->stream, a);
    HIPCHK(hipGetLastError());
    return 0;
}

template <int WT, int PRO, int EPI>
static int launch_gemm16_rb(tts_hip_ctx *c, const GemmArgs &a_in) {
    GemmArgs a = a_in;
    if (a.R <= 16) return launch_gemm16<WT, PRO, EPI, 1>(c, a);
    if (a.R <= 32) return launch_gemm16<WT, PRO, EPI, 2>(c, a);
    if (PRO == PRO_LN && (WT == 0 || a.R > 64)) return set_err("gemm16: %d rows with a fused LayerNorm prologue do not fit LDS", a.R);
    return launch_gemm16<WT, PRO, EPI, 4>(c, a);  // loops over groups of 64 rows, weights stay in registers
}

// rows one forward can carry: 256 through the 16-feature workgroups, 1024 when every decoder matrix has an LDS-tiled GEMM: fp16 (gemm_tile_kernel) or a
// GGUF-quantised one with its transposed scale table (qgemm_tile_kernel)
static bool any_qtile(const tts_hip_ctx *c) {
    if (c->heads.stoff) return true;
    for (const PLayer &y : c->layers)
        for (const W *w : {&y.qkv, &y.o, &y.cq, &y.co, &y.fc1, &y.fc2})
            if (w->stoff) return true;
    return false;
}
static int max_rows_for(const tts_hip_ctx *c) {
    if (c->tile_min_rows <= 0) return 256;
    auto tiled = [&](const W &w) { return w.type == TTS_HIP_F16 || (w.type == TTS_HIP_Q8I && w.stoff && c->qtile_min_rows > 0 && c->H <= 2048); };
    for (const PLayer &y : c->layers)
        for (const W *w : {&y.qkv, &y.o, &y.cq, &y.co, &y.fc1, &y.fc2})
            if (w->N && !tiled(*w)) return 256;
    if (!tiled(c->heads)) return 256;
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
int stream_slices(const tts_hip_ctx *c, const W &w, int R, int max_slabs) {
    if (!c->gemv_stream || w.type != TTS_HIP_F16 || R > 16 || w.K % 256 || w.N % 16 || (c->d.flags & TTS_HIP_FLAG_VALU_GEMM)) return 0;
    const int tiles = (int) w.N / 16;
    int ks = 1;
    while (ks * 2 <= max_slabs && (int) w.K % (ks * 2 * 256) == 0 && tiles * ks * 2 <= 4096) ks *= 2;
    if ((size_t) 16 * (w.K / ks + 32) * 2 > 96 * 1024) return 0;   // the slice of the rows must fit LDS
    return ks;
}

// a staging prologue that merges attention slices / applies silu * up exists for the four-wave instance of gemv_stream_kernel only
bool stream_fold_ok(const tts_hip_ctx *c, const W &w, int R, int max_slabs) {
    const int ks = stream_slices(c, w, R, max_slabs);
    return ks > 0 && (int) w.N / 16 * ks < 4096 && R <= 8;
}

template <int NWV, int PRO, int EPI, int NPI = 8>
static int launch_stream_one(tts_hip_ctx *c, const GemmArgs &a, StreamMap sm, int grid, size_t lds) {
    static std::atomic<uint64_t> attr{0};
    if (lds > 48 * 1024 && attr_needed(attr, c->device))
        HIPCHK(hipFuncSetAttribute((const void *) gemv_stream_kernel<NWV, PRO, EPI, NPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((gemv_stream_kernel<NWV, PRO, EPI, NPI>), dim3(grid), dim3(NWV * 64), lds, c->stream, a, sm);
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
    STREAM_CASE(4, PRO_ATTN8, EPI_STORE)   // Dia's step: four-wave workgroups only (the callers check stream_fold_ok)
    if (nwv == 4 && pro == PRO_SILU && epi == EPI_STORE) {   // an instance per slab count of the gate | up rows
        if (a.n_parts <= 1) return launch_stream_one<4, PRO_SILU, EPI_STORE, 1>(c, a, sm, grid, lds);
        if (a.n_parts <= 2) return launch_stream_one<4, PRO_SILU, EPI_STORE, 2>(c, a, sm, grid, lds);
        if (a.n_parts <= 4) return launch_stream_one<4, PRO_SILU, EPI_STORE, 4>(c, a, sm, grid, lds);
        return launch_stream_one<4, PRO_SILU, EPI_STORE, 8>(c, a, sm, grid, lds);
    }
#undef STREAM_CASE
    return set_err("gemv_stream: no kernel for pro=%d epi=%d", pro, epi);
}

// ------------------------------------------------------------------------------------------------
// many rows on a GGUF-quantised matrix: LayerNorm (if any) -> Q8_0 blocks of the activation rows -> LDS-tiled integer block GEMM
// (qgemm_tile_kernels.h / shim_qtile.hip).  The activations' blocks come from the producing kernel when it wrote them (c->aqt_src), from the
// LayerNorm launch (ln_rows_q8t_kernel) or from quant_rows_q8t_kernel.
// ------------------------------------------------------------------------------------------------
static bool qtile_ok(const tts_hip_ctx *c, const W &w, const GemmArgs &a, int pro) {
    return c->qtile_min_rows > 0 && a.R >= c->qtile_min_rows && w.stoff && c->adT && w.K % 128 == 0 && w.N % 16 == 0 && pro != PRO_F16 && pro != PRO_ATTN && pro != PRO_CROSS &&
           a.R <= c->ldr && !a.kchunk && !a.n_parts && (pro != PRO_LN || (a.K <= 2048 && a.K % 32 == 0));
}
static void choose_qtile(const tts_hip_ctx *c, int R, int N, int K, bool may_split, int *shape_out, int *ks_out) {
    // profiles/r06/qgemm_bench_*.txt: the launch is bound by the vector instructions its waves issue (48 scaling instructions per MFMA), so the shape that
    // gives every SIMD about four waves wins: 64 x 64 tiles of four waves (wave tile 32 x 32) when there are at least two tiles per CU; with fewer (N =
    // hidden size at 1024 rows: 256 tiles) two k groups inside the workgroup (shape 4: eight waves) — the epilogue (residual add, cross-attention fold)
    // stays in the launch; the K = 4096 residual GEMM (fc2) additionally writes two split-K slabs the next LayerNorm folds
    const int tiles = ((R + 63) / 64) * ((N + 63) / 64);
    int shape = tiles >= 512 ? c->qtile_big : 4, ks = 1;
    if (shape == 4 && may_split && K >= 2048 && K % 512 == 0) ks = 2;
    if (c->qtile_shape >= 0 && c->qtile_shape < N_QTILE_SHAPES) shape = c->qtile_shape;
    if (c->qtile_ks > 0 && may_split && K % (c->qtile_ks * 128) == 0) ks = c->qtile_ks;
    *shape_out = shape;
    *ks_out = ks;
}
static int run_qtile(tts_hip_ctx *c, int kclass, const W &w, GemmArgs a, int pro, int epi, QTileOut qo) {
    const bool have_q = c->aqt_src != nullptr && c->aqt_src == a.A && a.lda == a.K && pro != PRO_LN;
    const int in_set = have_q ? c->aqt_set : 0;
    c->aqt_src = nullptr;
    c->aq_src = nullptr;
    c->qtile_out = QTileOut{};
    if (pro == PRO_LN) {
        // LayerNorm (+ the pending split-K slabs folded into x) and the Q8_0 conversion in one launch
        CHK(prof_begin(c, TTS_HIP_K_LN, (double) a.R * a.K * 5.2, 0));
        const dim3 grid((unsigned) a.R), block(64);
        const float *parts = c->pending_parts ? (const float *) c->partials : (const float *) nullptr;
        const int np = c->pending_parts;
        const int64_t ss = (int64_t) c->RMAX * c->H;
#define LNQ_CASE(NIv, NPv) hipLaunchKernelGGL((ln_rows_q8t_kernel<NIv, NPv>), grid, block, 0, c->stream, (float *) a.A, a.K, a.ln_w, a.ln_b, c->aq, c->adT, c->ldr, a.R, parts, np, ss)
        if (a.K <= 1024) {
            if (np == 0) LNQ_CASE(4, 0); else if (np == 2) LNQ_CASE(4, 2); else if (np == 4) LNQ_CASE(4, 4); else LNQ_CASE(4, -1);
        } else {
            if (np == 0) LNQ_CASE(8, 0); else LNQ_CASE(8, -1);
        }
#undef LNQ_CASE
        HIPCHK(hipGetLastError());
        c->pending_parts = 0;
        CHK(prof_end(c));
    }
    const bool want_cross = epi == EPI_CROSS;
    const bool may_split = epi == EPI_RESID && a.H <= 2048 && a.N == a.H && a.out == c->x;
    int shape = 0, ks = 1;
    choose_qtile(c, a.R, a.N, a.K, may_split, &shape, &ks);
    if (want_cross) {
        const bool fold = QTILE_SHAPES[shape].BN == 64 && a.cross_E >= 1 && a.cross_E <= 32 && a.N == a.H && a.H % 64 == 0;
        c->cross_folded = fold;
        if (!fold) epi = EPI_STORE;
    }
    if (ks > 1) {
        a.kchunk = a.K / ks;
        a.slab_stride = (int64_t) c->RMAX * c->H;
        a.out = c->partials;
        epi = EPI_STORE;
        c->pending_parts = ks;
    }
    const double wbytes = (double) w.K * w.N * (1.0 + 2.0 / 32);
    CHK(prof_begin(c, kclass, wbytes + (double) a.R * a.K * 1.125 + (double) a.R * a.N * 4, 2.0 * a.R * (double) w.K * w.N));
    if (pro != PRO_LN && !have_q) CHK(qtile_quant_rows(c, (const float *) a.A, a.lda, a.K, a.R));
    QTileArgs qa{};
    qa.g = a;
    qa.wdT = (const float *) (c->arena + w.stoff);
    qa.ldw = w.ldw;
    qa.aq = in_set ? c->aq2 : c->aq;
    qa.adT = in_set ? c->adT2 : c->adT;
    qa.ldr = c->ldr;
    const bool q_out = qo.q && (epi == EPI_GELU || epi == EPI_CROSS) && qo.q != qa.aq;
    if (q_out) { qa.q_out = qo.q; qa.d_out = qo.dT; qa.ldq = qo.ldq; }
    CHK(launch_qtile(c, qa, epi, shape, ks));
    if (q_out) { c->aqt_src = qo.stands_for; c->aqt_set = qo.q == c->aq2 ? 1 : 0; }
    return prof_end(c);
}

// ------------------------------------------------------------------------------------------------
// weight-streaming integer GEMM for 5 .. 16 rows (qgemv_stream_kernel, gemv_stream_kernels.h): K slices -> fp32 slabs the consumer folds
// ------------------------------------------------------------------------------------------------
// K slices for an [N][K] quantised matrix: as many 256-column chunks per slice as keep about 4096 (tile, slice) items and the consumer's slab budget;
// 0 = the shape does not go through the kernel
// rows of one slab of the weight-streaming integer GEMM (= rows its workgroups keep in LDS; 8 of the 16 when R <= 8)
int qstream_slab_rows(int R) { return R <= 16 ? 16 : R <= 32 ? 32 : 64; }
static size_t qstream_lds(int R, int kslice) {
    const int RS = R <= 8 ? 8 : qstream_slab_rows(R);
    return (size_t) (RS + 1) * kslice + (size_t) RS * (kslice / 32) * 4;
}
// K slices of the weight-streaming integer GEMM for R rows on w (0: the shape does not qualify).  Measured at Orpheus-3B's shapes (profiles/qstream_bench.hip,
// profiles/r06/qstream_bench_r8d.txt): ~2000 (tile, slice) items fill the chip; a 192-tile projection prefers fewer, longer waves unless K is long.
int qstream_slices(const tts_hip_ctx *c, const W &w, int R, int max_slabs) {
    if (!c->q_stream || w.type != TTS_HIP_Q8I || R < 5 || R > 64 || w.K % 256 || (c->d.flags & (TTS_HIP_FLAG_VALU_GEMM | TTS_HIP_FLAG_DEQUANT_Q))) return 0;
    const int tiles = ((int) w.N + 15) / 16, chunks = (int) w.K / 256;
    const int want = std::min(max_slabs, tiles >= 4096 ? 1 : tiles >= 1024 ? 2 : tiles >= 256 ? 6 : w.K >= 8192 ? 16 : 4);
    int ks = 0;
    for (int k : {1, 2, 3, 4, 6, 8, 12, 16})
        if (k <= want && chunks % k == 0) ks = k;
    for (int k : {2, 3, 4, 6, 8, 12, 16})   // the rows' slice must fit LDS
        if (qstream_lds(R, (int) w.K / ks) > 64 * 1024 && k > ks && k <= max_slabs && chunks % k == 0) ks = k;
    return qstream_lds(R, (int) w.K / ks) > 64 * 1024 ? 0 : ks;
}
// activations: the Q8_0 blocks in c->aq / c->ad ([R][K], [R][K / 32]); slab kz of `out` receives the partial products of K slice kz
int launch_qstream(tts_hip_ctx *c, int kclass, const W &w, int R, float *out, int ldo, int64_t slab_stride, int ks) {
    QGemmArgs qa{};
    qa.g.W = c->arena + w.off; qa.g.K = (int) w.K; qa.g.N = (int) w.N; qa.g.R = R; qa.g.out = out; qa.g.ldo = ldo; qa.g.slab_stride = slab_stride;
    qa.wd = (const _Float16 *) (c->arena + w.soff);
    qa.aq = c->aq; qa.ad = c->ad;
    const StreamMap sm{ks, (int) w.K / ks};
    const int tiles = ((int) w.N + 15) / 16, items = tiles * ks;
    const int rt = qstream_slab_rows(R) / 16, nwv = (tiles >= 256 && rt < 4) ? 8 : 4;   // four row tiles: the eight-wave form (256 registers per wave at most) spills
    int grid = ((items + nwv - 1) / nwv + ks - 1) / ks * ks;
    grid = std::min(grid, 1024 / ks * ks);   // beyond four workgroups per CU the waves walk several tiles (the LM head)
    const size_t lds = qstream_lds(R, sm.kslice);
    CHK(prof_begin(c, kclass, (double) w.K * w.N * (1.0 + 2.0 / 32) + (double) R * w.K * 1.125 + (double) ks * R * w.N * 4, 2.0 * R * (double) w.K * w.N));
    typedef void (*kern_t)(QGemmArgs, StreamMap);
    static const kern_t kerns[2][3] = {{qgemv_stream_kernel<4, 2, 1>, qgemv_stream_kernel<4, 2, 2>, qgemv_stream_kernel<4, 2, 4>},
                                       {qgemv_stream_kernel<8, 2, 1>, qgemv_stream_kernel<8, 2, 2>, qgemv_stream_kernel<8, 2, 4>}};
    static std::atomic<uint64_t> attr{0};
    if (attr_needed(attr, c->device))
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 3; j++) HIPCHK(hipFuncSetAttribute((const void *) kerns[i][j], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(kerns[nwv == 8][rt == 1 ? 0 : rt == 2 ? 1 : 2], dim3(grid), dim3(nwv * 64), lds, c->stream, qa, sm);
    HIPCHK(hipGetLastError());
    return prof_end(c);
}

int run_qgemm(tts_hip_ctx *c, int kclass, const W &w, GemmArgs a, int pro, int epi) {
    if (pro == PRO_F16) return set_err("run_qgemm: fp16 activations are never produced for a quantised consumer");
    if (qtile_ok(c, w, a, pro)) return run_qtile(c, kclass, w, a, pro, epi, c->qtile_out);
    c->qtile_out = QTileOut{};
    if (epi == EPI_CROSS) { epi = EPI_STORE; c->cross_folded = false; }
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
            // NP = passes of 64 blocks a wave requests in one batch: 2 covers K <= 4096, 4 covers K <= 8192
            if (a.N >= 8192) hipLaunchKernelGGL((gemv_q4_rows_lds_kernel<4, 4>), dim3((a.N + 15) / 16), dim3(256), q4_lds, c->stream, qa, w.q4, epi);
            else if (a.K > 4096) hipLaunchKernelGGL((gemv_q4_rows_lds_kernel<4, 2, 0, 4>), dim3((a.N + 7) / 8), dim3(256), q4_lds, c->stream, qa, w.q4, epi);
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
        launch_ln_rows(c, 1, (float *) a.A, a.K, a.ln_w, a.ln_b, c->dbg, (_Float16 *) nullptr, a.R,
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
        const int ks = 4;
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

int run_gemm(tts_hip_ctx *c, int kclass, const W &w, GemmArgs a, int pro, int epi) {
    a.W = c->arena + w.off;
    a.K = (int) w.K;
    a.N = (int) w.N;
    // EPI_CROSS is a request: the cross-attention runs in the epilogue when this GEMM takes the 64 x 64 tile (one head per tile column); any other
    // path stores q as before and the caller launches the attention (c->cross_folded says which)
    const bool want_cross = epi == EPI_CROSS;
    if (want_cross) { epi = EPI_STORE; c->cross_folded = false; }
    if (w.type == TTS_HIP_Q8I) return run_qgemm(c, kclass, w, a, pro, want_cross ? (int) EPI_CROSS : epi);
    c->qtile_out = QTileOut{};
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
        launch_ln_rows(c, 1, (float *) a.A, a.K, a.ln_w, a.ln_b, h16 ? (float *) nullptr : c->dbg, h16 ? c->xn16 : (_Float16 *) nullptr, a.R,
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
        if (want_cross && shape == 3 && ks == 1 && a.cross_E >= 1 && a.cross_E <= 32 && a.N == a.H && a.H % 64 == 0) {
            TileMap tm{(a.R + 63) / 64, a.N / 64, 1};
            rc = launch_tile_shape<64, 64, 2, 4, EPI_CROSS>(c, a, tm, c->tile_deep && tm.m_tiles * tm.n_tiles <= 320);
            c->cross_folded = rc == 0;
        } else
        if (epi == EPI_STORE) rc = launch_tile<EPI_STORE>(c, a, shape, ks);
        else if (epi == EPI_QKV) rc = launch_tile<EPI_QKV>(c, a, shape, ks);
        else if (epi == EPI_RESID) rc = launch_tile<EPI_RESID>(c, a, shape, ks);
        else rc = launch_tile<EPI_GELU>(c, a, shape, ks);
        CHK(rc);
        return prof_end(c);
    }
    if (epi == EPI_RESID && a.R > c->ln_fuse_max && a.H <= 2048 && a.N == a.H && a.out == c->x && pro != PRO_CROSS && pro != PRO_ATTN) {   // (the staging prologues of the one-sequence chain have no K-slice form)
        // many rows: spread K over 4-8x more workgroups; the partial slabs are folded into x by the next LayerNorm
        const int ks = 4;
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
    GEMM_CASE(1, PRO_F16, EPI_RESID) GEMM_CASE(1, PRO_ATTN, EPI_RESID) GEMM_CASE(1, PRO_CROSS, EPI_RESID)
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
    if (nsplit == 1 && a.row_pos && a.row_seq && c->attn_rows_min > 0 && R >= c->attn_rows_min && c->NH * 16 <= 1024 && c->H == c->NH * 64) {
        // many rows: one workgroup per row reads whole K / V rows (attn_rows_kernel); below 1024 rows the keys of a row are cut into FOUR slices
        // (about a thousand workgroups again) that attn_combine_kernel folds — a fixed count per row-count class, so that a generation whose rows
        // are compacted from 384 to 256 keeps its summation order
        const int nzr = R >= 1024 ? 1 : 4;
        a.part = c->part;
        // eight keys in flight per lane: 4 / 8 / 12 / 16 measured 5.19 / 5.25 / 5.28 / 5.67 ms per 1024-row step (profiles/r04/attn_rows_u_call22.txt)
        hipLaunchKernelGGL(attn_rows_kernel<8>, dim3(R, nzr), dim3(c->NH * 16), 0, c->stream, a);
        HIPCHK(hipGetLastError());
        if (nzr > 1) {
            hipLaunchKernelGGL(attn_combine_kernel, dim3(c->NH, R), dim3(64), 0, c->stream, (const float *) a.part, nzr, c->H, c->NH, a.out, a.out16, a.out_q, a.out_dT, a.ldr);
            HIPCHK(hipGetLastError());
        }
        if (a.out_q) { c->aqt_src = a.out; c->aqt_set = a.out_q == c->aq2 ? 1 : 0; }   // the out projection finds its input rows as Q8_0 blocks
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
        // a quantised out projection on the tiled path: the attention writes the Q8_0 blocks of its rows (second set) instead of fp32
        if (c->qtile_fuse && !c->debug && c->adT && c->qtile_min_rows > 0 && R >= c->qtile_min_rows && y.o.type == TTS_HIP_Q8I && y.o.stoff && (int) y.o.K == H) {
            at.out_q = c->aq2; at.out_dT = c->adT2; at.ldr = c->ldr;
        }
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
            const bool co_half_ = y.co.type == TTS_HIP_F16 && !valu_mode && (H % 256 == 0);
            // many rows: the attention over the voice prompt inside the q projection's 64 x 64 tiles (gemm_tile_kernel<.., EPI_CROSS>), no launch of its own
            const bool try_fold = c->cross_fold && c->attn_short && c->E >= 1 && c->E <= 32 && H == c->NH * 64 && !c->debug;
            if (try_fold) {
                gq.cross_k = (const float *) (c->cross_kv_ptr() + ((size_t) l * 2 + 0) * c->ECAP * H * 4);
                gq.cross_v = (const float *) (c->cross_kv_ptr() + ((size_t) l * 2 + 1) * c->ECAP * H * 4);
                gq.cross_E = c->E; gq.cross_scale = at.scale;
                gq.cross_out = c->att; gq.cross_out16 = co_half_ ? c->att16 : nullptr;
            }
            // quantised out projection on the tiled path: the attended rows leave the fold as Q8_0 blocks (second set), not as fp32
            const bool q_fuse = c->qtile_fuse && !c->debug && c->adT && R >= c->qtile_min_rows && c->qtile_min_rows > 0;
            if (try_fold && q_fuse && y.co.type == TTS_HIP_Q8I && y.co.stoff && (int) y.co.K == H) c->qtile_out = QTileOut{c->aq2, c->adT2, H, c->att};
            CHK(run_gemm(c, TTS_HIP_K_GEMM_CROSS_Q, y.cq, gq, PRO_LN, try_fold ? EPI_CROSS : EPI_STORE));
            const bool folded = try_fold && c->cross_folded;
            AttnArgs ac{};
            ac.q = c->q;
            ac.kc = c->cross_kv_ptr() + ((size_t) l * 2 + 0) * c->ECAP * H * 4;
            ac.vc = c->cross_kv_ptr() + ((size_t) l * 2 + 1) * c->ECAP * H * 4;
            ac.kv_f16 = 0; ac.seq_stride = 0; ac.row_seq = nullptr; ac.row_pos = nullptr; ac.T_fixed = c->E;
            const bool co_half = y.co.type == TTS_HIP_F16 && !valu_mode && (H % 256 == 0);
            ac.H = H; ac.n_heads = c->NH; ac.scale = at.scale; ac.out = c->att; ac.out16 = co_half ? c->att16 : nullptr; ac.part = c->part;
            ac.stamps = stamp_slot();
            // one-sequence chain: the attention inside the out projection's prologue (gemm16_kernel<.., PRO_CROSS, ..>), no launch of its own
            const bool b1_fold = !folded && c->cross_fold && c->attn_short && R <= 4 && co_half && c->E >= 1 && c->E <= 32 && H == c->NH * 64 && H <= 2048 &&
                                 (int) y.co.K == H && !c->gemv_rows && !c->debug && !c->prof;
            if (!folded && !b1_fold) CHK(run_attn(c, TTS_HIP_K_ATTN_CROSS, ac, R, 1, 2.0 * c->E * H * 4));
            GemmArgs gc = go;
            gc.stamps = stamp_slot();
            gc.A = co_half ? (const void *) c->att16 : (const void *) c->att;
            if (b1_fold) {
                gc.A = c->q; gc.lda = H;
                gc.cross_k = (const float *) ac.kc; gc.cross_v = (const float *) ac.vc; gc.cross_E = c->E; gc.cross_scale = ac.scale;
            }
            CHK(run_gemm(c, TTS_HIP_K_GEMM_CROSS_OUT, y.co, gc, b1_fold ? PRO_CROSS : (co_half ? PRO_F16 : PRO_F32), EPI_RESID));
        }

        // FFN ------------------------------------------------------------------------------
        GemmArgs g1{};
        g1.R = R; g1.H = H; g1.gelu_mode = (int) c->d.gelu_mode; g1.A = c->x; g1.lda = H;
        g1.ln_w = (const float *) (c->arena + y.f_w); g1.ln_b = (const float *) (c->arena + y.f_b);
        const bool u_half = (y.fc2.type == TTS_HIP_F16) && !(c->d.flags & TTS_HIP_FLAG_VALU_GEMM) && (c->F % 256 == 0);
        g1.out = c->u32; g1.out16 = u_half ? c->u16 : nullptr; g1.ldo = c->F;
        g1.stamps = stamp_slot();
        if (c->qtile_fuse && !c->debug && c->adT && c->qtile_min_rows > 0 && R >= c->qtile_min_rows && y.fc2.type == TTS_HIP_Q8I && y.fc2.stoff && (int) y.fc2.K == c->F)
            c->qtile_out = QTileOut{c->aq2, c->adT2, c->F, c->u32};   // fc2's input rows as Q8_0 blocks straight from the GELU epilogue
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

// the block scales of the Parler decoder's quantised matrices once more, transposed, for qgemm_tile_kernel (arena space reserved by the planner)
static int derive_scale_tables(tts_hip_ctx *c) {
    if (!c->has_parler) return 0;
    HIPCHK(hipSetDevice(c->device));
    for (const PLayer &y : c->layers)
        for (const W *w : {&y.qkv, &y.o, &y.cq, &y.ck, &y.cv, &y.co, &y.fc1, &y.fc2}) CHK(qtile_transpose_scales(c, *w));
    CHK(qtile_transpose_scales(c, c->heads));
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
    if (external_arena) { c->arena = (char *) external_arena; c->arena_external = true; arena_share(c); }
    else { HIPCHK(hipMalloc((void **) &c->arena, c->arena_bytes)); arena_own(c); }
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
        if (any_qtile(c)) {
            c->ldr = (R + 255) & ~255;
            CHK(dmalloc(&c->adT, (size_t) c->ldr * (std::max(H, c->F) / 32)));
            CHK(dmalloc(&c->aq2, (size_t) R * std::max(H, c->F)));
            CHK(dmalloc(&c->adT2, (size_t) c->ldr * (std::max(H, c->F) / 32)));
        }
        CHK(dmalloc(&c->logits, (size_t) R * c->NO * c->V));
        CHK(dmalloc(&c->part, (size_t) R * c->NH * 16 * ATT_PS));
        CHK(dmalloc(&c->attn_cnt, (size_t) R * c->NH));
        if (c->b1_stamps_want || (getenv("TTS_HIP_B1_STAMPS") && atoi(getenv("TTS_HIP_B1_STAMPS")))) CHK(dmalloc(&c->b1_stamps, (size_t) 16 * (c->L * 8 + 8)));
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
        const size_t S = std::max<uint32_t>(1, c->lm.max_seqs);                 // cache slots of lock-step utterances; [slot][layer][position][kv width]
        const size_t kvb = S * c->L * NCTX * c->l_kvH;
        c->attn_part_cap = (size_t) 4 * c->NH * 16;   // up to 4 rows x heads x 16 splits (more rows take the unsplit kernel)
        CHK(dmalloc(&c->attn_part, c->attn_part_cap * ATTN_PART));
        CHK(dmalloc(&c->attn_cnt, (size_t) 256 + c->NH));   // arrival counters of the split attention's in-kernel merge (rows x heads <= 256)
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
        CHK(dmalloc(&c->l_logits, S * (size_t) c->l_Vpad));   // one row per utterance of a lock-step step
        CHK(dmalloc(&c->l_seq, (size_t) R));
        CHK(dmalloc(&c->l_btok, S));
        CHK(dmalloc(&c->l_bpv, S * ARGMAX_PARTS));
        CHK(dmalloc(&c->l_bpi, S * ARGMAX_PARTS));
        CHK(dmalloc(&c->l_bsmp, S * 3));
        CHK(dmalloc(&c->dbg, (size_t) R * std::max(H, F)));
        CHK(dmalloc(&c->aq, (size_t) R * std::max(std::max(H, F), c->NH * (int) c->lm.head_dim)));
        CHK(dmalloc(&c->ad, (size_t) R * std::max(std::max(H, F), c->NH * (int) c->lm.head_dim) / 32 + 1));
        CHK(dmalloc(&c->l_ids, (size_t) R));
        CHK(dmalloc(&c->l_pos, (size_t) R));
        CHK(dmalloc(&c->l_tok, (size_t) 1 + 2 * ARGMAX_PARTS + LLAMA_GREEDY_CHUNK + 1));
        HIPCHK(hipMalloc((void **) &c->l_cand, (size_t) TOPK_PARTS * TOPK_MAXK * 8 + 16));   // + the softmax total of a top_p < 1 step
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
    if (c->weights_present) CHK(derive_scale_tables(c));   // (a declare-only context on a shared arena finds the tables of the context that filled it)
    if (c->weights_present) CHK(compute_cross_kv(c));
    return 0;
}

extern "C" int tts_hip_arena_filled(tts_hip_ctx *c) {
    if (!c || !c->finalized) return set_err("tts_hip_arena_filled: context not finalized");
    c->weights_present = true;
    return 0;   // the arena image of a finalized context carries its transposed scale tables (derive_scale_tables) and cross K / V
}

// ------------------------------------------------------------------------------------------------
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
int ready(tts_hip_ctx *c, const char *who) {
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
            f.orig = c->d_seq; f.R_total = c->gen_total; f.dummy = c->gs.active ? (int) c->gs.n_slots : -1; f.max_steps = c->gs.active ? c->gs.max_steps : 0;
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
    if (c->gen_total != (int) n || c->gs_graphs) {   // the utterance count (and a stream's padding slot) is baked into the captured sampler / feed launches
        drop_gen_graphs(c);
        c->gen_total = (int) n;
        c->gs_graphs = false;
    }
    c->gs = tts_hip_ctx::GenStream{};   // a batch generation ends any stream of this context
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

int stage_uniforms(tts_hip_ctx *c, const float *uniforms, size_t count) {
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
int stage_penalty(tts_hip_ctx *c, float penalty, int n) {
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

// ------------------------------------------------------------------------------------------------
// Continuous batching (the queue of examples/server/server.cpp:126-158 widened): a generation that takes new utterances
// into freed rows while the others keep running.  generate_loop above runs ONE batch to its end: an utterance arriving a step later waits a
// whole generation, and a ragged batch idles (or, compacted, shrinks) instead of refilling.  Here the loop is cut at the host's 32-step
// look-in points into calls:
//   stream_begin(n_slots, max_steps, ...)   fixes the slot count (cache slots 0 .. n_slots-1; slot n_slots is the padding slot: the context
//                                           needs max_seqs >= n_slots + 1), the token buffer [max_steps][n_slots + 1][heads] and the sampler
//   stream_admit(slots, prompts ...)        prefills the newcomers as a side batch into their cache slots and enters them as rows at step 1
//   stream_run(n_steps)                     n_steps lock-step forwards over the live rows (padded to the row counts generate_loop captures:
//                                           multiples of 64, of 128 from 256 rows; padding rows sit on the padding slot and record nothing),
//                                           then reports the slots whose check_stopping() fired and frees them
//   stream_collect(slot, steps)             the finished utterance's tokens [steps][heads]
// Per-slot state (EOS flags, steps_done, sampler history, uniforms, token column) is indexed by slot, per-row state (ids, position, step
// counter) lives on the host between calls: the prefill of a newcomer uses the same device staging arrays as a step.
// An utterance's tokens do not depend on when it was admitted or on its neighbours: every kernel of the forward works row by row, and the
// GEMM tile a row count selects is pinned by the row-count rounding (tests/test_gpu_runner.py: admitted mid-flight == solo run).
// ------------------------------------------------------------------------------------------------
extern "C" int tts_hip_parler_stream_begin(tts_hip_ctx *c, uint32_t n_slots, uint32_t max_steps, uint32_t bos, uint32_t eos, const tts_hip_sampling *sp) {
    CHK(ready(c, "tts_hip_parler_stream_begin"));
    if (n_slots == 0 || n_slots + 1 > c->d.max_seqs || (int) n_slots > c->RMAX) return set_err("stream_begin: %u slots need a context with max_seqs >= %u (have %u) and <= %d rows", n_slots, n_slots + 1, c->d.max_seqs, c->RMAX);
    if (max_steps == 0) return set_err("stream_begin: max_steps == 0");
    if (bos >= (uint32_t) c->EROWS || eos >= (uint32_t) c->EROWS) return set_err("stream_begin: bos/eos outside the embedding table");
    auto &g = c->gs;
    g = tts_hip_ctx::GenStream{};
    g.n_slots = n_slots; g.max_steps = max_steps; g.bos = bos; g.eos = eos; g.sampled = sp != nullptr;
    const uint32_t RT = n_slots + 1;
    if (sp) {
        CHK(check_sampling(c, sp, "tts_hip_parler_stream_begin"));
        if (sp->top_k != c->smp.top_k || sp->top_p != c->smp.top_p || sp->temperature != c->smp.temperature ||
            (sp->repetition_penalty != 1.0f) != (c->smp.repetition_penalty != 1.0f))
            drop_gen_graphs(c);
        c->smp = *sp;
        const size_t count = (size_t) (max_steps + 1) * RT * c->NO;   // + 1: a row that spent its budget inside a chunk still reads the plane after its last
        if (count > c->uniforms_cap) {
            free_dev(c->d_uniforms); c->d_uniforms = nullptr;
            HIPCHK(hipMalloc((void **) &c->d_uniforms, count * 4));
            c->uniforms_cap = count;
            drop_gen_graphs(c);
        }
        HIPCHK(hipMemsetAsync(c->d_uniforms, 0, count * 4, c->stream));
        CHK(stage_penalty(c, sp->repetition_penalty, (int) max_steps));
    }
    const size_t need = (size_t) max_steps * RT * c->NO;
    if (need > c->tokens_out_cap) {
        free_dev(c->d_tokens_out); c->d_tokens_out = nullptr;
        HIPCHK(hipMalloc((void **) &c->d_tokens_out, need * 4));
        c->tokens_out_cap = need;
        drop_gen_graphs(c);
    }
    if (c->g_bos != bos || c->g_eos != eos) { drop_gen_graphs(c); c->g_bos = bos; c->g_eos = eos; }
    if (c->gen_total != (int) RT || !c->gs_graphs || c->gs_baked_steps != max_steps) { drop_gen_graphs(c); c->gen_total = (int) RT; c->gs_graphs = true; c->gs_baked_steps = max_steps; }   // feed_kernel's padding slot is baked in too
    HIPCHK(hipMemsetAsync(c->d_eos, 0, (size_t) RT * c->NO, c->stream));
    HIPCHK(hipMemsetAsync(c->d_steps_done, 0, (size_t) RT * 4, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    g.slot_live.assign(n_slots, 0);
    g.active = true;
    return 0;
}

extern "C" int tts_hip_parler_stream_admit(tts_hip_ctx *c, uint32_t n, const uint32_t *slots, const uint32_t *ids, const uint32_t *lens, const float *uniforms) {
    CHK(ready(c, "tts_hip_parler_stream_admit"));
    auto &g = c->gs;
    if (!g.active) return set_err("stream_admit: no stream (tts_hip_parler_stream_begin)");
    if (n == 0) return 0;
    if (!slots || !ids || !lens) return set_err("stream_admit: null argument");
    if (g.sampled && !uniforms) return set_err("stream_admit: a sampled stream needs the utterances' uniforms [n][max_steps][heads]");
    const uint32_t RT = g.n_slots + 1, max_pos = (uint32_t) std::min(c->KVPOS, c->NPOS);
    for (uint32_t i = 0; i < n; i++) {
        if (slots[i] >= g.n_slots) return set_err("stream_admit: slot %u >= %u", slots[i], g.n_slots);
        if (g.slot_live[slots[i]]) return set_err("stream_admit: slot %u is still generating", slots[i]);
        for (uint32_t j = 0; j < i; j++) if (slots[j] == slots[i]) return set_err("stream_admit: slot %u named twice", slots[i]);
        if (lens[i] == 0 || lens[i] >= max_pos) return set_err("stream_admit: prompt of %u ids leaves no room in %u cached positions", lens[i], max_pos);
    }
    {   // The newcomers' prompts as one side batch (prefill_batch's rows), padded with rows of the padding slot to the row counts the decode steps
        // use (multiples of 64): a lone 5-id prompt would otherwise run through the small-batch GEMM kernels (other summation order than the
        // tiled ones its neighbours' prompts went through), and the utterance's tokens would depend on how many others arrived with it.
        std::vector<uint32_t> rs, rp, ri;
        size_t off = 0;
        for (uint32_t i = 0; i < n; i++) {
            for (uint32_t j = 0; j < lens[i]; j++) {
                if (ids[off + j] >= (uint32_t) c->PV) return set_err("stream_admit: text id %u >= prompt vocab %d", ids[off + j], c->PV);
                rs.push_back(slots[i]); rp.push_back(j); ri.push_back(ids[off + j]);
            }
            off += lens[i];
        }
        for (size_t o = 0; o < rs.size(); o += (size_t) c->RMAX) {
            const int real = (int) std::min<size_t>((size_t) c->RMAX, rs.size() - o);
            const int R = std::min(c->RMAX, (real + 63) / 64 * 64);
            c->host_pos.resize(R);
            for (int r = 0; r < R; r++) {
                const bool pad = r >= real;
                c->h_ids[r] = pad ? ri[o] : ri[o + r];
                c->h_pos[r] = pad ? (uint32_t) ((r - real) % (int) max_pos) : rp[o + r];
                c->h_seq[r] = pad ? g.n_slots : rs[o + r];
                c->host_pos[r] = c->h_pos[r];
            }
            HIPCHK(hipMemcpyAsync(c->d_ids, c->h_ids, (size_t) R * 4, hipMemcpyHostToDevice, c->stream));
            HIPCHK(hipMemcpyAsync(c->d_pos, c->h_pos, (size_t) R * 4, hipMemcpyHostToDevice, c->stream));
            HIPCHK(hipMemcpyAsync(c->d_seq, c->h_seq, (size_t) R * 4, hipMemcpyHostToDevice, c->stream));
            CHK(enqueue_forward(c, R, /*audio=*/false, /*logits=*/false, /*same_seq=*/false));
            HIPCHK(hipStreamSynchronize(c->stream));
        }
    }
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t s = slots[i];
        HIPCHK(hipMemsetAsync(c->d_eos + (size_t) s * c->NO, 0, (size_t) c->NO, c->stream));
        HIPCHK(hipMemsetAsync(c->d_steps_done + s, 0, 4, c->stream));
        if (g.sampled) {
            if (c->smp.repetition_penalty != 1.0f) {   // sampler::reset for this utterance
                HIPCHK(hipMemsetAsync(c->d_last + (size_t) s * c->NO, 0xFF, (size_t) c->NO * 4, c->stream));
                HIPCHK(hipMemsetAsync(c->d_repc + (size_t) s * c->NO, 0, (size_t) c->NO * 4, c->stream));
            }
            // the utterance's draws [max_steps][heads] into its column of [max_steps][slots + 1][heads]
            HIPCHK(hipMemcpy2DAsync(c->d_uniforms + (size_t) s * c->NO, (size_t) RT * c->NO * 4, uniforms + (size_t) i * g.max_steps * c->NO, (size_t) c->NO * 4,
                                    (size_t) c->NO * 4, g.max_steps, hipMemcpyHostToDevice, c->stream));
        }
        g.slot_live[s] = 1;
        g.row_slot.push_back(s);
        g.pos.push_back(lens[i]);
        g.step.push_back(1);
        for (int h = 0; h < c->NO; h++) g.ids.push_back(g.bos);
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int tts_hip_parler_stream_run(tts_hip_ctx *c, uint32_t n_steps, uint32_t *n_finished, uint32_t *finished_slots, uint32_t *finished_steps) {
    CHK(ready(c, "tts_hip_parler_stream_run"));
    auto &g = c->gs;
    if (!g.active) return set_err("stream_run: no stream (tts_hip_parler_stream_begin)");
    if (!n_finished || !finished_slots || !finished_steps) return set_err("stream_run: null argument");
    *n_finished = 0;
    const uint32_t live = (uint32_t) g.row_slot.size();
    if (live == 0 || n_steps == 0) return 0;
    const uint32_t q = live >= 256 ? 128 : 64;
    const uint32_t R = std::min<uint32_t>((uint32_t) c->RMAX, (live + q - 1) / q * q);
    const int NO = c->NO;
    for (uint32_t r = 0; r < R; r++) {
        const bool pad = r >= live;
        for (int h = 0; h < NO; h++) c->h_ids[(size_t) r * NO + h] = pad ? g.bos : g.ids[(size_t) r * NO + h];
        c->h_pos[r] = pad ? 0 : g.pos[r];
        c->h_seq[r] = pad ? g.n_slots : g.row_slot[r];
        c->h_tok[r] = pad ? 1 : g.step[r];
    }
    HIPCHK(hipMemcpyAsync(c->d_ids, c->h_ids, (size_t) R * NO * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_pos, c->h_pos, (size_t) R * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_seq, c->h_seq, (size_t) R * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_step, c->h_tok, (size_t) R * 4, hipMemcpyHostToDevice, c->stream));
    const int mode = g.sampled ? MODE_GEN_SAMPLE : MODE_GEN;
    c->host_pos.resize(R);
    const uint32_t max_pos = (uint32_t) std::min(c->KVPOS, c->NPOS);
    for (uint32_t s = 0; s < n_steps; s++) {
        for (uint32_t r = 0; r < R; r++) c->host_pos[r] = r < live ? std::min(g.pos[r] + s, max_pos - 1) : 0;
        CHK(run_step(c, (int) R, mode, g.bos, g.eos));
    }
    // the rows' state back to the host (the next admit's prefill reuses the device staging arrays), and who has finished
    HIPCHK(hipMemcpyAsync(c->h_ids, c->d_ids, (size_t) live * NO * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(c->h_pos, c->d_pos, (size_t) live * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(c->h_seq, c->d_step, (size_t) live * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(c->h_tok, c->d_steps_done, (size_t) (g.n_slots + 1) * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    std::vector<uint32_t> slot2, pos2, step2, ids2;
    for (uint32_t r = 0; r < live; r++) {
        const uint32_t sl = g.row_slot[r];
        uint32_t done = c->h_tok[sl];
        if (done) {
            finished_slots[*n_finished] = sl;
            finished_steps[*n_finished] = std::min(done, g.max_steps);
            (*n_finished)++;
            g.slot_live[sl] = 0;
            continue;
        }
        slot2.push_back(sl); pos2.push_back(c->h_pos[r]); step2.push_back(c->h_seq[r]);
        ids2.insert(ids2.end(), c->h_ids + (size_t) r * NO, c->h_ids + (size_t) (r + 1) * NO);
    }
    g.row_slot.swap(slot2); g.pos.swap(pos2); g.step.swap(step2); g.ids.swap(ids2);
    return 0;
}

extern "C" int tts_hip_parler_stream_collect(tts_hip_ctx *c, uint32_t slot, uint32_t steps, uint32_t *tokens_out) {
    CHK(ready(c, "tts_hip_parler_stream_collect"));
    auto &g = c->gs;
    if (!g.active) return set_err("stream_collect: no stream");
    if (slot >= g.n_slots || steps > g.max_steps || !tokens_out) return set_err("stream_collect: slot %u / %u steps out of range", slot, steps);
    if (steps == 0) return 0;
    const size_t RT = g.n_slots + 1;
    HIPCHK(hipMemcpy2DAsync(tokens_out, (size_t) c->NO * 4, c->d_tokens_out + (size_t) slot * c->NO, RT * c->NO * 4, (size_t) c->NO * 4, steps, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int tts_hip_parler_stream_end(tts_hip_ctx *c) {
    if (!c) return set_err("null ctx");
    c->gs = tts_hip_ctx::GenStream{};
    return 0;
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
    if (c->has_llama && w == "l_logits") {   // the logits row the last Llama step left (row 0 of l_logits)
        const size_t n = std::min(max_floats, (size_t) c->l_V);
        if (hipMemcpy(out, c->l_logits, n * 4, hipMemcpyDeviceToHost) != hipSuccess) { set_err("copy failed"); return -1; }
        return (int64_t) n;
    }
    if (c->has_llama && w.size() > 4 && w[0] == 'l' && w[1] == '_' && (w[2] == 'k' || w[2] == 'v') && w[3] == ':') {   // "l_k:<layer>": slot 0's cache rows of a layer
        const int layer = atoi(w.c_str() + 4);
        if (layer < 0 || layer >= c->L) { set_err("debug_read(%s): bad layer", what); return -1; }
        const size_t per = (size_t) c->lm.n_ctx * c->l_kvH, n = std::min(max_floats, per);
        if (hipMemcpy(out, (w[2] == 'k' ? c->l_kc : c->l_vc) + (size_t) layer * per, n * 4, hipMemcpyDeviceToHost) != hipSuccess) { set_err("copy failed"); return -1; }
        return (int64_t) n;
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


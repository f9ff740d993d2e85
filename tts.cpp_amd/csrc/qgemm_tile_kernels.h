// qgemm_tile_kernels.h — LDS-tiled integer block GEMM for forwards that carry many rows and GGUF-quantised matrices.
//
//   y[r][n] = sum_b (float) sumi(r, n, b) * (d_w[n][b] * d_a[r][b]),     sumi = sum_{j < 32} q_w[n][32 b + j] * q_a[r][32 b + j]
//
// ggml_mul_mat with a Q4_0 / Q5_0 / Q8_0 weight (the reference's quantizer turns every decoder matrix into one of them,
// /root/reference/examples/quantize/quantize_impl.cpp:51-67; its best published configuration is Q5_0, README.md:103): the activation row
// is converted to Q8_0 blocks and every 32-wide block contributes an exact integer dot times the two fp16 block scales
// (ggml_vec_dot_q*_q8_0; oracle/tts_oracle.c dot_q_q8).  qgemm16_kernel (parler_kernels.h) does this for up to 256 rows with 16 features
// per workgroup; here a workgroup owns a BM x BN tile of the output like gemm_tile_kernel:
//   * a quantisation block is exactly one v_mfma_i32_32x32x32_i8 (K = 32): 32 features x 32 rows, 16 i32 results per lane.  Block sums
//     cannot be accumulated on the matrix pipe (every block has its own pair of scales), so each MFMA is followed by the scaling on
//     the vector pipe: u = d_w * d_a (exact in fp32: two 11-bit significands), acc = fma((float) sumi, u, acc) — ggml's
//     sumf += sumi * (d_w * d_a) with the product unrounded.  That is the kernel's bound: 24 vector instructions per 32-cycle... MFMA
//     (8 v_pk_add_f32 + 8 v_pk_mul_f32 + 8 v_pk_fma_f32, 4 cycles each), i.e. the matrix pipe idles two thirds of the time;
//   * int -> float without v_cvt: the MFMA's C operand is the integer 0x4B400000 in every element, so the result read as a float is
//     12582912 + sumi exactly (|sumi| <= 32 * 127 * 128 < 2^22), and one packed subtract yields (float) sumi for two results;
//   * the weight tile is laid into LDS with its 32-feature groups permuted (LDS row 8 g + 4 h + j holds feature 16 h + 4 g + j), so that
//     the MFMA's result layout (lane half h holds result rows 8 g + 4 h + j in register 4 g + j) leaves every lane with 16 CONSECUTIVE
//     features of one activation row: the 16 weight scales of a lane are one contiguous 64-byte LDS read (the same for the 32 lanes of
//     a half: a broadcast), the epilogue stores 64 contiguous bytes per lane, and a Q8_0 block of the OUTPUT (fused re-quantisation for
//     the next quantised matrix) is the lane pair (l, l ^ 32);
//   * scales travel with their k-tile: the weight scales are kept transposed in the arena (float [K/32][N_pad], written once by
//     transpose_scales_kernel at finalize), the activation scales are produced transposed (float [K/32][ldr]) by the kernels that
//     quantise rows for this path, so a k-tile's scales are a few 1-KiB global_load_lds pieces in the same in-order pipeline as the codes;
//   * staging, XOR swizzle, XCD-aware tile map and split-K slabs as gemm_tile_kernel.
// Summation order: blocks in k order inside a workgroup's k range, slabs in slab order (ln_rows folds them) — the reference adds block
// terms in k order too; the tolerance against the oracle is the one of the 16-feature kernel (tests/test_gpu_parler.py).
#pragma once
#include "gemm_tile_kernels.h"

typedef int int16v __attribute__((ext_vector_type(16)));
typedef float float16v __attribute__((ext_vector_type(16)));

// weight block scales [N][nb] fp16 -> float [nb][ldw] (ldw >= N, a multiple of 256; the pad columns are zero).  Floats: the scaling step is bound by
// the vector instructions it issues, and 16 v_cvt_f32_f16 per block and 32-feature group were a sixth of them; the LDS reads the wider table costs are
// not (profiles/r06/qgemm_bench_*.txt).
static __global__ void transpose_scales_kernel(const _Float16 *wd, float *wdT, int N, int nb, int ldw) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (n >= ldw) return;
    wdT[(int64_t) b * ldw + n] = n < N ? (float) wd[(int64_t) n * nb + b] : 0.0f;
}

// activation rows -> Q8_0 blocks for the tiled kernel: q int8 [R][K], d float [K/32][ldr] (the fp16-rounded scale, transposed).
// Same arithmetic as quant_rows_q8_kernel (ggml's quantize_row_q8_0_ref).  One wave per row piece of 256 values: 4 values per lane.
static __global__ __launch_bounds__(256) void quant_rows_q8t_kernel(const float *x, int lda, int K, int8_t *q, float *dT, int ldr, int R) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = blockIdx.y, k = (blockIdx.x * 4 + w) * 256 + lane * 4;
    if (r >= R || k >= K) return;
    const float4v y = *(const float4v *) (x + (int64_t) r * lda + k);
    float dd;
    const unsigned qq = quant4_q8(y, dd);
    *(unsigned *) (q + (int64_t) r * K + k) = qq;
    if ((lane & 7) == 0) dT[(int64_t) (k >> 5) * ldr + r] = dd;
}

struct QTileArgs {
    GemmArgs g;             // K, N, R, epilogue fields; g.W = int8 codes [N][K]
    const float *wdT;       // weight block scales, float [K/32][ldw]
    int ldw;
    const int8_t *aq;       // quantised activations [R][K]
    const float *adT;       // activation block scales, float [K/32][ldr]
    int ldr;
    // EPI_GELU / EPI_CROSS with a quantised consumer: the result rows leave as Q8_0 blocks (q_out int8 [R][ldq], d_out float [ldq/32][ldr])
    int8_t *q_out;
    float *d_out;
    int ldq;
    long long *stamps;      // micro-benchmark only: per-phase shader-clock sums of every workgroup's wave 0 (wait + barrier, stage issue, compute, epilogue)
    int dbg;                // micro-benchmark only (profiles/qgemm_bench.hip): 1 = no k loop, 2 = no epilogue stores, 4 = no scaling, 8 = no MFMA, 16 = no LDS reads / compute at all (staging pipeline alone)
};

#define QT_MAGIC_I 0x4B400000
#define QT_MAGIC_F 12582912.0f

// a lane's 16 consecutive results of one row are half of a Q8_0 block, the other half sits in lane ^ 32: the block's scale (fp16-rounded,
// ggml's quantize_row_q8_0_ref) and this lane's 16 codes.  Every lane of the wave must call it (v_permlane32_swap).
__device__ __forceinline__ int4v quant16_pair(const float (&y)[16], float &d_out) {
    float amax = 0.0f;
#pragma unroll
    for (int e = 0; e < 16; e++) amax = fmaxf(amax, fabsf(y[e]));
    {
        const unsigned u = __builtin_bit_cast(unsigned, amax);
        const auto s = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        amax = fmaxf(__builtin_bit_cast(float, (unsigned) s[0]), __builtin_bit_cast(float, (unsigned) s[1]));
    }
    const float dd = amax / 127.0f;
    const float id = dd ? 1.0f / dd : 0.0f;
    int4v pk;
#pragma unroll
    for (int w4 = 0; w4 < 4; w4++) {
        unsigned p = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) p |= ((unsigned) (int) roundf(y[w4 * 4 + e] * id) & 0xFFu) << (8 * e);
        pk[w4] = (int) p;
    }
    d_out = (float) (_Float16) dd;
    return pk;
}

// S: LDS buffers (S - 1 k-tiles in flight); a k-tile is 128 codes = 4 quantisation blocks = one 128-byte line per row.
// LDS image of a k-tile: [BN rows of W codes | BM rows of activation codes | W scales float [4][BN] | activation scales float [4][BM]].
// WPE: waves per SIMD the register allocation must allow (the scaling is issue-bound: a wave issues one vector instruction per ~4.6 cycles, the
// SIMD executes one per ~2.4, so the vector pipe is only full with >= 2, better 4 waves per SIMD — profiles/valu_rate.hip);
// PIPE: the MFMA of item i + 1 is issued before the scaling of item i and the LDS reads of block b + 1 before the items of block b (two result
// sets, two operand sets) — for few waves per SIMD; with four the other waves fill those gaps and the registers are worth more.
// KG: k groups inside the workgroup — wave (kg, wm, wn) takes the blocks b = kg (mod KG) of every k-tile for its 32 x 32 fragments and the groups' sums
// meet in LDS at the end (in group order): four times the waves for a GEMM with few tiles (N = hidden size at 1024 rows: 256 tiles of 64 x 64)
// without split-K slabs, i.e. with its epilogue (residual add, the cross-attention fold) still in the launch.
template <int BM, int BN, int WM, int WN, int S, int EPI, int WPE = 2, bool PIPE = true, bool DBG = false, int KG = 1>
__global__ __launch_bounds__(WM * WN * KG * 64) __attribute__((amdgpu_waves_per_eu(WPE, WPE > 4 ? WPE : 4))) void qgemm_tile_kernel(QTileArgs qa, TileMap tm) {
    constexpr int NW = WM * WN * KG, NWT = WM * WN;
    static_assert(KG == 1 || ((KG == 2 || KG == 4) && !PIPE), "k groups");
    constexpr int MI = BM / WM / 32, NI = BN / WN / 32;   // 32 x 32 fragments per wave: rows x features
    constexpr int T = MI * NI;
    constexpr int ROWB = 128;
    constexpr int DGROUPS = (BM + BN) / 8;                // 1-KiB pieces of codes: 8 rows each
    constexpr int WSG = BN / 64, ASG = BM / 64;           // 1-KiB pieces of scales: [4][BN] / [4][BM] floats
    constexpr int GROUPS = DGROUPS + WSG + ASG;
    constexpr int GPW = (GROUPS + NW - 1) / NW, GPW_MIN = GROUPS / NW;
    constexpr int SW_OFF = (BM + BN) * ROWB, SA_OFF = SW_OFF + WSG * 1024;
    constexpr int STAGE = SA_OFF + BM * 16;
    static_assert(MI >= 1 && NI >= 1 && BM % 64 == 0 && BN % 64 == 0 && S >= 2 && GPW_MIN >= 1, "tile / wave shape");
    static_assert((S - 2) * GPW_MIN < 64, "vmcnt range");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const GemmArgs &a = qa.g;

    // ---- workgroup -> (row tile, feature tile, k slice), XCD-aware (gemm_tile_kernel) --------------------
    const int total = tm.m_tiles * tm.n_tiles * tm.k_slices;
    const int per_xcd = (int) (gridDim.x >> 3);
    const int v = (int) (blockIdx.x & 7) * per_xcd + (int) (blockIdx.x >> 3);
    if (v >= total) return;
    const int mt = v % tm.m_tiles;
    const int nt = (v / tm.m_tiles) % tm.n_tiles;
    const int kz = v / (tm.m_tiles * tm.n_tiles);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kg = wave / NWT, wt = wave % NWT;
    const int wm = wt / WN, wn = wt % WN;
    const int r0 = mt * BM, n0 = nt * BN;
    const int kc = a.kchunk ? a.kchunk : a.K;
    const int k0 = kz * kc;
    const int n_kt = (DBG && (qa.dbg & 1)) ? 0 : kc / 128;

    // ---- staging: wave w copies pieces w, w + NW, ... of [W codes | activation codes | W scales | activation scales] ----
    const char *src[GPW];
    int inc[GPW];
#pragma unroll
    for (int i = 0; i < GPW; i++) {
        const int g = min(wave + i * NW, GROUPS - 1);
        if (g < DGROUPS) {
            const int rowl = g * 8 + (lane >> 3);          // row of the [BN + BM]-row LDS image
            // the swizzle goes on the source address (the LDS side is lane-linear).  Key: bits 1, 3, 4 of the row — a ds_read_b128 is served in the
            // lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32), 16 lanes over 256 bytes: with 128-byte rows the row's parity picks the
            // half and the 8 even (odd) rows of a group must land in 8 different 16-byte slots; bits 1, 3, 4 tell them apart in both groups
            const int schunk = (lane & 7) ^ (((rowl >> 1) & 1) | (((rowl >> 3) & 3) << 1));
            if (rowl < BN) {
                // LDS row 8 g + 4 h + j of a 32-feature group holds feature 16 h + 4 g + j
                const int x = rowl & 31, f = (rowl & ~31) + 16 * ((x >> 2) & 1) + 4 * (x >> 3) + (x & 3);
                const int n = min(n0 + f, a.N - 1);
                src[i] = (const char *) a.W + (int64_t) n * a.K + k0 + schunk * 16;
            } else {
                const int r = min(r0 + rowl - BN, a.R - 1);
                src[i] = (const char *) qa.aq + (int64_t) r * a.K + k0 + schunk * 16;
            }
            inc[i] = 128;
        } else if (g < DGROUPS + WSG) {
            const int o = (g - DGROUPS) * 256 + lane * 4;                // float index inside the [4][BN] image
            src[i] = (const char *) (qa.wdT + (int64_t) ((k0 >> 5) + o / BN) * qa.ldw + n0 + o % BN);
            inc[i] = 4 * qa.ldw * 4;
        } else {
            const int o = (g - DGROUPS - WSG) * 256 + lane * 4;          // float index inside the [4][BM] image
            src[i] = (const char *) (qa.adT + (int64_t) ((k0 >> 5) + o / BM) * qa.ldr + r0 + o % BM);
            inc[i] = 4 * qa.ldr * 4;
        }
    }
    auto stage = [&](int buf, int kt) {
#pragma unroll
        for (int i = 0; i < GPW; i++) {
            const int g = wave + i * NW;
            if (GROUPS % NW != 0 && g >= GROUPS) break;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (src[i] + (int64_t) kt * inc[i]),
                                             (__attribute__((address_space(3))) void *) (smem + buf * STAGE + g * 1024), 16, 0, 0);
        }
    };

    // ---- fragment read offsets ------------------------------------------------------------------------
    const int fl = lane & 31, fh = lane >> 5;
    int woff[NI], aoff[MI];
#pragma unroll
    for (int ni = 0; ni < NI; ni++) woff[ni] = (wn * (BN / WN) + ni * 32 + fl) * ROWB;
#pragma unroll
    for (int mi = 0; mi < MI; mi++) aoff[mi] = (BN + wm * (BM / WM) + mi * 32 + fl) * ROWB;
    const int sw = ((fl >> 1) & 1) | (((fl >> 3) & 3) << 1);

    float acc[NI][MI][16];
#pragma unroll
    for (int ni = 0; ni < NI; ni++)
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[ni][mi][e] = 0.f;
    int16v magic;
#pragma unroll
    for (int e = 0; e < 16; e++) magic[e] = QT_MAGIC_I;

    // One k-tile = 4 blocks x T fragments = 4 T "items", each one MFMA followed by its 24 scaling instructions.  The MFMA of item i + 1 is
    // issued before the scaling of item i (two result sets), the LDS reads of block b + 1 before the items of block b (two operand sets): the
    // matrix pipe and the LDS work under the vector pipe's instructions, which are the bound.
    struct Frag { int4v wf[NI], af[MI]; float4v wd[NI][4]; float ad[MI]; };
    auto load_frag = [&](const char *base, int b, Frag &f) {
        const int coff = ((b * 2 + fh) ^ sw) * 16;
#pragma unroll
        for (int ni = 0; ni < NI; ni++) f.wf[ni] = *(const int4v *) (base + woff[ni] + coff);
#pragma unroll
        for (int mi = 0; mi < MI; mi++) f.af[mi] = *(const int4v *) (base + aoff[mi] + coff);
#pragma unroll
        for (int ni = 0; ni < NI; ni++) {
            const float4v *p = (const float4v *) (base + SW_OFF + (b * BN + wn * (BN / WN) + ni * 32 + 16 * fh) * 4);
#pragma unroll
            for (int e = 0; e < 4; e++) f.wd[ni][e] = p[e];
        }
#pragma unroll
        for (int mi = 0; mi < MI; mi++) f.ad[mi] = *(const float *) (base + SA_OFF + (b * BM + wm * (BM / WM) + mi * 32 + fl) * 4);
    };
    auto scale_item = [&](const float16v &zz, const float (&wd)[16], float ad, float (&ac)[16]) {
        if (DBG && (qa.dbg & 4)) { ac[0] += zz[0] + zz[5] + zz[15] + wd[3] + ad; return; }
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const float u = wd[e] * ad;                       // exact: two 11-bit significands
            ac[e] = __builtin_fmaf(zz[e] - QT_MAGIC_F, u, ac[e]);   // (float) sumi * (d_w * d_a) + acc
        }
    };
    // 12582912 + block dot when read as floats.  (The whole vector is cast: __builtin_bit_cast(float, z[e]) on an element of an ext-vector
    // lvalue reads element 0 for every e with this compiler.)
    auto block_dot = [&](const int4v &wf, const int4v &af) { if (DBG && (qa.dbg & 8)) { float16v t = __builtin_bit_cast(float16v, magic); t[0] = __builtin_bit_cast(float, wf[0] ^ af[1]); t[7] = __builtin_bit_cast(float, wf[3] ^ af[2]); return t; } return __builtin_bit_cast(float16v, __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, af, magic, 0, 0, 0)); };
    auto compute = [&](int buf) {
        const char *base = smem + buf * STAGE;
        if constexpr (PIPE) {
            Frag fr[2];
            float16v z[2];
            load_frag(base, 0, fr[0]);
            z[0] = block_dot(fr[0].wf[0], fr[0].af[0]);
#pragma unroll
            for (int b = 0; b < 4; b++) {
                if (b + 1 < 4) load_frag(base, b + 1, fr[(b + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);   // the reads stay here: their wait then sits in front of the first MFMA of block b + 1, a block of scaling later
                const Frag &f = fr[b & 1];
#pragma unroll
                for (int ni = 0; ni < NI; ni++) {
                    float wd[16];
#pragma unroll
                    for (int e = 0; e < 16; e++) wd[e] = f.wd[ni][e >> 2][e & 3];
#pragma unroll
                    for (int mi = 0; mi < MI; mi++) {
                        const int it = b * T + ni * MI + mi;   // this item; the next one's MFMA goes first
                        if (it + 1 < 4 * T) {
                            const int b2 = (it + 1) / T, ni2 = ((it + 1) % T) / MI, mi2 = (it + 1) % MI;
                            z[(it + 1) & 1] = block_dot(fr[b2 & 1].wf[ni2], fr[b2 & 1].af[mi2]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        scale_item(z[it & 1], wd, f.ad[mi], acc[ni][mi]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        } else {
#pragma unroll
            for (int bb = 0; bb < 4 / KG; bb++) {
                const int b = bb * KG + kg;
                Frag f;
                load_frag(base, b, f);
#pragma unroll
                for (int ni = 0; ni < NI; ni++) {
                    float wd[16];
#pragma unroll
                    for (int e = 0; e < 16; e++) wd[e] = f.wd[ni][e >> 2][e & 3];
#pragma unroll
                    for (int mi = 0; mi < MI; mi++) scale_item(block_dot(f.wf[ni], f.af[mi]), wd, f.ad[mi], acc[ni][mi]);
                }
            }
        }
    };

    // ---- S-buffer pipeline (gemm_tile_kernel) -----------------------------------------------------------
#pragma unroll
    for (int t = 0; t < S - 1; t++)
        if (t < n_kt) stage(t, t);
    int cur = 0, nxt = S - 1;
    long long t_wait = 0, t_stage = 0, t_comp = 0, t0 = DBG && qa.stamps ? (long long) __builtin_amdgcn_s_memtime() : 0;
    for (int kt = 0; kt < n_kt; kt++) {
        if (kt + S - 2 < n_kt) wait_vmcnt<(S - 2) * GPW_MIN>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (DBG && qa.stamps) { const long long t = __builtin_amdgcn_s_memtime(); t_wait += t - t0; t0 = t; }
        if (kt + S - 1 < n_kt) stage(nxt, kt + S - 1);
        if (DBG && qa.stamps) { const long long t = __builtin_amdgcn_s_memtime(); t_stage += t - t0; t0 = t; }
        if (!(DBG && (qa.dbg & 16))) compute(cur);
        if (DBG && qa.stamps) { const long long t = __builtin_amdgcn_s_memtime(); t_comp += t - t0; t0 = t; }
        cur = cur + 1 == S ? 0 : cur + 1;
        nxt = nxt + 1 == S ? 0 : nxt + 1;
    }
    if (DBG && qa.stamps && tid == 0) { qa.stamps[v * 4 + 0] = t_wait; qa.stamps[v * 4 + 1] = t_stage; qa.stamps[v * 4 + 2] = t_comp; }

    if constexpr (KG > 1) {
        // the k groups' sums meet in LDS: groups 1 .. KG - 1 write, group 0 adds them in group order
        __syncthreads();   // every wave is done with the k-tile buffers (and no piece is in flight: the last tile waited for vmcnt(0))
        float *red = (float *) smem;   // [KG - 1][NWT][T * 16][64]
        if (kg > 0) {
#pragma unroll
            for (int ni = 0; ni < NI; ni++)
#pragma unroll
                for (int mi = 0; mi < MI; mi++)
#pragma unroll
                    for (int e = 0; e < 16; e++) red[((((kg - 1) * NWT + wt) * T + ni * MI + mi) * 16 + e) * 64 + lane] = acc[ni][mi][e];
        }
        __syncthreads();
        if (kg == 0) {
#pragma unroll
            for (int g = 1; g < KG; g++)
#pragma unroll
                for (int ni = 0; ni < NI; ni++)
#pragma unroll
                    for (int mi = 0; mi < MI; mi++)
#pragma unroll
                        for (int e = 0; e < 16; e++) acc[ni][mi][e] += red[((((g - 1) * NWT + wt) * T + ni * MI + mi) * 16 + e) * 64 + lane];
        }
        if (EPI != EPI_CROSS && kg > 0) return;
    }
    if (DBG && (qa.dbg & 2) && acc[0][0][0] != 1.2345f) return;
    // ---- epilogue: lane = row rr[mi], features nn[ni] .. nn[ni] + 15 -------------------------------------
    int rr[MI], nn[NI];
#pragma unroll
    for (int mi = 0; mi < MI; mi++) rr[mi] = r0 + wm * (BM / WM) + mi * 32 + fl;
#pragma unroll
    for (int ni = 0; ni < NI; ni++) nn[ni] = n0 + wn * (BN / WN) + ni * 32 + 16 * fh;
    auto quad = [&](int ni, int mi, int q) { return (float4v){acc[ni][mi][4 * q], acc[ni][mi][4 * q + 1], acc[ni][mi][4 * q + 2], acc[ni][mi][4 * q + 3]}; };

    if constexpr (EPI == EPI_RESID) {
        // loads first, then stores (gemm_tile_kernel)
        float4v xo[NI][MI][4];
#pragma unroll
        for (int ni = 0; ni < NI; ni++)
#pragma unroll
            for (int mi = 0; mi < MI; mi++)
#pragma unroll
                for (int q = 0; q < 4; q++)
                    xo[ni][mi][q] = *(const float4v *) (a.out + (int64_t) min(rr[mi], a.R - 1) * a.ldo + min(nn[ni] + 4 * q, a.N - 4));
#pragma unroll
        for (int ni = 0; ni < NI; ni++)
#pragma unroll
            for (int mi = 0; mi < MI; mi++)
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if (rr[mi] < a.R && nn[ni] + 4 * q < a.N) *(float4v *) (a.out + (int64_t) rr[mi] * a.ldo + nn[ni] + 4 * q) = xo[ni][mi][q] + quad(ni, mi, q);
    } else if constexpr (EPI == EPI_QKV) {
        int64_t rowoff[MI];
#pragma unroll
        for (int mi = 0; mi < MI; mi++) {
            const int r = min(rr[mi], a.R - 1);
            rowoff[mi] = (int64_t) a.row_seq[r] * a.seq_stride + (int64_t) a.row_pos[r] * a.H;
        }
#pragma unroll
        for (int ni = 0; ni < NI; ni++) {
            const int which = nn[ni] / a.H, c = nn[ni] - which * a.H;   // 16 consecutive features never straddle q / k / v (H % 16 == 0)
#pragma unroll
            for (int mi = 0; mi < MI; mi++) {
                if (rr[mi] >= a.R || nn[ni] >= a.N) continue;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float4v t = quad(ni, mi, q);
                    if (which == 0) {
                        *(float4v *) (a.q + (int64_t) rr[mi] * a.H + c + 4 * q) = t;
                    } else {
                        void *cb = which == 1 ? a.kc : a.vc;
                        if (a.kv_f16) {
                            half4 h;
#pragma unroll
                            for (int e = 0; e < 4; e++) h[e] = (_Float16) t[e];
                            *(half4 *) ((_Float16 *) cb + rowoff[mi] + c + 4 * q) = h;
                        } else {
                            *(float4v *) ((float *) cb + rowoff[mi] + c + 4 * q) = t;
                        }
                    }
                }
            }
        }
    } else if constexpr (EPI == EPI_GELU) {
#pragma unroll
        for (int ni = 0; ni < NI; ni++)
#pragma unroll
            for (int mi = 0; mi < MI; mi++) {
                float y[16];
#pragma unroll
                for (int e = 0; e < 16; e++) y[e] = gelu_apply(acc[ni][mi][e], a.gelu_mode);
                const bool ok = rr[mi] < a.R && nn[ni] < a.N;
                if (qa.q_out) {
                    // ggml quantises fc2's input rows: the block of 32 features is this lane's 16 and lane ^ 32's 16
                    float dd;
                    const int4v pk = quant16_pair(y, dd);
                    if (ok) {
                        *(int4v *) (qa.q_out + (int64_t) rr[mi] * qa.ldq + nn[ni]) = pk;
                        if (fh == 0) qa.d_out[(int64_t) (nn[ni] >> 5) * qa.ldr + rr[mi]] = dd;
                    }
                } else if (ok) {
#pragma unroll
                    for (int q = 0; q < 4; q++) *(float4v *) (a.out + (int64_t) rr[mi] * a.ldo + nn[ni] + 4 * q) = (float4v){y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]};
                }
            }
    } else if constexpr (EPI == EPI_CROSS) {
        // the tile's 64 features are ONE head's query for BM rows: cross-attention over the voice prompt here (gemm_tile_kernel's EPI_CROSS,
        // parler/model.cpp:586-593), 16 lanes per row with 4 channels each; the attended rows leave as fp32 or as Q8_0 blocks (8 lanes = a block)
        static_assert(BN == 64, "EPI_CROSS: one head per tile column");
        __syncthreads();                                   // the k-tile buffers (or the k groups' sums) have been read: the LDS is reused
        float *qs = (float *) smem;                        // [BM][68]
        float *ks = qs + BM * 68, *vs = ks + 32 * 64;      // [E][64] each
        const int E = a.cross_E;
        if (kg == 0) {
#pragma unroll
            for (int ni = 0; ni < NI; ni++)
#pragma unroll
                for (int mi = 0; mi < MI; mi++)
#pragma unroll
                    for (int q = 0; q < 4; q++) *(float4v *) (qs + (rr[mi] - r0) * 68 + (nn[ni] - n0) + 4 * q) = quad(ni, mi, q);
        }
        for (int i = tid; i < E * 16; i += NW * 64) {
            const int e = i >> 4, c4 = (i & 15) * 4;
            *(float4v *) (ks + e * 64 + c4) = *(const float4v *) (a.cross_k + (int64_t) e * a.H + n0 + c4);
            *(float4v *) (vs + e * 64 + c4) = *(const float4v *) (a.cross_v + (int64_t) e * a.H + n0 + c4);
        }
        __syncthreads();
        const int cl = tid & 15;
        constexpr int RPP = NW * 4;   // rows per pass
#pragma unroll
        for (int pass = 0; pass < BM / RPP; pass++) {
            const int rl = pass * RPP + (tid >> 4), r = r0 + rl;
            const float4v q4 = *(const float4v *) (qs + rl * 68 + cl * 4);
            float sc[32];
            float m = -INFINITY;
#pragma unroll
            for (int e = 0; e < 32; e++) {
                if (e < E) {
                    const float4v k4 = *(const float4v *) (ks + e * 64 + cl * 4);
                    float d = q4[0] * k4[0] + q4[1] * k4[1] + q4[2] * k4[2] + q4[3] * k4[3];
                    sc[e] = row16_sum(d) * a.cross_scale;
                    m = fmaxf(m, sc[e]);
                }
            }
            float l = 0.0f;
            float4v o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 32; e++) {
                if (e < E) {
                    const float p = expf(sc[e] - m);
                    const float4v v4 = *(const float4v *) (vs + e * 64 + cl * 4);
                    l += p;
#pragma unroll
                    for (int j = 0; j < 4; j++) o[j] += p * v4[j];
                }
            }
            float4v res;
#pragma unroll
            for (int j = 0; j < 4; j++) res[j] = o[j] / l;
            if (qa.q_out) {
                float dd;
                const unsigned qq = quant4_q8(res, dd);   // 8 neighbouring lanes = one block of 32 channels
                if (r < a.R) {
                    *(unsigned *) (qa.q_out + (int64_t) r * qa.ldq + n0 + cl * 4) = qq;
                    if ((cl & 7) == 0) qa.d_out[(int64_t) ((n0 + cl * 4) >> 5) * qa.ldr + r] = dd;
                }
            } else if (r < a.R) {
                *(float4v *) (a.cross_out + (int64_t) r * a.H + n0 + cl * 4) = res;
            }
        }
    } else {   // EPI_STORE (slab kz)
#pragma unroll
        for (int ni = 0; ni < NI; ni++)
#pragma unroll
            for (int mi = 0; mi < MI; mi++)
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if (rr[mi] < a.R && nn[ni] + 4 * q < a.N)
                        *(float4v *) (a.out + (int64_t) kz * a.slab_stride + (int64_t) rr[mi] * a.ldo + nn[ni] + 4 * q) = quad(ni, mi, q);
    }
}

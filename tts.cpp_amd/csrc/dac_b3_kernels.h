// dac_b3_kernels.h — the DAC residual unit and the convs around it on the bf16 matrix pipe at fp32 accuracy (gfx950).
//
// Reference ops: build_residual_unit (/root/reference/src/decoder/general_neural_audio_codec.cpp:133-149):
//     snake_1d -> conv_1d(k = 7, dilation d, padding 3 d) + bias -> snake_1d -> conv_1d(k = 1) + bias -> + x
// with F32 tensors.  Every fp32 operand is carried as three bf16 terms (split_bf16x3 in dac_kernels.h: x = x1 + x2 + x3 exactly) and a
// product is the six partial products of weight >= 2^-16 on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: fp32-level error
// (tests/test_oracle_cpu.py::test_bf16x3_split_is_fp32_accurate) at 16/6 of the fp32 matrix rate.
//
//   resunit_b3_kernel<MI, KS, KS2>   one launch per residual unit for C = 32 MI <= 192 channels: a workgroup of 8 waves owns ALL
//        channels of 256 positions, so (a) the input tile is staged (snake + split) once per element, (b) the k = 7 conv's accumulators,
//        after bias + snake + split, ARE the B fragments of the k = 1 conv — with the reduction index of the second GEMM permuted to
//        the accumulator layout (register e of a lane <-> channel (e & 3) + 8 (e >> 2) + 4 (lane >> 5)) there is no data movement at all,
//        the k = 1 weights are packed in that order — and (c) the intermediate activation never exists in memory: per unit the kernel
//        reads x (conv input + residual) and writes x', 8-12 B per element instead of 20.
//   pack_resunit_b3_kernel            the unit's two weight tensors as one stream of LDS stage images (bf16 planes)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

// Stage stream of one residual unit (C channels, KS k-steps per k = 7 stage, KS2 per k = 1 stage), every stage padded to WST elements:
//   k = 7 stages  g = chunk * (4 / KS) + sub : [plane][s < KS][hi][co < C][j < 8] = plane of w7[co][ci = 8 chunk + j][tap = 2 (sub KS + s) + hi]  (tap 7: zero)
//   k = 1 stages  g2 = pass * (C / 16 / KS2) + q (after all k = 7 stages; a pass covers 96 output channels, so that its accumulators and the
//                 k = 7 accumulators fit the register file side by side):
//                 [plane][s < KS2][hi][co' < 96][m < 8] = plane of w1[co = 96 pass + co'][ch], k-step ks = q KS2 + s, i = ks / 2, qq = ks % 2,
//                                                          ch = 32 i + 16 qq + (m < 4 ? 4 hi + m : 8 + 4 hi + m - 4)
struct ResUnitGeom { int C, KS, KS2, WST, n7, ns1, n1; };
__host__ __device__ inline ResUnitGeom resunit_geom(int C, int KS, int KS2) {
    ResUnitGeom g;
    g.C = C; g.KS = KS; g.KS2 = KS2;
    const int a = 3 * KS * 2 * C * 8, b = 3 * KS2 * 2 * 96 * 8;
    g.WST = a > b ? a : b;
    g.n7 = (C / 8) * (4 / KS);
    g.ns1 = (C / 16) / KS2;
    g.n1 = (C / 96) * g.ns1;
    return g;
}

static __global__ void pack_resunit_b3_kernel(const float *w7, const float *w1, __bf16 *dst, int C, int KS, int KS2) {
    const ResUnitGeom g = resunit_geom(C, KS, KS2);
    const int64_t per7 = (int64_t) KS * 2 * C * 8, per1 = (int64_t) KS2 * 2 * 96 * 8;
    const int64_t total = (int64_t) g.n7 * per7 + (int64_t) g.n1 * per1;
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t) gridDim.x * blockDim.x) {
        float v = 0.0f;
        int64_t base, plane_sz;
        if (i < (int64_t) g.n7 * per7) {
            const int st = (int) (i / per7);
            int64_t r = i % per7;
            const int j = (int) (r % 8); r /= 8;
            const int co = (int) (r % C); r /= C;
            const int hi = (int) (r % 2); r /= 2;
            const int s = (int) r;
            const int chunk = st / (4 / KS), sub = st % (4 / KS);
            const int tap = 2 * (sub * KS + s) + hi, ci = chunk * 8 + j;
            if (tap < 7) v = w7[((int64_t) co * C + ci) * 7 + tap];
            plane_sz = per7;
            base = (int64_t) st * g.WST + (i % per7);
        } else {
            const int64_t i1 = i - (int64_t) g.n7 * per7;
            const int st = (int) (i1 / per1);
            int64_t r = i1 % per1;
            const int m = (int) (r % 8); r /= 8;
            const int col = (int) (r % 96); r /= 96;
            const int hi = (int) (r % 2); r /= 2;
            const int s = (int) r;
            const int pass = st / g.ns1, q = st % g.ns1;
            const int ks = q * KS2 + s, ib = ks / 2, qq = ks % 2;
            const int ch = 32 * ib + 16 * qq + (m < 4 ? 4 * hi + m : 8 + 4 * hi + m - 4);
            v = w1[(int64_t) (96 * pass + col) * C + ch];
            plane_sz = per1;
            base = (int64_t) (g.n7 + st) * g.WST + (i1 % per1);
        }
        __bf16 h1, h2, h3;
        split_bf16x3(v, h1, h2, h3);
        dst[base] = h1;
        dst[base + plane_sz] = h2;
        dst[base + 2 * plane_sz] = h3;
    }
}

template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

struct ResUnitArgs {
    const float *x;          // [n][C][L] fp32: conv input and residual
    float *y;                // [n][C][L] fp32 output
    const __bf16 *w;         // stage stream (pack_resunit_b3_kernel)
    const float *b7, *b1;    // biases [C]
    const float *alpha_in;   // snake in front of the k = 7 conv [C]
    const float *alpha_mid;  // snake in front of the k = 1 conv [C]
    int L, dil, pad;         // L = row stride
    const uint32_t *frames;  // per-utterance frames (grid.z) or NULL; valid length = frames[z] * mult
    int mult;
    long long *stamps;       // profiling (profiles/ru_bench.hip): 5 cycle-counter stamps per workgroup, or NULL
};
#define RU_STAMP(k) do { if (a.stamps && tid == 0) a.stamps[((int64_t) blockIdx.z * gridDim.x + blockIdx.x) * 5 + (k)] = (long long) __builtin_readcyclecounter(); } while (0)

// SCHED: 0 = fragments read just in time (the compiler's order), 1 = double-buffered fragment sets, 2 = double-buffered and the reads of
// group q + 1 pinned between the MFMAs of group q (sched_group_barrier)
// ABL (timing experiments only, results invalid): 1 = no snake / split in the input staging, 2 = no weight streaming, 4 = no barriers,
// 8 = fragments read once instead of per group
template <int MI, int KS, int KS2, int SCHED = 1, int ABL = 0>
__global__ __launch_bounds__(512, 2) void resunit_b3_kernel(ResUnitArgs a) {
    constexpr int C = 32 * MI, WN = 8, NT = 512, T_T = 32 * WN;
    constexpr int SPC = 4 / KS;                               // k = 7 stages per 8-channel chunk
    constexpr int WPL7 = KS * 2 * C * 8, WPL1 = KS2 * 2 * 96 * 8;
    constexpr int WST = 3 * (WPL7 > WPL1 ? WPL7 : WPL1);      // bf16 per stage (stream stride and LDS buffer)
    constexpr int NCH = C / 8, N7 = NCH * SPC, NP = MI / 3, NS1 = (C / 16) / KS2, N1 = NP * NS1;
    constexpr int WV = (WST / 8 + NT - 1) / NT;               // 16-byte vectors per thread per stage
    static_assert(4 % KS == 0 && (C / 16) % KS2 == 0 && MI % 3 == 0, "stage shapes");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int xw = T_T + 6 * a.dil;
    const int xpl = xw * 8;                                   // bf16 per input plane
    __bf16 *wsb = (__bf16 *) smem;                            // [2][WST]
    __bf16 *xsb = wsb + 2 * WST;                              // [2][3][xpl]
    float4 *tab = (float4 *) (xsb + 2 * 3 * xpl);             // [C] {b7, alpha_mid, 1/alpha_mid, b1}
    float2 *tin = (float2 *) (tab + C);                       // [C] {alpha_in, 1/alpha_in}
    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int t0 = blockIdx.x * T_T;
    const int LS = a.L, L = valid_len(a.frames, a.mult, a.L);
    if (t0 >= L) return;
    const float *xg = a.x + (int64_t) blockIdx.z * C * LS;
    float *yg = a.y + (int64_t) blockIdx.z * C * LS;
    const uint4d *wg = (const uint4d *) a.w;

    RU_STAMP(0);
    for (int i = tid; i < C; i += NT) {
        const float am = a.alpha_mid[i], ai = a.alpha_in[i];
        tab[i] = make_float4(a.b7[i], am, 1.0f / am, a.b1[i]);
        tin[i] = make_float2(ai, 1.0f / ai);
    }

    float16d acc[MI];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[i][e] = 0.0f;

    uint4d wreg[WV];
    float xreg[8];
    const int xp = tid;                                       // the position row this thread stages (xw <= 310 < NT)
    auto prefetch_w = [&](int g) __attribute__((always_inline)) {
        const uint4d *wp = wg + (int64_t) g * (WST / 8);
#pragma unroll
        for (int j = 0; j < WV; j++) {
            const int i = tid + j * NT;
            if (i < WST / 8) wreg[j] = wp[i];
        }
    };
    auto commit_w = [&](int buf) __attribute__((always_inline)) {
        uint4d *wd = (uint4d *) (wsb + buf * WST);
#pragma unroll
        for (int j = 0; j < WV; j++) {
            const int i = tid + j * NT;
            if (i < WST / 8) wd[i] = wreg[j];
        }
    };
    auto prefetch_x = [&](int c) __attribute__((always_inline)) {
        const int t = t0 + xp - a.pad;
        const bool ok = xp < xw && t >= 0 && t < L;
#pragma unroll
        for (int e = 0; e < 8; e++) xreg[e] = ok ? xg[(int64_t) (c * 8 + e) * LS + t] : 0.0f;
    };
    auto commit_x = [&](int c, int buf) __attribute__((always_inline)) {
        if (xp < xw) {
            bf16x8d h1, h2, h3;
            float v[8], al[8], ral[8];
#pragma unroll
            for (int e = 0; e < 8; e++) { const float2 t2 = tin[c * 8 + e]; v[e] = xreg[e]; al[e] = t2.x; ral[e] = t2.y; }
            if (!(ABL & 1)) snake_vec<8>(v, al, ral);   // snake(0) == 0: zero padding is preserved
#pragma unroll
            for (int e = 0; e < 8; e++) {
                __bf16 b1, b2, b3;
                split_bf16x3(v[e], b1, b2, b3);
                h1[e] = b1; h2[e] = b2; h3[e] = b3;
            }
            __bf16 *xd = xsb + buf * 3 * xpl + xp * 8;
            *(bf16x8d *) xd = h1;
            *(bf16x8d *) (xd + xpl) = h2;
            *(bf16x8d *) (xd + 2 * xpl) = h3;
        }
    };

    prefetch_w(0);
    prefetch_x(0);
    __syncthreads();   // tables visible
    commit_w(0);
    commit_x(0, 0);
    __syncthreads();
    RU_STAMP(1);

    // ---- k = 7 conv: N7 stages --------------------------------------------------------------------------------------------------
    for (int g = 0; g < N7; g++) {
        const int c = g / SPC, sub = g - c * SPC;
        if (!(ABL & 2)) prefetch_w(g + 1);                     // the stream continues into the k = 1 stages
        if (!(ABL & 16) && sub == 0 && c + 1 < NCH) prefetch_x(c + 1);
        const __bf16 *ws = wsb + (g & 1) * WST;
        const __bf16 *xs = xsb + (c & 1) * 3 * xpl;
        // fragments of group q + 1 (three 32-channel blocks of one k-step) are requested before the MFMAs of group q: with two waves per
        // SIMD a just-in-time ds_read + wait in front of every few MFMAs left the matrix pipe 53 % idle (profiles/r03/pmc_busy_dac_b64_fused.txt)
        constexpr int GPS = MI / 3, NG = KS * GPS;
        constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};   // the six partial products, smallest first; term-major over
                                                                                // the three accumulators: consecutive MFMAs write different registers
        auto load_b = [&](bf16x8d (&bf)[3], int s) __attribute__((always_inline)) {
            const int st = sub * KS + s;
            const int tap = (2 * st + hi < 7) ? 2 * st + hi : 6;   // the eighth tap: zero weights, any valid rows
#pragma unroll
            for (int pl = 0; pl < 3; pl++) bf[pl] = *(const bf16x8d *) (xs + pl * xpl + (wn * 32 + l31 + tap * a.dil) * 8);
        };
        auto load_a = [&](bf16x8d (&af)[3][3], int s, int ig) __attribute__((always_inline)) {
#pragma unroll
            for (int ii = 0; ii < 3; ii++)
#pragma unroll
                for (int pl = 0; pl < 3; pl++)
                    af[ii][pl] = *(const bf16x8d *) (ws + pl * WPL7 + (((s * 2 + hi) * C) + (ig + ii) * 32 + l31) * 8);
        };
        if constexpr (SCHED == 0) {
            static_for<NG>([&](auto Q) __attribute__((always_inline)) {
                constexpr int q = decltype(Q)::value, s = q / GPS, ig = 3 * (q % GPS);
                bf16x8d af[3][3], bf[3];
                load_b(bf, s);
                load_a(af, s, ig);
#pragma unroll
                for (int tm = 0; tm < 6; tm++)
#pragma unroll
                    for (int ii = 0; ii < 3; ii++)
                        acc[ig + ii] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ii][TA[tm]], bf[TB[tm]], acc[ig + ii], 0, 0, 0);
            });
        } else {
            bf16x8d afs[2][3][3], bfs[2][3];
            load_b(bfs[0], 0);
            load_a(afs[0], 0, 0);
            if constexpr ((ABL & 8) != 0) { load_b(bfs[1], 1); load_a(afs[1], 1, 0); }
            static_for<NG>([&](auto Q) __attribute__((always_inline)) {
                constexpr int q = decltype(Q)::value, s = q / GPS, ig = 3 * (q % GPS);
                constexpr int s1 = (q + 1) / GPS, ig1 = 3 * ((q + 1) % GPS);
                constexpr int nrd = q + 1 < NG ? (ig1 == 0 ? 12 : 9) : 0;
                if constexpr (q + 1 < NG && !(ABL & 8)) {
                    if constexpr (ig1 == 0) load_b(bfs[s1 & 1], s1);
                    load_a(afs[(q + 1) & 1], s1, ig1);
                }
#pragma unroll
                for (int tm = 0; tm < 6; tm++)
#pragma unroll
                    for (int ii = 0; ii < 3; ii++)
                        acc[ig + ii] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afs[q & 1][ii][TA[tm]], bfs[s & 1][TB[tm]], acc[ig + ii], 0, 0, 0);
                if constexpr (SCHED == 2 && nrd > 0) {   // 2 reads behind every MFMA until the next group's reads are out, then the rest of the MFMAs
#pragma unroll
                    for (int k = 0; k < (nrd + 1) / 2; k++) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 18 - (nrd + 1) / 2, 0);
                }
            });
        }
        if (!(ABL & 2)) commit_w((g + 1) & 1);
        if (!(ABL & 16) && sub == SPC - 1 && c + 1 < NCH) commit_x(c + 1, (c + 1) & 1);
        if (!(ABL & 4)) __syncthreads();
    }

    RU_STAMP(2);
    // ---- k = 1 conv: the accumulators (bias, snake, split) are its B fragments; 96 output channels per pass --------------------------
    // (static_for: the stage index must be a compile-time constant, it selects accumulator registers)
    const int t = t0 + wn * 32 + l31;
    float16d acc2[3];
    static_for<N1>([&](auto G2) __attribute__((always_inline)) {
        constexpr int g2 = decltype(G2)::value, p = g2 / NS1, q = g2 % NS1;
        if constexpr (q == 0) {
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc2[i][e] = 0.0f;
        }
        if constexpr (g2 + 1 < N1) prefetch_w(N7 + g2 + 1);
        const __bf16 *ws = wsb + ((N7 + g2) & 1) * WST;
        static_for<(ABL & 32) ? 0 : KS2>([&](auto S) __attribute__((always_inline)) {
            constexpr int s = decltype(S)::value, ks = q * KS2 + s, ib = ks / 2, qq = ks % 2;
            bf16x8d bf[3];
            float hv[8], al[8], ral[8];
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const int e = 8 * qq + m;
                const int ch = 32 * ib + (e & 3) + 8 * (e >> 2) + 4 * hi;
                const float4 tb = tab[ch];
                hv[m] = acc[ib][e] + tb.x; al[m] = tb.y; ral[m] = tb.z;
            }
            snake_vec<8>(hv, al, ral);
#pragma unroll
            for (int m = 0; m < 8; m++) {
                __bf16 b1, b2, b3;
                split_bf16x3(hv[m], b1, b2, b3);
                bf[0][m] = b1; bf[1][m] = b2; bf[2][m] = b3;
            }
            bf16x8d af[3][3];
#pragma unroll
            for (int ii = 0; ii < 3; ii++)
#pragma unroll
                for (int pl = 0; pl < 3; pl++)
                    af[ii][pl] = *(const bf16x8d *) (ws + pl * WPL1 + (((s * 2 + hi) * 96) + ii * 32 + l31) * 8);
            constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int tm = 0; tm < 6; tm++)
#pragma unroll
                for (int ii = 0; ii < 3; ii++)
                    acc2[ii] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ii][TA[tm]], bf[TB[tm]], acc2[ii], 0, 0, 0);
        });
        if constexpr (g2 + 1 < N1) {
            commit_w((N7 + g2 + 1) & 1);
            __syncthreads();
        }
        if constexpr (q == NS1 - 1) {   // + bias + x
            if constexpr (p == NP - 1) RU_STAMP(3);
            if (t < L) {
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    float rv[16];
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const int co = 96 * p + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * hi;
                        rv[e] = xg[(int64_t) co * LS + t];
                    }
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const int co = 96 * p + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * hi;
                        yg[(int64_t) co * LS + t] = (acc2[i][e] + tab[co].w) + rv[e];
                    }
                }
            }
        }
    });
    RU_STAMP(4);
}

// ================================================================================================================================
// Split planes in memory.  For the wide classes (768 / 384 / 1536 channels) an output-channel tile cannot hold all channels, so a
// conv kernel that stages fp32 inputs applies snake + split once per OUTPUT-CHANNEL TILE (12 x per element at 768 channels) and the
// matrix pipe waits for the vector pipe.  There the activation a conv consumes is kept in memory already snaked and split:
//     planes [n][3][C / 8][L][8] bf16       (plane, 8-channel group, position, channel in group)
// — one 16-byte row per (plane, group, position), consecutive positions consecutive rows, i.e. exactly the LDS image of a chunk:
// staging is a straight 16-byte copy and the B fragment of v_mfma_f32_32x32x16_bf16 one ds_read_b128.  The producer's epilogue
// writes the planes (snake with the CONSUMER's alpha, then the split), plus the fp32 tensor where a residual add needs it.
//   snake_split_kernel      fp32 [n][C][L] -> planes (for producers that are not plane-aware)
//   conv_b3p_kernel<KT,...> k = 7 / k = 1 conv: planes in; fp32 and / or planes out; bias, residual, snake for the consumer fused
//   pack_conv_w_b3p_kernel  weights as bf16 planes in stage order
// ================================================================================================================================
struct SplitArgs {
    const float *x;          // [n][C][L]
    const float *alpha;      // [C] snake before the split, or NULL
    __bf16 *yp;              // [n][NPL][C/8][L][8]
    int C, L;
    const uint32_t *frames; int mult;
};

template <typename SP = SplitB3>
__global__ __launch_bounds__(256) void snake_split_kernel(SplitArgs a) {
    constexpr int NPL = SP::NPL;
    const int t = blockIdx.x * 256 + threadIdx.x, cg = blockIdx.y, z = blockIdx.z;
    const int LS = a.L, L = valid_len(a.frames, a.mult, a.L);
    if (t >= L) return;
    const int CG = a.C / 8;
    const float *xg = a.x + ((int64_t) z * a.C + cg * 8) * LS + t;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = xg[(int64_t) e * LS];
    if (a.alpha) {   // one wave-uniform range test for the eight values (snake_vec) instead of a divergent branch per value
        const float4d a0 = *(const float4d *) (a.alpha + cg * 8), a1 = *(const float4d *) (a.alpha + cg * 8 + 4);
        float al[8], ral[8];
#pragma unroll
        for (int e = 0; e < 4; e++) { al[e] = a0[e]; al[4 + e] = a1[e]; }
#pragma unroll
        for (int e = 0; e < 8; e++) ral[e] = 1.0f / al[e];
        snake_vec<8>(v, al, ral);
    }
    bf16x8d hp[NPL];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        __bf16 pv[NPL];
        SP::split(v[e], pv);
#pragma unroll
        for (int pl = 0; pl < NPL; pl++) hp[pl][e] = pv[pl];
    }
    const int64_t pst = (int64_t) CG * LS * 8;
    __bf16 *yp = a.yp + (int64_t) z * NPL * pst + ((int64_t) cg * LS + t) * 8;
#pragma unroll
    for (int pl = 0; pl < NPL; pl++) *(bf16x8d *) (yp + pl * pst) = hp[pl];
}

//   src [cout][cin][KT] -> dst [co_tile][chunk][plane][s < NS][hi][CO_T][8]
//   KT = 7, NS = 4: ci = 8 chunk + j, tap = 2 s + hi (tap 7: zero);   KT = 7, NS = 7: ci = 16 chunk + 8 hi + j, tap = s;
//   KT = 1: ci = 16 NS chunk + 8 (2 s + hi) + j
static __global__ void pack_conv_w_b3p_kernel(const float *src, __bf16 *dst, int cout, int cin, int KT, int CO_T, int NS, int n_chunks, int scheme = 0) {
    const int64_t plane_sz = (int64_t) NS * 2 * CO_T * 8;
    const int64_t total = (int64_t) ((cout + CO_T - 1) / CO_T) * n_chunks * plane_sz;
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t) gridDim.x * blockDim.x) {
        const int j = (int) (i % 8);
        int64_t r = i / 8;
        const int col = (int) (r % CO_T); r /= CO_T;
        const int hi = (int) (r % 2); r /= 2;
        const int st = (int) (r % NS); r /= NS;
        const int ch = (int) (r % n_chunks);
        const int ct = (int) (r / n_chunks);
        const int co = ct * CO_T + col;
        int ci, tap;
        if (KT == 7 && NS == 7) { ci = ch * 16 + 8 * hi + j; tap = st; }
        else if (KT == 7) { ci = ch * 8 + j; tap = 2 * st + hi; }
        else { ci = ch * 16 * NS + 8 * (2 * st + hi) + j; tap = 0; }
        float v = 0.0f;
        if (co < cout && ci < cin && tap < KT) v = src[((int64_t) co * cin + ci) * KT + tap];
        const int64_t base = ((int64_t) ct * n_chunks + ch) * split_planes(scheme) * plane_sz + (i % plane_sz);
        split_store(scheme, v, dst, base, plane_sz);
    }
}

// XCD-aware tile order.  Workgroup ids go round-robin over the 8 XCDs (each with its own L2).  With the plain (position tile, channel tile,
// utterance) grid the channel tiles of one position tile are dispatched far apart, each on another XCD, and every one of them pulls the input
// tile from memory again: 5 x the algorithmic traffic on the k = 7 convs of the wide classes, 4.5 x on the stride-8 transposed convs
// (profiles/r03/pmc_fetch_write_round3.txt).  Here the grid is one-dimensional, 8 * ceil(columns / 8) * nco workgroups, and workgroup g is
// item w = g / 8 of XCD g % 8: its channel tile is w % nco and its column (position tile, utterance) is (w / nco) * 8 + g % 8, so the nco
// channel tiles of a column are consecutive items of ONE XCD: the input tile leaves memory once, the other channel tiles take it from that L2.
// When the layer's weights outweigh its input (few positions per utterance: the first conv, the first transposed conv) the plain order —
// channel tile slowest, so that the workgroups in flight share one weight tile — moves less (nco < 0 selects it).
struct TileId { int pos, co, z; bool valid; };
__device__ __forceinline__ TileId xcd_tile(int npos, int nco_signed, int nz) {
    const unsigned g = blockIdx.x, xcd = g & 7u, w = g >> 3;
    const unsigned nco = (unsigned) (nco_signed < 0 ? -nco_signed : nco_signed), ncols = (unsigned) npos * (unsigned) nz;
    TileId t;
    unsigned col;
    if (nco_signed < 0) { col = g % ncols; t.co = (int) (g / ncols); t.valid = g < ncols * nco; }
    else { col = (w / nco) * 8u + xcd; t.co = (int) (w % nco); t.valid = col < ncols; }
    t.z = (int) (col / (unsigned) npos);
    t.pos = (int) (col % (unsigned) npos);
    return t;
}
// weights_outweigh: 4 x the weight bytes >= the input bytes of the launch -> plain order
__host__ inline int xcd_order(int nco, double input_bytes, double weight_bytes) { return 4.0 * weight_bytes >= input_bytes ? -nco : nco; }
__host__ inline unsigned xcd_grid(int npos, int nco, int nz) {
    const unsigned cols = (unsigned) npos * (unsigned) nz;
    return 8u * ((cols + 7u) / 8u) * (unsigned) nco;
}
__device__ __forceinline__ int valid_len_z(const uint32_t *frames, int mult, int L, int z) { return frames ? (int) frames[z] * mult : L; }

struct PConvArgs {
    const __bf16 *xp;        // input planes [n][3][cin/8][L][8]
    const __bf16 *w;         // packed weights (pack_conv_w_b3p_kernel)
    const float *b;          // [cout]
    const float *resid;      // fp32 [n][cout][L] or NULL
    float *y;                // fp32 out [n][cout][L] or NULL
    __bf16 *yp;              // planes out [n][3][cout/8][L][8] or NULL
    const float *alpha_out;  // snake applied before the split of yp, or NULL
    int cin, cout, L, dil, pad;   // L = row stride
    const uint32_t *frames; int mult;
    int npos, nco, nz;       // tiles along positions / output channels, utterances (xcd_tile)
};


// k = 7, NS = 7 (round 3, "tap per k-step"): chunk = 16 input channels, k-step s = tap s, the half-wave takes 8-channel group hi — 7 k-steps per
// 16 channels where the NS = 4 form (chunk = 8 channels, half-wave = tap parity) spends 8 with the eighth tap slot on zero weights.
// NB = 1: one LDS buffer, the next chunk waits in registers and is committed between two barriers; with two workgroups per CU the other one
// computes meanwhile (NB = 2: the next chunk is committed into the other buffer while the workgroup's own waves still compute).
template <int KT, int MI, int NI, int WM, int WN, int NS, int MINW, int NB = 2, typename SP = SplitB3>
__global__ __launch_bounds__(64 * WM * WN, MINW) void conv_b3p_kernel(PConvArgs a) {
    constexpr int NPL = SP::NPL;
    constexpr int CO_T = 32 * MI * WM, T_T = 32 * NI * WN, NT = 64 * WM * WN;
    constexpr bool TAPK = KT == 7 && NS == 7;
    constexpr int NCG = KT == 7 ? (TAPK ? 2 : 1) : 2 * NS;   // 8-channel groups per chunk
    constexpr int WPL = NS * 2 * CO_T * 8;                   // bf16 per weight plane of a chunk
    constexpr int WV = (NPL * WPL / 8 + NT - 1) / NT;        // 16-byte vectors per thread per chunk (all planes)
    constexpr int XP = (T_T + (KT - 1) * 9 + NT - 1) / NT;   // position rows per thread per (plane, group), dilation <= 9
    static_assert(KT == 1 || NS == 4 || NS == 7, "k = 7: four k-steps (tap pairs) per 8-channel chunk, or seven (taps) per 16-channel chunk");
    static_assert(NB == 1 || NB == 2, "LDS buffers");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int xw = T_T + (KT - 1) * a.dil;
    const int xsz = NPL * NCG * xw * 8;                      // 16-bit values per input chunk image [plane][group][position][8]
    __bf16 *wsb = (__bf16 *) smem;                           // [NB][NPL][WPL]
    __bf16 *xsb = wsb + NB * NPL * WPL;                      // [NB][xsz]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv / WN, wn = wv % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    const TileId tile = xcd_tile(a.npos, a.nco, a.nz);
    if (!tile.valid) return;
    const int t0 = tile.pos * T_T, co0 = tile.co * CO_T;
    const int CGI = a.cin / 8;
    const int n_chunks = CGI / NCG;
    const int LS = a.L, L = valid_len_z(a.frames, a.mult, a.L, tile.z);
    if (t0 >= L) return;
    const int64_t psti = (int64_t) CGI * LS * 8;             // bf16 per input plane
    const __bf16 *xg = a.xp + (int64_t) tile.z * NPL * psti;
    const uint4d *wg = (const uint4d *) (a.w + (int64_t) tile.co * n_chunks * NPL * WPL);

    float16d acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NI; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

    uint4d wreg[WV];
    uint4d xreg[NPL * NCG * XP];
    auto prefetch = [&](int c) __attribute__((always_inline)) {
        const uint4d *wp = wg + (int64_t) c * (NPL * WPL / 8);
#pragma unroll
        for (int j = 0; j < WV; j++) {
            const int i = tid + j * NT;
            if (i < NPL * WPL / 8) wreg[j] = wp[i];
        }
#pragma unroll
        for (int q = 0; q < XP; q++) {
            const int p = tid + q * NT;
            const int t = t0 + p - a.pad;
            const bool ok = p < xw && t >= 0 && t < L;
#pragma unroll
            for (int pl = 0; pl < NPL; pl++)
#pragma unroll
                for (int g = 0; g < NCG; g++) {
                    const uint4d zero = {0u, 0u, 0u, 0u};
                    xreg[(q * NPL + pl) * NCG + g] = ok ? *(const uint4d *) (xg + pl * psti + ((int64_t) (c * NCG + g) * LS + t) * 8) : zero;
                }
        }
    };
    auto commit = [&](int buf) __attribute__((always_inline)) {
        uint4d *wd = (uint4d *) (wsb + buf * NPL * WPL);
#pragma unroll
        for (int j = 0; j < WV; j++) {
            const int i = tid + j * NT;
            if (i < NPL * WPL / 8) wd[i] = wreg[j];
        }
        __bf16 *xd = xsb + buf * xsz;
#pragma unroll
        for (int q = 0; q < XP; q++) {
            const int p = tid + q * NT;
            if (p < xw) {
#pragma unroll
                for (int pl = 0; pl < NPL; pl++)
#pragma unroll
                    for (int g = 0; g < NCG; g++) *(uint4d *) (xd + ((pl * NCG + g) * xw + p) * 8) = xreg[(q * NPL + pl) * NCG + g];
            }
        }
    };

    prefetch(0);
    commit(0);
    __syncthreads();
    for (int c = 0; c < n_chunks; c++) {
        const int buf = NB == 2 ? (c & 1) : 0;
        if (c + 1 < n_chunks) prefetch(c + 1);
        const __bf16 *ws = wsb + buf * NPL * WPL;
        const __bf16 *xs = xsb + buf * xsz;
#pragma unroll
        for (int st = 0; st < NS; st++) {
            int xoff;   // bf16 offset of this half-wave's B rows inside a plane image
            if constexpr (TAPK) {
                xoff = (hi * xw + st * a.dil) * 8;                           // group hi, tap st
            } else if constexpr (KT == 7) {
                const int tap = (2 * st + hi < KT) ? 2 * st + hi : KT - 1;   // the eighth tap: zero weights, any valid rows
                xoff = tap * a.dil * 8;
            } else {
                xoff = (2 * st + hi) * xw * 8;
            }
            bf16x8d af[NPL][MI], bf[NPL][NI];
#pragma unroll
            for (int pl = 0; pl < NPL; pl++) {
#pragma unroll
                for (int i = 0; i < MI; i++)
                    af[pl][i] = *(const bf16x8d *) (ws + pl * WPL + (((st * 2 + hi) * CO_T) + (wm * MI + i) * 32 + l31) * 8);
#pragma unroll
                for (int j = 0; j < NI; j++)
                    bf[pl][j] = *(const bf16x8d *) (xs + pl * NCG * xw * 8 + xoff + ((wn * NI + j) * 32 + l31) * 8);
            }
#pragma unroll
            for (int tm = 0; tm < SP::NT; tm++)
#pragma unroll
                for (int i = 0; i < MI; i++)
#pragma unroll
                    for (int j = 0; j < NI; j++)
                        acc[i][j] = SP::mfma(af[SP::ta(tm)][i], bf[SP::tb(tm)][j], acc[i][j]);
        }
        if constexpr (NB == 2) {
            if (c + 1 < n_chunks) commit(buf ^ 1);
            __syncthreads();
        } else {
            __syncthreads();                     // every wave is done with the buffer
            if (c + 1 < n_chunks) {
                commit(0);
                __syncthreads();
            }
        }
    }

    const float *rg = a.resid ? a.resid + (int64_t) tile.z * a.cout * LS : nullptr;
    float *yg = a.y ? a.y + (int64_t) tile.z * a.cout * LS : nullptr;
    const int64_t psto = (int64_t) (a.cout / 8) * LS * 8;
    __bf16 *ypz = a.yp ? a.yp + (int64_t) tile.z * NPL * psto : nullptr;
    // Epilogue in phases per (row tile, quad of 4 channels): every load of the phase is issued before anything waits (bias / alpha as one
    // 16-byte load each, the NI x 4 residual values with clamped indices), then the arithmetic for all lanes, then predicated stores.  The
    // per-value form — `if (t < L)`, `if (resid)`, `if (alpha)` around single loads — compiled to load, s_waitcnt vmcnt(0), store per value:
    // 128 dependent round trips per lane on a 256 x 256 tile (profiles/tools/isa_serial_loads.py).
    typedef __bf16 bf16x4d __attribute__((ext_vector_type(4)));
    const bool has_b = a.b != nullptr, has_al = a.alpha_out != nullptr;
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int co = co0 + (wm * MI + i) * 32 + 8 * q + 4 * hi;   // a multiple of 4; cout is a multiple of 8
            const bool co_ok = co < a.cout;
            const int coc = co_ok ? co : 0;
            float4d b4 = {0.f, 0.f, 0.f, 0.f}, al4 = {1.f, 1.f, 1.f, 1.f};
            if (has_b) b4 = *(const float4d *) (a.b + coc);
            if (has_al) al4 = *(const float4d *) (a.alpha_out + coc);
            float rv[NI][4];
            if (rg) {
#pragma unroll
                for (int j = 0; j < NI; j++) {
                    const int tc = min(t0 + (wn * NI + j) * 32 + l31, L - 1);
#pragma unroll
                    for (int r = 0; r < 4; r++) rv[j][r] = rg[(int64_t) (coc + r) * LS + tc];
                }
            }
            float al[4], ral[4];
#pragma unroll
            for (int r = 0; r < 4; r++) { al[r] = al4[r]; ral[r] = 1.0f / al4[r]; }
#pragma unroll
            for (int j = 0; j < NI; j++) {
                const int t = t0 + (wn * NI + j) * 32 + l31;
                const bool ok = co_ok && t < L;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    v[r] = acc[i][j][4 * q + r] + (has_b ? b4[r] : 0.0f);
                    if (rg) v[r] = v[r] + rv[j][r];
                }
                if (yg && ok) {
#pragma unroll
                    for (int r = 0; r < 4; r++) yg[(int64_t) (co + r) * LS + t] = v[r];
                }
                if (ypz) {
                    if (has_al) {
#pragma unroll
                        for (int r = 0; r < 4; r++) if (!ok) v[r] = 0.0f;   // lanes outside the tensor must not drag the wave into the slow sine
                        snake_vec<4>(v, al, ral);
                    }
                    bf16x4d hp[NPL];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        __bf16 pv[NPL];
                        SP::split(v[r], pv);
#pragma unroll
                        for (int pl = 0; pl < NPL; pl++) hp[pl][r] = pv[pl];
                    }
                    if (ok) {
                        __bf16 *p = ypz + ((int64_t) (co >> 3) * LS + t) * 8 + (co & 7);   // co is a multiple of 4: the lower or upper half of a 16-byte row
#pragma unroll
                        for (int pl = 0; pl < NPL; pl++) *(bf16x4d *) (p + pl * psto) = hp[pl];
                    }
                }
            }
        }
}

// ================================================================================================================================
// ConvTranspose1d(stride S, kernel 2 S) as bf16 x 3 products, phase-decomposed like convt1d_mfma_kernel (dac_kernels.h):
//   y[co][ti S + ph - p] = b[co] + sum_ci ( f(x[ci][ti]) w[ci][co][ph] + f(x[ci][ti - 1]) w[ci][co][ph + S] ),   f = snake
// One MFMA k-step = 8 input channels x the two tap slots (half-wave hi = 0: x[ti], hi = 1: x[ti - 1]); the S phases share the B operand.
// Workgroup = 8 waves side by side in ti: (32 MI) output channels x 256 ti x S phases; chunk = 16 input channels (two k-steps).
// The fp32 input is staged once per workgroup and chunk: snake (snake_vec) + split -> LDS planes [plane][group][ti0 - 1 + r][8].
//   pack_convt_w_b3_kernel   src [cin][cout][2 S] -> dst [co_tile][chunk][plane][ks < 2][ph < S][hi][CO_T][8]  (ci = 16 chunk + 8 ks + j, k = ph + hi S)
// ================================================================================================================================
static __global__ void pack_convt_w_b3_kernel(const float *src, __bf16 *dst, int cout, int cin, int S, int CO_T, int n_chunks, int scheme = 0) {
    const int64_t plane_sz = (int64_t) 2 * S * 2 * CO_T * 8;
    const int64_t total = (int64_t) ((cout + CO_T - 1) / CO_T) * n_chunks * plane_sz;
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t) gridDim.x * blockDim.x) {
        const int j = (int) (i % 8);
        int64_t r = i / 8;
        const int col = (int) (r % CO_T); r /= CO_T;
        const int hi = (int) (r % 2); r /= 2;
        const int ph = (int) (r % S); r /= S;
        const int ks = (int) (r % 2); r /= 2;
        const int ch = (int) (r % n_chunks);
        const int ct = (int) (r / n_chunks);
        const int co = ct * CO_T + col, ci = ch * 16 + ks * 8 + j, k = ph + hi * S;
        float v = 0.0f;
        if (co < cout && ci < cin) v = src[((int64_t) ci * cout + co) * 2 * S + k];
        const int64_t base = ((int64_t) ct * n_chunks + ch) * split_planes(scheme) * plane_sz + (i % plane_sz);
        split_store(scheme, v, dst, base, plane_sz);
    }
}

// PLANES: the input arrives as split planes written by its producer (snake with this layer's alpha and the bf16 x 3 split done ONCE per element);
// the fp32 form snakes and splits the staged tile in every workgroup — 24 / 12 times per element for the stride-8 layers (cout / 32 channel
// tiles), about as many VALU cycles per chunk as half the MFMAs of the chunk (0.40 of the bf16 peak where the k = 7 families reach 0.53).
template <int S, int MI, bool PLANES = false, typename SP = SplitB3>
__global__ __launch_bounds__(512, 2) void convt_b3_kernel(ConvTArgs a) {
    constexpr int NPL = SP::NPL;
    constexpr int CO_T = 32 * MI, WN = 8, TI_T = 32 * WN, NT = 64 * WN, K2 = 2 * S;
    constexpr int WPL = 2 * S * 2 * CO_T * 8;                // bf16 per weight plane of a chunk
    constexpr int WV = (NPL * WPL / 8 + NT - 1) / NT;        // 16-byte vectors per thread per chunk
    constexpr int xw = TI_T + 1;                             // rows ti0 - 1 .. ti0 + TI_T - 1
    constexpr int XR = (2 * xw + NT - 1) / NT;               // (group, row) units per thread per chunk
    constexpr int xpl = 2 * xw * 8;                          // bf16 per input plane of a chunk
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16 *wsb = (__bf16 *) smem;                           // [2][NPL][WPL]
    __bf16 *xsb = wsb + 2 * NPL * WPL;                       // [2][NPL][xpl]
    float2 *tin = (float2 *) (xsb + 2 * NPL * xpl);          // [cin] {alpha, 1 / alpha}
    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const TileId tile = xcd_tile(a.npos, a.nco, a.nz);
    if (!tile.valid) return;
    const int ti0 = tile.pos * TI_T, co0 = tile.co * CO_T;
    const int n_chunks = a.cin / 16;
    const int LS = a.L, L = valid_len_z(a.frames, a.mult, a.L, tile.z);
    const int LoS = a.Lout, Lout = a.frames ? (L - 1) * S - 2 * a.pad + K2 : a.Lout;
    if (ti0 > L) return;  // ti runs 0..L inclusive
    const float *xg = a.x + (int64_t) tile.z * a.cin * LS;
    const int64_t psti = (int64_t) (a.cin / 8) * LS * 8;     // bf16 per input plane
    const __bf16 *xpg = PLANES ? a.xp + (int64_t) tile.z * NPL * psti : nullptr;
    float *yg = a.y + (int64_t) tile.z * a.cout * LoS;
    const uint4d *wg = (const uint4d *) ((const __bf16 *) a.w + (int64_t) tile.co * n_chunks * NPL * WPL);

    if (!PLANES) {
        for (int i = tid; i < a.cin; i += NT) {
            const float al = a.alpha ? a.alpha[i] : 1.0f;
            tin[i] = make_float2(al, 1.0f / al);
        }
    }

    float16d acc[MI][S];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int ph = 0; ph < S; ph++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][ph][e] = 0.0f;

    uint4d wreg[WV];
    float xreg[PLANES ? 1 : XR][8];
    uint4d xpreg[PLANES ? XR : 1][NPL];
    auto prefetch = [&](int c) __attribute__((always_inline)) {
        const uint4d *wp = wg + (int64_t) c * (NPL * WPL / 8);
#pragma unroll
        for (int j = 0; j < WV; j++) {
            const int i = tid + j * NT;
            if (i < NPL * WPL / 8) wreg[j] = wp[i];
        }
#pragma unroll
        for (int q = 0; q < XR; q++) {
            const int u = tid + q * NT;
            const int g = u / xw, r = u - g * xw;           // lanes run along positions: coalesced rows
            const int ti = ti0 - 1 + r;
            const bool ok = g < 2 && ti >= 0 && ti < L;
            if constexpr (PLANES) {
                const uint4d zero = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int pl = 0; pl < NPL; pl++)
                    xpreg[q][pl] = ok ? *(const uint4d *) (xpg + pl * psti + ((int64_t) (c * 2 + g) * LS + ti) * 8) : zero;
            } else {
#pragma unroll
                for (int e = 0; e < 8; e++) xreg[q][e] = ok ? xg[(int64_t) (c * 16 + g * 8 + e) * LS + ti] : 0.0f;
            }
        }
    };
    auto commit = [&](int c, int buf) __attribute__((always_inline)) {
        uint4d *wd = (uint4d *) (wsb + buf * NPL * WPL);
#pragma unroll
        for (int j = 0; j < WV; j++) {
            const int i = tid + j * NT;
            if (i < NPL * WPL / 8) wd[i] = wreg[j];
        }
        __bf16 *xd = xsb + buf * NPL * xpl;
#pragma unroll
        for (int q = 0; q < XR; q++) {
            const int u = tid + q * NT;
            const int g = u / xw, r = u - g * xw;
            if constexpr (PLANES) {
                if (g < 2) {
                    __bf16 *p = xd + (g * xw + r) * 8;
#pragma unroll
                    for (int pl = 0; pl < NPL; pl++) *(uint4d *) (p + pl * xpl) = xpreg[q][pl];
                }
            } else if (g < 2) {
                if (a.alpha) {
                    float al[8], ral[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) { const float2 t2 = tin[c * 16 + g * 8 + e]; al[e] = t2.x; ral[e] = t2.y; }
                    snake_vec<8>(xreg[q], al, ral);          // snake(0) == 0: zero padding is preserved
                }
                bf16x8d hp[NPL];
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    __bf16 pv[NPL];
                    SP::split(xreg[q][e], pv);
#pragma unroll
                    for (int pl = 0; pl < NPL; pl++) hp[pl][e] = pv[pl];
                }
                __bf16 *p = xd + (g * xw + r) * 8;
#pragma unroll
                for (int pl = 0; pl < NPL; pl++) *(bf16x8d *) (p + pl * xpl) = hp[pl];
            }
        }
    };

    prefetch(0);
    __syncthreads();   // alpha table visible
    commit(0, 0);
    __syncthreads();
    for (int c = 0; c < n_chunks; c++) {
        const int buf = c & 1;
        if (c + 1 < n_chunks) prefetch(c + 1);
        const __bf16 *ws = wsb + buf * NPL * WPL;
        const __bf16 *xs = xsb + buf * NPL * xpl;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            bf16x8d bf[NPL];
#pragma unroll
            for (int pl = 0; pl < NPL; pl++) bf[pl] = *(const bf16x8d *) (xs + pl * xpl + (ks * xw + wn * 32 + l31 + 1 - hi) * 8);
#pragma unroll
            for (int ph = 0; ph < S; ph++) {
                bf16x8d af[MI][NPL];
#pragma unroll
                for (int i = 0; i < MI; i++)
#pragma unroll
                    for (int pl = 0; pl < NPL; pl++)
                        af[i][pl] = *(const bf16x8d *) (ws + pl * WPL + ((((ks * S + ph) * 2 + hi) * CO_T) + i * 32 + l31) * 8);
#pragma unroll
                for (int tm = 0; tm < SP::NT; tm++)
#pragma unroll
                    for (int i = 0; i < MI; i++)
                        acc[i][ph] = SP::mfma(af[i][SP::ta(tm)], bf[SP::tb(tm)], acc[i][ph]);
            }
        }
        if (c + 1 < n_chunks) commit(c + 1, buf ^ 1);
        __syncthreads();
    }

    // a lane holds the S consecutive outputs to = ti S - p .. ti S - p + S - 1 of every channel it owns: stored as wide as the alignment allows
    const int ti = ti0 + wn * 32 + l31;
    const int tob = ti * S - a.pad;
    // the lane's biases first, as 16-byte loads (one per quad of channels): a scalar load per value sat in front of every store group
    float4d b4[MI][4];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int cq = co0 + i * 32 + 8 * q + 4 * hi;   // a multiple of 4; cout is a multiple of 8
            b4[i][q] = (float4d){0.f, 0.f, 0.f, 0.f};
            if (a.b) b4[i][q] = *(const float4d *) (a.b + (cq < a.cout ? cq : 0));
        }
#pragma unroll
    for (int i = 0; i < MI; i++) {
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int co = co0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
            if (co >= a.cout) continue;
            const float bias = b4[i][e >> 2][e & 3];
            float *row = yg + (int64_t) co * LoS;
            if ((S % 4) == 0 && (a.pad % 4) == 0 && (LoS % 4) == 0) {
#pragma unroll
                for (int p4 = 0; p4 < S / 4; p4++) {
                    const int to = tob + 4 * p4;
                    if (to >= 0 && to + 3 < Lout) {
                        float4d v = {acc[i][4 * p4][e] + bias, acc[i][4 * p4 + 1][e] + bias, acc[i][4 * p4 + 2][e] + bias, acc[i][4 * p4 + 3][e] + bias};
                        *(float4d *) (row + to) = v;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            if (to + r >= 0 && to + r < Lout) row[to + r] = acc[i][4 * p4 + r][e] + bias;
                    }
                }
            } else {
#pragma unroll
                for (int ph = 0; ph < S; ph++) {
                    const int to = tob + ph;
                    if (to >= 0 && to < Lout) row[to] = acc[i][ph][e] + bias;
                }
            }
        }
    }
}

// ================================================================================================================================
// resunit_t7_kernel: resunit_b3_kernel with ONE TAP PER K-STEP in the k = 7 conv.  The 8-channel chunk of resunit_b3_kernel pairs taps
// (2 s, 2 s + 1) in a k-step, i.e. 8 slots for 7 taps: an eighth of the k = 7 MFMAs multiply zero weights.  Here a chunk is 16 input
// channels, k-step s is tap s and the half-wave takes 8-channel group hi: 7 k-steps per 16 channels.  The weight stream keeps 36.9 KB
// stages, now of unequal k-step counts per chunk: 96 channels {4, 3}, 192 channels {2, 2, 2, 1}.  The k = 1 phase, the epilogue and the
// packing of the k = 1 weights are resunit_b3_kernel's.
//   stage g = chunk * SPC + sub:  [plane][s < CNT[sub]][hi][co < C][j < 8] = plane of w7[co][ci = 16 chunk + 8 hi + j][tap = FIRST[sub] + s]
// ================================================================================================================================
template <int MI> struct ResT7 {
    static constexpr int SPC = MI == 3 ? 2 : 4;
    static constexpr int cnt(int sub) { return MI == 3 ? (sub == 0 ? 4 : 3) : (sub == 3 ? 1 : 2); }
    static constexpr int first(int sub) { return MI == 3 ? (sub == 0 ? 0 : 4) : 2 * sub; }
    static constexpr int MAXCNT = MI == 3 ? 4 : 2;
};

static __global__ void pack_resunit_t7_kernel(const float *w7, const float *w1, __bf16 *dst, int C, int KS2, int scheme = 0) {
    const int MI = C / 32, SPC = MI == 3 ? 2 : 4, MAXCNT = MI == 3 ? 4 : 2;
    const int64_t WST = (int64_t) split_planes(scheme) * MAXCNT * 2 * C * 8;
    const int n7 = (C / 16) * SPC, ns1 = (C / 16) / KS2, n1 = (C / 96) * ns1;
    const int64_t per1 = (int64_t) KS2 * 2 * 96 * 8;
    // k = 7 part: one thread per (stage, slot s < MAXCNT, hi, co, j); slots beyond the stage's k-step count are skipped
    const int64_t slot7 = (int64_t) MAXCNT * 2 * C * 8;
    const int64_t total = (int64_t) n7 * slot7 + (int64_t) n1 * per1;
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t) gridDim.x * blockDim.x) {
        float v = 0.0f;
        int64_t base, plane_sz;
        if (i < (int64_t) n7 * slot7) {
            const int st = (int) (i / slot7);
            int64_t r = i % slot7;
            const int j = (int) (r % 8); r /= 8;
            const int co = (int) (r % C); r /= C;
            const int hi = (int) (r % 2); r /= 2;
            const int s = (int) r;
            const int chunk = st / SPC, sub = st % SPC;
            const int cnt = MI == 3 ? (sub == 0 ? 4 : 3) : (sub == 3 ? 1 : 2), first = MI == 3 ? (sub == 0 ? 0 : 4) : 2 * sub;
            if (s >= cnt) continue;
            v = w7[((int64_t) co * C + chunk * 16 + hi * 8 + j) * 7 + first + s];
            plane_sz = (int64_t) cnt * 2 * C * 8;
            base = (int64_t) st * WST + ((int64_t) (s * 2 + hi) * C + co) * 8 + j;
        } else {
            const int64_t i1 = i - (int64_t) n7 * slot7;
            const int st = (int) (i1 / per1);
            int64_t r = i1 % per1;
            const int m = (int) (r % 8); r /= 8;
            const int col = (int) (r % 96); r /= 96;
            const int hi = (int) (r % 2); r /= 2;
            const int s = (int) r;
            const int pass = st / ns1, q = st % ns1;
            const int ks = q * KS2 + s, ib = ks / 2, qq = ks % 2;
            const int ch = 32 * ib + 16 * qq + (m < 4 ? 4 * hi + m : 8 + 4 * hi + m - 4);
            v = w1[(int64_t) (96 * pass + col) * C + ch];
            plane_sz = per1;
            base = (int64_t) (n7 + st) * WST + (i1 % per1);
        }
        split_store(scheme, v, dst, base, plane_sz);
    }
}

// SP: the operand split (dac_kernels.h: SplitB3 = bf16 x 3 / six products, SplitH2 = fp16 hi + lo / three products, SplitH1 = fp16 / one product)
// WDMA: the weight stages (already LDS images in memory) go memory -> LDS by global_load_lds_dwordx4 (1 KiB per wave instruction, lane-linear) instead
// of through 16-byte registers + ds_write_b128 (13 LDS-path cycles per wave instruction, 5 per thread and stage): no staging registers, no store phase.
__device__ __forceinline__ void dac_wait_vmcnt0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// VAR (bit mask):
//   2  the k = 7 accumulators are turned into the k = 1 conv's B operand ONCE (bias, snake, split; hi | lo << 16) instead of once per k-step and 96-channel
//      pass of the k = 1 conv; planes of 16 bits only (fp16 hi + lo, fp16).  (Kept as its own array: the same bits stored back into the float accumulators
//      were miscompiled.)
//   Measured and removed (profiles/r06/ru7_bench_variants.txt): 1 = waves that share a SIMD run the chunk's last stage in opposite orders (staging first /
//   MFMAs first): -4.5 % at 96 channels alone, nothing on top of 2; 4 = the staging's snake + split as straight-line code pinned between the MFMAs with
//   sched_group_barrier (range test collected, not branched on): k = 7 phase 152.5 K -> 179.9 K cycles, slower.
//   8 = the operand built tile by tile between the first pass's MFMAs, the SIMD's two waves one tile apart: 6.95 -> 7.41 ms at 192 channels; 16 = the 12 .. 108
//   halo units beyond 512 as one element per lane: -0.5 % at dilation 1, +1-2 % at dilation 9.  Whatever mixes vector work into the MFMA stream loses here.
// NI: position tiles of 32 per wave (workgroup = 256 NI positions).  NI = 2: a weight fragment read from LDS feeds two MFMAs instead of one, a stage carries twice
// the MFMAs per barrier and per weight byte; needs VAR 2 (the k = 1 phase then runs once per position tile, streaming its weight stages again).
template <int MI, int KS2, typename SP = SplitB3, bool WDMA = false, int VAR = 0, int NI = 1>
__global__ __launch_bounds__(512, 2) void resunit_t7_kernel(ResUnitArgs a) {
    constexpr int NPL = SP::NPL;
    using G = ResT7<MI>;
    constexpr int C = 32 * MI, WN = 8, NT = 512, T_T = 32 * WN * NI, NQ = NI + 1;
    constexpr int SPC = G::SPC, NCH = C / 16, N7 = NCH * SPC;
    constexpr int WPL1 = KS2 * 2 * 96 * 8;
    constexpr int WST = NPL * G::MAXCNT * 2 * C * 8;          // 16-bit values per stage (stream stride and LDS buffer) >= NPL * WPL1
    constexpr int NP = MI / 3, NS1 = (C / 16) / KS2, N1 = NP * NS1;
    constexpr int WV = (WST / 8 + NT - 1) / NT;               // 16-byte vectors per thread per stage
    static_assert(NPL * WPL1 <= WST && (C / 16) % KS2 == 0 && MI % 3 == 0, "stage shapes");
    static_assert(NI == 1 || ((VAR & 2) != 0 && NPL <= 2 && N1 % 2 == 0), "NI > 1: once-built operand, an even number of k = 1 stages");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int xw = T_T + 6 * a.dil;
    const int xpl = 2 * xw * 8;                               // bf16 per input plane of a chunk: [group 2][position][8]
    __bf16 *wsb = (__bf16 *) smem;                            // [2][WST]
    __bf16 *xsb = wsb + 2 * WST;                              // [2][NPL][xpl]
    float4 *tab = (float4 *) (xsb + 2 * NPL * xpl);             // [C] {b7, alpha_mid, 1/alpha_mid, b1}
    float2 *tin = (float2 *) (tab + C);                       // [C] {alpha_in, 1/alpha_in}
    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int t0 = blockIdx.x * T_T;
    const int LS = a.L, L = valid_len(a.frames, a.mult, a.L);
    if (t0 >= L) return;
    const float *xg = a.x + (int64_t) blockIdx.z * C * LS;
    float *yg = a.y + (int64_t) blockIdx.z * C * LS;
    const uint4d *wg = (const uint4d *) a.w;

    for (int i = tid; i < C; i += NT) {
        const float am = a.alpha_mid[i], ai = a.alpha_in[i];
        tab[i] = make_float4(a.b7[i], am, 1.0f / am, a.b1[i]);
        tin[i] = make_float2(ai, 1.0f / ai);
    }

    float16d acc[NI][MI];
#pragma unroll
    for (int j = 0; j < NI; j++)
#pragma unroll
        for (int i = 0; i < MI; i++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[j][i][e] = 0.0f;

    constexpr int NPIECE = WST * 2 / 1024;                    // 1-KiB pieces per stage (WDMA): piece wn + 8 j is wave wn's
    static_assert(!WDMA || (WST * 2) % 1024 == 0, "stage = whole 1-KiB pieces");
    uint4d wreg[WDMA ? 1 : WV];
    float xreg[NQ][8];                                        // (group, position) units u = tid, tid + 512, .. of the 2 * xw <= 512 NI + 108 of a chunk
    auto prefetch_w = [&](int g) __attribute__((always_inline)) {
        if constexpr (WDMA) {                                 // stage g lands in buffer g & 1 (free since the barrier that ended stage g - 1)
            const char *wp = (const char *) a.w + (int64_t) g * (WST * 2) + lane * 16;
            char *wd = (char *) (wsb + (g & 1) * WST);
#pragma unroll
            for (int j = 0; j < (NPIECE + 7) / 8; j++) {
                const int pc = wn + 8 * j;
                if (NPIECE % 8 == 0 || pc < NPIECE)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (wp + pc * 1024),
                                                     (__attribute__((address_space(3))) void *) (wd + pc * 1024), 16, 0, 0);
            }
        } else {
            const uint4d *wp = wg + (int64_t) g * (WST / 8);
#pragma unroll
            for (int j = 0; j < WV; j++) {
                const int i = tid + j * NT;
                if (i < WST / 8) wreg[j] = wp[i];
            }
        }
    };
    auto commit_w = [&](int buf) __attribute__((always_inline)) {
        if constexpr (WDMA) {
            dac_wait_vmcnt0();                                // this wave's pieces have landed; the barrier that follows publishes every wave's
        } else {
            uint4d *wd = (uint4d *) (wsb + buf * WST);
#pragma unroll
            for (int j = 0; j < WV; j++) {
                const int i = tid + j * NT;
                if (i < WST / 8) wd[i] = wreg[j];
            }
        }
    };
    auto prefetch_x = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int u = tid + q * NT;
            const int g = u >= xw ? 1 : 0, p = u - g * xw;   // lanes run along positions: coalesced rows
            const int t = t0 + p - a.pad;
            const bool ok = u < 2 * xw && t >= 0 && t < L;
#pragma unroll
            for (int e = 0; e < 8; e++) xreg[q][e] = ok ? xg[(int64_t) (c * 16 + g * 8 + e) * LS + t] : 0.0f;
        }
    };
    auto commit_x_q = [&](int c, int buf, int q) __attribute__((always_inline)) {
        {
            const int u = tid + q * NT;
            if (u < 2 * xw) {
                const int g = u >= xw ? 1 : 0;
                float al[8], ral[8];
#pragma unroll
                for (int e = 0; e < 8; e++) { const float2 t2 = tin[c * 16 + g * 8 + e]; al[e] = t2.x; ral[e] = t2.y; }
                snake_vec<8>(xreg[q], al, ral);               // snake(0) == 0: zero padding is preserved
                bf16x8d hp[NPL];
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    __bf16 pv[NPL];
                    SP::split(xreg[q][e], pv);
#pragma unroll
                    for (int pl = 0; pl < NPL; pl++) hp[pl][e] = pv[pl];
                }
                __bf16 *xd = xsb + buf * NPL * xpl + u * 8;   // u = g * xw + p: the [group][position] order of the image
#pragma unroll
                for (int pl = 0; pl < NPL; pl++) *(bf16x8d *) (xd + pl * xpl) = hp[pl];
            }
        }
    };
    auto commit_x = [&](int c, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NQ; q++) commit_x_q(c, buf, q);
    };

    constexpr bool ONCE = (VAR & 2) != 0 && NPL <= 2;
    RU_STAMP(0);
    prefetch_w(0);
    prefetch_x(0);
    __syncthreads();   // tables visible
    commit_w(0);
    commit_x(0, 0);
    __syncthreads();
    RU_STAMP(1);

    // ---- k = 7 conv: NCH chunks of 16 input channels, SPC stages each -----------------------------------------------------------------
    for (int c = 0; c < NCH; c++) {
        const __bf16 *xs = xsb + (c & 1) * NPL * xpl;
        static_for<SPC>([&](auto SUB) __attribute__((always_inline)) {
            constexpr int sub = decltype(SUB)::value, CNT = G::cnt(sub), FIRST = G::first(sub), WPL7 = CNT * 2 * C * 8;
            const int g = c * SPC + sub;
            prefetch_w(g + 1);                                   // the stream continues into the k = 1 stages
            if (sub == 0 && c + 1 < NCH) prefetch_x(c + 1);
            const __bf16 *ws = wsb + (g & 1) * WST;
            static_for<CNT>([&](auto S) __attribute__((always_inline)) {
                constexpr int s = decltype(S)::value;
                bf16x8d bf[NI][NPL];
#pragma unroll
                for (int j = 0; j < NI; j++)
#pragma unroll
                    for (int pl = 0; pl < NPL; pl++) bf[j][pl] = *(const bf16x8d *) (xs + pl * xpl + (hi * xw + (wn * NI + j) * 32 + l31 + (FIRST + s) * a.dil) * 8);
#pragma unroll
                for (int ig = 0; ig < MI; ig += 3) {
                    bf16x8d af[3][NPL];
#pragma unroll
                    for (int ii = 0; ii < 3; ii++)
#pragma unroll
                        for (int pl = 0; pl < NPL; pl++)
                            af[ii][pl] = *(const bf16x8d *) (ws + pl * WPL7 + (((s * 2 + hi) * C) + (ig + ii) * 32 + l31) * 8);
#pragma unroll
                    for (int j = 0; j < NI; j++)
#pragma unroll
                        for (int tm = 0; tm < SP::NT; tm++)
#pragma unroll
                            for (int ii = 0; ii < 3; ii++)
                                acc[j][ig + ii] = SP::mfma(af[ii][SP::ta(tm)], bf[j][SP::tb(tm)], acc[j][ig + ii]);
                }
            });
            commit_w((g + 1) & 1);
            if (sub == SPC - 1 && c + 1 < NCH) commit_x(c + 1, (c + 1) & 1);
            __syncthreads();
        });
    }
    RU_STAMP(2);
    uint32_t bop[NI][ONCE ? MI : 1][16];   // ONCE: the k = 1 conv's B operand, planes packed hi | lo << 16 (takes the accumulators' registers over)
    if constexpr (ONCE) {
#pragma unroll
      for (int j = 0; j < NI; j++)
#pragma unroll
        for (int i = 0; i < MI; i++)
#pragma unroll
            for (int hf = 0; hf < 2; hf++) {
                float hv[8], al[8], ral[8];
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    const int e = 8 * hf + m;
                    const float4 tb = tab[32 * i + (e & 3) + 8 * (e >> 2) + 4 * hi];
                    hv[m] = acc[j][i][e] + tb.x; al[m] = tb.y; ral[m] = tb.z;
                }
                snake_vec<8>(hv, al, ral);
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    __bf16 pv[NPL];
                    SP::split(hv[m], pv);
                    uint32_t w = __builtin_bit_cast(uint16_t, pv[0]);
                    if constexpr (NPL == 2) w |= (uint32_t) __builtin_bit_cast(uint16_t, pv[NPL - 1]) << 16;
                    bop[j][i][8 * hf + m] = w;
                }
            }
    }
    RU_STAMP(3);

    // ---- k = 1 conv: the accumulators (bias, snake, split) are its B fragments; 96 output channels per pass --------------------------
    float16d acc2[3];
    static_for<NI>([&](auto JT) __attribute__((always_inline)) {
    constexpr int jt = decltype(JT)::value;
    const int t = t0 + (wn * NI + jt) * 32 + l31;
    static_for<N1>([&](auto G2) __attribute__((always_inline)) {
        constexpr int g2 = decltype(G2)::value, p = g2 / NS1, q = g2 % NS1;
        if constexpr (q == 0) {
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc2[i][e] = 0.0f;
        }
        if constexpr (g2 + 1 < N1) prefetch_w(N7 + g2 + 1);
        else if constexpr (jt + 1 < NI) prefetch_w(N7);          // the next position tile streams the k = 1 stages again (N1 even: same buffers)
        const __bf16 *ws = wsb + ((N7 + g2) & 1) * WST;
        static_for<KS2>([&](auto S) __attribute__((always_inline)) {
            constexpr int s = decltype(S)::value, ks = q * KS2 + s, ib = ks / 2, qq = ks % 2;
            bf16x8d bf[NPL];
            if constexpr (ONCE) {
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    const uint32_t w = bop[jt][ib][8 * qq + m];
                    bf[0][m] = __builtin_bit_cast(__bf16, (uint16_t) (w & 0xffffu));
                    if constexpr (NPL == 2) bf[NPL - 1][m] = __builtin_bit_cast(__bf16, (uint16_t) (w >> 16));
                }
            } else {
                float hv[8], al[8], ral[8];
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    const int e = 8 * qq + m;
                    const int ch = 32 * ib + (e & 3) + 8 * (e >> 2) + 4 * hi;
                    const float4 tb = tab[ch];
                    hv[m] = acc[jt][ib][e] + tb.x; al[m] = tb.y; ral[m] = tb.z;
                }
                snake_vec<8>(hv, al, ral);
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    __bf16 pv[NPL];
                    SP::split(hv[m], pv);
#pragma unroll
                    for (int pl = 0; pl < NPL; pl++) bf[pl][m] = pv[pl];
                }
            }
            bf16x8d af[3][NPL];
#pragma unroll
            for (int ii = 0; ii < 3; ii++)
#pragma unroll
                for (int pl = 0; pl < NPL; pl++)
                    af[ii][pl] = *(const bf16x8d *) (ws + pl * WPL1 + (((s * 2 + hi) * 96) + ii * 32 + l31) * 8);
#pragma unroll
            for (int tm = 0; tm < SP::NT; tm++)
#pragma unroll
                for (int ii = 0; ii < 3; ii++)
                    acc2[ii] = SP::mfma(af[ii][SP::ta(tm)], bf[SP::tb(tm)], acc2[ii]);
        });
        if constexpr (g2 + 1 < N1 || jt + 1 < NI) {
            commit_w((N7 + g2 + 1) & 1);
            __syncthreads();
        }
        if constexpr (q == NS1 - 1) {   // + bias + x
            if (t < L) {
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    float rv[16];
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const int co = 96 * p + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * hi;
                        rv[e] = xg[(int64_t) co * LS + t];
                    }
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const int co = 96 * p + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * hi;
                        yg[(int64_t) co * LS + t] = (acc2[i][e] + tab[co].w) + rv[e];
                    }
                }
            }
        }
    });
    });
    RU_STAMP(4);
}

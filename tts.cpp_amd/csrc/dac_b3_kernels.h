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

__global__ void pack_resunit_b3_kernel(const float *w7, const float *w1, __bf16 *dst, int C, int KS, int KS2) {
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
};

template <int MI, int KS, int KS2>
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
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float2 al = tin[c * 8 + e];
                const float v = snake_f(xreg[e], al.x, al.y);   // snake(0) == 0: zero padding is preserved
                __bf16 b1, b2, b3;
                split_bf16x3(v, b1, b2, b3);
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

    // ---- k = 7 conv: N7 stages --------------------------------------------------------------------------------------------------
    for (int g = 0; g < N7; g++) {
        const int c = g / SPC, sub = g - c * SPC;
        prefetch_w(g + 1);                                     // the stream continues into the k = 1 stages
        if (sub == 0 && c + 1 < NCH) prefetch_x(c + 1);
        const __bf16 *ws = wsb + (g & 1) * WST;
        const __bf16 *xs = xsb + (c & 1) * 3 * xpl;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const int st = sub * KS + s;
            const int tap = (2 * st + hi < 7) ? 2 * st + hi : 6;   // the eighth tap: zero weights, any valid rows
            bf16x8d bf[3];
#pragma unroll
            for (int pl = 0; pl < 3; pl++) bf[pl] = *(const bf16x8d *) (xs + pl * xpl + (wn * 32 + l31 + tap * a.dil) * 8);
#pragma unroll
            for (int ig = 0; ig < MI; ig += 3) {
                bf16x8d af[3][3];
#pragma unroll
                for (int ii = 0; ii < 3; ii++)
#pragma unroll
                    for (int pl = 0; pl < 3; pl++)
                        af[ii][pl] = *(const bf16x8d *) (ws + pl * WPL7 + (((s * 2 + hi) * C) + (ig + ii) * 32 + l31) * 8);
                // term-major over the three accumulators: consecutive MFMAs write different registers
                constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int tm = 0; tm < 6; tm++)
#pragma unroll
                    for (int ii = 0; ii < 3; ii++)
                        acc[ig + ii] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ii][TA[tm]], bf[TB[tm]], acc[ig + ii], 0, 0, 0);
            }
        }
        commit_w((g + 1) & 1);
        if (sub == SPC - 1 && c + 1 < NCH) commit_x(c + 1, (c + 1) & 1);
        __syncthreads();
    }

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
        static_for<KS2>([&](auto S) __attribute__((always_inline)) {
            constexpr int s = decltype(S)::value, ks = q * KS2 + s, ib = ks / 2, qq = ks % 2;
            bf16x8d bf[3];
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const int e = 8 * qq + m;
                const int ch = 32 * ib + (e & 3) + 8 * (e >> 2) + 4 * hi;
                const float4 tb = tab[ch];
                const float v = snake_f(acc[ib][e] + tb.x, tb.y, tb.z);
                __bf16 b1, b2, b3;
                split_bf16x3(v, b1, b2, b3);
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
}

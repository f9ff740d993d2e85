// gemv_stream_kernels.h — weight-streaming GEMM for 1..16 rows (a decode step of one utterance, or of a few
// in lock-step: Dia's 4 utterances x 2 guidance rows, dia/model.cpp:806-870; Parler batch 1..16, parler/model.cpp:544-603).
//
//   y[r][n] = sum_k W[n][k] * act[r][k]        W fp16 [N][K] (ggml ne=[K,N]), act fp32 or fp16 [R][lda], R <= 16
//
// Same operation as gemm16_kernel (parler_kernels.h: activations rounded to fp16, fp32 accumulate), different
// schedule.  gemm16_kernel is one-shot: a workgroup owns 16 features, its K/256 waves each fetch one 256-column
// slice, meet in LDS, store, exit — the chip sees a burst of loads, then a tail of barriers and stores during which
// nothing is in flight, and every workgroup re-reads all R activation rows from L2 through the same per-CU load
// path the weights use.  Here the workgroups are persistent:
//   * a workgroup is bound to one K slice (kslice columns) and keeps the R activation rows of that slice in LDS
//     as fp16 (converted once), so the only global traffic of the loop is the weight stream;
//   * a wave walks (feature tile, 256-column chunk) pairs of its slice on its own: 8 x 16-byte loads per lane per
//     chunk, the next chunk's loads issued before the MFMAs of the current one (two register sets), accumulation
//     in the MFMA accumulator across the chunks of a tile — no cross-wave reduction, no barrier after the staging;
//   * K slices write fp32 slabs (EPI_STORE) that the consumer folds in slab order, as with gemm16_kernel's split-K.
// Fragment layout as in gemm16_kernel: A operand = 16 features, B operand = 16 rows, a lane ends with 4 consecutive
// features of row (lane & 15).
#pragma once
#include "parler_kernels.h"

struct StreamMap {
    int ks;      // K slices (gridDim.x is a multiple of ks)
    int kslice;  // columns per slice, a multiple of 256
};

template <int NWV, int PRO, int EPI>
__global__ __launch_bounds__(NWV * 64) void gemv_stream_kernel(GemmArgs a, StreamMap sm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int kz = (int) blockIdx.x % sm.ks, wg_in = (int) blockIdx.x / sm.ks, nwg_in = (int) gridDim.x / sm.ks;
    const int KS = sm.kslice, k0 = kz * KS;
    const int RS = a.R <= 8 ? 8 : 16;   // rows kept in LDS; B-operand columns >= RS repeat rows (never stored)
    const int ldx = KS + 32;            // +64 B per row: the 4 k-groups of a fragment read land in different banks
    _Float16 *xs = (_Float16 *) smem;

    const int tiles = a.N >> 4, nc = KS >> 8;
    const int tstep = nwg_in * NWV;
    int t = wg_in * NWV + wave, ch = 0;
    const _Float16 *wrow = (const _Float16 *) a.W + (int64_t) li * a.K + k0 + g * 8;
    half8 w0[8], w1[8];
    auto loadw = [&](half8 (&w)[8], int tile, int chunk) {
        const _Float16 *p = wrow + (int64_t) tile * 16 * a.K + chunk * 256;
#pragma unroll
        for (int c = 0; c < 8; c++) w[c] = __builtin_nontemporal_load((const half8 *) (p + c * 32));
    };
    // in flight under the staging.  Unconditional (a wave with no tile re-reads the last one): under `if (t < tiles)` the loads were a block of their
    // own whose first result was copied on the way out — an s_waitcnt on the first weight load in front of the staging loads, two dependent
    // round trips at the top of every launch.
    loadw(w0, min(t, tiles - 1), 0);

    // ---- stage this slice of the R rows as fp16 ------------------------------------------------------
    const int c8n = KS >> 3;
    for (int i = tid; i < RS * c8n; i += NWV * 64) {
        const int r = i / c8n, c8 = i - r * c8n;
        half8 h = {0, 0, 0, 0, 0, 0, 0, 0};
        if (r < a.R) {
            if (PRO == PRO_F16) {
                h = *(const half8 *) ((const _Float16 *) a.A + (int64_t) r * a.lda + k0 + c8 * 8);
            } else if (PRO == PRO_ATTN8) {
                // K = heads x 128: eight values of one head, merged from the key slices (attn_gqa_combine_kernel: running max in slice order,
                // o = sum f_z o_z, l = sum f_z l_z with f_z = expf(m_z - m), o / l; an empty slice left (max = -inf, sum = 0) and adds + 0)
                const int k = k0 + c8 * 8;
                const float *p = a.att_part + ((int64_t) r * (a.K >> 7) + (k >> 7)) * ATTN_FOLD_NZ * ATTN_PART;
                const int t0 = k & 127;
                float2v ml[ATTN_FOLD_NZ], v[ATTN_FOLD_NZ][4];
#pragma unroll
                for (int z = 0; z < ATTN_FOLD_NZ; z++) {
                    ml[z] = *(const float2v *) (p + z * ATTN_PART);
#pragma unroll
                    for (int j = 0; j < 4; j++) v[z][j] = *(const float2v *) (p + z * ATTN_PART + 2 + t0 + 2 * j);
                }
                __builtin_amdgcn_sched_barrier(0);   // every slice requested before the first is used
                float m = -INFINITY;
#pragma unroll
                for (int z = 0; z < ATTN_FOLD_NZ; z++) m = fmaxf(m, ml[z][0]);
                float o[8], l = 0.0f;
#pragma unroll
                for (int e = 0; e < 8; e++) o[e] = 0.0f;
#pragma unroll
                for (int z = 0; z < ATTN_FOLD_NZ; z++) {
                    const float mz = ml[z][0];
                    const bool live = mz != -INFINITY;
                    const float f = live ? expf(mz - m) : 0.0f;
#pragma unroll
                    for (int e = 0; e < 8; e++) o[e] += f * (live ? v[z][e >> 1][e & 1] : 0.0f);
                    l += f * ml[z][1];
                }
#pragma unroll
                for (int e = 0; e < 8; e++) h[e] = (_Float16) (o[e] / l);
            } else if (PRO == PRO_SILU) {
                // a.A = gate | up rows [R][2 K] as a.n_parts <= 8 slabs (a.parts_stride floats apart) of the preceding projection: slabs added in
                // slab order (a slab beyond n_parts re-reads the last one and is never added), silu(gate) * up — silu_mul_kernel's arithmetic
                const float *pg = (const float *) a.A + (int64_t) r * a.lda + k0 + c8 * 8;
                float4v gx[8][2], ux[8][2];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const float *pq = pg + (int64_t) min(q, a.n_parts - 1) * a.parts_stride;
                    gx[q][0] = *(const float4v *) pq; gx[q][1] = *(const float4v *) (pq + 4);
                    ux[q][0] = *(const float4v *) (pq + a.K); ux[q][1] = *(const float4v *) (pq + a.K + 4);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    float x = gx[0][e >> 2][e & 3], u = ux[0][e >> 2][e & 3];
#pragma unroll
                    for (int q = 1; q < 8; q++)
                        if (q < a.n_parts) { x += gx[q][e >> 2][e & 3]; u += ux[q][e >> 2][e & 3]; }
                    h[e] = (_Float16) ((x / (1.0f + expf(-x))) * u);
                }
            } else {
                const float *p = (const float *) a.A + (int64_t) r * a.lda + k0 + c8 * 8;
                const float4v f0 = *(const float4v *) p, f1 = *(const float4v *) (p + 4);
#pragma unroll
                for (int e = 0; e < 4; e++) { h[e] = (_Float16) f0[e]; h[4 + e] = (_Float16) f1[e]; }
            }
        }
        *(half8 *) (xs + (size_t) r * ldx + c8 * 8) = h;
    }
    __syncthreads();

    const _Float16 *xb = xs + (size_t) (li & (RS - 1)) * ldx + g * 8;
    float4v acc = {0.f, 0.f, 0.f, 0.f};
    // one pipeline step: issue the loads of the pair after (t, ch) into `nxt`, run the MFMAs of (t, ch) from `cur`
    auto step = [&](half8 (&cur)[8], half8 (&nxt)[8]) {
        int t2 = t, ch2 = ch + 1;
        if (ch2 == nc) { ch2 = 0; t2 = t + tstep; }
        if (t2 < tiles) loadw(nxt, t2, ch2);
        const _Float16 *xp = xb + ch * 256;
#pragma unroll
        for (int c = 0; c < 8; c++) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[c], *(const half8 *) (xp + c * 32), acc, 0, 0, 0);
        if (ch2 == 0) {
            if (li < a.R) gemm_epilogue4(a, EPI, li, t * 16 + g * 4, acc, kz);
            acc = (float4v){0.f, 0.f, 0.f, 0.f};
        }
        t = t2; ch = ch2;
    };
    while (t < tiles) {
        step(w0, w1);
        if (t >= tiles) break;
        step(w1, w0);
    }
}

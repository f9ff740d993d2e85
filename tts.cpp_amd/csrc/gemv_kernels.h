// gemv_kernels.h — streaming matrix-vector kernels for 1..4 activation rows (the batch-1 / two-stream steps of the Orpheus and
// Dia decoders: mul_mat of [N][K] weights with 1-2 rows, orpheus/model.cpp:240-283, dia/model.cpp:557-648).
//
// The 16-feature MFMA workgroups of gemm16_kernel / qgemm16_kernel are shaped for lock-step batches (the MFMA tile gives up to 16
// rows for free); with one or two rows and matrices of 2048..16384 features they leave few, large workgroups with a long
// dependent chain each (measured: 1.3 TB/s on the Orpheus-3B matrices, profiles/r01/kernel_stats_orpheus_3b_q4_0_first_version.csv).
// Here one wave owns one output feature: every lane streams 16-byte pieces of that weight row (all loads of a row are independent
// and issued back to back), the activation rows are tiny and stay in L2/L1, and a wave-wide sum finishes the feature.  N waves
// (thousands) fill the chip; no LDS, no barriers.
//   gemv_rows_kernel<WT, NR>    F32 / F16 weights; F16: activations rounded to fp16 first (ggml's vec_dot_type conversion), fp32 accumulate
//   gemv_q8_rows_kernel<NR>     int8 block integers + fp16 block scales (Q4_0/Q5_0/Q8_0 expanded at upload) x Q8_0-quantised activations:
//                               exact integer block dots (v_dot4_i32_i8), scaled by d_w * d_a and summed in fp32 (ggml's vec_dot_q*_q8_0)
//   gemv_q4_rows_kernel<NR>     the same for matrices whose GGUF type is Q4_0, reading the 4-bit codes themselves (16 bytes per block of 32
//                               instead of the 32 of the int8 expansion): sum (n - 8) x = sum n x - 8 sum x, both by dot4 on unpacked nibbles
//   repack_i8_to_q4_kernel      int8 expansion [N][K] -> nibble blocks [N][K/32][16] (ggml's Q4_0 packing: code j | code j+16 << 4)
// Opt-in (TTS_HIP_GEMV_ROWS=1, TTS_HIP_Q4_NATIVE=1) until measured against the MFMA path on the GPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "llama_kernels.h"   // q8_block_store

template <int WT, int NR>
__global__ __launch_bounds__(256) void gemv_rows_kernel(GemmArgs a, int epi) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= a.N) return;
    const int K = a.K;
    float acc[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) acc[r] = 0.0f;
    if (WT == 1) {
        const _Float16 *w = (const _Float16 *) a.W + (int64_t) n * K;
#pragma unroll 4
        for (int k = lane * 8; k < K; k += 512) {   // K % 256 == 0 on this path: a lane's 8 columns never straddle the end
            const half8 wv = *(const half8 *) (w + k);
#pragma unroll
            for (int r = 0; r < NR; r++) {
                if (r < a.R) {
                    const float *x = (const float *) a.A + (int64_t) r * a.lda + k;
                    const float4v x0 = *(const float4v *) x, x1 = *(const float4v *) (x + 4);
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        acc[r] += (float) wv[e] * (float) (_Float16) x0[e];
                        acc[r] += (float) wv[4 + e] * (float) (_Float16) x1[e];
                    }
                }
            }
        }
    } else {
        const float *w = (const float *) a.W + (int64_t) n * K;
#pragma unroll 4
        for (int k = lane * 4; k < K; k += 256) {
            const float4v wv = *(const float4v *) (w + k);
#pragma unroll
            for (int r = 0; r < NR; r++) {
                if (r < a.R) {
                    const float4v x0 = *(const float4v *) ((const float *) a.A + (int64_t) r * a.lda + k);
#pragma unroll
                    for (int e = 0; e < 4; e++) acc[r] += wv[e] * x0[e];
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < NR; r++) {
        if (r < a.R) {
            const float s = wave_sum(acc[r]);
            if (lane == 0) {
                float *o = a.out + (int64_t) r * a.ldo + n;
                if (epi == EPI_RESID) *o += s;
                else *o = s;
            }
        }
    }
}

template <int NR>
__global__ __launch_bounds__(256) void gemv_q8_rows_kernel(QGemmArgs qa, int epi) {
    const GemmArgs &a = qa.g;
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= a.N) return;
    const int K = a.K, nb = K >> 5;
    const int8_t *w = (const int8_t *) a.W + (int64_t) n * K;
    const _Float16 *wd = qa.wd + (int64_t) n * nb;
    float acc[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) acc[r] = 0.0f;
#pragma unroll 2
    for (int b = lane; b < nb; b += 64) {   // one 32-value block per lane and pass: two 16-byte loads
        const int4v w0 = *(const int4v *) (w + b * 32), w1 = *(const int4v *) (w + b * 32 + 16);
        const float dw = (float) wd[b];
#pragma unroll
        for (int r = 0; r < NR; r++) {
            if (r < a.R) {
                const int8_t *xq = qa.aq + (int64_t) r * K + b * 32;
                const int4v x0 = *(const int4v *) xq, x1 = *(const int4v *) (xq + 16);
                int s = 0;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    s = __builtin_amdgcn_sdot4(w0[e], x0[e], s, false);
                    s = __builtin_amdgcn_sdot4(w1[e], x1[e], s, false);
                }
                acc[r] += (float) s * (dw * qa.ad[(int64_t) r * nb + b]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < NR; r++) {
        if (r < a.R) {
            const float s = wave_sum(acc[r]);
            if (lane == 0) {
                float *o = a.out + (int64_t) r * a.ldo + n;
                if (epi == EPI_RESID) *o += s;
                else *o = s;
            }
        }
    }
}


static __global__ void repack_i8_to_q4_kernel(const int8_t *q, uint8_t *out, int64_t n_bytes) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_bytes) return;
    const int64_t b = i >> 4;
    const int j = (int) (i & 15);
    const unsigned lo = (unsigned) (q[b * 32 + j] + 8) & 0xFu, hi = (unsigned) (q[b * 32 + 16 + j] + 8) & 0xFu;
    out[i] = (uint8_t) (lo | (hi << 4));
}

template <int NR>
__global__ __launch_bounds__(256) void gemv_q4_rows_kernel(QGemmArgs qa, const uint8_t *w4, int epi) {
    const GemmArgs &a = qa.g;
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= a.N) return;
    const int K = a.K, nb = K >> 5;
    const uint8_t *w = w4 + (int64_t) n * (K >> 1);
    const _Float16 *wd = qa.wd + (int64_t) n * nb;
    float acc[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) acc[r] = 0.0f;
#pragma unroll 2
    for (int b = lane; b < nb; b += 64) {   // one block of 32 codes per lane and pass: one 16-byte load
        const int4v wn = *(const int4v *) (w + b * 16);
        const float dw = (float) wd[b];
#pragma unroll
        for (int r = 0; r < NR; r++) {
            if (r < a.R) {
                const int8_t *xq = qa.aq + (int64_t) r * K + b * 32;
                const int4v x0 = *(const int4v *) xq, x1 = *(const int4v *) (xq + 16);
                int s = 0, sx = 0;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int lo = wn[e] & 0x0F0F0F0F, hi = (wn[e] >> 4) & 0x0F0F0F0F;   // codes 4e..4e+3 and 16+4e..16+4e+3 as bytes 0..15
                    s = __builtin_amdgcn_sdot4(lo, x0[e], s, false);
                    s = __builtin_amdgcn_sdot4(hi, x1[e], s, false);
                    sx = __builtin_amdgcn_sdot4(0x01010101, x0[e], sx, false);
                    sx = __builtin_amdgcn_sdot4(0x01010101, x1[e], sx, false);
                }
                acc[r] += (float) (s - 8 * sx) * (dw * qa.ad[(int64_t) r * nb + b]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < NR; r++) {
        if (r < a.R) {
            const float sm_ = wave_sum(acc[r]);
            if (lane == 0) {
                float *o = a.out + (int64_t) r * a.ldo + n;
                if (epi == EPI_RESID) *o += sm_;
                else *o = sm_;
            }
        }
    }
}


// The Q4_0 codes and block scales a wave needs for NF features over NP passes of 64 blocks (block b = lane + 64 p of every feature), requested
// in one batch of non-temporal loads BEFORE the workgroup's staging prologue, so that the HBM round trip runs under the rms norm / quantisation
// instead of after it (round 3: the loads were issued two passes at a time after the barrier — a few hundred bytes in flight per wave).  A pass
// beyond the row re-reads the row's last block and is never used.  The consumer keeps the old order (passes ascending per lane, then the wave
// sum): results are bit-identical to the kernels of round 2 / 3.
template <int NF, int NP>
struct Q4Frag {
    int4v wn[NF][NP];
    _Float16 dw[NF][NP];
};
template <int NF, int NP>
__device__ __forceinline__ void q4_frag_load(Q4Frag<NF, NP> &fr, const uint8_t *w4, const _Float16 *wd, const int (&nf)[NF], int K, int nb, int b0, int lane) {
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const int b = min(b0 + lane + 64 * p, nb - 1);
#pragma unroll
        for (int f = 0; f < NF; f++) {
            fr.wn[f][p] = __builtin_nontemporal_load((const int4v *) (w4 + (int64_t) nf[f] * (K >> 1) + b * 16));
            fr.dw[f][p] = __builtin_nontemporal_load(wd + (int64_t) nf[f] * nb + b);   // the scales are a ninth of the stream: as dead as the codes once used
        }
    }
}
// acc[f][r] += the block products of pass p for every row (activations as Q8_0 rows sx [R][K] + scales sd [R][nb] in LDS)
template <int NF, int NP, int NR>
__device__ __forceinline__ void q4_frag_dot(const Q4Frag<NF, NP> &fr, const int8_t *sx, const float *sd, int R, int K, int nb, int b0, int lane, float (&acc)[NF][NR]) {
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const int b = b0 + lane + 64 * p;
        if (b < nb) {
#pragma unroll
            for (int r = 0; r < NR; r++) {
                if (r < R) {
                    const int4v x0 = *(const int4v *) (sx + (size_t) r * K + b * 32), x1 = *(const int4v *) (sx + (size_t) r * K + b * 32 + 16);
                    const float da = sd[r * nb + b];
                    int sxs = 0;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        sxs = __builtin_amdgcn_sdot4(0x01010101, x0[e], sxs, false);
                        sxs = __builtin_amdgcn_sdot4(0x01010101, x1[e], sxs, false);
                    }
#pragma unroll
                    for (int f = 0; f < NF; f++) {
                        int s = 0;
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const int lo = fr.wn[f][p][e] & 0x0F0F0F0F, hi = (fr.wn[f][p][e] >> 4) & 0x0F0F0F0F;
                            s = __builtin_amdgcn_sdot4(lo, x0[e], s, false);
                            s = __builtin_amdgcn_sdot4(hi, x1[e], s, false);
                        }
                        acc[f][r] += (float) (s - 8 * sxs) * ((float) fr.dw[f][p] * da);
                    }
                }
            }
        }
    }
}

// The same Q4_0 x Q8_0 row products as gemv_q4_rows_kernel (identical per-feature arithmetic: lane -> block mapping, fp32 accumulation order,
// wave reduction), scheduled for the load path: that kernel issues five load instructions per 16 bytes of weights (codes, block scale,
// two activation vectors, activation scale — the last three are the same for every feature but still go through the CU's address /
// L1 path), and one feature per wave leaves one or two 16-byte loads in flight per lane.  Here the Q8_0 activation rows and their
// scales are staged in LDS once per workgroup, a wave owns FPW consecutive features and requests the codes and scales of all of them
// for a block column before the first integer dot.
// QSRC 1: the activations arrive as fp32 rows (a.A, lda == K) and are Q8_0-quantised while they are staged — ggml's quantize_row_q8_0_ref
// arithmetic, identical to quant_rows_q8_kernel / q8_block_store (d = amax / 127 kept as fp16, q = roundf(x / d)) — one 32-value block per
// thread and pass; saves the producer a separate quantisation (the silu * up product of gemv_q4_gateup_silu_kernel feeds the down projection).
//
// Order of the requests (round 5).  vmcnt retires in issue order: a wave that requests its streamed weights first and its staging inputs second
// cannot touch the inputs before the weights have landed — the staging prologue the weights were meant to fly under started one HBM round trip
// late in every kernel of the one-sequence chains (profiles/tools/isa_wait_order.py lists them; profiles/r05/isa_wait_order_before.txt).  So: the
// first pass of staging inputs (L2 hits, a few hundred bytes per lane) is requested FIRST, the weights right behind it, and a pass loop only
// handles what is left (nothing at the Orpheus / Dia shapes).
struct StageQ8 {      // QSRC 0: item tid of the Q8_0 rows (16 codes) and of their block scales
    int4v a0;
    float d0;
};
__device__ __forceinline__ void stage_q8_load(StageQ8 &st, const QGemmArgs &qa, int R, int K, int nb, int tid) {
    st.a0 = ((const int4v *) qa.aq)[min(tid, R * (K >> 4) - 1)];
    st.d0 = qa.ad[min(tid, R * nb - 1)];
}
__device__ __forceinline__ void stage_q8_store(const StageQ8 &st, const QGemmArgs &qa, int R, int K, int nb, int tid, int8_t *sx, float *sd) {
    if (tid < R * (K >> 4)) ((int4v *) sx)[tid] = st.a0;
    if (tid < R * nb) sd[tid] = st.d0;
    for (int i = tid + 256; i < R * (K >> 4); i += 256) ((int4v *) sx)[i] = ((const int4v *) qa.aq)[i];
    for (int i = tid + 256; i < R * nb; i += 256) sd[i] = qa.ad[i];
    __syncthreads();
}
struct StageF32 {     // QSRC 1: block item tid of the fp32 rows (32 values)
    float4v v[8];
};
__device__ __forceinline__ void stage_f32_load(StageF32 &st, const GemmArgs &a, int nb, int item) {
    const int r = item / nb, b = item - r * nb;
    const float4v *src = (const float4v *) ((const float *) a.A + (int64_t) r * a.lda + b * 32);
#pragma unroll
    for (int j = 0; j < 8; j++) st.v[j] = src[j];
}
__device__ __forceinline__ void stage_f32_quant(const StageF32 &st, int K, int nb, int item, int8_t *sx, float *sd) {
    const int r = item / nb, b = item - r * nb;
    float amax = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
        for (int e = 0; e < 4; e++) amax = fmaxf(amax, fabsf(st.v[j][e]));
    const float dd = amax / 127.0f;
    const float id = dd ? 1.0f / dd : 0.0f;
    int4v q[2];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        unsigned pk = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) pk |= ((unsigned) (int) (int8_t) roundf(st.v[j][e] * id) & 0xFFu) << (8 * e);
        q[j >> 2][j & 3] = (int) pk;
    }
    *(int4v *) (sx + (size_t) r * K + b * 32) = q[0];
    *(int4v *) (sx + (size_t) r * K + b * 32 + 16) = q[1];
    sd[item] = (float) (_Float16) dd;
}
template <int NR, int FPW, int QSRC = 0, int NP = 2>
__global__ __launch_bounds__(256) void gemv_q4_rows_lds_kernel(QGemmArgs qa, const uint8_t *w4, int epi) {
    extern __shared__ __attribute__((aligned(16))) char gq_sm[];
    const GemmArgs &a = qa.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.K, nb = K >> 5, R = a.R;
    // the workgroup's staging inputs first (see above), this wave's weights right behind them
    StageQ8 s8;
    StageF32 s32;
    if (QSRC == 0) stage_q8_load(s8, qa, R, K, nb, tid);
    else stage_f32_load(s32, a, nb, min(tid, R * nb - 1));
    __builtin_amdgcn_sched_barrier(0);
    const int n0 = ((int) blockIdx.x * 4 + wave) * FPW;
    int nf[FPW];
#pragma unroll
    for (int f = 0; f < FPW; f++) nf[f] = min(n0 + f, a.N - 1);
    Q4Frag<FPW, NP> fr;
    q4_frag_load<FPW, NP>(fr, w4, qa.wd, nf, K, nb, 0, lane);
    __builtin_amdgcn_sched_barrier(0);
    int8_t *sx = (int8_t *) gq_sm;                       // [R][K]
    float *sd = (float *) (gq_sm + (size_t) R * K);      // [R][nb]
    if (QSRC == 0) {
        stage_q8_store(s8, qa, R, K, nb, tid, sx, sd);
    } else {
        if (tid < R * nb) stage_f32_quant(s32, K, nb, tid, sx, sd);
        for (int i = tid + 256; i < R * nb; i += 256) {
            stage_f32_load(s32, a, nb, i);
            stage_f32_quant(s32, K, nb, i, sx, sd);
        }
        __syncthreads();
    }
    float acc[FPW][NR];
#pragma unroll
    for (int f = 0; f < FPW; f++)
#pragma unroll
        for (int r = 0; r < NR; r++) acc[f][r] = 0.0f;
    q4_frag_dot<FPW, NP, NR>(fr, sx, sd, R, K, nb, 0, lane, acc);
    for (int b0 = 64 * NP; b0 < nb; b0 += 64 * NP) {   // rows longer than 2048 NP values (none at the Orpheus shapes with the NP the host picks)
        q4_frag_load<FPW, NP>(fr, w4, qa.wd, nf, K, nb, b0, lane);
        q4_frag_dot<FPW, NP, NR>(fr, sx, sd, R, K, nb, b0, lane, acc);
    }
    if (n0 >= a.N) return;
#pragma unroll
    for (int f = 0; f < FPW; f++) {
        const int n = n0 + f;
#pragma unroll
        for (int r = 0; r < NR; r++) {
            if (r < R) {
                const float sm_ = wave_sum(acc[f][r]);
                if (lane == 0 && n < a.N) {
                    float *o = a.out + (int64_t) r * a.ldo + n;
                    if (epi == EPI_RESID) *o += sm_;
                    else *o = sm_;
                }
            }
        }
    }
}


// The q/k/v projection of a Llama decode step (orpheus/model.cpp:194-221) with ggml_rope_ext (NeoX pairs i, i + 64 of a 128-wide head,
// frequency factors) and the K/V cache append in the epilogue: gemv_q4_rows_lds_kernel with a wave owning features i and i + 64 of one head,
// so that lane 0 holds a rotation pair once the wave sums are done — the arithmetic of llama_rope_kv_kernel (iterated theta, cosf / sinf),
// one launch less per layer.  q goes to `out` (the attention kernels read it there), rotated k and plain v go straight to the cache rows
// of pos[r].
// Staging variant for the two projections that follow an rms norm (orpheus/model.cpp:122-125): the workgroup normalises the R rows itself —
// rms_fold_rows_kernel's arithmetic to the letter (thread t holds elements t + 256 k, per-thread sums, wave sums, (w0 + w1) + (w2 + w3),
// o = x * scale * weight, Q8_0 blocks by q8_block_store) — and leaves the Q8_0 rows in LDS: the norm needs no launch of its own, at the price
// of every workgroup reading the 12 KB row and its weight from L2.
struct RmsSrc {
    const float *x;   // [R][H] the residual stream
    const float *w;   // [H]
    float eps;
};
struct RmsRow {       // a thread's 16 elements of one residual row and of the norm weight (element tid + 256 k)
    float v[16], wv[16];
};
// straight-line loads with clamped indices (a chunk beyond the row re-reads its last element and is never used): under `if (i < H)` every
// chunk was its own basic block with a full s_waitcnt in front — 12 dependent L2 round trips for a 3072-wide row, most of this prologue
__device__ __forceinline__ void rms_row_load(RmsRow &row, const RmsSrc &rs, int r, int H, int tid) {
    const float *xr = rs.x + (int64_t) r * H;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int i = min(tid + k * 256, H - 1);
        row.v[k] = xr[i]; row.wv[k] = rs.w[i];
    }
}
__device__ __forceinline__ void rms_row_stage(const RmsRow &row, const RmsSrc &rs, int r, int H, int8_t *sx, float *sd, float *red, int tid) {
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; k++)
        if (tid + k * 256 < H) s += row.v[k] * row.v[k];
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    const float scale = 1.0f / sqrtf(s / (float) H + rs.eps);
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int i = tid + k * 256;
        if (i < H) q8_block_store(row.v[k] * scale * row.wv[k], (int64_t) r * H + i, sx, sd);
    }
    __syncthreads();   // red is reused by the next row; the blocks are visible
}
// row 0 was requested before the weights (`first`); further rows (R <= 4) are requested here, behind them
__device__ __forceinline__ void stage_rms_q8(RmsRow &first, const RmsSrc &rs, int R, int H, int8_t *sx, float *sd, float *red) {
    const int tid = threadIdx.x;
    rms_row_stage(first, rs, 0, H, sx, sd, red, tid);
    for (int r = 1; r < R; r++) {
        rms_row_load(first, rs, r, H, tid);
        rms_row_stage(first, rs, r, H, sx, sd, red, tid);
    }
}

struct RopeEpi {
    const uint32_t *pos;   // [R]
    const float *ff;       // [64] frequency factors or NULL
    float theta_scale;
    int NH, NKV;
    float *kcache, *vcache;   // this layer: [n_ctx][NKV * 128]
};
template <int NR, int QSRC = 0, int NP = 2>
__global__ __launch_bounds__(256) void gemv_q4_qkv_rope_kernel(QGemmArgs qa, const uint8_t *w4, RopeEpi re, RmsSrc rs = RmsSrc{}) {
    extern __shared__ __attribute__((aligned(16))) char gq_sm[];
    const GemmArgs &a = qa.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.K, nb = K >> 5, R = a.R;
    // rotation pair p = features i and i + 64 of one head
    const int p = (int) blockIdx.x * 4 + wave;
    const bool active = p * 2 < a.N;
    const int pc = active ? p : 0;
    const int head = pc >> 6, i = pc & 63;
    const int nf[2] = {head * 128 + i, head * 128 + i + 64};
    // requests in the order their consumers run (vmcnt retires in issue order, see gemv_q4_rows_lds_kernel): the staging inputs and what the
    // epilogue needs (positions, frequency factor: they were a round trip of their own after the wave sums) first, the weights behind them
    RmsRow row;
    StageQ8 s8;
    if (QSRC == 2) rms_row_load(row, rs, 0, K, tid);
    else stage_q8_load(s8, qa, R, K, nb, tid);
    uint32_t ps[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) ps[r] = re.pos[min(r, R - 1)];
    const float ffi = re.ff ? re.ff[i] : 1.0f;
    __builtin_amdgcn_sched_barrier(0);
    Q4Frag<2, NP> fr;
    q4_frag_load<2, NP>(fr, w4, qa.wd, nf, K, nb, 0, lane);
    __builtin_amdgcn_sched_barrier(0);
    int8_t *sx = (int8_t *) gq_sm;
    float *sd = (float *) (gq_sm + (size_t) R * K);
    if (QSRC == 2) stage_rms_q8(row, rs, R, K, sx, sd, sd + (size_t) R * nb);
    else stage_q8_store(s8, qa, R, K, nb, tid, sx, sd);
    // the rotation of this wave's pair (llama_rope_kv_kernel's arithmetic: iterated theta, cosf / sinf), computed while the weights are in flight
    float cs[NR], sn[NR];
    if (head < re.NH + re.NKV) {
#pragma unroll
        for (int r = 0; r < NR; r++) {
            if (r < R) {
                float theta = (float) ps[r];
                for (int j = 0; j < i; j++) theta *= re.theta_scale;
                const float ang = theta / ffi;
                cs[r] = cosf(ang); sn[r] = sinf(ang);
            }
        }
    }
    float acc[2][NR];
#pragma unroll
    for (int f = 0; f < 2; f++)
#pragma unroll
        for (int r = 0; r < NR; r++) acc[f][r] = 0.0f;
    q4_frag_dot<2, NP, NR>(fr, sx, sd, R, K, nb, 0, lane, acc);
    for (int b0 = 64 * NP; b0 < nb; b0 += 64 * NP) {
        q4_frag_load<2, NP>(fr, w4, qa.wd, nf, K, nb, b0, lane);
        q4_frag_dot<2, NP, NR>(fr, sx, sd, R, K, nb, b0, lane, acc);
    }
    if (!active) return;
    const int kvH = re.NKV * 128;
#pragma unroll
    for (int r = 0; r < NR; r++) {
        if (r < R) {
            const float x0 = wave_sum(acc[0][r]), x1 = wave_sum(acc[1][r]);
            if (lane == 0) {
                if (head < re.NH + re.NKV) {
                    const float y0 = x0 * cs[r] - x1 * sn[r], y1 = x0 * sn[r] + x1 * cs[r];
                    if (head < re.NH) {
                        float *o = a.out + (int64_t) r * a.ldo + nf[0];
                        o[0] = y0; o[64] = y1;
                    } else {
                        float *kc = re.kcache + (int64_t) ps[r] * kvH + (head - re.NH) * 128 + i;
                        kc[0] = y0; kc[64] = y1;
                    }
                } else {
                    float *vd = re.vcache + (int64_t) ps[r] * kvH + (head - re.NH - re.NKV) * 128 + i;
                    vd[0] = x0; vd[64] = x1;
                }
            }
        }
    }
}


// gate and up projections of a Llama FFN with silu(gate) * up in the epilogue (orpheus/model.cpp:274-279): gemv_q4_rows_lds_kernel with a
// wave owning rows i, i + 1 of gate and rows F + i, F + i + 1 of up (the stacked [gate; up] matrix), so that lane 0 holds both factors of
// two outputs once the wave sums are done; g [R][F] fp32 goes to the down projection, which quantises it while staging (QSRC 1) —
// silu_mul_kernel's arithmetic, one launch less per layer.
template <int NR, int QSRC = 0, int NP = 2>
__global__ __launch_bounds__(256) void gemv_q4_gateup_silu_kernel(QGemmArgs qa, const uint8_t *w4, int F, float *gout, RmsSrc rs = RmsSrc{}) {
    extern __shared__ __attribute__((aligned(16))) char gq_sm[];
    const GemmArgs &a = qa.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.K, nb = K >> 5, R = a.R;
    // A wave walks items (rows i, i + 1 of gate and of up) item, item + n_waves, ...: the host launches at most two workgroups per CU, all of them
    // resident from the start, so the staging prologue (rms norm + Q8_0 blocks: every workgroup normalises the row itself) runs once per
    // workgroup and never in a second round of workgroups behind the first (1024 one-item workgroups at 170 registers ran as two rounds: 19.7 us).
    // The weights of the next item are requested before the current one is consumed; the first item's fly under the prologue.
    const int n_items = (F + 1) / 2, n_waves = (int) gridDim.x * 4;
    int item = (int) blockIdx.x * 4 + wave;
    auto rows_of = [&](int it, int (&nf)[4]) __attribute__((always_inline)) {
        const int i0 = min(2 * it, F - 1), i1 = min(i0 + 1, F - 1);
        nf[0] = i0; nf[1] = i1; nf[2] = F + i0; nf[3] = F + i1;
    };
    int nfa[4], nfb[4];
    Q4Frag<4, NP> fa, fb;
    rows_of(min(item, n_items - 1), nfa);
    // the staging inputs first, the first item's weights behind them (vmcnt retires in issue order, see gemv_q4_rows_lds_kernel)
    RmsRow row;
    StageQ8 s8;
    if (QSRC == 2) rms_row_load(row, rs, 0, K, tid);
    else stage_q8_load(s8, qa, R, K, nb, tid);
    __builtin_amdgcn_sched_barrier(0);
    q4_frag_load<4, NP>(fa, w4, qa.wd, nfa, K, nb, 0, lane);
    __builtin_amdgcn_sched_barrier(0);
    int8_t *sx = (int8_t *) gq_sm;
    float *sd = (float *) (gq_sm + (size_t) R * K);
    if (QSRC == 2) stage_rms_q8(row, rs, R, K, sx, sd, sd + (size_t) R * nb);
    else stage_q8_store(s8, qa, R, K, nb, tid, sx, sd);
    auto finish = [&](Q4Frag<4, NP> &fr, const int (&nf)[4], int it) __attribute__((always_inline)) {
        float acc[4][NR];
#pragma unroll
        for (int f = 0; f < 4; f++)
#pragma unroll
            for (int r = 0; r < NR; r++) acc[f][r] = 0.0f;
        q4_frag_dot<4, NP, NR>(fr, sx, sd, R, K, nb, 0, lane, acc);
        for (int b0 = 64 * NP; b0 < nb; b0 += 64 * NP) {
            q4_frag_load<4, NP>(fr, w4, qa.wd, nf, K, nb, b0, lane);
            q4_frag_dot<4, NP, NR>(fr, sx, sd, R, K, nb, b0, lane, acc);
        }
        const int i0 = 2 * it;
#pragma unroll
        for (int r = 0; r < NR; r++) {
            if (r < R) {
                const float g0 = wave_sum(acc[0][r]), g1 = wave_sum(acc[1][r]), u0 = wave_sum(acc[2][r]), u1 = wave_sum(acc[3][r]);
                if (lane == 0) {
                    gout[(int64_t) r * F + i0] = (g0 / (1.0f + expf(-g0))) * u0;
                    if (i0 + 1 < F) gout[(int64_t) r * F + i0 + 1] = (g1 / (1.0f + expf(-g1))) * u1;
                }
            }
        }
    };
    while (item < n_items) {
        const int nxt = item + n_waves;
        if (nxt < n_items) { rows_of(nxt, nfb); q4_frag_load<4, NP>(fb, w4, qa.wd, nfb, K, nb, 0, lane); }
        finish(fa, nfa, item);
        if (nxt >= n_items) break;
        item = nxt + n_waves;
        if (item < n_items) { rows_of(item, nfa); q4_frag_load<4, NP>(fa, w4, qa.wd, nfa, K, nb, 0, lane); }
        finish(fb, nfb, nxt);
    }
}

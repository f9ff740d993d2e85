// llama_kernels.h — the non-GEMM pieces of the Orpheus decoder (Llama-3 blocks,
// /root/reference/src/models/orpheus/model.cpp:122-125,186-296).  The projections go through gemm16_kernel /
// qgemm16_kernel (parler_kernels.h) — Q4_0 GGUFs (BASELINE config 4) on the integer path.
//   rms_fold_rows_kernel   orpheus_build_layer_norm: ggml_rms_norm(eps 1e-5) * weight :122-125 (+ folds split-K slabs of
//                          the preceding down_proj into the residual stream, ggml_add :283)
//   llama_rope_kv_kernel   ggml_rope_ext(NEOX, frequency factors) on q and k, K/V cache append :186-221,248-251
//   attn_gqa_kernel        mul_mat(k, q) -> soft_max_ext(causal, 1/sqrt(d)) -> mul_mat(kq, v), kv head = q head / rep :228-259
//   silu_mul_kernel        silu(gate x) * (up x) :279
//   argmax_parts/fold      sampler::max over the 156 940-logit vocabulary, two stages
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "parler_kernels.h"   // smp_key: the candidate order of the device samplers

// Q8_0 block of 32 consecutive values held one per lane by 32 neighbouring lanes (ggml's quantize_row_q8_0_ref: d = amax / 127 kept
// as fp16, q = roundf(x / d)) — the same arithmetic as quant_rows_q8_kernel, for producers that quantise their own output row.
__device__ __forceinline__ void q8_block_store(float v, int64_t idx, int8_t *aq, float *ad) {
    const float amax = lanes32_max(fabsf(v));
    const float dd = amax / 127.0f;
    const float id = dd ? 1.0f / dd : 0.0f;
    aq[idx] = (int8_t) roundf(v * id);
    if ((idx & 31) == 0) ad[idx >> 5] = (float) (_Float16) dd;
}

// one workgroup (4 waves) per row; the row (H <= 4096) is held in registers: every load of the row, the slabs and the weight is in
// flight at once (the first version walked the row twice with dependent loads: 9.4 us for one 3072-wide row).
// aq / ad (optional): the normalised row is also written as Q8_0 blocks (aq int8 [R][H], ad [R][H/32]) for an integer GEMV that
// follows immediately (saves that GEMV's quant_rows_q8_kernel launch).
static __global__ __launch_bounds__(256) void rms_fold_rows_kernel(float *x, int H, const float *w, float *y, int R, float eps, const float *parts, int n_parts,
                                                            int64_t slab_stride, int8_t *aq, float *ad) {
    __shared__ float red[4];
    const int r = blockIdx.x, tid = threadIdx.x;
    float *xr = x + (int64_t) r * H;
    if (H <= 4096) {
        // Every load of the row, its weight and the slabs is issued straight-line with a clamped index (a chunk beyond the row re-reads the row's
        // last element and is never used); only arithmetic and stores sit under `i < H`.  Loads under that predicate were one basic block each
        // with a full s_waitcnt in front: 18 dependent round trips for one row (profiles/r03/isa_serial_loads_final.txt), 4.7-6 us per launch of
        // a kernel the Dia step launches three times per layer.
        float v[16], wv[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int i = min(tid + k * 256, H - 1);
            v[k] = xr[i]; wv[k] = w[i];
        }
        if (parts) {
            int p = 0;
            if (H <= 2048 && n_parts <= 8) {
                // every slab of the row in ONE round trip (round 5: two of four slabs each were two dependent trips in a kernel the Dia step launches
                // 55 times, profiles/r05/dia_step_kernels_call15.txt); a slab beyond n_parts re-reads the last one and is never added; slab order kept
                float t[8][8];
#pragma unroll
                for (int q = 0; q < 8; q++)
#pragma unroll
                    for (int k = 0; k < 8; k++) t[q][k] = parts[min(q, n_parts - 1) * slab_stride + (int64_t) r * H + min(tid + k * 256, H - 1)];
#pragma unroll
                for (int q = 0; q < 8; q++)
#pragma unroll
                    for (int k = 0; k < 8; k++)
                        if (q < n_parts && tid + k * 256 < H) v[k] += t[q][k];
                p = n_parts;
            } else if (H <= 2048) {   // four slabs per round trip (8 values each): the adds still run in slab order
                for (; p + 4 <= n_parts; p += 4) {
                    float t[4][8];
#pragma unroll
                    for (int q = 0; q < 4; q++)
#pragma unroll
                        for (int k = 0; k < 8; k++) t[q][k] = parts[(p + q) * slab_stride + (int64_t) r * H + min(tid + k * 256, H - 1)];
#pragma unroll
                    for (int q = 0; q < 4; q++)
#pragma unroll
                        for (int k = 0; k < 8; k++)
                            if (tid + k * 256 < H) v[k] += t[q][k];
                }
            }
            if (H > 2048 && H <= 3072 && n_parts > 2) {
                // eight slabs of a 3072-wide row per round trip (round 6: the integer streaming GEMM leaves up to 16 K slices of the o / down projections;
                // one slab per trip was 8.2 us per launch, 57 launches per lock-step Orpheus step); a slab beyond n_parts re-reads the last one, never added
                for (; p < n_parts; p += 8) {
                    float t[8][12];
#pragma unroll
                    for (int q = 0; q < 8; q++)
#pragma unroll
                        for (int k = 0; k < 12; k++) t[q][k] = parts[min(p + q, n_parts - 1) * slab_stride + (int64_t) r * H + min(tid + k * 256, H - 1)];
#pragma unroll
                    for (int q = 0; q < 8; q++)
#pragma unroll
                        for (int k = 0; k < 12; k++)
                            if (p + q < n_parts && tid + k * 256 < H) v[k] += t[q][k];
                }
            }
            for (; p < n_parts; p++) {
                float t[16];
#pragma unroll
                for (int k = 0; k < 16; k++) t[k] = parts[p * slab_stride + (int64_t) r * H + min(tid + k * 256, H - 1)];
#pragma unroll
                for (int k = 0; k < 16; k++)
                    if (tid + k * 256 < H) v[k] += t[k];
            }
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int i = tid + k * 256;
                if (i < H) xr[i] = v[k];
            }
        }
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; k++)
            if (tid + k * 256 < H) s += v[k] * v[k];
        s = wave_sum(s);
        if ((tid & 63) == 0) red[tid >> 6] = s;
        __syncthreads();
        s = (red[0] + red[1]) + (red[2] + red[3]);
        const float scale = 1.0f / sqrtf(s / (float) H + eps);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int i = tid + k * 256;
            if (i < H) {   // H is a multiple of 32 whenever aq is given: a wave's 32-lane halves are whole blocks
                const float o = v[k] * scale * wv[k];
                y[(int64_t) r * H + i] = o;
                if (aq) q8_block_store(o, (int64_t) r * H + i, aq, ad);
            }
        }
        return;
    }
    float s = 0.0f;
    for (int i = tid; i < H; i += 256) {
        float v = xr[i];
        if (parts) {
            for (int p = 0; p < n_parts; p++) v += parts[p * slab_stride + (int64_t) r * H + i];
            xr[i] = v;
        }
        s += v * v;
    }
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    const float scale = 1.0f / sqrtf(s / (float) H + eps);
    for (int i = tid; i < H; i += 256) {
        const float o = xr[i] * scale * w[i];   // xr[i] was written by this thread
        y[(int64_t) r * H + i] = o;
        if (aq) q8_block_store(o, (int64_t) r * H + i, aq, ad);
    }
}

// qkv [R][(NH + 2 NKV) * HD] (q | k | v).  One wave per (row, q or k head); lane = pair i (and i + 64, ... for HD > 128).
// theta walks pos, pos*s, (pos*s)*s, ... in fp32 exactly like ggml_rope_cache_init (theta *= theta_scale), theta_scale =
// powf(base, -2/HD) from the host; angle = theta / freq_factor[i]; NEOX pairing (i, i + HD/2).  The k-head waves write the
// rotated key into the cache and copy their head's slice of v next to it.
// row_seq (optional): row r appends to the cache of sequence row_seq[r], seq_stride floats apart (Dia's two streams).
// n_parts > 1: qkv holds n_parts fp32 slabs (part_stride floats apart, K slices of the projection written by gemv_stream_kernel); they are
// summed in slab order on the way in and the rotated q lands in slab 0, where the attention kernels read it.
static __global__ __launch_bounds__(64) void llama_rope_kv_kernel(float *qkv, const uint32_t *pos, const float *ff, float theta_scale, int NH, int NKV, int HD,
                                                           float *kcache, float *vcache, const uint32_t *row_seq, int64_t seq_stride, int n_parts = 1,
                                                           int64_t part_stride = 0) {
    const int r = blockIdx.x, h = blockIdx.y;
    if (row_seq) { kcache += (int64_t) row_seq[r] * seq_stride; vcache += (int64_t) row_seq[r] * seq_stride; }
    const int half = HD >> 1;
    const int ld = (NH + 2 * NKV) * HD, kvH = NKV * HD;
    float *row = qkv + (int64_t) r * ld;
    float *v = row + (int64_t) h * HD;      // heads NH.. are the k heads (they follow q in the row)
    // Order (round 5): the slab loads of the wave's first pair (and of its v slice) do not depend on the position: they go out before pos[r] is waited
    // for and fly under the theta chain (up to 63 dependent multiplies, ggml's iterated theta) and sincos.
    const bool kw = h >= NH;
    const float *vsrc = row + (int64_t) (NH + NKV) * HD + (int64_t) (h - NH) * HD;
    auto slabs2 = [&](const float *b, int i, int j, float &a0, float &a1, float (&t0)[7], float (&t1)[7]) __attribute__((always_inline)) {
        a0 = b[i]; a1 = b[j];
#pragma unroll
        for (int q = 1; q < 8; q++) {   // straight-line (a slab beyond n_parts re-reads the last one and is never added): under `if (q < n_parts)` each
            const int64_t qo = (int64_t) min(q, n_parts - 1) * part_stride;   // pair of loads was a basic block behind its own wait
            t0[q - 1] = b[qo + i]; t1[q - 1] = b[qo + j];
        }
    };
    auto fold2 = [&](const float *b, int i, int j, float &a0, float &a1, const float (&t0)[7], const float (&t1)[7]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 1; q < 8; q++)
            if (q < n_parts) { a0 += t0[q - 1]; a1 += t1[q - 1]; }
        for (int q = 8; q < n_parts; q++) { a0 += b[q * part_stride + i]; a1 += b[q * part_stride + j]; }
    };
    const uint32_t p = pos[r];   // first in the queue: the theta chain needs nothing else
    const int i0 = min((int) threadIdx.x, half - 1);
    const float ff0 = ff ? ff[i0] : 1.0f;
    float fx0, fx1, ft0[7], ft1[7], vx0 = 0.0f, vx1 = 0.0f, vt0[7], vt1[7];
    slabs2(v, i0, i0 + half, fx0, fx1, ft0, ft1);
    const int vi0 = min((int) threadIdx.x, HD - 1), vi1 = min((int) threadIdx.x + 64, HD - 1);
    if (kw) slabs2(vsrc, vi0, vi1, vx0, vx1, vt0, vt1);
    __builtin_amdgcn_sched_barrier(0);
    auto angle = [&](int i, float fi, float &cs, float &sn) __attribute__((always_inline)) {
        float theta = (float) p;
        for (int j = 0; j < i; j++) theta *= theta_scale;
        const float ang = theta / fi;
        cs = cosf(ang); sn = sinf(ang);
    };
    auto rotate = [&](int i, float x0, float x1, float cs, float sn) __attribute__((always_inline)) {
        const float y0 = x0 * cs - x1 * sn, y1 = x0 * sn + x1 * cs;
        if (h < NH) { v[i] = y0; v[i + half] = y1; }
        else {
            float *kc = kcache + (int64_t) p * kvH + (h - NH) * HD;
            kc[i] = y0; kc[i + half] = y1;
        }
    };
    {   // the wave's first pair: its loads are in flight while the angle is computed
        float cs, sn;
        angle(i0, ff0, cs, sn);
        __builtin_amdgcn_sched_barrier(0);
        fold2(v, i0, i0 + half, fx0, fx1, ft0, ft1);
        if ((int) threadIdx.x < half) rotate(i0, fx0, fx1, cs, sn);
    }
    for (int i = threadIdx.x + 64; i < half; i += 64) {   // HD > 128: the later pairs
        float cs, sn, x0, x1, t0[7], t1[7];
        slabs2(v, i, i + half, x0, x1, t0, t1);
        angle(i, ff ? ff[i] : 1.0f, cs, sn);
        fold2(v, i, i + half, x0, x1, t0, t1);
        rotate(i, x0, x1, cs, sn);
    }
    if (kw) {
        float *vdst = vcache + (int64_t) p * kvH + (h - NH) * HD;
        if (HD <= 128) {
            fold2(vsrc, vi0, vi1, vx0, vx1, vt0, vt1);
            if ((int) threadIdx.x < HD) vdst[threadIdx.x] = vx0;
            if ((int) threadIdx.x + 64 < HD) vdst[threadIdx.x + 64] = vx1;
        } else {
            for (int i = threadIdx.x; i < HD; i += 64) {
                float t = vsrc[i], tp[7];
#pragma unroll
                for (int q = 1; q < 8; q++) tp[q - 1] = vsrc[(int64_t) min(q, n_parts - 1) * part_stride + i];
#pragma unroll
                for (int q = 1; q < 8; q++)
                    if (q < n_parts) t += tp[q - 1];
                for (int q = 8; q < n_parts; q++) t += vsrc[q * part_stride + i];
                vdst[i] = t;
            }
        }
    }
}

// One workgroup (4 waves) per (q head, row).  Scores: 16 lanes per key, each lane 8 of the 128 dims (two 16-byte loads, a
// key row is read as one contiguous 512 bytes), 16 keys per pass.  Softmax statistics over the block.  P.V: 8 groups of
// 32 lanes walk the keys, a lane owns 4 consecutive output dims (16-byte loads, 512 contiguous bytes per key and group).
extern __shared__ __align__(16) float attn_gqa_sm[];

// Optional work on the query as an attention workgroup loads it (saves the llama_rope_kv_kernel launch of a projection that has no
// k/v to append — Dia's cross-attention q, dia/model.cpp:609-633): fold n_parts K-slice slabs (gemv_stream_kernels.h) in slab order,
// then ggml_rope NEOX at rope_pos[r] with the same iterated theta as llama_rope_kv_kernel.
struct QPre {
    int n_parts = 1;
    int64_t part_stride = 0;
    const uint32_t *rope_pos = nullptr;
    float theta_scale = 0.0f;
};
// Two phases (round 5): the requests go out first — ahead of the K / V rows of the first keys, which are requested right behind them and fly while the
// query is folded and rotated (vmcnt retires in issue order: with the K rows requested behind attn_load_q's three barriers, as until round 4, a
// 128-key slice of Dia's cross-attention was position -> q slabs -> rope position -> K -> K: five dependent round trips before the softmax).
struct QRaw {
    float t[8];
    uint32_t rp;
};
template <int HD>
__device__ __forceinline__ void attn_q_request(QRaw &qr, const float *q, int r, const QPre &qp, int tid) {
    const int i = min(tid, HD - 1);   // straight-line: lanes beyond the head re-read its last element, a slab beyond n_parts the last slab; neither is used
#pragma unroll
    for (int p = 0; p < 8; p++) qr.t[p] = q[(int64_t) min(p, qp.n_parts - 1) * qp.part_stride + i];
    qr.rp = qp.rope_pos ? qp.rope_pos[r] : 0u;
}
template <int HD>
__device__ __forceinline__ void attn_q_finish(float *qs, const QRaw &qr, const float *q, const QPre &qp, int tid) {
    if (tid < HD) {
        float x = qr.t[0];
#pragma unroll
        for (int p = 1; p < 8; p++)
            if (p < qp.n_parts) x += qr.t[p];
        for (int p = 8; p < qp.n_parts; p++) x += q[p * qp.part_stride + tid];
        qs[tid] = x;
    }
    __syncthreads();
    if (qp.rope_pos) {
        float y0 = 0.0f, y1 = 0.0f;
        if (tid < HD / 2) {
            float theta = (float) qr.rp;
            for (int j = 0; j < tid; j++) theta *= qp.theta_scale;
            const float cs = cosf(theta), sn = sinf(theta);
            const float x0 = qs[tid], x1 = qs[tid + HD / 2];
            y0 = x0 * cs - x1 * sn; y1 = x0 * sn + x1 * cs;
        }
        __syncthreads();
        if (tid < HD / 2) { qs[tid] = y0; qs[tid + HD / 2] = y1; }
        __syncthreads();
    }
}
// The two key passes of the decode attention with their loads in batches of 64 keys: every lane requests the rows of its 4 (scores) / 8 (P.V)
// keys of a batch back to back — clamped indices, nothing under a predicate — and only then computes.  One dependent round trip per 64 keys
// instead of one per 16 (scores) / 8 (P.V) keys: a 40-key slice of a split took 3 + 5 round trips (15-18 us per launch at the Orpheus-3B shapes,
// profiles/r04/kernel_stats_orpheus_baseline.csv).  Key -> lane group and the order of every sum are the ones of round 3 (key j: group j % 16 of
// the score pass, group j % 8 of the P.V pass, keys ascending inside a group; row16_sum reproduces the xor butterfly), so results are unchanged.
template <int HD>
__device__ __forceinline__ void attn_v_batch(float4 (&v)[8], const float *vbase, int kvH, int T, int j0, int tid) {
    const int grp = tid >> 5, e4 = tid & 31;
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const int j = max(0, min(j0 + grp + 8 * u, T - 1));
        v[u] = *(const float4 *) (vbase + (int64_t) j * kvH + e4 * 4);
    }
}
struct KBatch {   // the rows of a lane's 4 keys of a 64-key batch (its 2 x 16 bytes of each)
    float4 a[4], b[4];
};
template <int HD>
__device__ __forceinline__ void attn_k_request(KBatch &kb, const float *kbase, int kvH, int T, int j0, int tid) {
    const int g = tid >> 4, sub = tid & 15;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int j = max(0, min(j0 + g + 16 * u, T - 1));
        const float4 *kr = (const float4 *) (kbase + (int64_t) j * kvH);
        kb.a[u] = kr[sub]; kb.b[u] = kr[16 + sub];
    }
}
// kpre: the first NPRE batches, requested by the caller before the query was ready
template <int HD, int NPRE>
__device__ __forceinline__ void attn_scores_batched(KBatch (&kpre)[NPRE], const float *kbase, int kvH, int T, const float *qs, float scale, float *ps, int tid) {
    const int g = tid >> 4, sub = tid & 15;
    const float4 q0 = *(const float4 *) (qs + sub * 4), q1 = *(const float4 *) (qs + 64 + sub * 4);
    auto scores = [&](const KBatch &kb, int j0) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + g + 16 * u;
            float d = (kb.a[u].x * q0.x + kb.a[u].y * q0.y + kb.a[u].z * q0.z + kb.a[u].w * q0.w) + (kb.b[u].x * q1.x + kb.b[u].y * q1.y + kb.b[u].z * q1.z + kb.b[u].w * q1.w);
            d = row16_sum(d);
            if (sub == 0 && j < T) ps[j] = d * scale;
        }
    };
#pragma unroll
    for (int b = 0; b < NPRE; b++)
        if (b * 64 < T) scores(kpre[b], b * 64);
    for (int j0 = NPRE * 64; j0 < T; j0 += 64) {
        KBatch kb;
        attn_k_request<HD>(kb, kbase, kvH, T, j0, tid);
        scores(kb, j0);
    }
}
template <int HD>
__device__ __forceinline__ float4 attn_pv_batched(float4 (&v)[8], const float *vbase, int kvH, int T, const float *ps, int tid) {   // v: the first batch, already requested
    const int grp = tid >> 5;
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int j0 = 0; j0 < T; j0 += 64) {
        if (j0) attn_v_batch<HD>(v, vbase, kvH, T, j0, tid);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int j = j0 + grp + 8 * u;
            if (j < T) {
                const float p = ps[j];
                acc.x += p * v[u].x; acc.y += p * v[u].y; acc.z += p * v[u].z; acc.w += p * v[u].w;
            }
        }
    }
    return acc;
}

// Keys [kbeg[r], kend[r]) of the row's sequence (defaults: 0 and pos[r] + 1 = causal over the cache); row_seq / seq_stride as above.
template <int HD>
__global__ __launch_bounds__(256) void attn_gqa_kernel(const float *qkv, int ld, const uint32_t *pos, const float *kcache, const float *vcache, int NH, int NKV,
                                                       float scale, float *out, const uint32_t *kbeg, const uint32_t *kend, const uint32_t *row_seq,
                                                       int64_t seq_stride, QPre qp = QPre{}, int8_t *aq = nullptr, float *ad = nullptr) {
    static_assert(HD == 128, "lane mapping below is written for head_dim 128 (orpheus/model.h:28)");
    __shared__ float red[8];
    __shared__ float4 accs[8][HD / 4];
    float *qs = attn_gqa_sm, *ps = attn_gqa_sm + HD;   // [HD] q, then [T] scores -> probabilities
    const int h = blockIdx.x, r = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k0 = kbeg ? (int) kbeg[r] : 0;
    const int T = (kend ? (int) kend[r] : (int) pos[r] + 1) - k0;
    const int kvH = NKV * HD, kh = h / (NH / NKV);
    if (row_seq) { kcache += (int64_t) row_seq[r] * seq_stride; vcache += (int64_t) row_seq[r] * seq_stride; }
    kcache += (int64_t) k0 * kvH;
    vcache += (int64_t) k0 * kvH;
    // requests in the order of their use: the query (slabs, rope position), the K rows of the first 64 keys (two batches would be 140 registers: three workgroups per CU instead of four), the V rows of the first 64 (they do
    // not depend on the scores); the query is folded and rotated under the K / V rows
    QRaw qr;
    attn_q_request<HD>(qr, qkv + (int64_t) r * ld + h * HD, r, qp, tid);
    KBatch kpre[1];
    attn_k_request<HD>(kpre[0], kcache + kh * HD, kvH, T, 0, tid);
    float4 vfirst[8];
    attn_v_batch<HD>(vfirst, vcache + kh * HD, kvH, T, 0, tid);
    __builtin_amdgcn_sched_barrier(0);
    attn_q_finish<HD>(qs, qr, qkv + (int64_t) r * ld + h * HD, qp, tid);
    attn_scores_batched<HD, 1>(kpre, kcache + kh * HD, kvH, T, qs, scale, ps, tid);
    __syncthreads();
    float mx = -INFINITY;
    for (int j = tid; j < T; j += 256) mx = fmaxf(mx, ps[j]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.0f;
    for (int j = tid; j < T; j += 256) {
        const float p = expf(ps[j] - mx);
        ps[j] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();   // probabilities and the four partial sums are in LDS
    const float inv = 1.0f / ((red[4] + red[5]) + (red[6] + red[7]));
    accs[tid >> 5][tid & 31] = attn_pv_batched<HD>(vfirst, vcache + kh * HD, kvH, T, ps, tid);
    __syncthreads();
    if (tid < HD) {
        const float *a = (const float *) &accs[0][0];
        float o = 0.0f;
#pragma unroll
        for (int g = 0; g < 8; g++) o += a[g * HD + tid];
        const float res = o * inv;
        out[(int64_t) r * NH * HD + h * HD + tid] = res;
        // the o projection's activation blocks (a head is four Q8_0 blocks; waves 0 and 1 are whole here): saves its quant_rows_q8_kernel launch
        if (aq) q8_block_store(res, (int64_t) r * NH * HD + h * HD + tid, aq, ad);
    }
}

// Decode-time form of attn_gqa_kernel for few (head, row) pairs and many keys: one workgroup can pull only ~1/256 of the HBM
// bandwidth (a CU's load path), so 24 workgroups reading 0.5 MB each take 20 us whatever the chip could do.  The keys of a row are
// split over gridDim.z workgroups; each leaves (max, sum, unnormalised out[128]) and attn_gqa_combine_kernel merges them — the same
// softmax, associated differently.  An empty split (short sequences under a graph captured for long ones) leaves max = -inf.
// ATTN_PART (parler_kernels.h) floats per partial: max, sum, out[128]
template <int HD>
__global__ __launch_bounds__(256) void attn_gqa_split_kernel(const float *qkv, int ld, const uint32_t *pos, const float *kcache, const float *vcache, int NH, int NKV,
                                                             float scale, float *part, const uint32_t *kbeg, const uint32_t *kend, const uint32_t *row_seq,
                                                             int64_t seq_stride, QPre qp = QPre{}) {
    static_assert(HD == 128, "lane mapping below is written for head_dim 128");
    __shared__ float red[8];
    __shared__ float4 accs[8][HD / 4];
    float *qs = attn_gqa_sm, *ps = attn_gqa_sm + HD;
    // blockIdx.x = key slice: workgroups go to the XCDs round-robin by their linear index, so with 8 slices an XCD owns ONE slice of every head and row —
    // whole contiguous K / V rows per XCD.  With the head in x (until round 5) an XCD saw heads x and x + 8 only: two 512-byte pieces of every 8 KB row.
    // (Measured equal on Dia's cross-attention, 134 MB per launch: 28.9 us = 4.6 TB/s either way, as is the request order below: that launch is
    // neither latency- nor channel-bound, profiles/r05/dia_step_kernels_call18.txt.)
    const int z = blockIdx.x, r = blockIdx.y, h = blockIdx.z, nz = gridDim.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kb = kbeg ? (int) kbeg[r] : 0;
    const int Tall = (kend ? (int) kend[r] : (int) pos[r] + 1) - kb;
    const int chunk = (Tall + nz - 1) / nz;
    const int k0 = kb + z * chunk;
    const int T = max(0, min(chunk, Tall - z * chunk));
    float *pz = part + (((int64_t) r * NH + h) * nz + z) * ATTN_PART;
    if (T <= 0) {
        if (tid == 0) { pz[0] = -INFINITY; pz[1] = 0.0f; }
        return;
    }
    const int kvH = NKV * HD, kh = h / (NH / NKV);
    if (row_seq) { kcache += (int64_t) row_seq[r] * seq_stride; vcache += (int64_t) row_seq[r] * seq_stride; }
    kcache += (int64_t) k0 * kvH;
    vcache += (int64_t) k0 * kvH;
    // requests in the order of their use: the query (slabs, rope position), the K rows of the first 64 keys (two batches would be 140 registers: three workgroups per CU instead of four), the V rows of the first 64 (they do
    // not depend on the scores); the query is folded and rotated under the K / V rows
    QRaw qr;
    attn_q_request<HD>(qr, qkv + (int64_t) r * ld + h * HD, r, qp, tid);
    KBatch kpre[1];
    attn_k_request<HD>(kpre[0], kcache + kh * HD, kvH, T, 0, tid);
    float4 vfirst[8];
    attn_v_batch<HD>(vfirst, vcache + kh * HD, kvH, T, 0, tid);
    __builtin_amdgcn_sched_barrier(0);
    attn_q_finish<HD>(qs, qr, qkv + (int64_t) r * ld + h * HD, qp, tid);
    attn_scores_batched<HD, 1>(kpre, kcache + kh * HD, kvH, T, qs, scale, ps, tid);
    __syncthreads();
    float mx = -INFINITY;
    for (int j = tid; j < T; j += 256) mx = fmaxf(mx, ps[j]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.0f;
    for (int j = tid; j < T; j += 256) {
        const float p = expf(ps[j] - mx);
        ps[j] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    accs[tid >> 5][tid & 31] = attn_pv_batched<HD>(vfirst, vcache + kh * HD, kvH, T, ps, tid);
    __syncthreads();
    if (tid < HD) {
        const float *a = (const float *) &accs[0][0];
        float o = 0.0f;
#pragma unroll
        for (int g = 0; g < 8; g++) o += a[g * HD + tid];
        pz[2 + tid] = o;
    }
    if (tid == 0) { pz[0] = mx; pz[1] = (red[4] + red[5]) + (red[6] + red[7]); }
}

// attn_gqa_split_kernel's job for the captured one-row step of a Llama decoder, restructured so that the launch is ONE memory round trip with
// one workgroup barrier (round 5; the split kernel above is a chain of seven: position -> q -> K batch -> scores -> max -> exp / sum -> V -> merge,
// 8.5 us per launch for 2.4 MB of cache at the Orpheus-3B shapes, profiles/r05/kernel_stats_orpheus_call1.csv):
//   * the keys a (slice, wave, 16-lane group) reads do not depend on the position: key j belongs to slice (j / 16) % nz, inside the slice to group
//     j % 16 (= wave * 4 + lane / 16) and pass j / (16 nz).  Every lane therefore requests the K and V rows of its first U passes the moment the
//     kernel starts — together with q and the position, not behind them; keys at or beyond the position are masked afterwards (they are rows of the
//     cache allocation: readable, never used);
//   * a 16-lane group owns its keys from first to last with a running max / sum / output (soft_max_ext + mul_mat as one pass, as attn_rows_kernel
//     does for Parler): no score buffer, no barrier inside the key loop;
//   * the 4 groups of a wave merge by permlane swaps, the 4 waves through LDS with the only barrier of the kernel.
// A lane holds dims 4 sub .. + 3 and 64 + 4 sub .. + 3 of the 128-wide head (sub = lane % 16): a K or V row is read as 512 contiguous bytes.
// The partials have attn_gqa_split_kernel's meaning (max, sum, unnormalised out[128] of the slice's keys; an empty slice leaves -inf, 0), so
// attn_gqa_combine_kernel follows unchanged.  Another association of the same softmax than the kernel above: results agree to rounding.
// EXT (Dia's cross-attention over the 1024 text positions, 8 rows x 16 heads x 8 slices of 128 keys = U = 8 passes): the keys end at kend[r], row r reads
// the cache of sequence row_seq[r], and the query arrives as qp.n_parts K-slice slabs to be folded and rotated (QPre) — folded by the first 128
// threads through LDS (one more barrier, under the rows in flight), rotated in the lanes (a lane holds both halves of its NEOX pairs).
template <int HD, int U, bool EXT = false>
__global__ __launch_bounds__(256) void attn_gqa_wave_kernel(const float *qkv, int ld, const uint32_t *pos, const float *kcache, const float *vcache, int NH, int NKV,
                                                            float scale, float *part, int n_ctx, const uint32_t *kend = nullptr, const uint32_t *row_seq = nullptr,
                                                            int64_t seq_stride = 0, QPre qp = QPre{}) {
    static_assert(HD == 128, "lane mapping below is written for head_dim 128");
    __shared__ float s_m[4], s_l[4];
    __shared__ __attribute__((aligned(16))) float s_acc[4][HD];
    __shared__ __attribute__((aligned(16))) float s_q[EXT ? HD : 4];
    const int h = blockIdx.x, r = blockIdx.y, z = blockIdx.z, nz = gridDim.z, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, sub = lane & 15, gi = wave * 4 + g;
    const int kvH = NKV * HD, kh = h / (NH / NKV);
    if (EXT && row_seq) { kcache += (int64_t) row_seq[r] * seq_stride; vcache += (int64_t) row_seq[r] * seq_stride; }
    // wave-uniform bases + one 32-bit byte offset per row (a sequence's cache is far below 4 GB): K and V of a key share the offset register
    const char *kp = (const char *) (kcache + kh * HD), *vp = (const char *) (vcache + kh * HD);
    auto key_of = [&](int p) { return 16 * (nz * p + z) + gi; };
    float4 ka[U], kb[U], va[U], vb[U];
    auto request_one = [&](int u, int pass) __attribute__((always_inline)) {
        const uint32_t off = ((uint32_t) min(key_of(pass), n_ctx - 1) * (uint32_t) kvH + (uint32_t) sub * 4u) * 4u;
        if (EXT) {
            // non-temporal: 2.4 GB of cross K / V per Dia step, nothing of it is met again before it has left every cache — and with plain loads the
            // 134 MB of a launch pushed the step's small hot set (rows, slabs, partials) out of L2 / the memory-side cache: the launch itself gains
            // 0.8 us, the step 5 % (1.90 -> 1.80 ms, same box, profiles/r05/dia_step_kernels_call21.txt)
            auto ntl = [](const char *a) __attribute__((always_inline)) { return __builtin_bit_cast(float4, __builtin_nontemporal_load((const float4v *) a)); };
            ka[u] = ntl(kp + off); kb[u] = ntl(kp + off + 256);
            va[u] = ntl(vp + off); vb[u] = ntl(vp + off + 256);
        } else {
            ka[u] = *(const float4 *) (kp + off); kb[u] = *(const float4 *) (kp + off + 256);
            va[u] = *(const float4 *) (vp + off); vb[u] = *(const float4 *) (vp + off + 256);
        }
    };
    auto request = [&](int p0) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < U; u++) request_one(u, p0 + u);
    };
    // the small inputs first (vmcnt retires in issue order: the position and q are usable while the rows below are still in flight)
    const float *qr = qkv + (int64_t) r * ld + h * HD + sub * 4;
    float4 q0, q1;
    float tq[4];       // EXT: threads 0-127 hold slabs 0-3 of query element tid, threads 128-255 slabs 4-7 of element tid - 128 (four registers, not eight:
    uint32_t rp = 0;   // the kernel sits at the 128-register line of four workgroups per CU)
    if (EXT) {
        const float *qe = qkv + (int64_t) r * ld + h * HD + (tid & (HD - 1));
#pragma unroll
        for (int p = 0; p < 4; p++) tq[p] = qe[(int64_t) min((tid >> 7) * 4 + p, qp.n_parts - 1) * qp.part_stride];
        if (qp.rope_pos) rp = qp.rope_pos[r];
    } else { q0 = *(const float4 *) qr; q1 = *(const float4 *) (qr + 64); }
    const int T = EXT && kend ? (int) kend[r] : (int) pos[r] + 1;
    __builtin_amdgcn_sched_barrier(0);
    request(0);
    __builtin_amdgcn_sched_barrier(0);
    if (EXT) {
        // attn_q_finish's arithmetic: slabs in slab order (the upper half of the workgroup continues the lower half's sum), then ggml_rope NEOX with
        // the iterated theta (pair i, i + 64: both in this lane)
        if (tid < HD) {
            float x = tq[0];
#pragma unroll
            for (int p = 1; p < 4; p++)
                if (p < qp.n_parts) x += tq[p];
            s_q[tid] = x;
        }
        __syncthreads();
        if (tid >= HD) {
            float x = s_q[tid - HD];
#pragma unroll
            for (int p = 0; p < 4; p++)
                if (4 + p < qp.n_parts) x += tq[p];   // (the host sends at most 8 slabs here)
            s_q[tid - HD] = x;
        }
        __syncthreads();
        q0 = *(const float4 *) (s_q + sub * 4); q1 = *(const float4 *) (s_q + 64 + sub * 4);
        if (qp.rope_pos) {
            float theta = (float) rp;
            for (int j = 0; j < sub * 4; j++) theta *= qp.theta_scale;
            float *a = (float *) &q0, *b = (float *) &q1;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float cs = cosf(theta), sn = sinf(theta);
                const float x0 = a[e], x1 = b[e];
                a[e] = x0 * cs - x1 * sn; b[e] = x0 * sn + x1 * cs;
                theta *= qp.theta_scale;
            }
        }
    }
    float m = -INFINITY, l = 0.0f;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    auto consume = [&](int u, int pass) __attribute__((always_inline)) {
        if (key_of(pass) < T) {
            float d = (ka[u].x * q0.x + ka[u].y * q0.y + ka[u].z * q0.z + ka[u].w * q0.w) + (kb[u].x * q1.x + kb[u].y * q1.y + kb[u].z * q1.z + kb[u].w * q1.w);
            d = row16_sum(d) * scale;
            const float mn = fmaxf(m, d);
            const float f = expf(m - mn);   // 0 on the group's first key (m = -inf)
            const float pr = expf(d - mn);
            l = l * f + pr;
            a0.x = a0.x * f + pr * va[u].x; a0.y = a0.y * f + pr * va[u].y; a0.z = a0.z * f + pr * va[u].z; a0.w = a0.w * f + pr * va[u].w;
            a1.x = a1.x * f + pr * vb[u].x; a1.y = a1.y * f + pr * vb[u].y; a1.z = a1.z * f + pr * vb[u].z; a1.w = a1.w * f + pr * vb[u].w;
            m = mn;
        }
    };
    if (EXT) {
        // eight passes (the host checks: at most 16 nz 8 keys), U register slots: a slot is requested again for pass p + U the moment pass p has been
        // consumed, so every workgroup keeps U passes in flight from its first instruction to its last key.  Measured on Dia's cross-attention (134 MB
        // per launch; a pure read of the same bytes: 23 us, profiles/r05/stride_read_bench_call19.txt): all eight passes requested at once (198
        // registers, two workgroups per CU) 30.3 us, four slots (130 registers, three per CU) 27.7, three slots (112, four per CU) 26.9, the split
        // kernel's three phases 28.1-28.9 (profiles/r05/dia_cross_attention_forms.txt, dia_step_kernels_call21.txt).
#pragma unroll
        for (int p = 0; p < 8; p++) {
            consume(p % U, p);
            __builtin_amdgcn_sched_barrier(0);   // the slot's registers are free only now: a request hoisted above costs the 128-register line
            if (p + U < 8) request_one(p % U, p + U);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        // (rolling slots here too — a history beyond the first 512 keys is one more dependent round trip per 512 keys — measured equal to this loop and
        //  to the split kernel at positions 1120..1568 of an Orpheus-3B step: 1.33 / 1.33 / 1.32 ms, profiles/r05/orpheus_bench_call22_roll_rejected.txt)
        for (int p0 = 0; 16 * nz * p0 < T; p0 += U) {
            if (p0) request(p0);
#pragma unroll
            for (int u = 0; u < U; u++) consume(u, p0 + u);
        }
    }
    // the wave's four groups (lanes sub, 16 + sub, 32 + sub, 48 + sub hold the same dims): common max, rescale, add
    auto x16 = [](float v) { const unsigned w = __builtin_bit_cast(unsigned, v); const auto b = __builtin_amdgcn_permlane16_swap(w, w, false, false);
                             return make_float2(__builtin_bit_cast(float, (unsigned) b[0]), __builtin_bit_cast(float, (unsigned) b[1])); };
    auto x32 = [](float v) { const unsigned w = __builtin_bit_cast(unsigned, v); const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
                             return make_float2(__builtin_bit_cast(float, (unsigned) b[0]), __builtin_bit_cast(float, (unsigned) b[1])); };
    float mw;
    { float2 t = x16(m); mw = fmaxf(t.x, t.y); t = x32(mw); mw = fmaxf(t.x, t.y); }
    const float fg = m == -INFINITY ? 0.0f : expf(m - mw);
    float vals[9] = {l * fg, a0.x * fg, a0.y * fg, a0.z * fg, a0.w * fg, a1.x * fg, a1.y * fg, a1.z * fg, a1.w * fg};
#pragma unroll
    for (int i = 0; i < 9; i++) {
        float2 t = x16(vals[i]); float v = t.x + t.y;
        t = x32(v); vals[i] = t.x + t.y;
    }
    if (g == 0) {
        *(float4 *) (&s_acc[wave][sub * 4]) = make_float4(vals[1], vals[2], vals[3], vals[4]);
        *(float4 *) (&s_acc[wave][64 + sub * 4]) = make_float4(vals[5], vals[6], vals[7], vals[8]);
        if (sub == 0) { s_m[wave] = mw; s_l[wave] = vals[0]; }
    }
    __syncthreads();
    if (tid < HD) {
        const float M = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
        float o = 0.0f, L = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const float f = s_m[w] == -INFINITY ? 0.0f : expf(s_m[w] - M);
            o += f * s_acc[w][tid];
            L += f * s_l[w];
        }
        float *pz = part + (((int64_t) r * NH + h) * nz + z) * ATTN_PART;
        pz[2 + tid] = o;
        if (tid == 0) { pz[0] = M; pz[1] = L; }
    }
}

// one 128-thread workgroup per (head, row): out = sum_z e^(m_z - m) o_z / sum_z e^(m_z - m) l_z, splits in order.
// Every slice is requested straight-line with a clamped index (a slice beyond nz re-reads the last one and is never used): as `for (z < nz)` loops
// with the -inf test on a loaded value this kernel was nz dependent L2 round trips — 4.9 us per launch for 24 KB of partials in the Orpheus step
// (profiles/r04/kernel_stats_orpheus_call7.csv).  Same sums in the same order.
static __global__ __launch_bounds__(128) void attn_gqa_combine_kernel(const float *part, int nz, int NH, float *out, int8_t *aq, float *ad) {
    const int h = blockIdx.x, r = blockIdx.y, t = threadIdx.x;
    const float *p = part + ((int64_t) r * NH + h) * nz * ATTN_PART;
    float mm[16], ll[16], oo[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const float *pi = p + min(i, nz - 1) * ATTN_PART;
        mm[i] = pi[0]; ll[i] = pi[1]; oo[i] = pi[2 + t];
    }
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; i++) if (i < nz) m = fmaxf(m, mm[i]);
    float o = 0.0f, l = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        if (i < nz && mm[i] != -INFINITY) {   // -inf marks an empty slice
            const float f = expf(mm[i] - m);
            o += f * oo[i];
            l += f * ll[i];
        }
    }
    const float res = o / l;
    out[(int64_t) r * NH * 128 + h * 128 + t] = res;
    if (aq) q8_block_store(res, (int64_t) r * NH * 128 + h * 128 + t, aq, ad);   // the o projection's activation blocks
}

// arg-max over a large vocabulary (156 940 logits), sampler::max semantics (src/sampler.cpp:185-204: the first maximum
// wins).  Stage 1: ARGMAX_PARTS workgroups over contiguous chunks; stage 2: one wave folds the partial results.
#define ARGMAX_PARTS 128
__device__ __forceinline__ void argmax_merge(float &best, uint32_t &besti, float ov, uint32_t oi) {
    if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
}
static __global__ __launch_bounds__(256) void argmax_parts_kernel(const float *logits, int V, float *pv, uint32_t *pi) {
    __shared__ float bv[4];
    __shared__ uint32_t bi[4];
    const int chunk = (V + ARGMAX_PARTS - 1) / ARGMAX_PARTS;
    const int i0 = blockIdx.x * chunk, i1 = min(V, i0 + chunk);
    float best = -INFINITY;
    uint32_t besti = 0xffffffffu;
    for (int i = i0 + threadIdx.x; i < i1; i += 256) {
        const float v = logits[i];
        if (v > best) { best = v; besti = (uint32_t) i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) argmax_merge(best, besti, __shfl_xor(best, o), __shfl_xor(besti, o));
    if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = besti; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) argmax_merge(best, besti, bv[w], bi[w]);
        pv[blockIdx.x] = best;
        pi[blockIdx.x] = besti;
    }
}
// hist / next_id / next_pos (optional): the device-resident greedy loop feeds the token straight back as the next input
static __global__ __launch_bounds__(64) void argmax_fold_kernel(const float *pv, const uint32_t *pi, uint32_t *token, uint32_t *hist, uint32_t *next_id,
                                                         uint32_t *next_pos) {
    float best = -INFINITY;
    uint32_t besti = 0xffffffffu;
    for (int i = threadIdx.x; i < ARGMAX_PARTS; i += 64) argmax_merge(best, besti, pv[i], pi[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) argmax_merge(best, besti, __shfl_xor(best, o), __shfl_xor(besti, o));
    if (threadIdx.x == 0) {
        const uint32_t t = besti == 0xffffffffu ? 0u : besti;   // all -inf / NaN: index 0 like sampler::max
        token[0] = t;
        if (hist) hist[0] = t;
        if (next_id) { next_id[0] = t; next_pos[0] += 1; }
    }
}

// the two stages for R rows of logits (lock-step utterances): blockIdx.y = row, logits rows ld floats apart; partials [R][ARGMAX_PARTS]
static __global__ __launch_bounds__(256) void argmax_rows_parts_kernel(const float *logits, int V, int ld, float *pv, uint32_t *pi) {
    __shared__ float bv[4];
    __shared__ uint32_t bi[4];
    const float *lg = logits + (int64_t) blockIdx.y * ld;
    const int chunk = (V + ARGMAX_PARTS - 1) / ARGMAX_PARTS;
    const int i0 = blockIdx.x * chunk, i1 = min(V, i0 + chunk);
    float best = -INFINITY;
    uint32_t besti = 0xffffffffu;
    for (int i = i0 + threadIdx.x; i < i1; i += 256) {
        const float v = lg[i];
        if (v > best) { best = v; besti = (uint32_t) i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) argmax_merge(best, besti, __shfl_xor(best, o), __shfl_xor(besti, o));
    if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = besti; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) argmax_merge(best, besti, bv[w], bi[w]);
        pv[blockIdx.y * ARGMAX_PARTS + blockIdx.x] = best;
        pi[blockIdx.y * ARGMAX_PARTS + blockIdx.x] = besti;
    }
}
static __global__ __launch_bounds__(64) void argmax_rows_fold_kernel(const float *pv, const uint32_t *pi, uint32_t *token) {
    const int r = blockIdx.x;
    float best = -INFINITY;
    uint32_t besti = 0xffffffffu;
    for (int i = threadIdx.x; i < ARGMAX_PARTS; i += 64) argmax_merge(best, besti, pv[r * ARGMAX_PARTS + i], pi[r * ARGMAX_PARTS + i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) argmax_merge(best, besti, __shfl_xor(best, o), __shfl_xor(besti, o));
    if (threadIdx.x == 0) token[r] = besti == 0xffffffffu ? 0u : besti;
}

// the same fold for a captured step (one graph replayed for every position): the history slot comes from a device counter
static __global__ __launch_bounds__(64) void argmax_fold_graph_kernel(const float *pv, const uint32_t *pi, uint32_t *token, uint32_t *hist, uint32_t *hist_idx, uint32_t *next_id,
                                                               uint32_t *next_pos) {
    float best = -INFINITY;
    uint32_t besti = 0xffffffffu;
    for (int i = threadIdx.x; i < ARGMAX_PARTS; i += 64) argmax_merge(best, besti, pv[i], pi[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) argmax_merge(best, besti, __shfl_xor(best, o), __shfl_xor(besti, o));
    if (threadIdx.x == 0) {
        const uint32_t t = besti == 0xffffffffu ? 0u : besti;
        token[0] = t;
        hist[hist_idx[0]] = t;
        hist_idx[0] += 1;
        next_id[0] = t;
        next_pos[0] += 1;
    }
}

// ------------------------------------------------------------------------------------------------
// sampler::sample over the 156 940-logit vocabulary (orpheus/model.cpp:389-398, sampler.cpp:3-69 with topk :152-183 and softmax :82-116)
// for top_k in 1..TOPK_MAXK (top_p < 1: + softmax_total_kernel, below).  The reference sorts all 156 940 indices by
// (penalised) value on the host at every step; only the first top_k of that order are ever read, so two stages find them:
//   topk_parts_kernel   TOPK_PARTS workgroups, each sorts its slice of the vocabulary (bitonic, 64-bit keys = value descending, index
//                       ascending — the total order of sample_kernel / smp_key) and keeps its first k keys;
//   topk_sample_kernel  one workgroup sorts the TOPK_PARTS * k survivors, takes the first k (the reference's `picks`), and runs the
//                       rest of sampler::sample exactly as sample_kernel does: softmax over the picks in pick order (sequential fp32
//                       sum, exp(v / T - top) with top = the penalised maximum / T), inverse-CDF scan against the host-drawn uniform,
//                       repetition state update (sampler.cpp:57-63); it then feeds the token back like argmax_fold_*_kernel.
// The token sampled last takes part with v / pow(penalty, count) evaluated in double (sampler.cpp:89-90,172-175); pen_table[c] =
// pow(penalty, c) comes from the host libm.
// ------------------------------------------------------------------------------------------------
#define TOPK_PARTS 64
#define TOPK_SLICE 4096   // keys one part sorts: vocabularies up to TOPK_PARTS * TOPK_SLICE = 262 144
#define TOPK_MAXK 64

__device__ __forceinline__ void bitonic_sort_keys(unsigned long long *keys, int P) {
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < (P >> 1); t += blockDim.x) {
                const int lo = ((t / stride) * (stride << 1)) + (t % stride), hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
            }
            __syncthreads();
        }
    }
}
__device__ __forceinline__ float smp_key_value(unsigned long long key) {   // inverse of smp_key's value field
    const unsigned u = ~(unsigned) (key >> 32);
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

static __global__ __launch_bounds__(512) void topk_parts_kernel(const float *logits, int V, int k, const double *pen_table, int pen_len, const int32_t *last_id,
                                                         const uint32_t *rep_count, unsigned long long *cand) {
    __shared__ unsigned long long keys[TOPK_SLICE];
    const int chunk = (V + TOPK_PARTS - 1) / TOPK_PARTS;
    const int i0 = (int) blockIdx.x * chunk;
    const int last = pen_table ? last_id[0] : -1;
    for (int j = threadIdx.x; j < TOPK_SLICE; j += blockDim.x) {
        const int i = i0 + j;
        unsigned long long key = ~0ull;
        if (j < chunk && i < V) {
            float v = logits[i];
            if (i == last) {
                const uint32_t cnt = rep_count[0];
                v = (float) ((double) v / pen_table[cnt < (uint32_t) pen_len ? cnt : (uint32_t) pen_len - 1]);
            }
            key = smp_key(v, i);
        }
        keys[j] = key;
    }
    __syncthreads();
    bitonic_sort_keys(keys, TOPK_SLICE);
    for (int j = threadIdx.x; j < k; j += blockDim.x) cand[(int) blockIdx.x * TOPK_MAXK + j] = keys[j];
}

// top_p < 1 (sampler.cpp:22-27): the reference runs softmax over the WHOLE vocabulary before it looks for the top k — every probability is
// exp(v / T - top) / cumsum with cumsum accumulated over the 156 940 entries in index order, in fp32.  The order of an fp32 sum is part of its
// value, so the total is accumulated here by ONE thread in that order (the exponentials are computed by the whole workgroup, a chunk at a time,
// into LDS): ~0.4 ms per token — the price of drawing exactly the reference's nucleus on the device instead of shipping 628 KB of logits to
// the host and sorting them there (~10 ms per token).  `top` is the penalised maximum / T, read off the part winners of topk_parts_kernel.
#define SOFTMAX_CHUNK 8192
static __global__ __launch_bounds__(1024) void softmax_total_kernel(const float *logits, int V, const unsigned long long *cand, float temperature, const double *pen_table,
                                                             int pen_len, const int32_t *last_id, const uint32_t *rep_count, float *total_out) {
    __shared__ __attribute__((aligned(16))) float ex[SOFTMAX_CHUNK];
    __shared__ float s_top;
    const bool temp = temperature != 1.0f;
    if (threadIdx.x == 0) {
        unsigned long long best = ~0ull;
        for (int p = 0; p < TOPK_PARTS; p++) best = cand[p * TOPK_MAXK] < best ? cand[p * TOPK_MAXK] : best;
        float top = smp_key_value(best);
        if (temp) top /= temperature;
        s_top = top;
    }
    __syncthreads();
    const float top = s_top;
    const int last = pen_table ? last_id[0] : -1;
    float total = 0.0f;
    for (int c0 = 0; c0 < V; c0 += SOFTMAX_CHUNK) {
        const int n = min(SOFTMAX_CHUNK, V - c0);
        for (int j = threadIdx.x; j < n; j += blockDim.x) {
            const int i = c0 + j;
            float v = logits[i];
            if (i == last) {
                const uint32_t cnt = rep_count[0];
                v = (float) ((double) v / pen_table[cnt < (uint32_t) pen_len ? cnt : (uint32_t) pen_len - 1]);
            }
            if (temp) v /= temperature;
            ex[j] = expf(v - top);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int j = 0;
            for (; j + 64 <= n; j += 64) {   // 16 LDS reads in flight, then 64 adds in index order (one read per four adds was 25 clocks per add)
                float4 e[16];
#pragma unroll
                for (int q = 0; q < 16; q++) e[q] = *(const float4 *) (ex + j + 4 * q);
#pragma unroll
                for (int q = 0; q < 16; q++) { total += e[q].x; total += e[q].y; total += e[q].z; total += e[q].w; }
            }
            for (; j < n; j++) total += ex[j];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) total_out[0] = total;
}

// uniforms[call[0]] is this call's draw; call[0] advances.  hist_idx != NULL: captured step, the history slot is hist[hist_idx[0]++];
// otherwise hist (may be NULL) is the slot itself.  next_id / next_pos (may be NULL): the token goes straight back as the next input.
// total (NULL: top_p >= 1): the full-vocabulary softmax total of softmax_total_kernel; then the picks keep their full-vocabulary probabilities
// exp(v - top) / total, topp() trims them at top_p (sampler.cpp:118-150) and the draw is scaled by the nucleus mass (sampler.cpp:47).
static __global__ __launch_bounds__(1024) void topk_sample_kernel(const unsigned long long *cand, int k, float temperature, const float *uniforms, uint32_t *call,
                                                           const double *pen_table, int32_t *last_id, uint32_t *rep_count, uint32_t *token, uint32_t *hist,
                                                           uint32_t *hist_idx, uint32_t *next_id, uint32_t *next_pos, float top_p = 1.0f, const float *total_in = nullptr) {
    __shared__ unsigned long long keys[TOPK_PARTS * TOPK_MAXK];
    __shared__ float prob[TOPK_MAXK];
    const int n = TOPK_PARTS * k;
    int P = 1;
    while (P < n) P <<= 1;
    for (int j = threadIdx.x; j < P; j += blockDim.x) keys[j] = j < n ? cand[(j / k) * TOPK_MAXK + (j % k)] : ~0ull;
    __syncthreads();
    bitonic_sort_keys(keys, P);
    const bool temp = temperature != 1.0f;
    float top = smp_key_value(keys[0]);     // sampler::max over the penalised values: first maximum wins = smallest key
    if (temp) top /= temperature;
    if ((int) threadIdx.x < k) {
        float v = smp_key_value(keys[threadIdx.x]);
        if (temp) v /= temperature;
        prob[threadIdx.x] = keys[threadIdx.x] == ~0ull ? 0.0f : expf(v - top);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int m = k;                            // fewer than k real candidates only when the vocabulary is smaller than k
        while (m > 1 && keys[m - 1] == ~0ull) m--;
        float target = uniforms[call[0]];
        call[0] += 1;
        float cum = 0.0f;
        int chosen;
        if (total_in) {
            // softmax ran over the whole vocabulary (prob / total_in are the reference's logits[] after it); topp(): the first prefix of the picks
            // whose mass reaches top_p, the draw scaled by min(mass, top_p)
            const float total = total_in[0];
            float mass = 0.0f;
            int trim = -1;
            for (int j = 0; j < m; j++) {
                mass += prob[j] / total;
                if (mass >= top_p) { trim = j + 1; break; }
            }
            if (trim > 0) m = trim;
            target *= fminf(mass, top_p);
            chosen = (int) (unsigned) keys[m - 1];
            for (int j = 0; j < m; j++) {
                cum += prob[j] / total;
                if (target <= cum || j + 1 >= m) { chosen = (int) (unsigned) keys[j]; break; }
            }
        } else {
            float total = 0.0f;
            for (int j = 0; j < m; j++) total += prob[j];
            chosen = (int) (unsigned) keys[m - 1];
            for (int j = 0; j < m; j++) {
                cum += prob[j] / total;
                if (target <= cum || j + 1 >= m) { chosen = (int) (unsigned) keys[j]; break; }
            }
        }
        if (pen_table) {
            uint32_t cnt = rep_count[0];
            if (last_id[0] != chosen) cnt = 0;
            last_id[0] = chosen;
            rep_count[0] = cnt + 1;
        }
        token[0] = (uint32_t) chosen;
        if (hist_idx) { hist[hist_idx[0]] = (uint32_t) chosen; hist_idx[0] += 1; }
        else if (hist) hist[0] = (uint32_t) chosen;
        if (next_id) { next_id[0] = (uint32_t) chosen; next_pos[0] += 1; }
    }
}

// gu [R][2F] (gate | up) -> g [R][F] = silu(gate) * up; aq / ad (optional, F % 32 == 0): g also as Q8_0 blocks for the down projection
// n_parts > 1: gu holds n_parts fp32 slabs part_stride floats apart (gemv_stream_kernel), summed in slab order on the way in
static __global__ void silu_mul_kernel(const float *gu, int F, int R, float *g, int8_t *aq, float *ad, int n_parts = 1, int64_t part_stride = 0) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t) R * F) return;
    const int64_t r = i / F, c = i - r * F;
    float x = gu[r * 2 * F + c], u = gu[r * 2 * F + F + c];
    {   // slabs 1..7 requested together (clamped: a slab beyond n_parts re-reads the last one), added in slab order; a loop of dependent loads before
        float tx[7], tu[7];
#pragma unroll
        for (int q = 1; q < 8; q++) {
            const int64_t qo = (int64_t) min(q, n_parts - 1) * part_stride + r * 2 * F + c;
            tx[q - 1] = gu[qo]; tu[q - 1] = gu[qo + F];
        }
#pragma unroll
        for (int q = 1; q < 8; q++)
            if (q < n_parts) { x += tx[q - 1]; u += tu[q - 1]; }
        for (int q = 8; q < n_parts; q++) { x += gu[q * part_stride + r * 2 * F + c]; u += gu[q * part_stride + r * 2 * F + F + c]; }
    }
    const float o = (x / (1.0f + expf(-x))) * u;
    g[i] = o;
    if (aq) q8_block_store(o, i, aq, ad);
}

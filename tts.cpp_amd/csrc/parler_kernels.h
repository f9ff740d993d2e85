// parler_kernels.h — gfx950 (CDNA4) kernels for the Parler-TTS decoder step.
//
// One forward over R "rows" (R = utterances decoded in lock-step, or the S tokens of one text
// prompt) replaces the ggml graph that parler_tts_runner::build_parler_graph
// (/root/reference/src/models/parler/model.cpp:520-614) rebuilds every step.  Kernel ↔ reference:
//   embed_rows_kernel   parler_build_inp_embd            model.cpp:387-410
//   gemm16_kernel<PRO_LN,...>  parler_build_layer_norm + ggml_mul_mat   :412-418, :544-546, :583, :601
//        EPI_QKV        parler_build_kv_store            :420-439   (cache append in the epilogue)
//        EPI_GELU       ggml_gelu                        :602
//        EPI_RESID      ggml_add(residual)               :574, :595, :604
//   attn_kernel         mul_mat(K,q) -> soft_max_ext -> mul_mat(kq,V)   :549-570 / :583-595
//   argmax_kernel       sampler::max                     src/sampler.cpp:185-204
//   sample_kernel       sampler::sample (softmax/topk/topp/inverse CDF)   src/sampler.cpp:3-69,82-183
//   feed_kernel         delay-pattern feed + EOS flags   model.cpp:715-732, :778-785
//
// Design notes (MI355X): a decode step is pure weight streaming (≈725 MB of fp16 weights per
// step, 0.7 GFLOP per row), so the GEMMs are shaped for HBM, not for MFMA peak: one workgroup
// owns 16 output features, its waves split K in 256-wide slices, every lane issues all of its
// weight loads (16 rows x one 64-byte sector per wave instruction) before the LayerNorm prologue
// runs, so HBM latency overlaps the prologue.  MFMA is used because a 16x16 tile gives up to 16
// rows for free at the same weight traffic (lock-step utterances / prompt tokens), not because
// the op is compute bound.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "wave_ops.h"

// ------------------------------------------------------------------------------------------------
// embeddings: x[r] = (audio ? sum_i Emb_i[ids[r][i]] : EmbPrompt[ids[r]]) + Pos[pos[r]]
// ------------------------------------------------------------------------------------------------
struct EmbedArgs {
    const void *tab;        // [n_tabs][rows][H] (audio) or [rows][H] (text)
    int         tab_f16;    // element type of tab
    int64_t     tab_stride; // elements between consecutive tables
    int         n_tabs;     // n_output_heads (audio) or 1 (text)
    const uint32_t *ids;    // [R][n_tabs]
    const float *pos_embed; // [n_pos][H] fp32
    const uint32_t *row_pos;
    float      *x;          // [R][H]
    int         H;
};

static __global__ void embed_rows_kernel(EmbedArgs a) {
    const int r = blockIdx.x;
    const uint32_t pos = a.row_pos[r];
    uint32_t id[16];
#pragma unroll
    for (int i = 0; i < 16; i++) id[i] = a.ids[r * a.n_tabs + min(i, a.n_tabs - 1)];  // all id loads in flight at once (straight-line: a table beyond n_tabs repeats the last)
    // gridDim.y column blocks: every thread does one column per table, all of its loads in one round trip (one workgroup walking the
    // row in 4 dependent passes took 17 us at batch 1)
    for (int c = blockIdx.y * blockDim.x + threadIdx.x; c < a.H; c += gridDim.y * blockDim.x) {
        float v[16];
        if (a.tab_f16) {   // the element type is tested once, not around every load (each test was its own basic block with a wait)
#pragma unroll
            for (int i = 0; i < 16; i++)
                v[i] = (float) ((const _Float16 *) a.tab)[(int64_t) min(i, a.n_tabs - 1) * a.tab_stride + (int64_t) id[i] * a.H + c];
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++)
                v[i] = ((const float *) a.tab)[(int64_t) min(i, a.n_tabs - 1) * a.tab_stride + (int64_t) id[i] * a.H + c];
        }
        const float pe = a.pos_embed[(int64_t) pos * a.H + c];
        float acc = v[0];
#pragma unroll
        for (int i = 1; i < 16; i++)
            if (i < a.n_tabs) acc = v[i] + acc;  // ggml_add(get_rows(i), input_embs), model.cpp:401
        a.x[(int64_t) r * a.H + c] = acc + pe;
    }
}

// ------------------------------------------------------------------------------------------------
// GEMM  y[r][n] = sum_k W[n][k] * act[r][k]   for R <= 16*RB rows, 16 features per workgroup
// ------------------------------------------------------------------------------------------------
// PRO_ATTN: the activations are the split-T self-attention's partials (attn_kernel with nsplit > 1 and no in-kernel combine); the
// workgroup folds them — attn_combine_kernel's arithmetic — into the fp16 rows its MFMAs read (batch-1 chain)
enum { PRO_F32 = 0, PRO_LN = 1, PRO_F16 = 2, PRO_ATTN = 3,
       // gemv_stream_kernel only (Dia's step): the activations are the eight key slices of attn_gqa_split_kernel (merged while they are staged:
       // attn_gqa_combine_kernel's arithmetic) / the gate | up slabs of the preceding projection (silu(gate) * up while staged: silu_mul_kernel's)
       PRO_ATTN8 = 4, PRO_SILU = 5,
       PRO_CROSS = 6 };   // gemm16_kernel (<= 4 rows): the activations are the cross-attention's query rows; the attention over the voice prompt runs in the prologue
#define ATTN_PART 130     // floats per partial of attn_gqa_split_kernel: max, sum, out[128]
#define ATTN_FOLD_NZ 8    // slices a consumer's staging prologue merges (the default split count of the captured steps)
// key-split partials of one (row, head): [nsplit][ATT_PS] floats = max, sum, pad, pad, acc[64] (acc 16-byte aligned)
constexpr int ATT_PS = 68, ATT_PO = 4;
enum { EPI_STORE = 0, EPI_QKV = 1, EPI_RESID = 2, EPI_GELU = 3,
       EPI_CROSS = 4 };   // gemm_tile_kernel<64, 64, ..> only: the tile is one head's cross-attention query for 64 rows; the attention runs in the epilogue

struct GemmArgs {
    const void *W;      // [N][K], fp16 or fp32 (ggml ne=[K,N])
    int K, N, R;
    const void *A;      // activations [R][lda] fp32 (PRO_F32 / PRO_LN) or fp16 (PRO_F16)
    int lda;
    const float *ln_w, *ln_b;  // PRO_LN (K == hidden size)
    float *out;         // EPI_STORE / EPI_RESID / EPI_GELU(fp32): [R][ldo]
    _Float16 *out16;    // EPI_GELU with fp16 activations
    int ldo;
    // EPI_QKV: n in [0,H) -> q, [H,2H) -> K cache, [2H,3H) -> V cache
    float *q;           // [R][H]
    void *kc, *vc;      // this layer's cache base [seq][n_ctx][H]
    int kv_f16;
    int64_t seq_stride; // elements between sequences in the cache
    const uint32_t *row_seq, *row_pos;
    int H;
    int gelu_mode;
    // split-K over workgroups (blockIdx.y = K slice of `kchunk` columns): EPI_STORE writes slab blockIdx.y of
    // `out` (slab_stride floats apart); the consumer (ln_rows_kernel) adds the slabs in a fixed order.
    int kchunk;
    int64_t slab_stride;
    // rows split over workgroups too (blockIdx.z owns rows [z*rows_per_z, (z+1)*rows_per_z)); 0 = one workgroup
    // walks all rows.  Many lock-step rows: more resident workgroups per CU hide the L2 latency of the activation loads.
    int rows_per_z;
    // host-side hint, not read by any kernel: the caller has room for K-slice slabs and a consumer that folds them ->
    // run_gemm may take gemv_stream_kernel (gemv_stream_kernels.h) for <= 16 rows
    int stream;
    // <= 4 rows: the preceding fc2 ran split-K (four times the workgroups, a quarter of the weight bytes each) and left n_parts fp32
    // slabs [n_parts][rows][H]: a PRO_LN prologue normalises x + slabs (x is not written: other workgroups still read it), the next
    // EPI_RESID epilogue stores x + slabs + its own result (every element of x has one owner there).  Fixed slab order in both.
    const float *parts;
    int n_parts;
    int64_t parts_stride;
    // PRO_ATTN: partials [R][att_heads][att_nz][ATT_PS]
    const float *att_part;
    int att_nz, att_heads;
    // EPI_CROSS: cross-attention over the voice prompt inside the cross-q GEMM (parler/model.cpp:576-593): K_c / V_c [cross_E][H] fp32 of this layer,
    // the attended rows go to cross_out (fp32 [R][H]) or cross_out16 (fp16, for an fp16 out projection); q itself is never written
    const float *cross_k, *cross_v;
    int cross_E;
    float cross_scale;
    float *cross_out;
    _Float16 *cross_out16;
    // debug (TTS_HIP_B1_STAMPS=1): s_memrealtime stamps (100 MHz) written by the first and the last workgroup of the launch, 8 slots each
    long long *stamps;
};

#define B1_STAMP(st, i) do { if ((st) != nullptr && threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) \
    (st)[(blockIdx.x ? 8 : 0) + (i)] = (long long) __builtin_amdgcn_s_memrealtime(); } while (0)

__device__ __forceinline__ void gemm_epilogue4(const GemmArgs &a, int EPI, int r, int n, float4v v, int slab) {
    if (EPI == EPI_STORE) {
        *(float4v *) (a.out + (int64_t) slab * a.slab_stride + (int64_t) r * a.ldo + n) = v;
    } else if (EPI == EPI_RESID) {
        float4v *p = (float4v *) (a.out + (int64_t) r * a.ldo + n);
        float4v o = *p;
        if (a.n_parts == 4) {   // the residual stream still lacks the previous fc2's four K-slice slabs
            float4v pv[4];
#pragma unroll
            for (int sp = 0; sp < 4; sp++) pv[sp] = *(const float4v *) (a.parts + sp * a.parts_stride + (int64_t) r * a.ldo + n);
#pragma unroll
            for (int sp = 0; sp < 4; sp++) o += pv[sp];
        }
        o += v;  // ggml_add(cur, residual)
        *p = o;
    } else if (EPI == EPI_GELU) {
        float4v g;
#pragma unroll
        for (int e = 0; e < 4; e++) g[e] = gelu_apply(v[e], a.gelu_mode);
        if (a.out16) {
            half4 h;
#pragma unroll
            for (int e = 0; e < 4; e++) h[e] = (_Float16) g[e];
            *(half4 *) (a.out16 + (int64_t) r * a.ldo + n) = h;
        } else {
            *(float4v *) (a.out + (int64_t) r * a.ldo + n) = g;
        }
    } else {  // EPI_QKV
        const int which = n / a.H;
        const int c = n - which * a.H;
        if (which == 0) {
            *(float4v *) (a.q + (int64_t) r * a.H + c) = v;
        } else {
            const int64_t off = (int64_t) a.row_seq[r] * a.seq_stride + (int64_t) a.row_pos[r] * a.H + c;
            void *base = which == 1 ? a.kc : a.vc;
            if (a.kv_f16) {
                half4 h;
#pragma unroll
                for (int e = 0; e < 4; e++) h[e] = (_Float16) v[e];
                *(half4 *) ((_Float16 *) base + off) = h;
            } else {
                *(float4v *) ((float *) base + off) = v;
            }
        }
    }
}

// WT: 0 = fp32 weights (exact-fp32 MFMA 16x16x4), 1 = fp16 weights (MFMA 16x16x32, activations rounded
// to fp16 like ggml's vec_dot_type conversion).  blockDim.x = 64 * K/256.
template <int WT, int PRO, int EPI, int RB>
__global__ __launch_bounds__(PRO == PRO_LN || PRO == PRO_ATTN || PRO == PRO_CROSS ? 512 : 1024) void gemm16_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // nw waves split this workgroup's K range in 256-wide slices; when the forward carries several groups of
    // 16*RB rows the workgroup holds `ngs` such wave sets and the row groups are dealt round-robin to them,
    // so groups are processed in parallel instead of one after the other.
    const int nw = (a.kchunk ? a.kchunk : a.K) >> 8;
    const int ngs = (blockDim.x >> 6) / nw;
    const int w = wave % nw, gs = wave / nw;
    const int n0 = blockIdx.x * 16;
    const int li = lane & 15, g = lane >> 4;
    const int K = a.K;
    const int kz = blockIdx.y * a.kchunk;  // 0 unless K is split over workgroups
    B1_STAMP(a.stamps, 0);

    // ---- 1. every weight load of this lane, issued RIGHT BEHIND the first staging loads (HBM latency overlaps the prologue) -----
    // MFMA-natural K order: for load c, the 4 lanes (g = 0..3) that share a weight row read one
    // contiguous 64-byte sector of it, so a wave instruction touches 16 rows x 64 B (whole sectors).
    // Round 5: vmcnt retires in issue order, so with the weights requested first the prologue's own inputs (the LayerNorm row, the attention
    // partials, the activation fragments: L2 hits) could not be used before the weights had landed — every launch of the one-sequence chain
    // started its prologue one HBM round trip late (profiles/r05/isa_wait_order_before.txt).  Each prologue now requests its first inputs,
    // calls load_weights(), and only then computes; sched_barriers keep the compiler from merging the two groups again.
    half8   wh[8];
    float4v wf[16];
    auto load_weights = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        if (WT == 1) {
            const _Float16 *wp = (const _Float16 *) a.W + (int64_t) (n0 + li) * K + kz + w * 256 + g * 8;
#pragma unroll
            for (int c = 0; c < 8; c++) wh[c] = __builtin_nontemporal_load((const half8 *) (wp + c * 32));
        } else {
            const float *wp = (const float *) a.W + (int64_t) (n0 + li) * K + kz + w * 256 + g * 4;
#pragma unroll
            for (int c = 0; c < 16; c++) wf[c] = __builtin_nontemporal_load((const float4v *) (wp + c * 16));
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- 2. prologue: LayerNorm of the R rows into LDS (fp16 for WT=1, fp32 for WT=0) -----------
    // One pass: the row (K <= 2048) is held in registers; rows >= R are not computed (their MFMA
    // columns are never stored).
    const int ldx = K + (WT == 1 ? 8 : 4);  // +16 B per row: spreads rows over LDS banks
    _Float16 *xs16 = (_Float16 *) smem;
    float    *xs32 = (float *) smem;
    size_t    red_off = 0;
    if (PRO == PRO_LN) {
        const float *A = (const float *) a.A;
        const int nk = K >> 8;   // chunks of 256 columns in a row (K % 256 == 0 on this path); wave-uniform
        // NI = chunks held in registers: 4 for K <= 1024 (leaves room for the slab fold), 8 up to K = 2048
        auto ln_rows = [&](auto ni_c) __attribute__((always_inline)) {
            constexpr int NI = decltype(ni_c)::value;
            float4v v[NI], lwv[NI], lbv[NI];
            float4v pp[4][4];
            const bool fold = WT == 1 && NI == 4 && a.n_parts == 4;   // (the host splits fc2 only when every matrix is fp16)
            auto load = [&](int r) __attribute__((always_inline)) {
                const float *xr = A + (int64_t) r * a.lda;
#pragma unroll
                for (int i = 0; i < NI; i++) {  // x row and the affine parameters in one round trip
                    // straight-line loads: a chunk beyond the row (i >= nk, wave-uniform) re-reads chunk 0 and is never used.  Loads predicated on
                    // `k < K` became one basic block each and hipcc put an s_waitcnt between them: the row arrived in four dependent round trips.
                    const int k = (i < nk ? i * 256 : 0) + lane * 4;
                    v[i] = *(const float4v *) (xr + k);
                    lwv[i] = *(const float4v *) (a.ln_w + k);
                    lbv[i] = *(const float4v *) (a.ln_b + k);
                }
                if (fold) {   // x + the previous fc2's slabs in slab order; same round trip (K <= 1024: i < 4)
#pragma unroll
                    for (int sp = 0; sp < 4; sp++)
#pragma unroll
                        for (int i = 0; i < 4; i++)
                            pp[sp][i] = *(const float4v *) (a.parts + sp * a.parts_stride + (int64_t) r * a.lda + (i < nk ? i * 256 : 0) + lane * 4);
                }
            };
            auto finish = [&](int r) __attribute__((always_inline)) {
                if (fold) {
#pragma unroll
                    for (int sp = 0; sp < 4; sp++)
#pragma unroll
                        for (int i = 0; i < 4; i++) v[i] += pp[sp][i];
                }
                float s = 0.0f;
#pragma unroll
                for (int i = 0; i < NI; i++)
                    if (i < nk) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
                const float mean = wave_sum(s) / (float) K;
                float s2 = 0.0f;
#pragma unroll
                for (int i = 0; i < NI; i++) {
                    if (i < nk) {
#pragma unroll
                        for (int e = 0; e < 4; e++) { const float d = v[i][e] - mean; s2 += d * d; }
                    }
                }
                const float rstd = 1.0f / sqrtf(wave_sum(s2) / (float) K + LN_EPS);
#pragma unroll
                for (int i = 0; i < NI; i++) {
                    const int k = i * 256 + lane * 4;
                    if (i < nk) {
                        const float4v lw = lwv[i], lb = lbv[i];
                        float4v y;
#pragma unroll
                        for (int e = 0; e < 4; e++) y[e] = (v[i][e] - mean) * rstd * lw[e] + lb[e];
                        if (WT == 1) {
                            half4 h;
#pragma unroll
                            for (int e = 0; e < 4; e++) h[e] = (_Float16) y[e];
                            *(half4 *) (xs16 + (size_t) r * ldx + k) = h;
                        } else {
                            *(float4v *) (xs32 + (size_t) r * ldx + k) = y;
                        }
                    }
                }
            };
            // this wave's first row (a wave without one re-reads the last row and drops it), the weights behind it, then the arithmetic
            // (fp32 weights: 64 registers of fragments beside a 2048-wide row — the old order, or the pair spills)
            if (WT == 0) load_weights();
            load(min(wave, a.R - 1));
            if (WT == 1) load_weights();
            if (wave < a.R) finish(wave);
            for (int r = wave + nw * ngs; r < a.R; r += nw * ngs) { load(r); finish(r); }
        };
        if (K <= 1024) ln_rows(std::integral_constant<int, 4>{});
        else ln_rows(std::integral_constant<int, 8>{});
        red_off = (size_t) RB * 16 * ldx * (WT == 1 ? 2 : 4);
        red_off = (red_off + 15) & ~(size_t) 15;
        __syncthreads();
    }
    if (PRO == PRO_ATTN) {   // WT == 1
        const int nq = K >> 2, nz = a.att_nz;   // K = att_heads * 64
        const int n_items = a.R * nq;
        float mm[16], ss[16];
        float4v oo[16];
        auto load = [&](int idx) __attribute__((always_inline)) {
            const int r = idx / nq, c4 = (idx - r * nq) * 4, h = c4 >> 6, d = c4 & 63;
            const float *p = a.att_part + ((int64_t) r * a.att_heads + h) * nz * ATT_PS;
#pragma unroll
            for (int i = 0; i < 16; i++) {   // straight-line loads (a split beyond nz re-reads the last one, see the LayerNorm prologue)
                const int ic = i < nz ? i : nz - 1;
                mm[i] = p[ic * ATT_PS]; ss[i] = p[ic * ATT_PS + 1]; oo[i] = *(const float4v *) (p + ic * ATT_PS + ATT_PO + d);
            }
        };
        auto finish = [&](int idx) __attribute__((always_inline)) {
            const int r = idx / nq, c4 = (idx - r * nq) * 4;
            float gmx = -INFINITY;
#pragma unroll
            for (int i = 0; i < 16; i++) if (i < nz) gmx = fmaxf(gmx, mm[i]);
            float4v ot = {0.f, 0.f, 0.f, 0.f};
            float st = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (i < nz && mm[i] != -INFINITY) {  // -inf marks an empty chunk
                    const float f2 = expf(mm[i] - gmx);
#pragma unroll
                    for (int e = 0; e < 4; e++) ot[e] += f2 * oo[i][e];
                    st += f2 * ss[i];
                }
            }
            half4 hh;
#pragma unroll
            for (int e = 0; e < 4; e++) hh[e] = (_Float16) (ot[e] / st);
            *(half4 *) (xs16 + (size_t) r * ldx + c4) = hh;
        };
        // this thread's first item (a thread without one re-reads the last item and drops it), the weights behind it, then the merge
        load(min(tid, n_items - 1));
        load_weights();
        if (tid < n_items) finish(tid);
        for (int idx = tid + (int) blockDim.x; idx < n_items; idx += blockDim.x) { load(idx); finish(idx); }
        red_off = (size_t) RB * 16 * ldx * 2;
        red_off = (red_off + 15) & ~(size_t) 15;
        __syncthreads();
    }
    if (PRO == PRO_CROSS) {   // WT == 1, no K split, <= 4 rows, blockDim.x == K / 4 == 16 lanes per 64-wide head
        // The out projection of the cross-attention block takes the attention itself into its prologue (one-sequence chain, round 5): a.A holds the
        // query rows (the cross-q GEMM's output), cross_k / cross_v the layer's K_c / V_c [E][H].  Thread t owns channels 4 t .. 4 t + 3 (head t / 16):
        // score of a key = row16_sum of a 4-term partial dot, running max / sum / output per row (soft_max_ext + mul_mat as one pass over the <= 32
        // keys; attn_short_kernel's mathematics up to the order of the sums); the attended rows go to LDS as fp16, where the MFMA loop reads them as
        // it does a LayerNorm prologue's.  Every workgroup repeats the attention (68 KB of K_c / V_c from L2 at 8 positions): ~0.8 us of prologue
        // against the 3 us of a launch (attn_short_kernel's span + the gap in front of it, profiles/r05/b1_chain_call2.txt).
        const int E = a.cross_E, col = tid * 4;
        const float *qa = (const float *) a.A;
        float4v q4[4], kf[8], vf[8];
#pragma unroll
        for (int r = 0; r < 4; r++) q4[r] = *(const float4v *) (qa + (int64_t) min(r, a.R - 1) * a.lda + col);
        auto request = [&](int e0) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int64_t off = (int64_t) min(e0 + u, E - 1) * K + col;
                kf[u] = *(const float4v *) (a.cross_k + off);
                vf[u] = *(const float4v *) (a.cross_v + off);
            }
        };
        request(0);
        load_weights();
        float m[4], l[4];
        float4v acc[4];
#pragma unroll
        for (int r = 0; r < 4; r++) { m[r] = -INFINITY; l[r] = 0.0f; acc[r] = (float4v){0.f, 0.f, 0.f, 0.f}; }
        auto chunk = [&](int e0) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (r < a.R) {
                    // the eight scores of a chunk are independent of each other: one merge with the running state per chunk (per key it was a chain
                    // of eight dependent max / exp / fma groups: 2.3 us of prologue, profiles/r05/b1_cross_fold_call13.txt)
                    float sc[8], mc = -INFINITY;
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const float d = q4[r][0] * kf[u][0] + q4[r][1] * kf[u][1] + q4[r][2] * kf[u][2] + q4[r][3] * kf[u][3];
                        sc[u] = e0 + u < E ? row16_sum(d) * a.cross_scale : -INFINITY;
                        mc = fmaxf(mc, sc[u]);
                    }
                    const float mn = fmaxf(m[r], mc);
                    const float f = expf(m[r] - mn);   // 0 on the first chunk (m = -inf)
                    float ls = 0.0f;
                    float4v as = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const float p = expf(sc[u] - mn);   // 0 for a key beyond E
                        ls += p;
#pragma unroll
                        for (int e = 0; e < 4; e++) as[e] += p * vf[u][e];
                    }
                    l[r] = l[r] * f + ls;
#pragma unroll
                    for (int e = 0; e < 4; e++) acc[r][e] = acc[r][e] * f + as[e];
                    m[r] = mn;
                }
            }
        };
        chunk(0);   // outside the loop: a loop header waits for every outstanding load, the weights included (profiles/tools/isa_wait_order.py)
        for (int e0 = 8; e0 < E; e0 += 8) { request(e0); chunk(e0); }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (r < a.R) {
                half4 hh;
#pragma unroll
                for (int e = 0; e < 4; e++) hh[e] = (_Float16) (acc[r][e] / l[r]);
                *(half4 *) (xs16 + (size_t) r * ldx + col) = hh;
            }
        }
        red_off = (size_t) RB * 16 * ldx * 2;
        red_off = (red_off + 15) & ~(size_t) 15;
        __syncthreads();
    }
    // activations straight from memory (PRO_F16 / PRO_F32): the fragments of the first row group ahead of the weights when a group is one
    // row block (<= 16 rows per group: the one-sequence chain and the small lock-step batches); larger groups keep the weights in front
    const int r_lo = a.rows_per_z ? (int) blockIdx.z * a.rows_per_z : 0;
    const int r_hi = a.rows_per_z ? min(a.R, r_lo + a.rows_per_z) : a.R;
    constexpr bool PRELOAD = WT == 1 && RB == 1 && (PRO == PRO_F16 || PRO == PRO_F32);
    half8 bpre[8];
    if (PRO != PRO_LN && PRO != PRO_ATTN && PRO != PRO_CROSS) {
        if (PRELOAD) {
            const int r = r_lo + gs * 16 + li;
            const int rr = max(0, min(r, r_hi - 1));
            const int kb = kz + w * 256 + g * 8;
            if (PRO == PRO_F16) {
#pragma unroll
                for (int c = 0; c < 8; c++) bpre[c] = *(const half8 *) ((const _Float16 *) a.A + (int64_t) rr * a.lda + kb + c * 32);
                load_weights();
            } else {
                float4v f[16];
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    const float *p = (const float *) a.A + (int64_t) rr * a.lda + kb + c * 32;
                    f[2 * c] = *(const float4v *) p; f[2 * c + 1] = *(const float4v *) (p + 4);
                }
                load_weights();
#pragma unroll
                for (int c = 0; c < 8; c++)
#pragma unroll
                    for (int e = 0; e < 4; e++) { bpre[c][e] = (_Float16) f[2 * c][e]; bpre[c][4 + e] = (_Float16) f[2 * c + 1][e]; }
            }
        } else {
            load_weights();
        }
    }
    B1_STAMP(a.stamps, 1);

    // ---- 3-5. for every group of 16*RB rows: MFMA over this wave's 256-wide K slice, reduce the K slices
    //           across waves, epilogue.  The weight fragments stay in registers across row groups, so many
    //           lock-step utterances cost one pass over the weights.
    const int n_groups = (r_hi - r_lo + 16 * RB - 1) / (16 * RB);
    const int n_rounds = (n_groups + ngs - 1) / ngs;   // uniform trip count: every wave reaches every barrier
    for (int rd = 0; rd < n_rounds; rd++) {
        const int rg = r_lo + (rd * ngs + gs) * 16 * RB;      // may lie beyond R for the last round: nothing is stored then
        float4v acc[RB];
#pragma unroll
        for (int rb = 0; rb < RB; rb++) acc[rb] = (float4v){0.f, 0.f, 0.f, 0.f};

#pragma unroll
        for (int rb = 0; rb < RB; rb++) {
            const int r  = rg + rb * 16 + li;
            const int rr = r < r_hi ? r : r_hi - 1;
            if (WT == 1) {
                const int kb = kz + w * 256 + g * 8;
                if (PRELOAD && rd == 0) {   // (wave-uniform) the first group's fragments were requested ahead of the weights
#pragma unroll
                    for (int c = 0; c < 8; c++) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c], bpre[c], acc[rb], 0, 0, 0);
                } else {
#pragma unroll
                    for (int c = 0; c < 8; c++) {
                        half8 b;
                        const int k = kb + c * 32;
                        if (PRO == PRO_LN || PRO == PRO_ATTN || PRO == PRO_CROSS) {
                            b = *(const half8 *) (xs16 + (size_t) r * ldx + k);
                        } else if (PRO == PRO_F16) {
                            b = *(const half8 *) ((const _Float16 *) a.A + (int64_t) rr * a.lda + k);
                        } else {
                            const float *p = (const float *) a.A + (int64_t) rr * a.lda + k;
                            const float4v f0 = *(const float4v *) p, f1 = *(const float4v *) (p + 4);
#pragma unroll
                            for (int e = 0; e < 4; e++) { b[e] = (_Float16) f0[e]; b[4 + e] = (_Float16) f1[e]; }
                        }
                        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c], b, acc[rb], 0, 0, 0);
                    }
                }
            } else {
                const int kb = kz + w * 256 + g * 4;
#pragma unroll
                for (int c = 0; c < 16; c++) {
                    float4v b;
                    const int k = kb + c * 16;
                    if (PRO == PRO_LN) b = *(const float4v *) (xs32 + (size_t) r * ldx + k);
                    else               b = *(const float4v *) ((const float *) a.A + (int64_t) rr * a.lda + k);
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[c][e], b[e], acc[rb], 0, 0, 0);
                }
            }
        }

        if (a.stamps != nullptr) { asm volatile("" :: "v"(acc[0][0])); B1_STAMP(a.stamps, 2); }
        // reduce the K slices across waves (fixed order: deterministic) and run the epilogue; the row blocks
        // of the group are spread over the waves (wave w owns row blocks w, w+nw, ...) so that neither the
        // reduction nor the scattered epilogue stores serialise on one wave
        if (nw > 1) {
            float *red = (float *) (smem + red_off) + (size_t) gs * nw * RB * 256;  // [ngs][nw][RB][4][64]
            if (rd > 0) __syncthreads();              // previous round's partials have been consumed
#pragma unroll
            for (int rb = 0; rb < RB; rb++)
#pragma unroll
                for (int e = 0; e < 4; e++) red[((w * RB + rb) * 4 + e) * 64 + lane] = acc[rb][e];
            __syncthreads();
#pragma unroll
            for (int rb = 0; rb < RB; rb++) {
                if ((rb % nw) != w) continue;
                float4v t;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float sum = red[((0 * RB + rb) * 4 + e) * 64 + lane];
                    for (int ww = 1; ww < nw; ww++) sum += red[((ww * RB + rb) * 4 + e) * 64 + lane];
                    t[e] = sum;
                }
                const int r = rg + rb * 16 + li;
                if (r < r_hi) gemm_epilogue4(a, EPI, r, n0 + g * 4, t, blockIdx.y);
            }
        } else {
#pragma unroll
            for (int rb = 0; rb < RB; rb++) {
                const int r = rg + rb * 16 + li;
                if (r < r_hi) gemm_epilogue4(a, EPI, r, n0 + g * 4, acc[rb], blockIdx.y);
            }
        }
    }
    B1_STAMP(a.stamps, 3);
}

// Scalar-FMA reference GEMV (debug / parity cross-check on the device, and shapes with K % 256 != 0).
// One wave per output feature; activations already normalised by ln_rows_kernel when needed.
template <int WT>
__global__ void gemv_valu_kernel(GemmArgs a, int EPI, int act_f16_src) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n = blockIdx.x * (blockDim.x >> 6) + w;
    if (n >= a.N) return;
    for (int r = 0; r < a.R; r++) {
        float acc = 0.0f;
        for (int k = lane; k < a.K; k += 64) {
            float wv, xv;
            if (WT == 1) wv = (float) ((const _Float16 *) a.W)[(int64_t) n * a.K + k];
            else         wv = ((const float *) a.W)[(int64_t) n * a.K + k];
            if (act_f16_src) xv = (float) ((const _Float16 *) a.A)[(int64_t) r * a.lda + k];
            else {
                xv = ((const float *) a.A)[(int64_t) r * a.lda + k];
                if (WT == 1) xv = (float) (_Float16) xv;  // vec_dot_type conversion of src1
            }
            acc += wv * xv;
        }
        acc = wave_sum(acc);
        if (lane == 0) {
            if (EPI == EPI_STORE) a.out[(int64_t) r * a.ldo + n] = acc;
            else if (EPI == EPI_RESID) a.out[(int64_t) r * a.ldo + n] += acc;
            else if (EPI == EPI_GELU) {
                const float gl = gelu_apply(acc, a.gelu_mode);
                if (a.out16) a.out16[(int64_t) r * a.ldo + n] = (_Float16) gl;
                else a.out[(int64_t) r * a.ldo + n] = gl;
            } else {
                const int which = n / a.H, c = n - which * a.H;
                if (which == 0) a.q[(int64_t) r * a.H + c] = acc;
                else {
                    const int64_t off = (int64_t) a.row_seq[r] * a.seq_stride + (int64_t) a.row_pos[r] * a.H + c;
                    void *base = which == 1 ? a.kc : a.vc;
                    if (a.kv_f16) ((_Float16 *) base)[off] = (_Float16) acc;
                    else ((float *) base)[off] = acc;
                }
            }
        }
    }
}

// LayerNorm of R rows, one wave per row, row held in registers (H <= 2048 on this path, any H on the
// looped fallback).  Used when R is large enough that re-normalising every row inside every GEMM
// workgroup would dominate (and for the debug "hidden" read-back / scalar GEMV path).
// y32 and/or y16 may be NULL.
//
// Residual hand-off: when `parts` != NULL the preceding out_proj / fc2 GEMM ran split-K and left n_parts fp32
// slabs [n_parts][R][H]; this kernel performs the reference's ggml_add(cur, residual): x[r] += sum_s parts[s][r]
// (fixed order), stores the new residual stream in place (x is written by the wave that owns the row) and
// normalises it.  Requires H <= 2048.
// One wave normalises one row held in registers (NI float4 per lane: H <= 1024 for NI = 4, <= 2048 for NI = 8).
// NP >= 0: that many split-K slabs are added to x in slab order first, with every load in flight at once
// (the slab loop with a run-time trip count serialised 4 dependent round trips per float4: 13.8 us -> see DESIGN.md);
// NP < 0: run-time slab count.
__device__ __forceinline__ unsigned quant4_q8(float4v y, float &d_out);   // below (quantised weights)
// yq / ydT != NULL: the normalised row leaves as Q8_0 blocks for a quantised consumer (qgemm_tile_kernels.h): codes yq[k], block scale ydT[(k / 32) * ldr]
// (the caller passes the row's column of the transposed scale table).  A lane's float4 is an eighth of a block: quant4_q8, H % 32 == 0.
template <int NI, int NP>
__device__ __forceinline__ void ln_row_regs(float *xr, int H, int lane, const float *lw, const float *lb, float *yr32, _Float16 *yr16,
                                            const float *pr, int n_parts, int64_t slab_stride, int8_t *yq = nullptr, float *ydT = nullptr, int ldr = 0) {
    // straight-line loads with clamped offsets (a piece beyond the row re-reads the row's last float4 and is never used): under `if (k < H)`
    // every piece was its own basic block behind a full s_waitcnt — NI dependent round trips for a row that is one round trip of data
    float4v v[NI], w4[NI], b4[NI];
    float4v p[NP > 0 ? NP : 1][NI];
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const int k = min(i * 256 + lane * 4, H - 4);
        v[i] = *(const float4v *) (xr + k);
#pragma unroll
        for (int sp = 0; sp < NP; sp++) p[sp][i] = *(const float4v *) (pr + sp * slab_stride + k);
        w4[i] = *(const float4v *) (lw + k);
        b4[i] = *(const float4v *) (lb + k);
    }
    if (NP != 0) {
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const int k = i * 256 + lane * 4;
            if (k < H) {
                if (NP > 0) {
#pragma unroll
                    for (int sp = 0; sp < NP; sp++) v[i] += p[sp][i];
                } else {
                    for (int sp = 0; sp < n_parts; sp++) v[i] += *(const float4v *) (pr + sp * slab_stride + k);
                }
                *(float4v *) (xr + k) = v[i];
            }
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NI; i++)
        if (i * 256 + lane * 4 < H) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = wave_sum(s) / (float) H;
    float s2 = 0.0f;
#pragma unroll
    for (int i = 0; i < NI; i++)
        if (i * 256 + lane * 4 < H) {
#pragma unroll
            for (int e = 0; e < 4; e++) { const float d = v[i][e] - mean; s2 += d * d; }
        }
    const float rstd = 1.0f / sqrtf(wave_sum(s2) / (float) H + LN_EPS);
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const int k = i * 256 + lane * 4;
        if (k < H) {
            float4v y;
#pragma unroll
            for (int e = 0; e < 4; e++) y[e] = (v[i][e] - mean) * rstd * w4[i][e] + b4[i][e];
            if (yr32) *(float4v *) (yr32 + k) = y;
            if (yr16) {
                half4 h;
#pragma unroll
                for (int e = 0; e < 4; e++) h[e] = (_Float16) y[e];
                *(half4 *) (yr16 + k) = h;
            }
            if (yq) {
                float dd;
                const unsigned qq = quant4_q8(y, dd);
                *(unsigned *) (yq + k) = qq;
                if ((lane & 7) == 0) ydT[(int64_t) (k >> 5) * ldr] = dd;
            }
        }
    }
}

// One kernel per (row width class, slab count): a single kernel that branches over the variants is allocated the registers of the
// largest one (256 VGPRs + 96 AGPRs: one wave per SIMD, and next to the codec's workgroups a launch waited for two thirds of a
// SIMD's register file to drain — 29.7 us per launch with three runners against 7.5 alone).  NI: float4 pieces per lane (4: H <= 1024,
// 8: H <= 2048); NP: slabs folded (-1: any count, looped).
template <int NI, int NP>
__global__ __launch_bounds__(256) void ln_rows_t_kernel(float *x, int H, const float *lw, const float *lb, float *y32, _Float16 *y16, int R, const float *parts, int n_parts,
                                                        int64_t slab_stride) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= R) return;
    const float *pr = parts ? parts + (int64_t) r * H : nullptr;
    ln_row_regs<NI, NP>(x + (int64_t) r * H, H, lane, lw, lb, y32 ? y32 + (int64_t) r * H : nullptr, y16 ? y16 + (int64_t) r * H : nullptr, pr, parts ? n_parts : 0, slab_stride);
}

// the same with Q8_0 output for the tiled integer GEMM: codes q [R][H], block scales dT float [H / 32][ldr]
template <int NI, int NP>
__global__ __launch_bounds__(256) void ln_rows_q8t_kernel(float *x, int H, const float *lw, const float *lb, int8_t *q, float *dT, int ldr, int R, const float *parts, int n_parts,
                                                          int64_t slab_stride) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= R) return;
    const float *pr = parts ? parts + (int64_t) r * H : nullptr;
    ln_row_regs<NI, NP>(x + (int64_t) r * H, H, lane, lw, lb, (float *) nullptr, (_Float16 *) nullptr, pr, parts ? n_parts : 0, slab_stride, q + (int64_t) r * H, dT + r, ldr);
}

// any H: three passes over the row
static __global__ __launch_bounds__(256) void ln_rows_kernel(float *x, int H, const float *lw, const float *lb, float *y32,
                                                      _Float16 *y16, int R, const float *parts, int n_parts, int64_t slab_stride) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= R) return;
    float *xr = x + (int64_t) r * H;
    if (parts)
        for (int k = lane; k < H; k += 64) {
            float v = xr[k];
            for (int sp = 0; sp < n_parts; sp++) v += parts[sp * slab_stride + (int64_t) r * H + k];
            xr[k] = v;
        }
    float s = 0.0f;
    for (int k = lane; k < H; k += 64) s += xr[k];
    const float mean = wave_sum(s) / (float) H;
    float s2 = 0.0f;
    for (int k = lane; k < H; k += 64) { const float d = xr[k] - mean; s2 += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(s2) / (float) H + LN_EPS);
    for (int k = lane; k < H; k += 64) {
        const float y = (xr[k] - mean) * rstd * lw[k] + lb[k];
        if (y32) y32[(int64_t) r * H + k] = y;
        if (y16) y16[(int64_t) r * H + k] = (_Float16) y;
    }
}

// ------------------------------------------------------------------------------------------------
// attention for one query row per (head, row): softmax(q·K^T / sqrt(d)) · V over T cached positions.
// head_dim is 64 (Parler-Mini/Large).  16 lanes own one key (4 channels each, one 256-B line per
// key per head), 16 keys in flight per workgroup pass; the causal mask of model.cpp:623-631 is
// implicit in T = pos+1, the all-zero cross mask (:633-641) in T = n_encode_length.
// Split-T: blockIdx.z owns keys [z*chunk, (z+1)*chunk); partial (max, sum, acc[64]) are merged by
// attn_combine_kernel when gridDim.z > 1.
// ------------------------------------------------------------------------------------------------
struct AttnArgs {
    const float *q;        // [R][H]
    const void *kc, *vc;   // [seq][n_ctx][H]  (or [E][H] for cross)
    int kv_f16;
    int64_t seq_stride;    // 0 for cross attention
    const uint32_t *row_seq, *row_pos;  // NULL for cross attention
    int T_fixed;           // cross: n_encode_length
    int H, n_heads;
    float scale;
    float *out;            // [R][H]
    _Float16 *out16;       // [R][H] fp16 copy for an fp16-weight out_proj (same rounding the GEMM would apply), or NULL
    float *part;           // [R][n_heads][nsplit][ATT_PS] when nsplit > 1
    int max_T;             // LDS score capacity
    // nsplit > 1 with counters != NULL: the workgroup that finishes a (row, head) last folds the nsplit partials itself (same fixed
    // order as attn_combine_kernel) — no separate combine launch; counters [R][n_heads] start at 0 and are left at 0
    uint32_t *counters;
    long long *stamps;     // debug, as GemmArgs
    // attn_rows_kernel / attn_combine_kernel with a quantised out projection on the tiled path: the attended rows leave as Q8_0 blocks as well
    // (codes out_q [R][H], block scales out_dT float [H / 32][ldr]) — ggml's quantize_row_q8_0 of the row the fp32 store holds
    int8_t *out_q;
    float *out_dT;
    int ldr;
};

__device__ __forceinline__ float4v load_kv4(const void *base, int kv_f16, int64_t off) {
    if (kv_f16) {
        const half4 h = *(const half4 *) ((const _Float16 *) base + off);
        return (float4v){(float) h[0], (float) h[1], (float) h[2], (float) h[3]};
    }
    return *(const float4v *) ((const float *) base + off);
}

template <bool KVF16>
__device__ __forceinline__ void attn_key_pass(const AttnArgs &a, int64_t hb, float4v q4, int tbeg, int t1, int step, int NKG, float &m, float &l, float4v &acc) {
    for (int t = tbeg; t < t1; t += step) {
        // the eight rows of a pass requested back to back (a key beyond the slice re-reads the slice's last key and is never used): under
        // `if (t + u * NKG < t1)` each pair of loads was a basic block behind its own s_waitcnt, four dependent round trips per pass
        float4v k4[4], v4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int64_t off = hb + (int64_t) min(t + u * NKG, t1 - 1) * a.H;
            if (KVF16) {
                const half4 hk = *(const half4 *) ((const _Float16 *) a.kc + off), hv = *(const half4 *) ((const _Float16 *) a.vc + off);
                k4[u] = (float4v){(float) hk[0], (float) hk[1], (float) hk[2], (float) hk[3]};
                v4[u] = (float4v){(float) hv[0], (float) hv[1], (float) hv[2], (float) hv[3]};
            } else {
                k4[u] = *(const float4v *) ((const float *) a.kc + off);
                v4[u] = *(const float4v *) ((const float *) a.vc + off);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (t + u * NKG < t1) {
                float d = q4[0] * k4[u][0] + q4[1] * k4[u][1] + q4[2] * k4[u][2] + q4[3] * k4[u][3];
                d = row16_sum(d);
                d *= a.scale;  // soft_max_ext(kq, mask, 1/sqrt(d), 0)
                const float mn = fmaxf(m, d);
                const float f = expf(m - mn);   // 0 on the first key (m = -inf)
                const float p = expf(d - mn);
                l = l * f + p;
#pragma unroll
                for (int e = 0; e < 4; e++) acc[e] = acc[e] * f + p * v4[u][e];
                m = mn;
            }
        }
    }
}

// blockDim.x = 16 * NKG threads (NKG key groups of 16 lanes; 16 for 256 threads, 64 for 1024 threads: the
// wide form keeps a single workgroup per (head,row) fast enough that small batches need no split-T pass).
// Single pass: the K and V rows of a key are loaded together (one HBM round trip per 4*NKG keys instead of
// two dependent passes) and folded with a running max / sum per key group; the groups are merged through
// LDS in a fixed order.  Mathematically soft_max_ext + mul_mat; rounding differs from the two-pass form
// only in the order of fp32 operations.
// (8 keys per lane group and pass instead of 4 — 128 keys per workgroup and round trip — was measured for the batch-1 chain: 183 VGPRs and
// twice the loads in flight per CU made the key pass slower, 3.7 -> 6.0 us at T ~ 1000, profiles/r03/b1_chain_attn_variants.txt.)
static __global__ __launch_bounds__(1024) void attn_kernel(AttnArgs a) {   // U = 8 is launched with 256 threads only (nsplit > 1)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NKG = blockDim.x >> 4;
    float *red = (float *) smem;                 // [NKG][64] acc, [NKG] max, [NKG] sum
    const int h = blockIdx.x, r = blockIdx.y, z = blockIdx.z, nz = gridDim.z;
    const int tid = threadIdx.x, kg = tid >> 4, cl = tid & 15, c4 = cl * 4;
    B1_STAMP(a.stamps, 0);
    const int T = a.row_pos ? (int) a.row_pos[r] + 1 : a.T_fixed;
    const int64_t sb = a.row_seq ? (int64_t) a.row_seq[r] * a.seq_stride : 0;
    const int chunk = (T + nz - 1) / nz;
    const int t0 = z * chunk, t1 = min(T, t0 + chunk);
    const int64_t hb = sb + h * 64 + c4;
    const float4v q4 = *(const float4v *) (a.q + (int64_t) r * a.H + h * 64 + c4);
    const int step = NKG * 4;

    float m = -INFINITY, l = 0.0f;
    float4v acc = {0.f, 0.f, 0.f, 0.f};
    // the cache type is a run-time flag of the context: one instance of the key pass per type, chosen once (a `flag ? half load : float load`
    // inside the pass made every load its own basic block behind a wait: 11 dependent load groups in the ISA of the dominant kernel)
    if (a.kv_f16) attn_key_pass<true>(a, hb, q4, t0 + kg, t1, step, NKG, m, l, acc);
    else attn_key_pass<false>(a, hb, q4, t0 + kg, t1, step, NKG, m, l, acc);
    if (a.stamps != nullptr) { asm volatile("" :: "v"(acc[0])); B1_STAMP(a.stamps, 1); }
    // merge the key groups
    if (cl == 0) red[NKG * 64 + kg] = m;
    __syncthreads();
    float mx = -INFINITY;
    for (int i = 0; i < NKG; i++) mx = fmaxf(mx, red[NKG * 64 + i]);
    const float f = (m == -INFINITY) ? 0.0f : expf(m - mx);
#pragma unroll
    for (int e = 0; e < 4; e++) acc[e] *= f;
    *(float4v *) (red + kg * 64 + c4) = acc;
    if (cl == 0) red[NKG * 65 + kg] = l * f;
    __syncthreads();
    if (tid < 64) {
        float o = 0.0f, s = 0.0f;
        for (int i = 0; i < NKG; i++) { o += red[i * 64 + tid]; s += red[NKG * 65 + i]; }
        if (nz == 1) {
            const float res = o / s;
            if (a.out16) a.out16[(int64_t) r * a.H + h * 64 + tid] = (_Float16) res;
            else a.out[(int64_t) r * a.H + h * 64 + tid] = res;
        } else {
            float *p = a.part + (((int64_t) r * a.n_heads + h) * nz + z) * ATT_PS;
            if (a.counters) {
                // agent-scope atomic stores: the partial is read inside this launch by a workgroup on another CU / XCD (L2 is per XCD)
                if (tid == 0) {
                    __hip_atomic_store(p, mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(p + 1, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __hip_atomic_store(p + ATT_PO + tid, o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {   // read by a later launch (attn_combine_kernel or out_proj's PRO_ATTN prologue)
                if (tid == 0) { p[0] = mx; p[1] = s; }
                p[ATT_PO + tid] = o;
            }
        }
    }
    B1_STAMP(a.stamps, 2);
    if (nz == 1 || !a.counters) return;
    // last workgroup of this (row, head): fold the partials in split order (attn_combine_kernel's arithmetic)
    uint32_t *s_last = (uint32_t *) (red + NKG * 66);   // one of the 16 spare words behind [NKG][66]
    __syncthreads();   // the 64 stores above are issued
    if (tid == 0) {
        const uint32_t prev = __hip_atomic_fetch_add(a.counters + r * a.n_heads + h, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        *s_last = prev == (uint32_t) nz - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!*s_last || tid >= 64) return;
    const float *p = a.part + ((int64_t) r * a.n_heads + h) * nz * ATT_PS;
    float mm[16], ss[16], oo[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        if (i < nz) {
            mm[i] = __hip_atomic_load(p + i * ATT_PS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ss[i] = __hip_atomic_load(p + i * ATT_PS + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            oo[i] = __hip_atomic_load(p + i * ATT_PS + ATT_PO + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    float gmx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; i++) if (i < nz) gmx = fmaxf(gmx, mm[i]);
    float ot = 0.0f, st = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        if (i < nz && mm[i] != -INFINITY) {  // -inf marks an empty chunk
            const float f2 = expf(mm[i] - gmx);
            ot += f2 * oo[i];
            st += f2 * ss[i];
        }
    }
    const float res = ot / st;
    if (a.out16) a.out16[(int64_t) r * a.H + h * 64 + tid] = (_Float16) res;
    else a.out[(int64_t) r * a.H + h * 64 + tid] = res;
    if (tid == 0) __hip_atomic_store(a.counters + r * a.n_heads + h, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.stamps != nullptr && tid == 0 && (h == 0 || h == (int) gridDim.x - 1) && r == 0) a.stamps[(h ? 8 : 0) + 3] = (long long) __builtin_amdgcn_s_memrealtime();
}

// Self-attention of a many-row forward, one workgroup per ROW: 16 lanes per head, all heads of the row side by side (threads = 16 n_heads), the
// keys of the row one after the other with the rows of eight keys in flight per lane.  A wave-instruction of the workgroup then reads ONE key's
// whole K (or V) row — 4 KB contiguous for Parler-Mini — where attn_kernel's one-head workgroups read 256-byte pieces 4 KB apart and leave it to
// the dispatcher to run the 16 heads of a row close enough in time for the DRAM pages to be shared (5.7 TB/s; a plain streaming read reaches
// 7.1 TB/s on this chip, profiles/r04/overlap_bench_synthetic.txt).  Every 16-lane group owns its head from the first key to the last: one running
// max / sum, no LDS, no barrier, no merge.  soft_max_ext + mul_mat as one pass in key order (attn_kernel folds sixteen interleaved key groups).
// nz > 1: the keys of a row in nz slices (blockIdx.y), partials as attn_kernel writes them for attn_combine_kernel.
template <bool KVF16, int U>
__device__ __forceinline__ void attn_row_pass(const AttnArgs &a, int64_t hb, float4v q4, int t0, int t1, float &m, float &l, float4v &acc) {
    for (int t = t0; t < t1; t += U) {
        float4v k4[U], v4[U];
#pragma unroll
        for (int u = 0; u < U; u++) {   // straight-line: a key beyond the slice re-reads its last key and is never used
            const int64_t off = hb + (int64_t) min(t + u, t1 - 1) * a.H;
            if (KVF16) {
                const half4 hk = __builtin_nontemporal_load((const half4 *) ((const _Float16 *) a.kc + off)), hv = __builtin_nontemporal_load((const half4 *) ((const _Float16 *) a.vc + off));
                k4[u] = (float4v){(float) hk[0], (float) hk[1], (float) hk[2], (float) hk[3]};
                v4[u] = (float4v){(float) hv[0], (float) hv[1], (float) hv[2], (float) hv[3]};
            } else {
                k4[u] = __builtin_nontemporal_load((const float4v *) ((const float *) a.kc + off));
                v4[u] = __builtin_nontemporal_load((const float4v *) ((const float *) a.vc + off));
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (t + u < t1) {
                float d = q4[0] * k4[u][0] + q4[1] * k4[u][1] + q4[2] * k4[u][2] + q4[3] * k4[u][3];
                d = row16_sum(d);
                d *= a.scale;  // soft_max_ext(kq, mask, 1/sqrt(d), 0)
                const float mn = fmaxf(m, d);
                const float f = expf(m - mn);   // 0 on the first key (m = -inf)
                const float p = expf(d - mn);
                l = l * f + p;
#pragma unroll
                for (int e = 0; e < 4; e++) acc[e] = acc[e] * f + p * v4[u][e];
                m = mn;
            }
        }
    }
}
template <int U>
__global__ __launch_bounds__(1024) void attn_rows_kernel(AttnArgs a) {
    const int r = blockIdx.x, z = blockIdx.y, nz = gridDim.y, tid = threadIdx.x;
    const int h = tid >> 4, cl = tid & 15;
    const int T = (int) a.row_pos[r] + 1;
    const int64_t sb = (int64_t) a.row_seq[r] * a.seq_stride;
    const int chunk = (T + nz - 1) / nz;
    const int t0 = z * chunk, t1 = min(T, t0 + chunk);
    const int64_t hb = sb + tid * 4;                        // head h = tid / 16, channels 4 (tid % 16) .. + 3 of it: element tid * 4 of the row
    const float4v q4 = *(const float4v *) (a.q + (int64_t) r * a.H + tid * 4);
    float m = -INFINITY, l = 0.0f;
    float4v acc = {0.f, 0.f, 0.f, 0.f};
    if (t0 < t1) {
        if (a.kv_f16) attn_row_pass<true, U>(a, hb, q4, t0, t1, m, l, acc);
        else attn_row_pass<false, U>(a, hb, q4, t0, t1, m, l, acc);
    }
    if (nz == 1) {
        const float inv = 1.0f / l;
        if (a.out16) {
            half4 o;
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = (_Float16) (acc[e] * inv);
            *(half4 *) (a.out16 + (int64_t) r * a.H + tid * 4) = o;
        } else {
            float4v o;
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = acc[e] * inv;
            if (a.out_q) {   // eight neighbouring threads hold one block of 32 channels
                float dd;
                const unsigned qq = quant4_q8(o, dd);
                *(unsigned *) (a.out_q + (int64_t) r * a.H + tid * 4) = qq;
                if ((tid & 7) == 0) a.out_dT[(int64_t) (tid >> 3) * a.ldr + r] = dd;
            } else {
                *(float4v *) (a.out + (int64_t) r * a.H + tid * 4) = o;
            }
        }
    } else {   // read by attn_combine_kernel: max (-inf: an empty slice), sum, unnormalised out[64] of this slice
        float *p = a.part + (((int64_t) r * a.n_heads + h) * nz + z) * ATT_PS;
        if (cl == 0) { p[0] = m; p[1] = l; }
        *(float4v *) (p + ATT_PO + cl * 4) = acc;
    }
}

// Cross-attention over a short voice prompt (T_fixed <= 32 encoder positions; Parler-Mini: 8..40): the general kernel
// spends its time in workgroup merges and idle key groups there (18 us per layer at 384 rows).  One wave per (row, head):
// lane = channel of the 64-wide head, every K / V row is one coalesced 256-B load, the score of a key is a wave reduction,
// softmax runs redundantly in every lane (no LDS, no barrier).  Same mathematics as soft_max_ext + mul_mat (model.cpp:586-593).
// (Keeping a head's K / V in registers for 8 consecutive rows of a 1024-row forward — an eighth of the L2 reads — measured slower, 18.3 -> 22.6 us per
// launch: a row is ~2 us of dependent cross-lane reductions and wants its own wave, profiles/r03/attn_short_rows_rejected.txt.)
// TN = the prompt length rounded up to 8 / 16 / 32 (compile time): the K / V rows and the scores live in registers, TN loads per array
template <int TN>
__device__ __forceinline__ void attn_short_body(const AttnArgs &a, int r, int h, int lane, int T) {
    const int64_t hb = h * 64 + lane;
    const float qv = a.q[(int64_t) r * a.H + hb];
    float kv[TN], vv[TN], sc[TN];
    const float *kc = (const float *) a.kc, *vc = (const float *) a.vc;   // the cross K / V are fp32 (compute_cross_kv); the launcher refuses kv_f16
#pragma unroll
    for (int t = 0; t < TN; t++) {   // K and V rows requested together, straight-line: one round trip (a position beyond T re-reads the last row)
        const int64_t off = hb + (int64_t) min(t, T - 1) * a.H;
        kv[t] = kc[off];
        vv[t] = vc[off];
    }
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < TN; t++)
        if (t < T) { sc[t] = wave_sum(qv * kv[t]) * a.scale; m = fmaxf(m, sc[t]); }
    float l = 0.0f, o = 0.0f;
#pragma unroll
    for (int t = 0; t < TN; t++)
        if (t < T) { const float p = expf(sc[t] - m); l += p; o += p * vv[t]; }
    const float res = o / l;
    if (a.out16) a.out16[(int64_t) r * a.H + hb] = (_Float16) res;
    else a.out[(int64_t) r * a.H + hb] = res;
}
static __global__ __launch_bounds__(256) void attn_short_kernel(AttnArgs a) {
    const int lane = threadIdx.x & 63;
    const int h = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), r = blockIdx.y;
    if (h >= a.n_heads) return;
    B1_STAMP(a.stamps, 0);
    const int T = a.T_fixed;
    if (T <= 8) attn_short_body<8>(a, r, h, lane, T);
    else if (T <= 16) attn_short_body<16>(a, r, h, lane, T);
    else attn_short_body<32>(a, r, h, lane, T);
    B1_STAMP(a.stamps, 3);
}

// (A second lane layout — key = lane / 4, a quarter of the head's channels per lane: all scores from 16 FMAs and two quad steps, 14 cross-lane steps
// per row instead of six per key — measured 14.8 us per launch at 1024 rows against 13.7 for this kernel with its compile-time prompt bound:
// the launch is bound by the latency of its loads, not by the reductions; profiles/r03/attn_short16_rejected.txt.)

static __global__ void attn_combine_kernel(const float *part, int nz, int H, int n_heads, float *out, _Float16 *out16, int8_t *out_q = nullptr, float *out_dT = nullptr, int ldr = 0) {
    const int h = blockIdx.x, r = blockIdx.y, c = threadIdx.x;  // 64 threads; nz <= 16
    const float *p = part + ((int64_t) r * n_heads + h) * nz * ATT_PS;
    float m[16], s[16], o[16];
#pragma unroll
    for (int z = 0; z < 16; z++) {  // every load issued before any use
        if (z < nz) { m[z] = p[z * ATT_PS]; s[z] = p[z * ATT_PS + 1]; o[z] = p[z * ATT_PS + ATT_PO + c]; }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int z = 0; z < 16; z++) if (z < nz) mx = fmaxf(mx, m[z]);
    float oo = 0.0f, ss = 0.0f;
#pragma unroll
    for (int z = 0; z < 16; z++) {
        if (z < nz && m[z] != -INFINITY) {  // -inf marks an empty chunk
            const float f = expf(m[z] - mx);
            oo += f * o[z];
            ss += f * s[z];
        }
    }
    const float res = oo / ss;
    if (out_q) {   // a half-wave holds one block of 32 channels
        const float amax = lanes32_max(fabsf(res));
        const float dd = amax / 127.0f;
        const float id = dd ? 1.0f / dd : 0.0f;
        out_q[(int64_t) r * H + h * 64 + c] = (int8_t) roundf(res * id);
        if ((c & 31) == 0) out_dT[(int64_t) (h * 2 + (c >> 5)) * ldr + r] = (float) (_Float16) dd;
    } else if (out16) out16[(int64_t) r * H + h * 64 + c] = (_Float16) res;
    else out[(int64_t) r * H + h * 64 + c] = res;
}

// ------------------------------------------------------------------------------------------------
// sampler::max on the device (src/sampler.cpp:185-204): first maximum wins (v > max).
// ------------------------------------------------------------------------------------------------
static __global__ void argmax_kernel(const float *logits, int V, uint32_t *tokens) {
    __shared__ float bv[4];
    __shared__ uint32_t bi[4];
    const int idx = blockIdx.x;  // r * n_out + head
    const float *lg = logits + (int64_t) idx * V;
    float best = -INFINITY;
    uint32_t besti = 0;
    for (int i = threadIdx.x; i < V; i += blockDim.x) {
        const float v = lg[i];
        if (v > best) { best = v; besti = (uint32_t) i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const uint32_t oi = __shfl_xor(besti, o);
        if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { bv[w] = best; bi[w] = besti; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < (int) (blockDim.x >> 6); i++)
            if (bv[i] > best || (bv[i] == best && bi[i] < besti)) { best = bv[i]; besti = bi[i]; }
        tokens[idx] = besti;
    }
}

// ------------------------------------------------------------------------------------------------
// sampler::sample on the device (src/sampler.cpp:3-69 with softmax :82-116, topk :152-183, topp :118-150) for
// repetition_penalty == 1: one workgroup per (row, head).  Everything whose result depends on the order of fp32
// operations follows the reference's order: the softmax denominator and the inverse-CDF scan are sequential sums
// over the candidates in candidate order (one thread), probabilities are exp(v/T - top)/total with the same
// operation sequence; only the embarrassingly parallel parts (exp, the candidate sort) use the whole workgroup.
// Candidate order = descending value; equal values are ordered by index (the reference's std::sort leaves the
// order of equal keys unspecified).  The uniform draws come from the host (std::minstd_rand, sampler.cpp:47-48).
// ------------------------------------------------------------------------------------------------
#define SMP_VMAX 2048
struct SampleArgs {
    const float *logits;       // [R][n_out][V]
    int V, n_out, R;
    uint32_t top_k;
    float top_p, temperature;
    const float *uniforms;     // [calls][R][n_out]
    const uint32_t *row_step;  // [R] 1-based index of the sampler call this step is (NULL: call 1)
    uint32_t *out;             // [R][n_out]
    // repetition penalty (sampler.cpp:89-90,99-100,172-175,194-195: v /= pow(penalty, count) for the token sampled last,
    // in double): pen_table[c] = pow(penalty, c) evaluated on the host, NULL when the penalty is 1
    const double *pen_table;
    int pen_len;
    int32_t *last_ids;         // [R][n_out] sampler::last_token_ids (-1 after reset)
    uint32_t *rep_counts;      // [R][n_out] sampler::repetition_counts
    // generation loop with row compaction: row r of this forward is utterance orig[r] of R_total; the uniforms and the sampler state stay
    // indexed by utterance (NULL: row == utterance, R_total == R)
    const uint32_t *orig;
    int R_total;
};

// Candidate order = descending value, equal values by ascending index: a total order, so a bitonic sort of the
// 64-bit keys (~monotone(value) << 32 | index) over the vocabulary padded to a power of two gives it directly.
__device__ __forceinline__ unsigned long long smp_key(float v, int i) {
    unsigned u = __float_as_uint(v == 0.0f ? 0.0f : v);  // -0 == +0
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);      // monotone: larger float -> larger unsigned
    return ((unsigned long long) (~u) << 32) | (unsigned) i;  // ascending key == descending value, ascending index
}

// keys[0..P) sorted ascending (P = power of two >= V, padding keys = ~0); then picks[rank] = index for rank < k
__device__ __forceinline__ void smp_rank_select(const float *val, int V, int k, unsigned short *picks, unsigned long long *keys) {
    int P = 1;
    while (P < V) P <<= 1;
    for (int i = threadIdx.x; i < P; i += blockDim.x) keys[i] = i < V ? smp_key(val[i], i) : ~0ull;
    __syncthreads();
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
    for (int j = threadIdx.x; j < k; j += blockDim.x) picks[j] = (unsigned short) (keys[j] & 0xFFFFu);
}

static __global__ __launch_bounds__(256) void sample_kernel(SampleArgs a) {
    __shared__ float val[SMP_VMAX];
    __shared__ float tmp[SMP_VMAX];
    __shared__ unsigned short picks[SMP_VMAX];
    __shared__ unsigned long long keys[SMP_VMAX];
    __shared__ float bv[4];
    __shared__ uint32_t bi[4];
    __shared__ float s_top, s_total, s_mhp;
    __shared__ int s_n;
    const int h = blockIdx.x, r = blockIdx.y, tid = threadIdx.x, V = a.V;
    const float *row = a.logits + ((int64_t) r * a.n_out + h) * V;
    const bool temp = a.temperature != 1.0f;
    const bool use_topk = a.top_k > 0 && a.top_k < (uint32_t) V;
    const bool use_topp = a.top_p < 1.0f;

    // the token sampled last enters every comparison and the softmax with its penalised value
    const int ro = a.orig ? (int) a.orig[r] : r;   // utterance of this row
    const int last = a.pen_table ? a.last_ids[ro * a.n_out + h] : -1;
    float pen_v = 0.0f;
    if (last >= 0 && last < V) {
        const uint32_t cnt = a.rep_counts[ro * a.n_out + h];
        pen_v = (float) ((double) row[last] / a.pen_table[cnt < (uint32_t) a.pen_len ? cnt : (uint32_t) a.pen_len - 1]);
    }
    // sampler::max (first maximum wins)
    float best = -INFINITY;
    uint32_t besti = 0;
    for (int i = tid; i < V; i += 256) {
        const float v = i == last ? pen_v : row[i];
        val[i] = v;
        if (v > best) { best = v; besti = (uint32_t) i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const uint32_t oi = __shfl_xor(besti, o);
        if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
    }
    if ((tid & 63) == 0) { bv[tid >> 6] = best; bi[tid >> 6] = besti; }
    __syncthreads();
    if (tid == 0) {
        for (int i = 1; i < 4; i++)
            if (bv[i] > best || (bv[i] == best && bi[i] < besti)) { best = bv[i]; besti = bi[i]; }
        float top = val[besti];
        if (temp) top /= a.temperature;
        s_top = top;
    }
    __syncthreads();
    const float top = s_top;
    int n = V;
    bool nucleus = false;

    if (use_topp) {  // softmax over the whole vocabulary first (sampler.cpp:20-23)
        for (int i = tid; i < V; i += 256) {
            float v = val[i];
            if (temp) v /= a.temperature;
            tmp[i] = expf(v - top);
        }
        __syncthreads();
        if (tid == 0) {
            float total = 0.0f;
            for (int i = 0; i < V; i++) total += tmp[i];
            s_total = total;
        }
        __syncthreads();
        for (int i = tid; i < V; i += 256) val[i] = tmp[i] / s_total;
        __syncthreads();
    }
    if (use_topk) {  // on logits, or on probabilities when the softmax already ran
        smp_rank_select(val, V, (int) a.top_k, picks, keys);
        n = (int) a.top_k;
        nucleus = true;
        __syncthreads();
    }
    if (!use_topp) {  // softmax over the candidates, in candidate order
        for (int j = tid; j < n; j += 256) {
            float v = val[nucleus ? picks[j] : j];
            if (temp) v /= a.temperature;
            tmp[j] = expf(v - top);
        }
        __syncthreads();
        if (tid == 0) {
            float total = 0.0f;
            for (int j = 0; j < n; j++) total += tmp[j];
            s_total = total;
        }
        __syncthreads();
        for (int j = tid; j < n; j += 256) val[nucleus ? picks[j] : j] = tmp[j] / s_total;
        __syncthreads();
    } else {
        if (!nucleus) {  // topp sorts the whole vocabulary by probability (sampler.cpp:119-131)
            smp_rank_select(val, V, V, picks, keys);
            nucleus = true;
            __syncthreads();
        }
        if (tid == 0) {
            float mass = 0.0f;
            int keep = -1;
            for (int j = 0; j < n; j++) {
                mass += val[picks[j]];
                if (mass >= a.top_p) { keep = j + 1; break; }
            }
            s_mhp = fminf(mass, a.top_p);
            s_n = keep > 0 ? keep : n;
        }
        __syncthreads();
        n = s_n;
    }
    if (tid == 0) {
        const uint32_t call = a.row_step ? a.row_step[r] - 1 : 0;
        const float u = a.uniforms[((int64_t) call * (a.orig ? a.R_total : a.R) + ro) * a.n_out + h];
        const float target = use_topp ? u * s_mhp : u;
        float cum = 0.0f;
        int chosen = n ? (nucleus ? (int) picks[n - 1] : n - 1) : 0;
        for (int j = 0; j < n; j++) {
            const int i = nucleus ? (int) picks[j] : j;
            cum += val[i];
            if (target <= cum || j + 1 >= n) { chosen = i; break; }
        }
        a.out[r * a.n_out + h] = (uint32_t) chosen;
        if (a.pen_table) {  // sampler.cpp:57-63
            uint32_t cnt = a.rep_counts[ro * a.n_out + h];
            if (last != chosen) cnt = 0;
            a.last_ids[ro * a.n_out + h] = chosen;
            a.rep_counts[ro * a.n_out + h] = cnt + 1;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// device-resident AR loop bookkeeping for greedy generation:
//   record the sampled tokens, update eos_seen (check_stopping, model.cpp:715-732), build the next
//   input ids with the delay pattern (model.cpp:778-785) and advance positions.
// ------------------------------------------------------------------------------------------------
struct FeedArgs {
    const uint32_t *tokens;   // [R][n_out] argmax of this step
    uint32_t *ids;            // [R][n_out] next step's input ids (in/out)
    uint32_t *row_pos;        // [R]
    uint32_t *row_step;       // [R] current_step of the step that just ran (>= 1), incremented here
    uint8_t  *eos_seen;       // [R][n_out]
    uint32_t *steps_done;     // [R] number of audio steps after which check_stopping() would return true (0 = not yet)
    uint32_t *tokens_out;     // [n_steps][R][n_out]
    int R, n_out;
    uint32_t bos, eos;
    uint32_t max_pos;         // a row whose next position would reach this has finished (check_stopping's current_position >= max_generation_size,
                              // model.cpp:720-722, with max_generation = the cached positions of a lock-step context); it idles on its last position
    // row compaction (generate_loop drops finished utterances from the lock-step forward): row r is utterance orig[r] of R_total; eos_seen,
    // steps_done and tokens_out stay indexed by utterance, ids / row_pos / row_step by row (NULL: row == utterance)
    const uint32_t *orig;
    int R_total;
    uint32_t max_steps;       // continuous batching: an utterance's step budget = planes of tokens_out (0: the loop's own n_steps bounds it); at
                              // max_steps the utterance is marked finished and records nothing further
    int dummy;                // continuous batching (tts_hip_parler_stream_*): rows of this cache slot are padding of the lock-step forward and record nothing (-1: none)
};

// rows dst[r] = src[map[r]] of the per-row loop state (ids, position, cache slot, step counter): the compaction of generate_loop, in two
// launches through a scratch copy (rows move towards lower indices while other workgroups still read them)
struct GatherArgs {
    const uint32_t *map;      // [R2] old row of new row r
    uint32_t *ids, *pos, *seq, *step;          // live arrays
    uint32_t *s_ids, *s_pos, *s_seq, *s_step;  // scratch
    int R2, n_out;
};
static __global__ void gather_rows_kernel(GatherArgs a, int phase) {
    const int r = blockIdx.x, t = threadIdx.x;
    if (r >= a.R2) return;
    if (phase == 0) {
        const int o = (int) a.map[r];
        if (t < a.n_out) a.s_ids[r * a.n_out + t] = a.ids[o * a.n_out + t];
        if (t == 0) { a.s_pos[r] = a.pos[o]; a.s_seq[r] = a.seq[o]; a.s_step[r] = a.step[o]; }
    } else {
        if (t < a.n_out) a.ids[r * a.n_out + t] = a.s_ids[r * a.n_out + t];
        if (t == 0) { a.pos[r] = a.s_pos[r]; a.seq[r] = a.s_seq[r]; a.step[r] = a.s_step[r]; }
    }
}

// one 64-thread workgroup per row
static __global__ void feed_kernel(FeedArgs a) {
    __shared__ int not_seen;
    const int r = blockIdx.x, hd = threadIdx.x;
    const int ro = a.orig ? (int) a.orig[r] : r, RT = a.orig ? a.R_total : a.R;
    if (ro == a.dummy) return;   // workgroup-uniform
    const uint32_t step = a.row_step[r];
    if (a.max_steps && step > a.max_steps) return;   // budget spent (reported at step == max_steps below); workgroup-uniform
    if (hd == 0) not_seen = 0;
    __syncthreads();
    if (hd < a.n_out) {
        const int idx = r * a.n_out + hd, io = ro * a.n_out + hd;
        const uint32_t tok = a.tokens[idx];
        a.tokens_out[((int64_t) (step - 1) * RT + ro) * a.n_out + hd] = tok;
        const uint8_t seen = a.eos_seen[io] | (tok == a.eos ? 1 : 0);
        a.eos_seen[io] = seen;
        a.ids[idx] = ((int) step > hd) ? (seen ? a.eos : tok) : a.bos;
        if (!seen) atomicOr(&not_seen, 1);
    }
    __syncthreads();
    if (hd == 0) {
        const uint32_t np = a.row_pos[r] + 1;
        if (np < a.max_pos) a.row_pos[r] = np;
        a.row_step[r] = step + 1;
        if ((!not_seen || np >= a.max_pos || (a.max_steps && step >= a.max_steps)) && a.steps_done[ro] == 0) a.steps_done[ro] = step;
    }
}

// ================================================================================================
// Quantised weights (GGUF Q4_0 / Q5_0 / Q8_0) with ggml's CPU semantics (upstream ggml knowledge, SURVEY.md
// A.3): ggml_compute_forward_mul_mat converts the activation row to Q8_0 blocks (d = max|x|/127 per 32
// values, q = roundf(x/d), d kept as fp16) and each 32-wide block contributes
//       (fp16 d_w * fp16 d_a) * sum_j q_w[j] * q_a[j]        with the integer dot exact.
// On the device the block integers are held as int8 (Q4_0: nibble-8, Q5_0: 5-bit value-16, Q8_0: as stored;
// expanded once at upload, the scales stay fp16), a 32-wide block is exactly one v_mfma_i32_16x16x32_i8, and
// the per-block scaling/accumulation runs in fp32 like ggml_vec_dot_q*_q8_0.
// ================================================================================================
typedef int int4v __attribute__((ext_vector_type(4)));

// activation rows -> Q8_0 blocks: q int8 [R][K], d (fp16-rounded, stored as float) [R][K/32]
static __global__ void quant_rows_q8_kernel(const float *x, int lda, int K, int8_t *q, float *d, int R) {
    const int r = blockIdx.y;
    const int b = blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5);  // 32 threads per block of 32 values
    const int j = threadIdx.x & 31;
    if (r >= R || b >= K / 32) return;
    const float v = x[(int64_t) r * lda + b * 32 + j];
    float amax = fabsf(v);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    const float dd = amax / 127.0f;
    const float id = dd ? 1.0f / dd : 0.0f;
    q[(int64_t) r * K + b * 32 + j] = (int8_t) roundf(v * id);
    if (j == 0) d[(int64_t) r * (K / 32) + b] = (float) (_Float16) dd;
}

struct QGemmArgs {
    GemmArgs g;             // K, N, R, epilogue fields (out/ldo/q/kc/vc/...); g.W = int8 weights [N][K]
    const _Float16 *wd;     // weight block scales [N][K/32]
    const int8_t *aq;       // quantised activations [R][K]
    const float *ad;        // activation block scales [R][K/32]
};

// 4 consecutive values of a Q8_0 block held by each of 8 neighbouring lanes -> the lane's 4 int8 (packed) and the
// block scale; same arithmetic as quant_rows_q8_kernel / ggml's quantize_row_q8_0_ref.
__device__ __forceinline__ unsigned quant4_q8(float4v y, float &d_out) {
    float amax = fmaxf(fmaxf(fabsf(y[0]), fabsf(y[1])), fmaxf(fabsf(y[2]), fabsf(y[3])));
    amax = fmaxf(amax, __shfl_xor(amax, 1));
    amax = fmaxf(amax, __shfl_xor(amax, 2));
    amax = fmaxf(amax, __shfl_xor(amax, 4));
    const float dd = amax / 127.0f;
    const float id = dd ? 1.0f / dd : 0.0f;
    unsigned q = 0;
#pragma unroll
    for (int e = 0; e < 4; e++) q |= ((unsigned) (int) roundf(y[e] * id) & 0xFFu) << (8 * e);
    d_out = (float) (_Float16) dd;
    return q;
}

// QPRO 0: activations pre-quantised by quant_rows_q8_kernel (qa.aq / qa.ad).
// QPRO 1: (R <= 16) the workgroup quantises the fp32 rows itself: every wave its own 256-column slice of each row,
//         4 values per lane, into LDS (int8 + block scales); the MFMA operands are then read from LDS.
// QPRO 2: (R <= 8, no split-K, K <= 2048) LayerNorm first: wave r % nw holds row r in registers (as gemm16_kernel's
//         PRO_LN), normalises and quantises it into LDS.
// split-K like gemm16_kernel: blockIdx.y owns `kchunk` columns, EPI_STORE writes slab blockIdx.y.
template <int EPI, int RB, int QPRO>
__global__ __launch_bounds__(QPRO == 2 ? 512 : 1024) void qgemm16_kernel(QGemmArgs qa) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const GemmArgs &a = qa.g;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6;
    const int n0 = blockIdx.x * 16;
    const int li = lane & 15, g = lane >> 4;
    const int K = a.K, nb = K / 32;
    const int k0 = a.kchunk ? blockIdx.y * a.kchunk : 0;
    const int kb = k0 + w * 256 + g * 8;  // this lane's 8 consecutive k inside each 32-wide block
    const int b0 = (k0 >> 5) + w * 8;     // first of this wave's 8 blocks

    // weights: 8 blocks x 8 int8 (one 64-bit load each); scales: 8 fp16 per owned output row
    long wq[8];
    const int8_t *wp = (const int8_t *) a.W + (int64_t) (n0 + li) * K + kb;
#pragma unroll
    for (int c = 0; c < 8; c++) wq[c] = __builtin_nontemporal_load((const long *) (wp + c * 32));
    half8 wds[4];
#pragma unroll
    for (int e = 0; e < 4; e++) wds[e] = *(const half8 *) (qa.wd + (int64_t) (n0 + g * 4 + e) * nb + b0);

    // LDS: qs int8 [R][Kc] | ds float [R][Kc/32] | red   (Kc = the K range of this workgroup)
    const int Kc = nw * 256;
    int8_t *qs = (int8_t *) smem;
    float *ds = (float *) (smem + (((size_t) a.R * Kc + 15) & ~(size_t) 15));
    size_t red_off = 0;
    if (QPRO >= 1) {
        red_off = (((size_t) a.R * Kc + 15) & ~(size_t) 15) + (((size_t) a.R * (Kc / 32) * 4 + 15) & ~(size_t) 15);
        const float *A = (const float *) a.A;
        if (QPRO == 2) {
            for (int r = w; r < a.R; r += nw) {
                const float *xr = A + (int64_t) r * a.lda;
                float4v v[8], lwv[8], lbv[8];
                float s = 0.0f;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int k = i * 256 + lane * 4;
                    if (k < K) {
                        v[i] = *(const float4v *) (xr + k);
                        lwv[i] = *(const float4v *) (a.ln_w + k);
                        lbv[i] = *(const float4v *) (a.ln_b + k);
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; i++)
                    if (i * 256 + lane * 4 < K) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
                const float mean = wave_sum(s) / (float) K;
                float s2 = 0.0f;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    if (i * 256 + lane * 4 < K) {
#pragma unroll
                        for (int e = 0; e < 4; e++) { const float d = v[i][e] - mean; s2 += d * d; }
                    }
                }
                const float rstd = 1.0f / sqrtf(wave_sum(s2) / (float) K + LN_EPS);
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int k = i * 256 + lane * 4;
                    if (k < K) {
                        float4v y;
#pragma unroll
                        for (int e = 0; e < 4; e++) y[e] = (v[i][e] - mean) * rstd * lwv[i][e] + lbv[i][e];
                        float dd;
                        const unsigned q = quant4_q8(y, dd);
                        *(unsigned *) (qs + (size_t) r * Kc + k) = q;
                        if ((lane & 7) == 0) ds[r * (Kc / 32) + (k >> 5)] = dd;
                    }
                }
            }
        } else {
            const int kk = w * 256 + lane * 4;  // inside this workgroup's K range
            for (int r0 = 0; r0 < a.R; r0 += 4) {
                float4v x[4];
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (r0 + j < a.R) x[j] = *(const float4v *) (A + (int64_t) (r0 + j) * a.lda + k0 + kk);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (r0 + j < a.R) {
                        float dd;
                        const unsigned q = quant4_q8(x[j], dd);
                        *(unsigned *) (qs + (size_t) (r0 + j) * Kc + kk) = q;
                        if ((lane & 7) == 0) ds[(r0 + j) * (Kc / 32) + (kk >> 5)] = dd;
                    }
                }
            }
        }
        __syncthreads();
    }

    for (int rg = 0; rg < a.R; rg += 16 * RB) {
        float4v acc[RB];
#pragma unroll
        for (int rb = 0; rb < RB; rb++) {
            acc[rb] = (float4v){0.f, 0.f, 0.f, 0.f};
            const int r = rg + rb * 16 + li;
            const int rr = r < a.R ? r : a.R - 1;
            if (QPRO >= 1) {
                const int8_t *ap = qs + (size_t) rr * Kc + w * 256 + g * 8;
                const float *adp = ds + rr * (Kc / 32) + w * 8;
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    const long bq = *(const long *) (ap + c * 32);
                    int4v z = {0, 0, 0, 0};
                    z = __builtin_amdgcn_mfma_i32_16x16x32_i8(wq[c], bq, z, 0, 0, 0);
                    const float da = adp[c];
#pragma unroll
                    for (int e = 0; e < 4; e++) acc[rb][e] += (float) z[e] * ((float) wds[e][c] * da);
                }
            } else {
                const int8_t *ap = qa.aq + (int64_t) rr * K + kb;
                const float *adp = qa.ad + (int64_t) rr * nb + b0;
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    const long bq = *(const long *) (ap + c * 32);
                    int4v z = {0, 0, 0, 0};
                    z = __builtin_amdgcn_mfma_i32_16x16x32_i8(wq[c], bq, z, 0, 0, 0);  // exact block dot
                    const float da = adp[c];
#pragma unroll
                    for (int e = 0; e < 4; e++) acc[rb][e] += (float) z[e] * ((float) wds[e][c] * da);
                }
            }
        }
        if (nw > 1) {
            float *red = (float *) (smem + red_off);  // [nw][RB][4][64]
            if (rg > 0) __syncthreads();
#pragma unroll
            for (int rb = 0; rb < RB; rb++)
#pragma unroll
                for (int e = 0; e < 4; e++) red[((w * RB + rb) * 4 + e) * 64 + lane] = acc[rb][e];
            __syncthreads();
#pragma unroll
            for (int rb = 0; rb < RB; rb++) {
                if ((rb % nw) != w) continue;
                float4v t;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float sum = red[((0 * RB + rb) * 4 + e) * 64 + lane];
                    for (int ww = 1; ww < nw; ww++) sum += red[((ww * RB + rb) * 4 + e) * 64 + lane];
                    t[e] = sum;
                }
                const int r = rg + rb * 16 + li;
                if (r < a.R) gemm_epilogue4(a, EPI, r, n0 + g * 4, t, blockIdx.y);
            }
        } else {
#pragma unroll
            for (int rb = 0; rb < RB; rb++) {
                const int r = rg + rb * 16 + li;
                if (r < a.R) gemm_epilogue4(a, EPI, r, n0 + g * 4, acc[rb], blockIdx.y);
            }
        }
    }
}

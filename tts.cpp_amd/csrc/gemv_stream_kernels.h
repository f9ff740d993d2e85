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

// NPI (PRO_SILU only): slabs of the gate | up rows an item requests (>= a.n_parts; 8 until round 5, when the launch got an instance per slab count:
// an item of 32 16-byte loads left one item per thread in front of the weights and the other three as dependent round trips)
template <int NWV, int PRO, int EPI, int NPI = 8>
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
    // ---- stage this slice of the R rows as fp16 ------------------------------------------------------
    // Order of the requests (round 5): vmcnt retires in issue order, so the staging inputs (L2 hits) go out FIRST and the first weight set right
    // behind them — with the weights in front (rounds 2-4) the staging could not touch its inputs before the weights had landed, one HBM round
    // trip at the top of every launch (profiles/r05/isa_wait_order_before.txt).  PRE items per thread are requested ahead of the weights (all of
    // them at Dia's and Parler's shapes for the plain prologues; one for the folding prologues, whose item is 80-128 registers), the rest behind.
    const int c8n = KS >> 3;
    const int total = RS * c8n;
    constexpr int NT = NWV * 64;
    constexpr int PRE = (PRO == PRO_F16 || PRO == PRO_F32) ? (NWV <= 4 ? 8 : 4) : PRO == PRO_SILU ? (NPI == 1 ? 4 : NPI == 2 ? 4 : NPI == 4 ? 2 : 1) : 1;
    constexpr int RAWQ = PRO == PRO_F16 ? 1 : PRO == PRO_SILU ? 4 * NPI : PRO == PRO_ATTN8 ? 1 : 2;   // what an item holds between its request and its use
    constexpr int RAWD = PRO == PRO_ATTN8 ? 5 * ATTN_FOLD_NZ : 1;
    struct Raw { float4v q[RAWQ]; float2v d[RAWD]; };
    auto load_item = [&](int i, Raw &rw) __attribute__((always_inline)) {
        const int r = min(i / c8n, a.R - 1), c8 = i % c8n;   // a row slot beyond R holds zeros: its loads re-read the last row and are dropped
        if (PRO == PRO_F16) {
            const half8 h = *(const half8 *) ((const _Float16 *) a.A + (int64_t) r * a.lda + k0 + c8 * 8);
            rw.q[0] = __builtin_bit_cast(float4v, h);
        } else if (PRO == PRO_ATTN8) {
            // K = heads x 128: eight values of one head from every key slice (max, sum | 8 values): 5 x float2v per slice
            const int k = k0 + c8 * 8;
            const float *p = a.att_part + ((int64_t) r * (a.K >> 7) + (k >> 7)) * ATTN_FOLD_NZ * ATTN_PART;
            const int t0 = k & 127;
#pragma unroll
            for (int z = 0; z < ATTN_FOLD_NZ; z++) {
                rw.d[z * 5] = *(const float2v *) (p + z * ATTN_PART);
#pragma unroll
                for (int j = 0; j < 4; j++) rw.d[z * 5 + 1 + j] = *(const float2v *) (p + z * ATTN_PART + 2 + t0 + 2 * j);
            }
        } else if (PRO == PRO_SILU) {
            // a.A = gate | up rows [R][2 K] as a.n_parts <= 8 slabs (a.parts_stride floats apart) of the preceding projection
            // (a slab beyond n_parts re-reads the last one and is never added)
            const float *pg = (const float *) a.A + (int64_t) r * a.lda + k0 + c8 * 8;
#pragma unroll
            for (int q = 0; q < NPI; q++) {
                const float *pq = pg + (int64_t) min(q, a.n_parts - 1) * a.parts_stride;
                rw.q[q * 4 + 0] = *(const float4v *) pq; rw.q[q * 4 + 1] = *(const float4v *) (pq + 4);
                rw.q[q * 4 + 2] = *(const float4v *) (pq + a.K); rw.q[q * 4 + 3] = *(const float4v *) (pq + a.K + 4);
            }
        } else {
            const float *p = (const float *) a.A + (int64_t) r * a.lda + k0 + c8 * 8;
            rw.q[0] = *(const float4v *) p; rw.q[1] = *(const float4v *) (p + 4);
        }
    };
    auto finish_item = [&](int i, const Raw &rw) __attribute__((always_inline)) {
        const int r = i / c8n, c8 = i - r * c8n;
        half8 h = {0, 0, 0, 0, 0, 0, 0, 0};
        if (r < a.R) {
            if (PRO == PRO_F16) {
                h = __builtin_bit_cast(half8, rw.q[0]);
            } else if (PRO == PRO_ATTN8) {
                // merged from the key slices (attn_gqa_combine_kernel: running max in slice order, o = sum f_z o_z, l = sum f_z l_z with
                // f_z = expf(m_z - m), o / l; an empty slice left (max = -inf, sum = 0) and adds + 0)
                const float2v (&d)[RAWD] = rw.d;
                float m = -INFINITY;
#pragma unroll
                for (int z = 0; z < ATTN_FOLD_NZ; z++) m = fmaxf(m, d[z * 5][0]);
                float o[8], l = 0.0f;
#pragma unroll
                for (int e = 0; e < 8; e++) o[e] = 0.0f;
#pragma unroll
                for (int z = 0; z < ATTN_FOLD_NZ; z++) {
                    const float mz = d[z * 5][0];
                    const bool live = mz != -INFINITY;
                    const float f = live ? expf(mz - m) : 0.0f;
#pragma unroll
                    for (int e = 0; e < 8; e++) o[e] += f * (live ? d[z * 5 + 1 + (e >> 1)][e & 1] : 0.0f);
                    l += f * d[z * 5][1];
                }
#pragma unroll
                for (int e = 0; e < 8; e++) h[e] = (_Float16) (o[e] / l);
            } else if (PRO == PRO_SILU) {
                // slabs added in slab order, silu(gate) * up — silu_mul_kernel's arithmetic
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    float x = rw.q[0 + (e >> 2)][e & 3], u = rw.q[2 + (e >> 2)][e & 3];
#pragma unroll
                    for (int q = 1; q < NPI; q++)
                        if (q < a.n_parts) { x += rw.q[q * 4 + (e >> 2)][e & 3]; u += rw.q[q * 4 + 2 + (e >> 2)][e & 3]; }
                    h[e] = (_Float16) ((x / (1.0f + expf(-x))) * u);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) { h[e] = (_Float16) rw.q[0][e]; h[4 + e] = (_Float16) rw.q[1][e]; }
            }
        }
        *(half8 *) (xs + (size_t) r * ldx + c8 * 8) = h;
    };
    Raw raw[PRE];
#pragma unroll
    for (int j = 0; j < PRE; j++) load_item(min(tid + j * NT, total - 1), raw[j]);
    __builtin_amdgcn_sched_barrier(0);
    // the first weight set.  Unconditional (a wave with no tile re-reads the last one): under `if (t < tiles)` the loads were a block of their
    // own whose first result was copied on the way out — an s_waitcnt on the first weight load in front of everything behind it
    loadw(w0, min(t, tiles - 1), 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < PRE; j++)
        if (tid + j * NT < total) finish_item(tid + j * NT, raw[j]);
    for (int i = tid + PRE * NT; i < total; i += NT) {
        load_item(i, raw[0]);
        finish_item(i, raw[0]);
    }
    __syncthreads();

    const _Float16 *xb = xs + (size_t) (li & (RS - 1)) * ldx + g * 8;
    float4v acc = {0.f, 0.f, 0.f, 0.f};
    // The pipeline: the loads of the pair after (t, ch) go out UNCONDITIONALLY in the block in front of the MFMAs of (t, ch) (round 6).  Until then they sat
    // under `if (t2 < tiles)`: at the join the compiler cannot know whether eight more loads are outstanding, so the MFMAs' waits counted down to
    // vmcnt(0) — the set just requested had to land before the current one was finished, one memory latency per chunk whatever was in flight
    // (profiles/r06/stream_loop_waits.txt).  The loop below decides first whether a next pair exists and only then issues + computes.
    // (all four chunks of a one-item wave requested before the staging — Dia's down projection — measured equal: profiles/r05/dia_step_kernels_call17_deep_rejected.txt;
    //  that launch is bound by its staging volume: every workgroup re-reads its slice of the gate | up slabs, 64 MB through L2 beside 33.5 MB of weights)
    auto compute = [&](half8 (&cur)[8]) __attribute__((always_inline)) {   // the MFMAs of (t, ch); the tile's epilogue behind its last chunk
        const _Float16 *xp = xb + ch * 256;
#pragma unroll
        for (int c = 0; c < 8; c++) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[c], *(const half8 *) (xp + c * 32), acc, 0, 0, 0);
        if (ch == nc - 1) {
            if (li < a.R) gemm_epilogue4(a, EPI, li, t * 16 + g * 4, acc, kz);
            acc = (float4v){0.f, 0.f, 0.f, 0.f};
        }
    };
    if (t < tiles) {   // (a wave without a tile requested a clamped set above and leaves)
        int t2, ch2;
        auto next = [&]() __attribute__((always_inline)) { t2 = t; ch2 = ch + 1; if (ch2 == nc) { ch2 = 0; t2 = t + tstep; } };
        for (;;) {     // the pending set is w0 = (t, ch) here
            next();
            if (t2 >= tiles) { compute(w0); break; }
            loadw(w1, t2, ch2);
            __builtin_amdgcn_sched_barrier(0);   // the requests first: the scheduler otherwise sinks them behind the first MFMAs, i.e. behind the wait for the current set
            compute(w0);
            t = t2; ch = ch2;
            next();
            if (t2 >= tiles) { compute(w1); break; }
            loadw(w0, t2, ch2);
            __builtin_amdgcn_sched_barrier(0);
            compute(w1);
            t = t2; ch = ch2;
        }
    }
}

// ================================================================================================================================
// qgemv_stream_kernel — the same schedule for GGUF-quantised matrices and 5 .. 16 RT rows (RT = 1, 2, 4 row tiles of 16; round 6: lock-step Orpheus utterances, orpheus/model.cpp:194-283
// for a handful of rows).  qgemm16_kernel is one-shot per 16 features like gemm16_kernel (K / 256 waves meet in LDS) and streamed the int8 expansion of
// the 3B matrices at ~0.15 of HBM at 8 rows (profiles/r06/orpheus_batch_call2.txt); the streaming Q4_0 kernels end at 4 rows.  Here:
//   * a workgroup is bound to one K slice and keeps the Q8_0 rows of that slice in LDS (codes int8 [RS][KS], 16-byte pieces XOR-swizzled by the row so
//     that the 16 rows of a fragment read land in 16 different bank groups; block scales float [RS][KS / 32]), taken from the producer's blocks
//     (aq [R][K], ad [R][K / 32]: rms norm, attention, silu * up write them) — no conversion here;
//   * a wave walks (16-feature tile, 256-column chunk) pairs: 4 x 16-byte code loads + 4 x 16-byte scale loads per lane and chunk — 64 contiguous bytes
//     per matrix row and instruction, as in gemv_stream_kernel (8-byte loads, 32 bytes per row: 2.6 TB/s at best, profiles/r06/qstream_bench_r8.txt) —
//     DEPTH - 1 pairs' loads in flight ahead of the MFMAs;
//   * 16 bytes per lane = v_mfma_i32_16x16x64_i8, whose 64 columns are TWO quantisation blocks (lane groups 0, 1 hold block 0, groups 2, 3 block 1):
//     two MFMAs per span, the activation operand of the other block's lane groups read from a row of zeros, give the two exact block dots;
//     then acc += (float) sumi * (d_w * d_a) — qgemm16_kernel's arithmetic (ggml_vec_dot_q*_q8_0); accumulation across the chunks of a tile in
//     registers, no cross-wave reduction;
//   * K slices write fp32 slabs (slab kz of `out`, slab_stride floats apart) that the consumer folds in slab order (llama_rope_kv_kernel, silu_mul_kernel,
//     rms_fold_rows_kernel), like gemv_stream_kernel.  N need not be a multiple of 16 (the LM head): a last tile re-reads feature N - 1 and stores nothing for it.
// ================================================================================================================================
// WL (experiment, profiles/qstream_bench.hip): the weight codes go through a per-wave LDS ring (global_load_lds, 256 contiguous bytes per matrix row and
// instruction instead of 64; XOR-swizzled by permuting the source pieces) and the fragments are read back with ds_read_b128.
template <int NWV, int DEPTH = 2, int RT = 1, bool WL = false>
__global__ __launch_bounds__(NWV * 64) void qgemv_stream_kernel(QGemmArgs qa, StreamMap sm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const GemmArgs &a = qa.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int kz = (int) blockIdx.x % sm.ks, wg_in = (int) blockIdx.x / sm.ks, nwg_in = (int) gridDim.x / sm.ks;
    const int KS = sm.kslice, k0 = kz * KS, nb = a.K >> 5, nbs = KS >> 5;
    const int RS = RT > 1 ? 16 * RT : (a.R <= 8 ? 8 : 16);   // rows kept in LDS (RT row tiles of 16); B-operand columns >= RS repeat rows (never stored)
    int8_t *xs = (int8_t *) smem;                    // [RS + 1][KS]: row RS is zeros
    float *sd = (float *) (smem + (size_t) (RS + 1) * KS);

    const int tiles = (a.N + 15) >> 4, nc = KS >> 8;
    const int tstep = nwg_in * NWV;
    const int8_t *wbase = (const int8_t *) a.W + k0 + g * 16;
    struct WSet { int4v w[WL ? 1 : 4]; half8 d[4]; };
    char *const ring = smem + ((((size_t) (RS + 1) * KS + (size_t) RS * nbs * 4) + 15) & ~(size_t) 15) + (size_t) wave * DEPTH * 4096;   // WL: DEPTH x [16 rows][256 B] per wave
    auto loadw = [&](WSet &s, int slot, int tile, int chunk) __attribute__((always_inline)) {
        if constexpr (WL) {
#pragma unroll
            for (int j = 0; j < 4; j++) {   // rows 4 j .. 4 j + 3 of the tile: lane -> (row, slot of the row's 256 bytes), which must hold source piece slot ^ (row & 15)
                const int r = 4 * j + (lane >> 4), piece = (lane & 15) ^ (r & 15);
                const int8_t *p = (const int8_t *) a.W + (int64_t) min(tile * 16 + r, a.N - 1) * a.K + k0 + chunk * 256 + piece * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) p, (__attribute__((address_space(3))) void *) (ring + slot * 4096 + j * 1024), 16, 0, 0);
            }
        } else {
            const int8_t *p = wbase + (int64_t) min(tile * 16 + li, a.N - 1) * a.K + chunk * 256;
#pragma unroll
            for (int c = 0; c < 4; c++) s.w[c] = __builtin_nontemporal_load((const int4v *) (p + c * 64));
        }
#pragma unroll
        // plain loads: a feature's scales of consecutive chunks share a 128-byte line; non-temporal 16-byte pieces fetched 64 bytes from HBM per chunk each
        // (FETCH_SIZE 618 MB for 512 MB on the LM head, profiles/r06/qstream_fetch_size.txt)
        for (int e = 0; e < 4; e++) s.d[e] = *(const half8 *) (qa.wd + (int64_t) min(tile * 16 + g * 4 + e, a.N - 1) * nb + (k0 >> 5) + chunk * 8);
    };
    // ---- stage this slice of the Q8_0 rows: the staging requests first, the first weight sets right behind them (vmcnt retires in issue order) ----
    constexpr int NTH = NWV * 64;
    const int cvec = KS >> 4, ctot = RS * cvec, stot = RS * nbs;   // 16-byte code vectors, block scales
    constexpr int PRE = NWV <= 4 ? 4 : 2;
    int4v craw[PRE];
    float sraw[PRE];
#pragma unroll
    for (int j = 0; j < PRE; j++) {
        const int i = min(tid + j * NTH, ctot - 1), r = min(i / cvec, a.R - 1), cv = i % cvec;
        craw[j] = *(const int4v *) (qa.aq + (int64_t) r * a.K + k0 + cv * 16);
        const int is = min(tid + j * NTH, stot - 1), rs = min(is / nbs, a.R - 1), bs = is % nbs;
        sraw[j] = qa.ad[(int64_t) rs * nb + (k0 >> 5) + bs];
    }
    __builtin_amdgcn_sched_barrier(0);
    // the wave's (tile, chunk) pairs in order; lt / lch: the pair whose loads go out next, t / ch: the pair computed next
    int t = wg_in * NWV + wave, ch = 0, lt = t, lch = 0;
    auto advance = [&](int &tt, int &cc) __attribute__((always_inline)) { if (++cc == nc) { cc = 0; tt += tstep; } };
    WSet w[DEPTH];
    int inflight = 0;   // WL: sets requested and not yet consumed (8 vector-memory operations each: the wait before a set's LDS reads counts them)
#pragma unroll
    for (int d = 0; d < DEPTH - 1; d++) {   // DEPTH - 1 sets in flight before the first MFMA
        if (d == 0 || lt < tiles) { loadw(w[d], d, min(lt, tiles - 1), lch); inflight++; }
        advance(lt, lch);
    }
    __builtin_amdgcn_sched_barrier(0);
    // piece cv of row r sits at piece cv ^ (r & 15) of the row's 256-byte group
#pragma unroll
    for (int j = 0; j < PRE; j++) {
        const int i = tid + j * NTH;
        if (i < ctot) *(int4v *) (xs + (size_t) (i / cvec) * KS + (((i % cvec) ^ ((i / cvec) & 15)) << 4)) = craw[j];
        if (i < stot) sd[i] = sraw[j];
    }
    for (int i = tid + PRE * NTH; i < ctot; i += NTH) {
        const int r = min(i / cvec, a.R - 1), cv = i % cvec;
        *(int4v *) (xs + (size_t) (i / cvec) * KS + ((cv ^ ((i / cvec) & 15)) << 4)) = *(const int4v *) (qa.aq + (int64_t) r * a.K + k0 + cv * 16);
    }
    for (int i = tid + PRE * NTH; i < stot; i += NTH) {
        const int rs = min(i / nbs, a.R - 1), bs = i % nbs;
        sd[i] = qa.ad[(int64_t) rs * nb + (k0 >> 5) + bs];
    }
    for (int i = tid; i < cvec; i += NTH) *(int4v *) (xs + (size_t) RS * KS + i * 16) = (int4v){0, 0, 0, 0};
    __syncthreads();

    const int row = li & (RS - 1);   // row tile j: row + 16 j (same swizzle key: the key is the row's low four bits)
    // operand of the even block of a span (lane groups 0, 1 hold its columns) and of the odd block (groups 2, 3): the other groups read zeros
    const int8_t *x_even = xs + (size_t) (g < 2 ? row : RS) * KS, *x_odd = xs + (size_t) (g < 2 ? RS : row) * KS;
    const int tile_even = g < 2 ? 16 * KS : 0, tile_odd = g < 2 ? 0 : 16 * KS;   // the zero row stays where it is
    int off[4];
#pragma unroll
    for (int c = 0; c < 4; c++) off[c] = ((c * 4 + g) ^ (row & 15)) << 4;
    const float *sb = sd + row * nbs;
    float4v acc[RT];
#pragma unroll
    for (int j = 0; j < RT; j++) acc[j] = (float4v){0.f, 0.f, 0.f, 0.f};
    // issue: the loads of the next pair, UNCONDITIONALLY in the block that precedes the MFMAs of the current one.  Under `if (lt < tiles)` the compiler cannot
    // know at the join whether eight more loads are outstanding and waits with vmcnt(0) for the older set — i.e. for the set just requested as well: the
    // loop then pays the full memory latency per chunk and no prefetch depth changes anything (that was the state until call 73 of round 6).
    auto issue = [&](WSet &nxt, int nslot) __attribute__((always_inline)) { loadw(nxt, nslot, lt, lch); inflight++; advance(lt, lch); };
    auto compute = [&](WSet &cur, int cslot) __attribute__((always_inline)) {
        int4v wl[4];
        if constexpr (WL) {
            // the current set's DMA has landed once at most the later sets' operations are outstanding (vmcnt counts in issue order)
            if (inflight >= 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (inflight == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            inflight--;
#pragma unroll
            for (int c = 0; c < 4; c++) wl[c] = *(const int4v *) (ring + cslot * 4096 + li * 256 + (((c * 4 + g) ^ li) << 4));
        }
#pragma unroll
        for (int j = 0; j < RT; j++) {
            const float4v da0 = *(const float4v *) (sb + j * 16 * nbs + ch * 8), da1 = *(const float4v *) (sb + j * 16 * nbs + ch * 8 + 4);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int4v zero = {0, 0, 0, 0};
                const int4v wc = WL ? wl[c] : cur.w[WL ? 0 : c];
                const int4v ze = __builtin_amdgcn_mfma_i32_16x16x64_i8(wc, *(const int4v *) (x_even + j * tile_even + ch * 256 + off[c]), zero, 0, 0, 0);   // exact block dots
                const int4v zo = __builtin_amdgcn_mfma_i32_16x16x64_i8(wc, *(const int4v *) (x_odd + j * tile_odd + ch * 256 + off[c]), zero, 0, 0, 0);
                const float dae = c < 2 ? da0[2 * c] : da1[2 * c - 4], dao = c < 2 ? da0[2 * c + 1] : da1[2 * c - 3];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    acc[j][e] += (float) ze[e] * ((float) cur.d[e][2 * c] * dae);
                    acc[j][e] += (float) zo[e] * ((float) cur.d[e][2 * c + 1] * dao);
                }
            }
            if (RT > 2) __builtin_amdgcn_sched_barrier(0);   // four row tiles: without the fence the scheduler keeps the MFMA results of all of them alive (256 registers + 192 bytes of scratch)
        }
        if (ch == nc - 1) {
            const int n0 = t * 16 + g * 4;
#pragma unroll
            for (int j = 0; j < RT; j++) {
                float *o = a.out + (int64_t) kz * a.slab_stride + (int64_t) (li + 16 * j) * a.ldo + n0;
                if (li + 16 * j < a.R) {
                    if (n0 + 3 < a.N) *(float4v *) o = acc[j];
                    else
#pragma unroll
                        for (int e = 0; e < 4; e++) if (n0 + e < a.N) o[e] = acc[j][e];
                }
                acc[j] = (float4v){0.f, 0.f, 0.f, 0.f};
            }
        }
        advance(t, ch);
    };
    static_assert(DEPTH == 2, "the loop below is written for two register sets");
    if (t < tiles) {   // (a wave without a tile requested a clamped set above and leaves)
        for (;;) {     // the pending set is w[0] here
            if (lt >= tiles) { compute(w[0], 0); break; }
            issue(w[1], 1);
            __builtin_amdgcn_sched_barrier(0);
            compute(w[0], 0);
            if (lt >= tiles) { compute(w[1], 1); break; }
            issue(w[0], 0);
            __builtin_amdgcn_sched_barrier(0);
            compute(w[1], 1);
        }
    }
}

// shim_llama.hip — the Orpheus (Llama-3) decoder step and Dia, on the row-streaming GEMV kernels.
#include "shim_internal.h"

#include "parler_kernels.h"
#include "gemv_stream_kernels.h"
#include "t5_kernels.h"
#include "llama_kernels.h"
#include "dia_kernels.h"
#include "gemv_kernels.h"
#include "shim_decoder.h"

// Orpheus decoder (src/models/orpheus/model.cpp:186-325)
// ------------------------------------------------------------------------------------------------
extern "C" tts_hip_ctx *tts_hip_orpheus_create(int device, const tts_hip_orpheus_desc *ld) {
    if (!ld || ld->struct_size != sizeof(tts_hip_orpheus_desc)) { set_err("tts_hip_orpheus_create: bad desc (struct_size mismatch)"); return nullptr; }
    tts_hip_desc d{};
    d.struct_size = sizeof(d);
    d.hidden_size = ld->hidden_size; d.n_layers = ld->n_layers; d.n_attn_heads = ld->n_attn_heads; d.max_ctx_length = ld->n_ctx;
    d.max_seqs = 1;
    d.flags = (ld->flags & (TTS_HIP_FLAG_VALU_GEMM | TTS_HIP_FLAG_DEQUANT_Q)) | TTS_HIP_FLAG_NO_PARLER | TTS_HIP_FLAG_NO_DAC;
    tts_hip_ctx *c = tts_hip_create(device, &d);
    if (!c) return nullptr;
    c->has_llama = true;
    c->lm = *ld;
    if (c->lm.max_seqs == 0) c->lm.max_seqs = 1;
    if (c->lm.max_seqs > 64) { set_err("tts_hip_orpheus_create: max_seqs %u > 64", c->lm.max_seqs); tts_hip_destroy(c); return nullptr; }
    // measured on MI355X at the orpheus-3b Q4_0 shapes (profiles/r02/first_call_orpheus_*.log): 3.84 ms/step through the
    // lock-step workgroups, 2.98 with the streaming 1-4 row kernels, 2.84 reading the Q4_0 codes themselves, 2.80 with the
    // step captured in one hipGraph -> all three are the default here; TTS_HIP_GEMV_ROWS / _Q4_NATIVE / _LLAMA_GRAPH=0 turn them off
    if (!getenv("TTS_HIP_GEMV_ROWS")) c->gemv_rows = true;
    if (!getenv("TTS_HIP_Q4_NATIVE")) c->q4_native = true;
    if (!getenv("TTS_HIP_LLAMA_GRAPH")) c->llama_graph = true;
    if (c->lm.rope_base == 0.0f) c->lm.rope_base = 500000.0f;
    // lock-step utterances (5 .. 64 rows per step): the rows are quantised once (quant_rows_q8_kernel), not by every 16-feature workgroup again — at 8 rows
    // of width 3072 a workgroup read 98 KB of fp32 rows beside its 49 KB of weights (4.11 -> 3.35 ms per step, profiles/r06/orpheus_batch_call2.txt)
    if (c->lm.max_seqs > 4) c->q_fuse_max = 4;
    return c;
}

// attention of the Llama / Dia steps: one workgroup per (head, row), or — few rows, many keys — the keys split over `nz` workgroups
// plus a combine launch (attn_gqa_split_kernel).  max_keys bounds the LDS score buffer.
static int launch_attn_gqa(tts_hip_ctx *c, int NHq, int rows, int max_keys, const float *qkv, int ld, const uint32_t *pos, const float *kc, const float *vc, int NKV,
                           float scale, float *out, const uint32_t *kbeg, const uint32_t *kend, const uint32_t *row_seq, int64_t seq_stride, bool fixed_split, bool q_out = false,
                           QPre qp = QPre{}, int *deferred = nullptr, int n_ctx_keys = 0) {
    int nz = 1;
    if (deferred) *deferred = 0;
    if (c->attn_split_max > 1 && NHq * rows <= 256) {
        // a graph captured once replays for every position: the split count must not depend on the position then
        nz = fixed_split ? c->attn_split_max : std::min(c->attn_split_max, std::max(1, max_keys / 128));
        while (nz > 1 && (size_t) rows * NHq * nz > c->attn_part_cap) nz--;
    }
    if (nz <= 1) {
        hipLaunchKernelGGL(attn_gqa_kernel<128>, dim3(NHq, rows), dim3(256), (size_t) (128 + max_keys) * 4, c->stream, qkv, ld, pos, kc, vc, NHq, NKV, scale, out, kbeg, kend,
                           row_seq, seq_stride, qp, q_out ? c->aq : (int8_t *) nullptr, q_out ? c->ad : (float *) nullptr);
        HIPCHK(hipGetLastError());
        if (q_out) c->aq_src = out;
        return 0;
    }
    const int chunk = (max_keys + nz - 1) / nz;
    const bool off32 = (uint64_t) std::max(n_ctx_keys, 1) * (uint64_t) NKV * 128 * 4 < (1ull << 32);   // attn_gqa_wave_kernel addresses a sequence's rows with 32-bit byte offsets
    if (c->attn_wave && off32 && !kbeg && !kend && !row_seq && qp.n_parts == 1 && !qp.rope_pos && n_ctx_keys > 0) {
        // the captured one-row step (Orpheus): every key row requested at kernel start, online softmax per 16-lane group, one barrier (attn_gqa_wave_kernel)
        hipLaunchKernelGGL((attn_gqa_wave_kernel<128, 4>), dim3(NHq, rows, nz), dim3(256), 0, c->stream, qkv, ld, pos, kc, vc, NHq, NKV, scale, c->attn_part, n_ctx_keys);
    } else if (c->attn_wave && off32 && !kbeg && kend && n_ctx_keys > 0 && max_keys <= 16 * nz * 8 && qp.n_parts <= 8) {
        // Dia's cross-attention (keys end at kend[r], per-row sequences, the query as slabs to fold and rotate): the same form, 8 passes through 3 rolling register slots
        hipLaunchKernelGGL((attn_gqa_wave_kernel<128, 3, true>), dim3(NHq, rows, nz), dim3(256), 0, c->stream, qkv, ld, pos, kc, vc, NHq, NKV, scale, c->attn_part, n_ctx_keys,
                           kend, row_seq, seq_stride, qp);
    } else {
        hipLaunchKernelGGL(attn_gqa_split_kernel<128>, dim3(nz, rows, NHq), dim3(256), (size_t) (128 + chunk + 1) * 4, c->stream, qkv, ld, pos, kc, vc, NHq, NKV, scale, c->attn_part,
                           kbeg, kend, row_seq, seq_stride, qp);
    }
    HIPCHK(hipGetLastError());
    if (deferred && c->attn_fold && nz == ATTN_FOLD_NZ && !c->prof && !c->debug) {
        *deferred = nz;   // the caller's next projection merges the slices in its staging prologue (gemv_stream_kernel<.., PRO_ATTN8, ..>); `out` is not written
        return 0;
    }
    hipLaunchKernelGGL(attn_gqa_combine_kernel, dim3(NHq, rows), dim3(128), 0, c->stream, (const float *) c->attn_part, nz, NHq, out, q_out ? c->aq : (int8_t *) nullptr, q_out ? c->ad : (float *) nullptr);
    if (q_out) c->aq_src = out;
    HIPCHK(hipGetLastError());
    return 0;
}

static int llama_gemm(tts_hip_ctx *c, const W &w, const float *A, int lda, float *out, int ldo, int n, int epi, int ksplit = 1) {
    for (int r0 = 0; r0 < n; r0 += c->RMAX) {
        GemmArgs g{};
        g.R = std::min(c->RMAX, n - r0); g.H = c->H;
        g.A = A + (size_t) r0 * lda; g.lda = lda;
        g.out = out + (size_t) r0 * ldo; g.ldo = ldo;
        if (ksplit > 1) {  // slabs [ksplit][RMAX][ldo], folded into the residual stream by the next rms_fold_rows_kernel
            g.kchunk = (int) w.K / ksplit;
            g.slab_stride = (int64_t) c->RMAX * ldo;
        }
        CHK(run_gemm(c, TTS_HIP_K_GEMM_OTHER, w, g, PRO_F32, epi));
    }
    return 0;
}

// one call of orpheus_runner::decode: n rows (<= RMAX) at pos0..; leaves the final-normed last row's logits in l_logits
// ids == nullptr: one row whose token id and position are already in l_ids[0] / l_pos[0] (the device-resident greedy loop)
// attn_positions != 0: size the attention scratch for that many cached positions instead of pos0 + n (a captured step is replayed
// at every position)
// row_seq != NULL (lock-step utterances, tts_hip_orpheus_step_batch / _generate_batch): row r belongs to cache slot row_seq[r] (device array); ids / positions /
// slots are already in l_ids / l_pos / l_seq (ids == NULL, any n), attn_positions bounds the keys of any row; logits_row: where the last row's logits go
// (l_logits + logits_row * Vpad), or -1: the logits of EVERY row r to l_logits + r * Vpad.  The one-sequence kernels that append to "the" cache
// (gemv_q4_qkv_rope_kernel) stay out of it.
static int llama_forward(tts_hip_ctx *c, const uint32_t *ids, int n, uint32_t pos0, int attn_positions = 0, const uint32_t *row_seq = nullptr, int logits_row = 0) {
    const int H = c->H, F = c->F, NH = c->NH, NKV = (int) c->lm.n_kv_heads, HD = (int) c->lm.head_dim;
    const int QKV = (NH + 2 * NKV) * HD, NCTX = (int) c->lm.n_ctx;
    if (n < 1 || n > c->RMAX) return set_err("tts_hip_orpheus_decode: %d tokens per call outside 1..%d", n, c->RMAX);
    if (pos0 + (uint32_t) n > (uint32_t) NCTX) return set_err("tts_hip_orpheus_decode: positions up to %u exceed the %d cached positions", pos0 + n, NCTX);
    auto f32 = [&](size_t off) { return (const float *) (c->arena + off); };
    if (ids) {
        std::vector<uint32_t> hp((size_t) n);
        for (int i = 0; i < n; i++) {
            if (ids[i] >= (uint32_t) c->l_V) return set_err("tts_hip_orpheus_decode: token id %u >= vocabulary %d", ids[i], c->l_V);
            hp[(size_t) i] = pos0 + (uint32_t) i;
        }
        HIPCHK(hipMemcpyAsync(c->l_ids, ids, (size_t) n * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->l_pos, hp.data(), (size_t) n * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));  // hp is a local
    } else if (n != 1 && !row_seq) {
        return set_err("llama_forward: device-resident inputs carry one row");
    }
    const int64_t seq_stride = (int64_t) c->L * NCTX * c->l_kvH;   // the cache is [slot][layer][position][kv width]: slot 0 is the one-sequence context's cache
    hipLaunchKernelGGL(t5_embed_kernel, dim3(n), dim3(256), 0, c->stream, f32(c->l_embd), (const uint32_t *) c->l_ids, H, c->l_x);
    HIPCHK(hipGetLastError());
    const float theta_scale = powf(c->lm.rope_base, -2.0f / (float) HD);
    c->l_pending = 0;
    // the streaming integer GEMV of 1..4 rows takes its Q8_0 activation blocks from the producing kernel where there is one
    enum { QS_QKV = 1, QS_O = 2, QS_GU = 4, QS_DOWN = 8, QS_HEAD = 16 };   // bits of tune("q_stream")
    auto slices = [&](const W &w, int bit, int rows, int max_slabs) { return (c->q_stream & bit) ? qstream_slices(c, w, rows, max_slabs) : 0; };
    auto q_for = [&](const W &w, int rows, int bit) {
        if (slices(w, bit, rows, std::min(16, 8 * c->RMAX / qstream_slab_rows(rows)))) return true;   // 5 .. 16 rows: the weight-streaming integer GEMM takes the producer's blocks as well
        return c->gemv_rows && rows <= 4 && w.type == TTS_HIP_Q8I && !(c->d.flags & (TTS_HIP_FLAG_VALU_GEMM | TTS_HIP_FLAG_DEQUANT_Q)) && w.K % 32 == 0;
    };
    // 5 .. 16 rows on a quantised matrix: qgemv_stream_kernel — K slices as fp32 slabs the consumer folds.  Returns the slab count (0: the shape does not
    // qualify, the caller takes the MFMA workgroups), < 0 on error.  A: the fp32 rows whose Q8_0 blocks the producer left in aq / ad (quantised here otherwise).
    const int srows = qstream_slab_rows(n), smax = std::min(16, c->RMAX / srows), pmax = std::min(16, 8 * c->RMAX / srows);   // rows per slab of the streaming integer GEMM; slabs l_qkv / l_gu and l_parts hold
    auto qstream = [&](const W &w, int bit, const float *A, int K, float *out, int ldo, int64_t slab_stride, int max_slabs) -> int {
        const int ks = slices(w, bit, n, max_slabs);
        if (!ks) return 0;
        if (c->aq_src != A) {
            hipLaunchKernelGGL(quant_rows_q8_kernel, dim3((K / 32 + 7) / 8, n), dim3(256), 0, c->stream, A, K, K, c->aq, c->ad, n);
            if (hipGetLastError() != hipSuccess) { set_err("quant_rows_q8_kernel launch failed"); return -1; }
        }
        c->aq_src = nullptr;
        if (launch_qstream(c, TTS_HIP_K_GEMM_OTHER, w, n, out, ldo, slab_stride, ks) != 0) return -1;
        return ks;
    };
    auto rms = [&](size_t w_off, int rows, float *x, float *y, const W *next, int bit = 0) {
        const bool q = next && q_for(*next, rows, bit);
        hipLaunchKernelGGL(rms_fold_rows_kernel, dim3(rows), dim3(256), 0, c->stream, x, H, f32(w_off), y, rows, 1e-5f,
                           c->l_pending ? (const float *) c->l_parts : (const float *) nullptr, c->l_pending, c->l_pstride ? c->l_pstride : (int64_t) c->RMAX * H,
                           q ? c->aq : (int8_t *) nullptr, q ? c->ad : (float *) nullptr);
        c->l_pending = 0; c->l_pstride = 0;
        c->aq_src = q ? y : nullptr;
        return hipGetLastError() == hipSuccess ? 0 : set_err("rms_fold_rows_kernel launch failed");
    };
    for (int l = 0; l < c->L; l++) {
        const auto &y = c->l_layers[l];
        float *kc = c->l_kc + (size_t) l * NCTX * c->l_kvH, *vc = c->l_vc + (size_t) l * NCTX * c->l_kvH;
        const size_t qkv_lds = (size_t) n * H + (size_t) n * (H / 32) * 4;
        const bool qkv_fused = !row_seq && n <= 4 && c->q4_rope && c->q4_lds && y.qkv.q4 && q_for(y.qkv, n, QS_QKV) && HD == 128 && H % 512 == 0 && qkv_lds <= 64 * 1024 && !c->prof;
        // the rms norm inside the consuming projection's staging (stage_rms_q8): no slabs may be pending, the row is held in registers
        const bool rms_fused = c->q4_rms && !c->l_pending && H <= 4096;
        if (qkv_fused && rms_fused) {
            QGemmArgs qa{};
            qa.g.W = c->arena + y.qkv.off; qa.g.K = H; qa.g.N = QKV; qa.g.R = n; qa.g.out = c->l_qkv; qa.g.ldo = QKV;
            qa.wd = (const _Float16 *) (c->arena + y.qkv.soff);
            RopeEpi re{(const uint32_t *) c->l_pos, f32(c->l_ropef), theta_scale, NH, NKV, kc, vc};
            RmsSrc rs{c->l_x, f32(y.in_norm), 1e-5f};
            hipLaunchKernelGGL((gemv_q4_qkv_rope_kernel<4, 2>), dim3((QKV / 2 + 3) / 4), dim3(256), qkv_lds + 16, c->stream, qa, y.qkv.q4, re, rs);
            HIPCHK(hipGetLastError());
            c->aq_src = nullptr;
        } else if (qkv_fused) {
            CHK(rms(y.in_norm, n, c->l_x, c->l_xn, &y.qkv, QS_QKV));
            // the projection, the rope of q and k and the cache append in one launch (gemv_q4_qkv_rope_kernel)
            QGemmArgs qa{};
            qa.g.W = c->arena + y.qkv.off; qa.g.K = H; qa.g.N = QKV; qa.g.R = n; qa.g.out = c->l_qkv; qa.g.ldo = QKV;
            qa.wd = (const _Float16 *) (c->arena + y.qkv.soff); qa.aq = c->aq; qa.ad = c->ad;
            RopeEpi re{(const uint32_t *) c->l_pos, f32(c->l_ropef), theta_scale, NH, NKV, kc, vc};
            hipLaunchKernelGGL(gemv_q4_qkv_rope_kernel<4>, dim3((QKV / 2 + 3) / 4), dim3(256), qkv_lds, c->stream, qa, y.qkv.q4, re);
            HIPCHK(hipGetLastError());
            c->aq_src = nullptr;
        } else {
            CHK(rms(y.in_norm, n, c->l_x, c->l_xn, &y.qkv, QS_QKV));
            const int sl = qstream(y.qkv, QS_QKV, c->l_xn, H, c->l_qkv, QKV, (int64_t) srows * QKV, smax);   // slabs of srows rows inside l_qkv ([RMAX][QKV])
            if (sl < 0) return -1;
            if (!sl) CHK(llama_gemm(c, y.qkv, c->l_xn, H, c->l_qkv, QKV, n, EPI_STORE));
            hipLaunchKernelGGL(llama_rope_kv_kernel, dim3(n, NH + NKV), dim3(64), 0, c->stream, c->l_qkv, (const uint32_t *) c->l_pos, f32(c->l_ropef), theta_scale, NH, NKV, HD, kc, vc,
                               row_seq, row_seq ? seq_stride : (int64_t) 0, std::max(sl, 1), (int64_t) srows * QKV);
            HIPCHK(hipGetLastError());
        }
        CHK(launch_attn_gqa(c, NH, n, (int) (attn_positions ? (uint32_t) attn_positions : pos0 + n), (const float *) c->l_qkv, QKV, (const uint32_t *) c->l_pos,
                            (const float *) kc, (const float *) vc, NKV, 1.0f / sqrtf((float) HD), c->l_att, nullptr, nullptr, row_seq, row_seq ? seq_stride : (int64_t) 0,
                            attn_positions != 0 && !row_seq, q_for(y.o, n, QS_O), QPre{}, nullptr, NCTX));
        {
            const int sl = qstream(y.o, QS_O, c->l_att, NH * HD, c->l_parts, H, (int64_t) srows * H, pmax);   // slabs of srows rows inside l_parts ([8][RMAX][H]), folded into the residual stream by the next rms norm
            if (sl < 0) return -1;
            if (sl) { c->l_pending = sl; c->l_pstride = (int64_t) srows * H; }
            else CHK(llama_gemm(c, y.o, c->l_att, NH * HD, c->l_x, H, n, EPI_RESID));
        }
        const size_t gu_lds = (size_t) n * H + (size_t) n * (H / 32) * 4, dn_lds = (size_t) n * F + (size_t) n * (F / 32) * 4;
        const bool gu_fused = n <= 4 && c->q4_silu && c->q4_lds && y.gu.q4 && y.down.q4 && q_for(y.gu, n, QS_GU) && q_for(y.down, n, QS_DOWN) && H % 512 == 0 && F % 512 == 0 &&
                              gu_lds <= 64 * 1024 && dn_lds <= 64 * 1024 && (int) y.gu.N == 2 * F && !c->prof;
        if (!(gu_fused && rms_fused)) CHK(rms(y.post_norm, n, c->l_x, c->l_xn, &y.gu, QS_GU));
        if (gu_fused) {
            // gate | up with silu * up in the epilogue, then the down projection quantising that product while it stages it: two launches
            // instead of three (gemv_q4_gateup_silu_kernel, gemv_q4_rows_lds_kernel<.., QSRC 1>)
            const int gu_grid = std::min((F / 2 + 3) / 4, 512);   // 256 / 384 / 512 / 1024 workgroups: 1.24 / 1.17 / 1.14 / 1.25 ms per step (profiles/r05/orpheus_gu_grid_call12.txt)   // two resident workgroups per CU; a wave walks its items (gemv_q4_gateup_silu_kernel)
            QGemmArgs qa{};
            qa.g.W = c->arena + y.gu.off; qa.g.K = H; qa.g.N = 2 * F; qa.g.R = n;
            qa.wd = (const _Float16 *) (c->arena + y.gu.soff); qa.aq = c->aq; qa.ad = c->ad;
            if (rms_fused) {
                RmsSrc rs{c->l_x, f32(y.post_norm), 1e-5f};
                hipLaunchKernelGGL((gemv_q4_gateup_silu_kernel<4, 2>), dim3(gu_grid), dim3(256), gu_lds + 16, c->stream, qa, y.gu.q4, F, c->l_g, rs);
            } else {
                hipLaunchKernelGGL(gemv_q4_gateup_silu_kernel<4>, dim3(gu_grid), dim3(256), gu_lds, c->stream, qa, y.gu.q4, F, c->l_g);
            }
            HIPCHK(hipGetLastError());
            c->aq_src = nullptr;
            QGemmArgs qd{};
            qd.g.W = c->arena + y.down.off; qd.g.K = F; qd.g.N = H; qd.g.R = n; qd.g.A = c->l_g; qd.g.lda = F; qd.g.out = c->l_x; qd.g.ldo = H;
            qd.wd = (const _Float16 *) (c->arena + y.down.soff);
            static std::atomic<uint64_t> attr{0};
            if (attr_needed(attr, c->device)) {
                HIPCHK(hipFuncSetAttribute((const void *) gemv_q4_rows_lds_kernel<4, 2, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                HIPCHK(hipFuncSetAttribute((const void *) gemv_q4_rows_lds_kernel<4, 2, 1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            }
            if (F > 4096) hipLaunchKernelGGL((gemv_q4_rows_lds_kernel<4, 2, 1, 4>), dim3((H + 7) / 8), dim3(256), dn_lds, c->stream, qd, y.down.q4, (int) EPI_RESID);
            else hipLaunchKernelGGL((gemv_q4_rows_lds_kernel<4, 2, 1, 2>), dim3((H + 7) / 8), dim3(256), dn_lds, c->stream, qd, y.down.q4, (int) EPI_RESID);
            HIPCHK(hipGetLastError());
            continue;
        }
        const int slg = qstream(y.gu, QS_GU, c->l_xn, H, c->l_gu, 2 * F, (int64_t) srows * 2 * F, smax);   // slabs of srows rows inside l_gu; silu_mul_kernel folds them
        if (slg < 0) return -1;
        if (!slg) CHK(llama_gemm(c, y.gu, c->l_xn, H, c->l_gu, 2 * F, n, EPI_STORE));
        const bool down_stream = slices(y.down, QS_DOWN, n, pmax) != 0;
        const int ks = ((c->gemv_rows && n <= 4) || down_stream) ? 1 : c->l_ksplit;   // the streaming kernels walk all of K themselves
        const bool qd = ks == 1 && q_for(y.down, n, QS_DOWN);
        hipLaunchKernelGGL(silu_mul_kernel, dim3((unsigned) (((size_t) n * F + 255) / 256)), dim3(256), 0, c->stream, (const float *) c->l_gu, F, n, c->l_g,
                           qd ? c->aq : (int8_t *) nullptr, qd ? c->ad : (float *) nullptr, std::max(slg, 1), (int64_t) srows * 2 * F);
        HIPCHK(hipGetLastError());
        c->aq_src = qd ? c->l_g : nullptr;
        if (down_stream) {
            const int sl = qstream(y.down, QS_DOWN, c->l_g, F, c->l_parts, H, (int64_t) srows * H, pmax);
            if (sl <= 0) return sl < 0 ? -1 : set_err("llama_forward: the down projection lost its streaming form");
            c->l_pending = sl; c->l_pstride = (int64_t) srows * H;
        } else if (ks > 1) {
            CHK(llama_gemm(c, y.down, c->l_g, F, c->l_parts, H, n, EPI_STORE, ks));
            c->l_pending = ks;
        } else {
            CHK(llama_gemm(c, y.down, c->l_g, F, c->l_x, H, n, EPI_RESID));
        }
    }
    // lm_head on the last token only (:287-290) — or, for lock-step utterances, on every row (each row is an utterance's last token)
    const bool head_stream = logits_row < 0 && slices(c->l_head, QS_HEAD, n, 1) != 0;
    CHK(rms(c->l_out_norm, n, c->l_x, c->l_xn, head_stream ? &c->l_head : nullptr, QS_HEAD));
    if (head_stream) {   // lock-step utterances, 5 .. 16 of them: the head streams once, whole K per wave (no slabs)
        const int sl = qstream(c->l_head, QS_HEAD, c->l_xn, H, c->l_logits, c->l_Vpad, 0, 1);
        if (sl < 0) return -1;
        if (sl) return 0;
    }
    GemmArgs g{};
    g.H = H; g.lda = H; g.ldo = c->l_Vpad;
    if (logits_row < 0) { g.R = n; g.A = c->l_xn; g.out = c->l_logits; }
    else { g.R = 1; g.A = c->l_xn + (size_t) (n - 1) * H; g.out = c->l_logits + (size_t) logits_row * c->l_Vpad; }
    CHK(run_gemm(c, TTS_HIP_K_GEMM_HEADS, c->l_head, g, PRO_F32, EPI_STORE));
    return 0;
}

extern "C" int tts_hip_orpheus_decode(tts_hip_ctx *c, const uint32_t *ids, uint32_t n, uint32_t pos0, float *logits_out, uint32_t *token_out) {
    if (!c || !c->has_llama) return set_err("tts_hip_orpheus_decode: not an Orpheus context (tts_hip_orpheus_create)");
    if (!c->finalized || !c->weights_present) return set_err("tts_hip_orpheus_decode: context not finalized");
    if (!ids || n == 0) return set_err("tts_hip_orpheus_decode: no tokens");
    HIPCHK(hipSetDevice(c->device));
    uint32_t done = 0;
    while (done < n) {   // a long prompt goes through in pieces of RMAX rows (same cache semantics as one call)
        const uint32_t m = std::min<uint32_t>((uint32_t) c->RMAX, n - done);
        CHK(llama_forward(c, ids + done, (int) m, pos0 + done));
        done += m;
    }
    if (token_out) {
        // l_tok: [0] the token, [1..] stage-1 indices, then stage-1 maxima
        uint32_t *pi = c->l_tok + 1;
        float *pv = (float *) (c->l_tok + 1 + ARGMAX_PARTS);
        hipLaunchKernelGGL(argmax_parts_kernel, dim3(ARGMAX_PARTS), dim3(256), 0, c->stream, (const float *) c->l_logits, c->l_V, pv, pi);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(argmax_fold_kernel, dim3(1), dim3(64), 0, c->stream, (const float *) pv, (const uint32_t *) pi, c->l_tok, (uint32_t *) nullptr,
                           (uint32_t *) nullptr, (uint32_t *) nullptr);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(token_out, c->l_tok, 4, hipMemcpyDeviceToHost, c->stream));
    }
    if (logits_out) HIPCHK(hipMemcpyAsync(logits_out, c->l_logits, (size_t) c->l_V * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// sampler::max or sampler::sample of l_logits -> l_tok[0]; captured: the history slot and the uniform come from device counters and the
// token is fed back; eager: hist_slot (may be NULL) receives the token, feed says whether it goes back as the next input
static int llama_select(tts_hip_ctx *c, const tts_hip_sampling *sp, bool captured, uint32_t *hist_slot, bool feed) {
    uint32_t *pi = c->l_tok + 1, *hist = c->l_tok + 1 + 2 * ARGMAX_PARTS, *hist_idx = hist + LLAMA_GREEDY_CHUNK;
    float *pv = (float *) (c->l_tok + 1 + ARGMAX_PARTS);
    if (!sp) {
        hipLaunchKernelGGL(argmax_parts_kernel, dim3(ARGMAX_PARTS), dim3(256), 0, c->stream, (const float *) c->l_logits, c->l_V, pv, pi);
        HIPCHK(hipGetLastError());
        if (captured)
            hipLaunchKernelGGL(argmax_fold_graph_kernel, dim3(1), dim3(64), 0, c->stream, (const float *) pv, (const uint32_t *) pi, c->l_tok, hist, hist_idx, c->l_ids, c->l_pos);
        else
            hipLaunchKernelGGL(argmax_fold_kernel, dim3(1), dim3(64), 0, c->stream, (const float *) pv, (const uint32_t *) pi, c->l_tok, hist_slot,
                               feed ? c->l_ids : (uint32_t *) nullptr, feed ? c->l_pos : (uint32_t *) nullptr);
        HIPCHK(hipGetLastError());
        return 0;
    }
    const double *pen = sp->repetition_penalty != 1.0f ? c->d_pen : nullptr;
    int32_t *last = (int32_t *) c->l_smp;
    uint32_t *repc = c->l_smp + 1, *call = c->l_smp + 2;
    hipLaunchKernelGGL(topk_parts_kernel, dim3(TOPK_PARTS), dim3(512), 0, c->stream, (const float *) c->l_logits, c->l_V, (int) sp->top_k, pen, c->pen_len, (const int32_t *) last,
                       (const uint32_t *) repc, c->l_cand);
    HIPCHK(hipGetLastError());
    float *total = nullptr;
    if (sp->top_p < 1.0f) {   // nucleus sampling: the softmax total over the whole vocabulary, accumulated in index order like the reference's
        total = (float *) (c->l_cand + (size_t) TOPK_PARTS * TOPK_MAXK);
        hipLaunchKernelGGL(softmax_total_kernel, dim3(1), dim3(1024), 0, c->stream, (const float *) c->l_logits, c->l_V, (const unsigned long long *) c->l_cand, sp->temperature, pen,
                           c->pen_len, (const int32_t *) last, (const uint32_t *) repc, total);
        HIPCHK(hipGetLastError());
    }
    hipLaunchKernelGGL(topk_sample_kernel, dim3(1), dim3(1024), 0, c->stream, (const unsigned long long *) c->l_cand, (int) sp->top_k, sp->temperature, (const float *) c->d_uniforms,
                       call, pen, last, repc, c->l_tok, captured ? hist : hist_slot, captured ? hist_idx : (uint32_t *) nullptr,
                       (captured || feed) ? c->l_ids : (uint32_t *) nullptr, (captured || feed) ? c->l_pos : (uint32_t *) nullptr, sp->top_p, (const float *) total);
    HIPCHK(hipGetLastError());
    return 0;
}

// what the two-stage device sampler covers (topk_parts_kernel / topk_sample_kernel, llama_kernels.h); anything else: sample on the host
// from tts_hip_orpheus_decode's logits
static int check_llama_sampling(const tts_hip_ctx *c, const tts_hip_sampling *sp, const char *what) {
    if (!sp) return set_err("%s: null sampling parameters", what);
    if (!(sp->temperature > 0.0f)) return set_err("%s: temperature must be > 0", what);
    if (!(sp->repetition_penalty > 0.0f)) return set_err("%s: repetition_penalty must be > 0 (1 = off)", what);
    if (!(sp->top_p > 0.0f)) return set_err("%s: top_p must be > 0", what);
    if (sp->top_k == 0 || sp->top_k > TOPK_MAXK || (int) sp->top_k >= c->l_V)
        return set_err("%s: the device sampler takes top_k in 1..%d (got %u): sample on the host", what, TOPK_MAXK, sp->top_k);
    if (c->l_V > TOPK_PARTS * TOPK_SLICE) return set_err("%s: vocabulary %d > %d", what, c->l_V, TOPK_PARTS * TOPK_SLICE);
    return 0;
}

// generate_from_batch (:378-392) with sampler::max (sp == NULL) or sampler::sample (sp, uniforms[max_new])
static int orpheus_generate(tts_hip_ctx *c, const char *what, const uint32_t *prompt, uint32_t n_prompt, uint32_t max_new, uint32_t stop_id, const tts_hip_sampling *sp,
                            const float *uniforms, uint32_t *tokens_out, uint32_t *n_out) {
    if (!c || !c->has_llama) return set_err("%s: not an Orpheus context (tts_hip_orpheus_create)", what);
    if (!c->finalized || !c->weights_present) return set_err("%s: context not finalized", what);
    if (!prompt || n_prompt == 0 || !tokens_out || !n_out) return set_err("%s: null argument", what);
    *n_out = 0;
    HIPCHK(hipSetDevice(c->device));
    if (sp) {
        CHK(check_llama_sampling(c, sp, what));
        if (!uniforms) return set_err("%s: null uniforms", what);
        if (max_new == 0) return 0;
        CHK(stage_uniforms(c, uniforms, (size_t) max_new));   // one sampler call per token, at most max_new tokens
        CHK(stage_penalty(c, sp->repetition_penalty, (int) max_new));
        const uint32_t init[3] = {0xFFFFFFFFu, 0u, 0u};   // sampler::reset (sampler.cpp:71-80): last token -1, count 0; call index 0
        HIPCHK(hipMemcpyAsync(c->l_smp, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        const void *pen = sp->repetition_penalty != 1.0f ? (const void *) c->d_pen : nullptr;
        if (c->l_smp_baked.uni != c->d_uniforms || c->l_smp_baked.pen != pen || c->l_smp_baked.k != sp->top_k || c->l_smp_baked.temp != sp->temperature || c->l_smp_baked.top_p != sp->top_p) {
            auto it = c->graphs.find(9000002);
            if (it != c->graphs.end()) { (void) hipGraphExecDestroy(it->second); c->graphs.erase(it); }
            c->l_smp_baked.uni = c->d_uniforms; c->l_smp_baked.pen = pen; c->l_smp_baked.k = sp->top_k; c->l_smp_baked.temp = sp->temperature; c->l_smp_baked.top_p = sp->top_p;
        }
    }
    uint32_t tok = 0, pos = n_prompt;
    {   // the prompt (pieces of RMAX rows), then the first selection
        uint32_t done = 0;
        while (done < n_prompt) {
            const uint32_t m = std::min<uint32_t>((uint32_t) c->RMAX, n_prompt - done);
            CHK(llama_forward(c, prompt + done, (int) m, done));
            done += m;
        }
        CHK(llama_select(c, sp, false, nullptr, false));
        HIPCHK(hipMemcpyAsync(&tok, c->l_tok, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    // stop once the last token is the stopping token or max_generation_size ids exist.  The token never leaves the device inside a
    // chunk of LLAMA_GREEDY_CHUNK steps (the selection writes it back as the next input and bumps the position); the host looks at
    // a chunk's tokens at once, so at most CHUNK-1 steps run past the stopping token (their cache rows are never read: the next
    // call starts at position 0; the sampler draws they consume belong to no token).
    uint32_t *hist = c->l_tok + 1 + 2 * ARGMAX_PARTS;
    uint32_t host_hist[LLAMA_GREEDY_CHUNK];
    while (*n_out < max_new) {
        tokens_out[(*n_out)++] = tok;
        if (tok == stop_id || *n_out >= max_new) break;
        if (pos >= c->lm.n_ctx) break;
        const uint32_t chunk = std::min<uint32_t>(std::min<uint32_t>(LLAMA_GREEDY_CHUNK, max_new - *n_out), c->lm.n_ctx - pos);
        HIPCHK(hipMemcpyAsync(c->l_ids, &tok, 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->l_pos, &pos, 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));  // tok / pos are reused below
        if (c->llama_graph && !c->prof) {
            // one captured step (forward + selection + feedback) replayed `chunk` times; the history slot is a device counter
            uint32_t *hist_idx = hist + LLAMA_GREEDY_CHUNK;
            HIPCHK(hipMemsetAsync(hist_idx, 0, 4, c->stream));
            if (pos + chunk > c->lm.n_ctx) return set_err("%s: positions exceed the cache", what);
            const int key = sp ? 9000002 : 9000001;
            auto it = c->graphs.find(key);
            if (it == c->graphs.end()) {
                // one eager pass first: per-kernel attributes are set outside the capture (it rewrites the cache row of `pos`
                // with the values the first replay writes again, nothing else)
                CHK(llama_forward(c, nullptr, 1, pos, (int) c->lm.n_ctx));
                HIPCHK(hipStreamSynchronize(c->stream));
                hipGraph_t graph = nullptr;
                HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
                int rc = llama_forward(c, nullptr, 1, 0, (int) c->lm.n_ctx);
                if (rc == 0) rc = llama_select(c, sp, true, nullptr, true);
                const hipError_t e = hipStreamEndCapture(c->stream, &graph);
                if (rc != 0) { if (graph) (void) hipGraphDestroy(graph); return rc; }
                if (e != hipSuccess) return set_err("hipStreamEndCapture: %s", hipGetErrorString(e));
                hipGraphExec_t exec = nullptr;
                HIPCHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
                (void) hipGraphDestroy(graph);
                it = c->graphs.emplace(key, exec).first;
            }
            for (uint32_t s = 0; s < chunk; s++) HIPCHK(hipGraphLaunch(it->second, c->stream));
        } else {
            for (uint32_t s = 0; s < chunk; s++) {
                CHK(llama_forward(c, nullptr, 1, pos + s));
                CHK(llama_select(c, sp, false, hist + s, true));
            }
        }
        HIPCHK(hipMemcpyAsync(host_hist, hist, (size_t) chunk * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        pos += chunk;
        // all but the chunk's last token are final here; the last one goes through the loop head like any other
        uint32_t s = 0;
        for (; s + 1 < chunk; s++) {
            tokens_out[(*n_out)++] = host_hist[s];
            if (host_hist[s] == stop_id || *n_out >= max_new) return 0;
        }
        tok = host_hist[s];
    }
    return 0;
}

extern "C" int tts_hip_orpheus_generate_greedy(tts_hip_ctx *c, const uint32_t *prompt, uint32_t n_prompt, uint32_t max_new, uint32_t stop_id,
                                               uint32_t *tokens_out, uint32_t *n_out) {
    return orpheus_generate(c, "tts_hip_orpheus_generate_greedy", prompt, n_prompt, max_new, stop_id, nullptr, nullptr, tokens_out, n_out);
}

extern "C" int tts_hip_orpheus_generate_sampled(tts_hip_ctx *c, const uint32_t *prompt, uint32_t n_prompt, uint32_t max_new, uint32_t stop_id,
                                                const tts_hip_sampling *sampling, const float *uniforms, uint32_t *tokens_out, uint32_t *n_out) {
    if (!sampling) return set_err("tts_hip_orpheus_generate_sampled: null sampling parameters");
    return orpheus_generate(c, "tts_hip_orpheus_generate_sampled", prompt, n_prompt, max_new, stop_id, sampling, uniforms, tokens_out, n_out);
}

extern "C" int tts_hip_orpheus_sample_logits(tts_hip_ctx *c, const float *logits, const tts_hip_sampling *sp, float uniform, int32_t *last_id, uint32_t *rep_count,
                                             uint32_t *token_out) {
    if (!c || !c->has_llama) return set_err("tts_hip_orpheus_sample_logits: not an Orpheus context (tts_hip_orpheus_create)");
    if (!c->finalized) return set_err("tts_hip_orpheus_sample_logits: context not finalized");
    if (!logits || !token_out) return set_err("tts_hip_orpheus_sample_logits: null argument");
    CHK(check_llama_sampling(c, sp, "tts_hip_orpheus_sample_logits"));
    HIPCHK(hipSetDevice(c->device));
    const bool rep = sp->repetition_penalty != 1.0f;
    if (rep && (!last_id || !rep_count)) return set_err("tts_hip_orpheus_sample_logits: repetition penalty needs last_id and rep_count");
    CHK(stage_uniforms(c, &uniform, 1));
    if (rep) CHK(stage_penalty(c, sp->repetition_penalty, (int) std::min<uint32_t>(*rep_count + 2, 1u << 20)));
    const uint32_t init[3] = {rep ? (uint32_t) *last_id : 0xFFFFFFFFu, rep ? *rep_count : 0u, 0u};
    HIPCHK(hipMemcpyAsync(c->l_smp, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->l_logits, logits, (size_t) c->l_V * 4, hipMemcpyHostToDevice, c->stream));
    CHK(llama_select(c, sp, false, nullptr, false));
    uint32_t back[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(token_out, c->l_tok, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(back, c->l_smp, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (rep) { *last_id = (int32_t) back[0]; *rep_count = back[1]; }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Dia (src/models/dia/model.cpp:383-659)
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// Lock-step utterances ("within a GPU, B utterances batched in lock-step"; the reference's only concurrency is N independent
// workers, each with its own model copy, examples/server/server.cpp:225-321).  The cache holds lm.max_seqs slots; a step carries one row per
// live utterance, every row with its own slot and position.
// ------------------------------------------------------------------------------------------------
static int llama_stage_rows(tts_hip_ctx *c, const char *what, uint32_t n, const uint32_t *slots, const uint32_t *ids, const uint32_t *pos, uint32_t *max_pos) {
    if (n == 0 || (int) n > c->RMAX) return set_err("%s: %u rows outside 1..%d", what, n, c->RMAX);
    *max_pos = 0;
    for (uint32_t r = 0; r < n; r++) {
        if (slots[r] >= c->lm.max_seqs) return set_err("%s: cache slot %u >= max_seqs %u", what, slots[r], c->lm.max_seqs);
        if (ids[r] >= (uint32_t) c->l_V) return set_err("%s: token id %u >= vocabulary %d", what, ids[r], c->l_V);
        if (pos[r] >= c->lm.n_ctx) return set_err("%s: position %u outside the %u cached positions", what, pos[r], c->lm.n_ctx);
        *max_pos = std::max(*max_pos, pos[r]);
    }
    HIPCHK(hipMemcpyAsync(c->l_ids, ids, (size_t) n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->l_pos, pos, (size_t) n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->l_seq, slots, (size_t) n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// sampler::max of the logits of rows 0 .. n-1 -> l_btok[r]
static int llama_argmax_rows(tts_hip_ctx *c, int n) {
    hipLaunchKernelGGL(argmax_rows_parts_kernel, dim3(ARGMAX_PARTS, n), dim3(256), 0, c->stream, (const float *) c->l_logits, c->l_V, c->l_Vpad, c->l_bpv, c->l_bpi);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(argmax_rows_fold_kernel, dim3(n), dim3(64), 0, c->stream, (const float *) c->l_bpv, (const uint32_t *) c->l_bpi, c->l_btok);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int tts_hip_orpheus_step_batch(tts_hip_ctx *c, uint32_t n, const uint32_t *slots, const uint32_t *ids, const uint32_t *pos, float *logits_out, uint32_t *tokens_out) {
    if (!c || !c->has_llama) return set_err("tts_hip_orpheus_step_batch: not an Orpheus context (tts_hip_orpheus_create)");
    if (!c->finalized || !c->weights_present) return set_err("tts_hip_orpheus_step_batch: context not finalized");
    if (!slots || !ids || !pos) return set_err("tts_hip_orpheus_step_batch: null argument");
    if (n > c->lm.max_seqs) return set_err("tts_hip_orpheus_step_batch: %u rows > max_seqs %u (one row per utterance)", n, c->lm.max_seqs);
    HIPCHK(hipSetDevice(c->device));
    uint32_t max_pos = 0;
    CHK(llama_stage_rows(c, "tts_hip_orpheus_step_batch", n, slots, ids, pos, &max_pos));
    CHK(llama_forward(c, nullptr, (int) n, 0, (int) max_pos + 1, c->l_seq, -1));
    if (tokens_out) {
        CHK(llama_argmax_rows(c, (int) n));
        HIPCHK(hipMemcpyAsync(tokens_out, c->l_btok, (size_t) n * 4, hipMemcpyDeviceToHost, c->stream));
    }
    if (logits_out) HIPCHK(hipMemcpy2DAsync(logits_out, (size_t) c->l_V * 4, c->l_logits, (size_t) c->l_Vpad * 4, (size_t) c->l_V * 4, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// the prompt of one utterance into its cache slot (pieces of RMAX rows); the last row's logits land in l_logits row `slot`
static int llama_prefill_slot(tts_hip_ctx *c, const char *what, uint32_t slot, const uint32_t *prompt, uint32_t n_prompt) {
    std::vector<uint32_t> sl, ps;
    uint32_t done = 0;
    while (done < n_prompt) {
        const uint32_t m = std::min<uint32_t>((uint32_t) c->RMAX, n_prompt - done);
        sl.assign(m, slot);
        ps.resize(m);
        for (uint32_t i = 0; i < m; i++) ps[i] = done + i;
        uint32_t max_pos = 0;
        CHK(llama_stage_rows(c, what, m, sl.data(), prompt + done, ps.data(), &max_pos));
        CHK(llama_forward(c, nullptr, (int) m, 0, (int) max_pos + 1, c->l_seq, (int) slot));
        done += m;
    }
    return 0;
}

// generate_from_batch (orpheus/model.cpp:378-392) for n_utt utterances in lock-step: every utterance gets exactly the tokens its own one-sequence
// generation gets (sampler::max, or sampler::sample with its own uniforms and repetition state); finished utterances leave the step.
extern "C" int tts_hip_orpheus_generate_batch(tts_hip_ctx *c, uint32_t n_utt, const uint32_t *prompts, const uint32_t *n_prompt, uint32_t max_new, uint32_t stop_id,
                                              const tts_hip_sampling *sp, const float *uniforms, uint32_t *tokens_out, uint32_t *n_out) {
    const char *what = "tts_hip_orpheus_generate_batch";
    if (!c || !c->has_llama) return set_err("%s: not an Orpheus context (tts_hip_orpheus_create)", what);
    if (!c->finalized || !c->weights_present) return set_err("%s: context not finalized", what);
    if (!prompts || !n_prompt || !tokens_out || !n_out) return set_err("%s: null argument", what);
    if (n_utt == 0 || n_utt > c->lm.max_seqs) return set_err("%s: %u utterances outside 1..max_seqs = %u", what, n_utt, c->lm.max_seqs);
    HIPCHK(hipSetDevice(c->device));
    for (uint32_t u = 0; u < n_utt; u++) { n_out[u] = 0; if (n_prompt[u] == 0 || n_prompt[u] >= c->lm.n_ctx) return set_err("%s: utterance %u: prompt of %u ids", what, u, n_prompt[u]); }
    if (max_new == 0) return 0;
    if (sp) {
        CHK(check_llama_sampling(c, sp, what));
        if (!uniforms) return set_err("%s: null uniforms", what);
        CHK(stage_uniforms(c, uniforms, (size_t) n_utt * max_new));   // utterance u draws uniforms[u * max_new + call]
        CHK(stage_penalty(c, sp->repetition_penalty, (int) max_new));
        std::vector<uint32_t> init((size_t) 3 * n_utt);
        for (uint32_t u = 0; u < n_utt; u++) { init[3 * u] = 0xFFFFFFFFu; init[3 * u + 1] = 0; init[3 * u + 2] = 0; }   // sampler::reset per utterance
        HIPCHK(hipMemcpyAsync(c->l_bsmp, init.data(), init.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    // select for logits row r on behalf of utterance u -> l_btok[r]
    auto select_rows = [&](const std::vector<uint32_t> &utt) -> int {
        const int n = (int) utt.size();
        if (!sp) return llama_argmax_rows(c, n);
        const double *pen = sp->repetition_penalty != 1.0f ? c->d_pen : nullptr;
        for (int r = 0; r < n; r++) {
            const uint32_t u = utt[(size_t) r];
            int32_t *last = (int32_t *) (c->l_bsmp + 3 * u);
            uint32_t *repc = c->l_bsmp + 3 * u + 1, *call = c->l_bsmp + 3 * u + 2;
            const float *lg = c->l_logits + (size_t) r * c->l_Vpad;
            hipLaunchKernelGGL(topk_parts_kernel, dim3(TOPK_PARTS), dim3(512), 0, c->stream, lg, c->l_V, (int) sp->top_k, pen, c->pen_len, (const int32_t *) last, (const uint32_t *) repc, c->l_cand);
            HIPCHK(hipGetLastError());
            float *total = nullptr;
            if (sp->top_p < 1.0f) {
                total = (float *) (c->l_cand + (size_t) TOPK_PARTS * TOPK_MAXK);
                hipLaunchKernelGGL(softmax_total_kernel, dim3(1), dim3(1024), 0, c->stream, lg, c->l_V, (const unsigned long long *) c->l_cand, sp->temperature, pen, c->pen_len,
                                   (const int32_t *) last, (const uint32_t *) repc, total);
                HIPCHK(hipGetLastError());
            }
            hipLaunchKernelGGL(topk_sample_kernel, dim3(1), dim3(1024), 0, c->stream, (const unsigned long long *) c->l_cand, (int) sp->top_k, sp->temperature,
                               (const float *) c->d_uniforms + (size_t) u * max_new, call, pen, last, repc, c->l_btok + r, (uint32_t *) nullptr, (uint32_t *) nullptr, (uint32_t *) nullptr,
                               (uint32_t *) nullptr, sp->top_p, (const float *) total);
            HIPCHK(hipGetLastError());
        }
        return 0;
    };
    // prompts, then the first selection from every utterance's last prompt row (logits row u)
    size_t off = 0;
    std::vector<uint32_t> utt(n_utt), pos(n_utt), tok(n_utt), slots, ids, ps;
    for (uint32_t u = 0; u < n_utt; u++) {
        CHK(llama_prefill_slot(c, what, u, prompts + off, n_prompt[u]));
        off += n_prompt[u];
        utt[u] = u;
        pos[u] = n_prompt[u];
    }
    CHK(select_rows(utt));
    HIPCHK(hipMemcpyAsync(tok.data(), c->l_btok, (size_t) n_utt * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    // the loop of orpheus_generate, per utterance: record the token; stop on the stopping token, at max_new ids or at the end of the cache
    std::vector<uint32_t> live;   // utterances still generating, in utterance order (= the rows of the next step)
    for (uint32_t u = 0; u < n_utt; u++) live.push_back(u);
    std::vector<uint32_t> cur(n_utt);
    for (uint32_t u = 0; u < n_utt; u++) cur[u] = tok[u];
    while (!live.empty()) {
        std::vector<uint32_t> next;
        for (uint32_t u : live) {
            tokens_out[(size_t) u * max_new + n_out[u]++] = cur[u];
            if (cur[u] == stop_id || n_out[u] >= max_new || pos[u] >= c->lm.n_ctx) continue;
            next.push_back(u);
        }
        live.swap(next);
        if (live.empty()) break;
        const uint32_t n = (uint32_t) live.size();
        slots.resize(n); ids.resize(n); ps.resize(n);
        for (uint32_t r = 0; r < n; r++) { slots[r] = live[r]; ids[r] = cur[live[r]]; ps[r] = pos[live[r]]; }
        uint32_t max_pos = 0;
        CHK(llama_stage_rows(c, what, n, slots.data(), ids.data(), ps.data(), &max_pos));
        CHK(llama_forward(c, nullptr, (int) n, 0, (int) max_pos + 1, c->l_seq, -1));
        CHK(select_rows(live));
        HIPCHK(hipMemcpyAsync(tok.data(), c->l_btok, (size_t) n * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        for (uint32_t r = 0; r < n; r++) { cur[live[r]] = tok[r]; pos[live[r]]++; }
    }
    return 0;
}

extern "C" tts_hip_ctx *tts_hip_dia_create(int device, const tts_hip_dia_desc *dd) {
    if (!dd || dd->struct_size != sizeof(tts_hip_dia_desc)) { set_err("tts_hip_dia_create: bad desc (struct_size mismatch)"); return nullptr; }
    tts_hip_desc d{};
    d.struct_size = sizeof(d);
    d.hidden_size = dd->dec_hidden_size; d.n_layers = dd->dec_layers; d.n_attn_heads = dd->dec_attn_heads; d.max_ctx_length = dd->max_gen;
    d.max_seqs = 1;
    d.flags = (dd->flags & (TTS_HIP_FLAG_VALU_GEMM | TTS_HIP_FLAG_DEQUANT_Q)) | TTS_HIP_FLAG_NO_PARLER | TTS_HIP_FLAG_NO_DAC;
    tts_hip_ctx *c = tts_hip_create(device, &d);
    if (!c) return nullptr;
    c->has_dia = true;
    c->dia = *dd;
    if (c->dia.cfg_scale == 0.0f) c->dia.cfg_scale = 3.0f;
    return c;
}

// rows in pieces of RMAX (the activation-quantisation scratch holds RMAX rows); ksplit > 1 only with n <= RMAX
static int dia_gemm(tts_hip_ctx *c, const W &w, const float *A, int lda, float *out, int ldo, int n, int epi, int ksplit = 1) {
    if (ksplit > 1 && n > c->RMAX) return set_err("dia_gemm: split-K needs all rows in one piece");
    for (int r0 = 0; r0 < n; r0 += c->RMAX) {
        GemmArgs g{};
        g.R = std::min(c->RMAX, n - r0); g.H = c->H;
        g.A = A + (size_t) r0 * lda; g.lda = lda;
        g.out = out + (size_t) r0 * ldo; g.ldo = ldo;
        if (ksplit > 1) {
            g.kchunk = (int) w.K / ksplit;
            g.slab_stride = (int64_t) c->RMAX * ldo;
        }
        CHK(run_gemm(c, TTS_HIP_K_GEMM_OTHER, w, g, PRO_F32, epi));
    }
    return 0;
}

// the encoder's GEMMs (2 x max_ctx rows): fp16 matrices take gemm_tile_kernel with all rows in one launch — the activations are rounded
// to fp16 once (what ggml_mul_mat does with them for an F16 weight), every matrix leaves HBM once instead of once per RMAX rows
static int dia_gemm_rows(tts_hip_ctx *c, const W &w, const float *A, int lda, float *out, int ldo, int n, int epi) {
    if (w.type != TTS_HIP_F16 || c->tile_min_rows <= 0 || n < c->tile_min_rows || w.K % 128 || w.N % 16 || (c->d.flags & TTS_HIP_FLAG_VALU_GEMM) || c->prof)
        return dia_gemm(c, w, A, lda, out, ldo, n, epi);
    const int64_t n8 = (int64_t) n * (int64_t) (w.K / 8);
    hipLaunchKernelGGL(rows_to_f16_kernel, dim3((unsigned) ((n8 + 255) / 256)), dim3(256), 0, c->stream, A, lda, (int) w.K, n8, c->di_e16);
    HIPCHK(hipGetLastError());
    GemmArgs g{};
    g.R = n; g.H = c->H;
    g.A = c->di_e16; g.lda = (int) w.K;
    g.out = out; g.ldo = ldo;
    return run_gemm(c, TTS_HIP_K_GEMM_OTHER, w, g, PRO_F16, epi);
}

// <= 16 rows through gemv_stream_kernel: `out` receives *slabs K-slice slabs 16 * ldo floats apart (the consumer folds them);
// *slabs = 0: the shape does not qualify and nothing was launched
// pro: PRO_F32 (A = fp32 rows), PRO_ATTN8 (A unused: the eight key slices in c->attn_part are merged while the rows are staged) or PRO_SILU (A = the
// gate | up rows [n][2 K] as in_parts slabs in_stride floats apart: silu(gate) * up while staged); the last two need stream_fold_ok()
static int dia_gemm_stream(tts_hip_ctx *c, const W &w, const float *A, int lda, float *out, int ldo, int n, int max_slabs, int64_t slab_stride, int *slabs,
                           int pro = PRO_F32, int in_parts = 1, int64_t in_stride = 0) {
    const int ks = stream_slices(c, w, n, max_slabs);
    *slabs = ks;
    if (!ks) return pro == PRO_F32 ? 0 : set_err("dia_gemm_stream: a folding prologue was promised to a shape the streaming kernel does not take");
    GemmArgs g{};
    g.R = n; g.H = c->H;
    g.A = A; g.lda = lda;
    g.out = out; g.ldo = ldo;
    g.stream = 1;
    g.kchunk = ks > 1 ? (int) w.K / ks : 0;
    g.slab_stride = slab_stride;
    if (pro == PRO_ATTN8) g.att_part = c->attn_part;
    if (pro == PRO_SILU) { g.n_parts = std::max(in_parts, 1); g.parts_stride = in_stride; }
    return run_gemm(c, TTS_HIP_K_GEMM_OTHER, w, g, pro, EPI_STORE);
}

static int dia_rms(tts_hip_ctx *c, size_t w_off, int rows, int H, float *x, float *y, bool fold) {
    const int pend = fold ? c->di_pending : 0;
    hipLaunchKernelGGL(rms_fold_rows_kernel, dim3(rows), dim3(256), 0, c->stream, x, H, (const float *) (c->arena + w_off), y, rows, 1e-5f,
                       pend ? (const float *) c->di_parts : (const float *) nullptr, pend, (int64_t) c->RMAX * H, (int8_t *) nullptr, (float *) nullptr);
    if (fold) c->di_pending = 0;
    return hipGetLastError() == hipSuccess ? 0 : set_err("rms_fold_rows_kernel launch failed");
}

extern "C" int tts_hip_dia_encode_slot(tts_hip_ctx *c, uint32_t slot, const uint32_t *tokens, uint32_t sentence_len, float *enc_out) {
    if (!c || !c->has_dia) return set_err("tts_hip_dia_encode: not a Dia context (tts_hip_dia_create)");
    if (slot >= (uint32_t) c->di_U) return set_err("tts_hip_dia_encode_slot: slot %u outside the %d utterance slots of this context (max_utterances)", slot, c->di_U);
    if (!c->finalized || !c->weights_present) return set_err("tts_hip_dia_encode: context not finalized");
    if (!tokens) return set_err("tts_hip_dia_encode: null argument");
    const int S = (int) c->dia.max_ctx, EH = c->di_EH, EF = c->di_EF, A = c->di_A, HD = (int) c->dia.head_dim, ENH = (int) c->dia.enc_attn_heads;
    const int NH = c->NH, n = 2 * S;
    if (sentence_len == 0 || sentence_len > (uint32_t) S) return set_err("tts_hip_dia_encode: sentence length %u outside 1..%d", sentence_len, S);
    std::vector<uint32_t> tok((size_t) n, 0u), epos((size_t) n), eseq((size_t) n), kbeg((size_t) n), kend((size_t) n);
    for (int t = 0; t < S; t++) {
        if (tokens[t] >= (uint32_t) c->di_evocab) return set_err("tts_hip_dia_encode: token %u >= encoder vocabulary %d", tokens[t], c->di_evocab);
        tok[(size_t) t] = tokens[t];
    }
    for (int t = 0; t < n; t++) {   // set_inputs :712-721: real positions see real positions, pad positions see pad positions
        const uint32_t p = (uint32_t) (t % S);
        epos[(size_t) t] = p; eseq[(size_t) t] = (uint32_t) (t / S);
        kbeg[(size_t) t] = p < sentence_len ? 0u : sentence_len;
        kend[(size_t) t] = p < sentence_len ? sentence_len : (uint32_t) S;
    }
    HIPCHK(hipSetDevice(c->device));
    const size_t nb = (size_t) n * 4;
    HIPCHK(hipMemcpyAsync(c->di_tok, tok.data(), nb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->di_epos, epos.data(), nb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->di_eseq, eseq.data(), nb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->di_kbeg, kbeg.data(), nb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->di_kend, kend.data(), nb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));  // the vectors are locals
    auto f32 = [&](size_t off) { return (const float *) (c->arena + off); };
    const float theta_scale = powf(10000.0f, -2.0f / (float) HD);   // ggml_rope(..., head_size, 2): default base
    const size_t attn_lds = (size_t) (128 + S) * 4;
    static std::atomic<uint64_t> attr{0};
    if (attn_lds > 48 * 1024 && attr_needed(attr, c->device))
        HIPCHK(hipFuncSetAttribute((const void *) attn_gqa_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8192));
    hipLaunchKernelGGL(t5_embed_kernel, dim3(n), dim3(256), 0, c->stream, f32(c->di_enc_embd), (const uint32_t *) c->di_tok, EH, c->di_ex);
    HIPCHK(hipGetLastError());
    for (const auto &y : c->di_enc) {
        CHK(dia_rms(c, y.sa_norm, n, EH, c->di_ex, c->di_exn, false));
        CHK(dia_gemm_rows(c, y.qkv, c->di_exn, EH, c->di_eqkv, 3 * A, n, EPI_STORE));
        hipLaunchKernelGGL(llama_rope_kv_kernel, dim3(n, 2 * ENH), dim3(64), 0, c->stream, c->di_eqkv, (const uint32_t *) c->di_epos, (const float *) nullptr, theta_scale,
                           ENH, ENH, HD, c->di_ek, c->di_ev, (const uint32_t *) c->di_eseq, (int64_t) S * A);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(attn_gqa_kernel<128>, dim3(ENH, n), dim3(256), attn_lds, c->stream, (const float *) c->di_eqkv, 3 * A, (const uint32_t *) c->di_epos,
                           (const float *) c->di_ek, (const float *) c->di_ev, ENH, ENH, 1.0f, c->di_eatt, (const uint32_t *) c->di_kbeg, (const uint32_t *) c->di_kend,
                           (const uint32_t *) c->di_eseq, (int64_t) S * A);
        HIPCHK(hipGetLastError());
        CHK(dia_gemm_rows(c, y.o, c->di_eatt, A, c->di_ex, EH, n, EPI_RESID));
        CHK(dia_rms(c, y.mlp_norm, n, EH, c->di_ex, c->di_exn, false));
        CHK(dia_gemm_rows(c, y.gu, c->di_exn, EH, c->di_egu, 2 * EF, n, EPI_STORE));
        hipLaunchKernelGGL(silu_mul_kernel, dim3((unsigned) (((size_t) n * EF + 255) / 256)), dim3(256), 0, c->stream, (const float *) c->di_egu, EF, n, c->di_eg, (int8_t *) nullptr, (float *) nullptr);
        HIPCHK(hipGetLastError());
        CHK(dia_gemm_rows(c, y.out, c->di_eg, EF, c->di_ex, EH, n, EPI_RESID));
    }
    CHK(dia_rms(c, c->di_enc_norm, n, EH, c->di_ex, c->di_exn, false));
    // cross K/V of every decoder layer (build_dia_cross_kv_store :505-541): V for all positions, K (rope'd with the encoder
    // positions) only for the sentence; the other K rows are zero as in the freshly cleared cache
    for (int l = 0; l < c->L; l++) {
        const auto &y = c->di_dec[(size_t) l];
        float *ck = c->di_ck + ((size_t) l * c->di_U + slot) * n * A, *cv = c->di_cv + ((size_t) l * c->di_U + slot) * n * A;   // rows 2*slot, 2*slot+1
        CHK(dia_gemm_rows(c, y.ckv, c->di_exn, EH, c->di_ckv, 2 * A, n, EPI_STORE));
        hipLaunchKernelGGL(llama_rope_kv_kernel, dim3(n, NH), dim3(64), 0, c->stream, c->di_ckv, (const uint32_t *) c->di_epos, (const float *) nullptr, theta_scale, 0, NH,
                           HD, ck, cv, (const uint32_t *) c->di_eseq, (int64_t) S * A);
        HIPCHK(hipGetLastError());
        if ((int) sentence_len < S)
            for (int b = 0; b < 2; b++)
                HIPCHK(hipMemsetAsync(ck + ((size_t) b * S + sentence_len) * A, 0, (size_t) (S - (int) sentence_len) * A * 4, c->stream));
    }
    if (enc_out) HIPCHK(hipMemcpyAsync(enc_out, c->di_exn, (size_t) n * EH * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->di_slot_encoded[slot] = 1;
    return 0;
}

extern "C" int tts_hip_dia_encode(tts_hip_ctx *c, const uint32_t *tokens, uint32_t sentence_len, float *enc_out) {
    return tts_hip_dia_encode_slot(c, 0, tokens, sentence_len, enc_out);
}

// the decoder step for the U utterances whose input ids / positions / cache rows are in di_ids / di_pos / di_seq; leaves the guided
// logits in di_guided.  self_keys sizes the self-attention scratch; fixed_split: a captured step is replayed at every position, so the
// key-split count must not depend on it (the kernels read the true extent from di_pos)
static int dia_forward(tts_hip_ctx *c, int U, int self_keys, bool fixed_split) {
    const int S = (int) c->dia.max_ctx, G = (int) c->dia.max_gen, DH = c->H, DF = c->di_DF, A = c->di_A, kvH = c->di_kvH, HD = (int) c->dia.head_dim;
    const int NH = c->NH, NKV = (int) c->dia.dec_kv_heads, NO = c->NO, V = c->di_V, QKV = A + 2 * kvH;
    const int R = 2 * U, RS = 2 * c->di_U;
    auto f32 = [&](size_t off) { return (const float *) (c->arena + off); };
    const float theta_scale = powf(10000.0f, -2.0f / (float) HD);
    static std::atomic<uint64_t> attr{0};
    if ((size_t) (128 + std::max(S, G)) * 4 > 48 * 1024 && attr_needed(attr, c->device))
        HIPCHK(hipFuncSetAttribute((const void *) attn_gqa_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8192));
    DiaEmbedArgs ea{};
    for (int i = 0; i < NO; i++) ea.table[i] = f32(c->di_embd[i]);
    ea.ids = c->di_ids; ea.n_out = NO; ea.H = DH; ea.x = c->di_x;
    hipLaunchKernelGGL(dia_embed_kernel, dim3((DH + 255) / 256, U), dim3(256), 0, c->stream, ea);
    HIPCHK(hipGetLastError());
    c->di_pending = 0;
    const uint32_t *nul = nullptr;
    for (int l = 0; l < c->L; l++) {
        const auto &y = c->di_dec[(size_t) l];
        float *kc = c->di_k + (size_t) l * RS * G * kvH, *vc = c->di_v + (size_t) l * RS * G * kvH;
        const float *ck = c->di_ck + (size_t) l * RS * S * A, *cv = c->di_cv + (size_t) l * RS * S * A;
        // every projection: gemv_stream_kernel slabs folded by its consumer when the step has <= 16 rows and fp16 matrices
        // (sl = slabs written, 0 = shape does not qualify -> gemm16_kernel as before)
        int sl = 0;
        const int64_t st16 = 16;   // slab stride in rows
        CHK(dia_rms(c, y.sa_norm, R, DH, c->di_x, c->di_xn, true));
        CHK(dia_gemm_stream(c, y.sqkv, c->di_xn, DH, c->di_qkv, QKV, R, DIA_STREAM_SLABS, st16 * QKV, &sl));
        if (!sl) CHK(dia_gemm(c, y.sqkv, c->di_xn, DH, c->di_qkv, QKV, R, EPI_STORE));
        hipLaunchKernelGGL(llama_rope_kv_kernel, dim3(R, NH + NKV), dim3(64), 0, c->stream, c->di_qkv, (const uint32_t *) c->di_pos, (const float *) nullptr, theta_scale, NH,
                           NKV, HD, kc, vc, (const uint32_t *) c->di_seq, (int64_t) G * kvH, std::max(sl, 1), st16 * QKV);
        HIPCHK(hipGetLastError());
        int slices = 0;   // key slices the attention left unmerged for the projection's staging prologue (0: di_att holds the rows)
        CHK(launch_attn_gqa(c, NH, R, self_keys, (const float *) c->di_qkv, QKV, (const uint32_t *) c->di_pos, (const float *) kc, (const float *) vc, NKV, 1.0f,
                            c->di_att, nul, nul, (const uint32_t *) c->di_seq, (int64_t) G * kvH, fixed_split, false, QPre{},
                            A % 128 == 0 && stream_fold_ok(c, y.so, R, DIA_STREAM_SLABS) ? &slices : nullptr));
        CHK(dia_gemm_stream(c, y.so, c->di_att, A, c->di_parts, DH, R, DIA_STREAM_SLABS, (int64_t) c->RMAX * DH, &sl, slices ? PRO_ATTN8 : PRO_F32));
        if (sl) c->di_pending = sl;
        else CHK(dia_gemm(c, y.so, c->di_att, A, c->di_x, DH, R, EPI_RESID));
        CHK(dia_rms(c, y.ca_norm, R, DH, c->di_x, c->di_xn, true));
        CHK(dia_gemm_stream(c, y.cq, c->di_xn, DH, c->di_q, A, R, DIA_STREAM_SLABS, st16 * A, &sl));
        if (!sl) CHK(dia_gemm(c, y.cq, c->di_xn, DH, c->di_q, A, R, EPI_STORE));
        QPre qp;   // slab fold + rope of the cross-attention query happen as the attention workgroups load it
        qp.n_parts = std::max(sl, 1); qp.part_stride = st16 * A; qp.rope_pos = c->di_pos; qp.theta_scale = theta_scale;
        slices = 0;
        CHK(launch_attn_gqa(c, NH, R, S, (const float *) c->di_q, A, (const uint32_t *) c->di_pos, ck, cv, NH, 1.0f, c->di_att, nul, (const uint32_t *) c->di_cend,
                            (const uint32_t *) c->di_seq, (int64_t) S * A, false, false, qp,
                            A % 128 == 0 && stream_fold_ok(c, y.co, R, DIA_STREAM_SLABS) ? &slices : nullptr, S));
        CHK(dia_gemm_stream(c, y.co, c->di_att, A, c->di_parts, DH, R, DIA_STREAM_SLABS, (int64_t) c->RMAX * DH, &sl, slices ? PRO_ATTN8 : PRO_F32));
        if (sl) c->di_pending = sl;
        else CHK(dia_gemm(c, y.co, c->di_att, A, c->di_x, DH, R, EPI_RESID));
        CHK(dia_rms(c, y.mlp_norm, R, DH, c->di_x, c->di_xn, true));
        CHK(dia_gemm_stream(c, y.gu, c->di_xn, DH, c->di_gu, 2 * DF, R, DIA_STREAM_SLABS, st16 * 2 * DF, &sl));
        if (!sl) CHK(dia_gemm(c, y.gu, c->di_xn, DH, c->di_gu, 2 * DF, R, EPI_STORE));
        if (c->attn_fold && !c->prof && !c->debug && std::max(sl, 1) <= 8 && stream_fold_ok(c, y.out, R, DIA_STREAM_SLABS)) {
            // silu(gate) * up while the down projection stages its rows: no launch of its own, di_g is not written
            CHK(dia_gemm_stream(c, y.out, c->di_gu, 2 * DF, c->di_parts, DH, R, DIA_STREAM_SLABS, (int64_t) c->RMAX * DH, &sl, PRO_SILU, std::max(sl, 1), st16 * 2 * DF));
        } else {
            hipLaunchKernelGGL(silu_mul_kernel, dim3((unsigned) (((size_t) R * DF + 255) / 256)), dim3(256), 0, c->stream, (const float *) c->di_gu, DF, R, c->di_g, (int8_t *) nullptr,
                               (float *) nullptr, std::max(sl, 1), st16 * 2 * DF);
            HIPCHK(hipGetLastError());
            CHK(dia_gemm_stream(c, y.out, c->di_g, DF, c->di_parts, DH, R, DIA_STREAM_SLABS, (int64_t) c->RMAX * DH, &sl));
        }
        if (sl) {
            c->di_pending = sl;
        } else if (c->di_ksplit > 1) {
            CHK(dia_gemm(c, y.out, c->di_g, DF, c->di_parts, DH, R, EPI_STORE, c->di_ksplit));
            c->di_pending = c->di_ksplit;
        } else {
            CHK(dia_gemm(c, y.out, c->di_g, DF, c->di_x, DH, R, EPI_RESID));
        }
    }
    CHK(dia_rms(c, c->di_dec_norm, R, DH, c->di_x, c->di_xn, true));
    GemmArgs g{};
    g.R = R; g.H = DH; g.A = c->di_xn; g.lda = DH; g.out = c->di_logits; g.ldo = c->di_Vpad;
    CHK(run_gemm(c, TTS_HIP_K_GEMM_HEADS, c->di_heads, g, PRO_F32, EPI_STORE));
    hipLaunchKernelGGL(dia_cfg_kernel, dim3((NO * V + 255) / 256, U), dim3(256), 0, c->stream, (const float *) c->di_logits, c->di_Vpad, NO * V, c->dia.cfg_scale, c->di_guided);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int tts_hip_dia_step_batch(tts_hip_ctx *c, uint32_t n_utt, const uint32_t *slots, const uint32_t *ids, const uint32_t *pos, float *logits_out,
                                      float *raw_out) {
    if (!c || !c->has_dia) return set_err("tts_hip_dia_step: not a Dia context (tts_hip_dia_create)");
    if (!c->finalized || !c->weights_present) return set_err("tts_hip_dia_step: context not finalized");
    if (!ids || !pos || !logits_out) return set_err("tts_hip_dia_step: null argument");
    if (n_utt == 0 || n_utt > (uint32_t) c->di_U) return set_err("tts_hip_dia_step_batch: %u utterances outside 1..%d (max_utterances)", n_utt, c->di_U);
    const int G = (int) c->dia.max_gen, NO = c->NO, V = c->di_V;
    const int U = (int) n_utt, R = 2 * U, RS = 2 * c->di_U;   // rows of this step, row slots of the caches
    uint32_t max_pos = 0;
    uint32_t *h_ids = c->h_di, *h_pos = c->h_di + (size_t) c->di_U * 16, *h_seq = h_pos + RS;
    for (int u = 0; u < U; u++) {
        const uint32_t slot = slots ? slots[u] : (uint32_t) u;
        if (slot >= (uint32_t) c->di_U) return set_err("tts_hip_dia_step_batch: slot %u outside the %d utterance slots", slot, c->di_U);
        if (!c->di_slot_encoded[slot]) return set_err("tts_hip_dia_step: tts_hip_dia_encode has not run%s", c->di_U > 1 ? " for this slot" : "");
        if (pos[u] >= (uint32_t) G) return set_err("tts_hip_dia_step: position %u outside the %d cached positions", pos[u], G);
        for (int i = 0; i < NO; i++) {
            if (ids[u * NO + i] >= (uint32_t) V) return set_err("tts_hip_dia_step: id %u >= output vocabulary %d", ids[u * NO + i], V);
            h_ids[u * NO + i] = ids[u * NO + i];
        }
        h_pos[2 * u] = h_pos[2 * u + 1] = pos[u];
        h_seq[2 * u] = 2 * slot; h_seq[2 * u + 1] = 2 * slot + 1;
        max_pos = std::max(max_pos, pos[u]);
    }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(c->di_ids, h_ids, (size_t) U * NO * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->di_pos, h_pos, (size_t) R * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->di_seq, h_seq, (size_t) R * 4, hipMemcpyHostToDevice, c->stream));
    CHK(dia_forward(c, U, (int) max_pos + 1, false));
    HIPCHK(hipMemcpyAsync(logits_out, c->di_guided, (size_t) U * NO * V * 4, hipMemcpyDeviceToHost, c->stream));
    if (raw_out)
        for (int b = 0; b < R; b++)
            HIPCHK(hipMemcpyAsync(raw_out + (size_t) b * NO * V, c->di_logits + (size_t) b * c->di_Vpad, (size_t) NO * V * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

#define DIA_LOOP_CHUNK 16
extern "C" int tts_hip_dia_generate(tts_hip_ctx *c, uint32_t n_utt, uint32_t max_gen, const tts_hip_dia_codes *codes, const tts_hip_sampling *sp, const float *uniforms,
                                    uint32_t *tokens_out, uint32_t *steps_out) {
    if (!c || !c->has_dia) return set_err("tts_hip_dia_generate: not a Dia context (tts_hip_dia_create)");
    if (!c->finalized || !c->weights_present) return set_err("tts_hip_dia_generate: context not finalized");
    if (!codes || !tokens_out || !steps_out) return set_err("tts_hip_dia_generate: null argument");
    if (n_utt == 0 || n_utt > (uint32_t) c->di_U) return set_err("tts_hip_dia_generate: %u utterances outside 1..%d (max_utterances)", n_utt, c->di_U);
    const int G = (int) c->dia.max_gen, NO = c->NO, V = c->di_V, U = (int) n_utt;
    if (max_gen == 0 || max_gen > (uint32_t) G) return set_err("tts_hip_dia_generate: max_gen %u outside 1..%d cached positions", max_gen, G);
    if (codes->max_delay >= max_gen) return set_err("tts_hip_dia_generate: max_gen %u must exceed max_delay %u", max_gen, codes->max_delay);
    if (codes->bos >= (uint32_t) V || codes->eos >= (uint32_t) V || codes->pad >= (uint32_t) V) return set_err("tts_hip_dia_generate: special ids outside the vocabulary %d", V);
    for (int u = 0; u < U; u++)
        if (!c->di_slot_encoded[(size_t) u]) return set_err("tts_hip_dia_generate: slot %d has not been encoded (tts_hip_dia_encode_slot)", u);
    if (sp) {
        if (V > SMP_VMAX) return set_err("tts_hip_dia_generate: output vocabulary %d > %d", V, SMP_VMAX);
        if (!(sp->temperature > 0.0f) || !(sp->top_p > 0.0f) || !(sp->repetition_penalty > 0.0f)) return set_err("tts_hip_dia_generate: temperature, top_p, repetition_penalty must be > 0");
        if (!uniforms) return set_err("tts_hip_dia_generate: null uniforms");
    }
    HIPCHK(hipSetDevice(c->device));
    const bool rep = sp && sp->repetition_penalty != 1.0f;
    if (sp) {
        CHK(stage_uniforms(c, uniforms, (size_t) max_gen * U * NO));
        CHK(stage_penalty(c, sp->repetition_penalty, (int) max_gen));
    }
    // loop state: ids = BOS everywhere, positions 0, countdown -1, nothing done; sampler::reset (sampler.cpp:71-80)
    {
        std::vector<uint32_t> ids((size_t) U * NO, codes->bos), zero((size_t) 3 * c->di_U, 0u), seq((size_t) 2 * U);
        for (int u = 0; u < U; u++) { zero[(size_t) u] = 0xFFFFFFFFu; seq[(size_t) 2 * u] = 2 * u; seq[(size_t) 2 * u + 1] = 2 * u + 1; }
        HIPCHK(hipMemcpyAsync(c->di_ids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemsetAsync(c->di_pos, 0, (size_t) 2 * U * 4, c->stream));
        HIPCHK(hipMemcpyAsync(c->di_seq, seq.data(), seq.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->di_loop, zero.data(), zero.size() * 4, hipMemcpyHostToDevice, c->stream));
        if (rep) {
            HIPCHK(hipMemsetAsync(c->d_last, 0xFF, (size_t) U * NO * 4, c->stream));
            HIPCHK(hipMemsetAsync(c->d_repc, 0, (size_t) U * NO * 4, c->stream));
        }
        HIPCHK(hipStreamSynchronize(c->stream));   // the vectors are locals
    }
    DiaLoopArgs la{};
    la.n_utt = U; la.n_out = NO;
    la.bos = codes->bos; la.eos = codes->eos; la.pad = codes->pad; la.max_delay = codes->max_delay; la.max_gen = max_gen;
    for (int i = 0; i < 16; i++) la.delay_pattern[i] = codes->delay_pattern[i];
    la.ids = c->di_ids; la.pos = c->di_pos;
    la.delay = (int32_t *) c->di_loop; la.done = c->di_loop + c->di_U; la.call = c->di_loop + 2 * c->di_U;
    la.tok = c->di_stok; la.hist = c->di_hist;
    auto one_step = [&](bool captured) -> int {
        hipLaunchKernelGGL(dia_prestep_kernel, dim3((U + 63) / 64), dim3(64), 0, c->stream, la);
        HIPCHK(hipGetLastError());
        CHK(dia_forward(c, U, G, captured));
        if (sp) {
            SampleArgs sa{};
            sa.logits = c->di_guided; sa.V = V; sa.n_out = NO; sa.R = U;
            sa.top_k = sp->top_k; sa.top_p = sp->top_p; sa.temperature = sp->temperature;
            sa.uniforms = c->d_uniforms; sa.row_step = la.call; sa.out = c->di_stok;
            if (rep) { sa.pen_table = c->d_pen; sa.pen_len = c->pen_len; sa.last_ids = c->d_last; sa.rep_counts = c->d_repc; }
            hipLaunchKernelGGL(sample_kernel, dim3(NO, U), dim3(256), 0, c->stream, sa);
        } else {
            hipLaunchKernelGGL(argmax_kernel, dim3(U * NO), dim3(256), 0, c->stream, (const float *) c->di_guided, V, c->di_stok);
        }
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(dia_poststep_kernel, dim3((U + 63) / 64), dim3(64), 0, c->stream, la);
        HIPCHK(hipGetLastError());
        return 0;
    };
    // everything the captured launches hold by value: a change drops the graph
    const int mode = sp ? 1 : 0;
    const void *pen = rep ? (const void *) c->d_pen : nullptr;
    const tts_hip_sampling spv = sp ? *sp : tts_hip_sampling{};
    auto &bk = c->di_baked;
    const bool same = bk.mode == mode && bk.U == n_utt && bk.max_gen == max_gen && memcmp(&bk.codes, codes, sizeof(*codes)) == 0 &&
                      (!sp || (bk.uni == c->d_uniforms && bk.pen == pen && memcmp(&bk.sp, &spv, sizeof(spv)) == 0));
    const int key = 9100001;
    if (!same) {
        auto it = c->graphs.find(key);
        if (it != c->graphs.end()) { (void) hipGraphExecDestroy(it->second); c->graphs.erase(it); }
        bk.mode = mode; bk.U = n_utt; bk.max_gen = max_gen; bk.codes = *codes; bk.uni = c->d_uniforms; bk.pen = pen; bk.sp = spv;
    }
    const bool use_graph = !(c->d.flags & TTS_HIP_FLAG_NO_GRAPH) && !c->prof;
    std::vector<uint32_t> done((size_t) U);
    uint32_t ran = 0;
    while (ran < max_gen + 1) {   // at most max_gen sampler calls, then one pre-step that ends the countdown
        const uint32_t chunk = std::min<uint32_t>(DIA_LOOP_CHUNK, max_gen + 1 - ran);
        if (use_graph) {
            auto it = c->graphs.find(key);
            if (it == c->graphs.end()) {
                // the first step runs eagerly (per-kernel attributes are set outside a capture), the capture follows
                CHK(one_step(false));
                HIPCHK(hipStreamSynchronize(c->stream));
                hipGraph_t graph = nullptr;
                HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
                const int rc = one_step(true);
                const hipError_t e = hipStreamEndCapture(c->stream, &graph);
                if (rc != 0) { if (graph) (void) hipGraphDestroy(graph); return rc; }
                if (e != hipSuccess) return set_err("hipStreamEndCapture: %s", hipGetErrorString(e));
                hipGraphExec_t exec = nullptr;
                HIPCHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
                (void) hipGraphDestroy(graph);
                it = c->graphs.emplace(key, exec).first;
                for (uint32_t s = 1; s < chunk; s++) HIPCHK(hipGraphLaunch(it->second, c->stream));
            } else {
                for (uint32_t s = 0; s < chunk; s++) HIPCHK(hipGraphLaunch(it->second, c->stream));
            }
        } else {
            for (uint32_t s = 0; s < chunk; s++) CHK(one_step(false));
        }
        ran += chunk;
        HIPCHK(hipMemcpyAsync(done.data(), la.done, (size_t) U * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        bool all = true;
        for (int u = 0; u < U; u++) all = all && done[(size_t) u] != 0;
        if (all) break;
    }
    std::vector<uint32_t> pos((size_t) 2 * U);
    HIPCHK(hipMemcpyAsync(pos.data(), c->di_pos, pos.size() * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(tokens_out, c->di_hist, (size_t) U * max_gen * NO * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int u = 0; u < U; u++) steps_out[u] = pos[(size_t) 2 * u];
    return 0;
}

extern "C" int tts_hip_dia_step(tts_hip_ctx *c, const uint32_t *ids, uint32_t pos, float *logits_out, float *raw_out) {
    return tts_hip_dia_step_batch(c, 1, nullptr, ids, &pos, logits_out, raw_out);
}

// ------------------------------------------------------------------------------------------------

// shim_qtile.hip — the LDS-tiled integer block GEMM for many rows on GGUF-quantised matrices (qgemm_tile_kernels.h): kernel instances and launchers.
// A unit of its own because of two code-generation switches the rest of the shim does not want (tts.cpp_amd/build.py):
//   -fno-slp-vectorize                  the scaling step is 48 scalar fp32 instructions per MFMA; SLP packs them into v_pk_mul / v_pk_add / v_pk_fma_f32,
//                                       which gfx950 runs no faster than the scalar forms (two passes each) and slower next to MFMAs
//                                       (profiles/valu_rate.hip, MI355X_MICROARCH.md "packed f32 VALU ... an anti-lever beside MFMAs")
//   -mllvm -amdgpu-mfma-vgpr-form=1     MFMA results straight into VGPRs: the default allocation put them into AGPRs and moved all 16 results of
//                                       every MFMA through v_accvgpr_read (27 extra instructions per MFMA)
#include "shim_internal.h"
#include "qgemm_tile_kernels.h"
#include "shim_qtile.h"



const QTileShape QTILE_SHAPES[N_QTILE_SHAPES] = {{64, 64, 256}, {128, 128, 512}, {128, 64, 512}, {64, 128, 512}, {64, 64, 512}};   // the last one: two k groups

template <int BM, int BN, int WM, int WN, int S, int EPI, int WPE, int KG = 1>
static int launch_shape(tts_hip_ctx *c, const QTileArgs &qa, const TileMap &tm) {
    const int total = tm.m_tiles * tm.n_tiles * tm.k_slices;
    const int grid = (total + 7) / 8 * 8;
    size_t lds = (size_t) S * (BM + BN) * 144;
    if (EPI == EPI_CROSS) lds = std::max(lds, (size_t) (BM * 68 + 2 * 32 * 64) * 4);
    if (KG > 1) lds = std::max(lds, (size_t) (KG - 1) * (BM / 32) * (BN / 32) * 4096);   // the k groups' sums
    static std::atomic<uint64_t> attr{0};
    if (attr_needed(attr, c->device))
        HIPCHK(hipFuncSetAttribute((const void *) qgemm_tile_kernel<BM, BN, WM, WN, S, EPI, WPE, false, false, KG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((qgemm_tile_kernel<BM, BN, WM, WN, S, EPI, WPE, false, false, KG>), dim3(grid), dim3(WM * WN * KG * 64), lds, c->stream, qa, tm);
    HIPCHK(hipGetLastError());
    return 0;
}

template <int EPI>
static int launch_epi(tts_hip_ctx *c, const QTileArgs &qa, int shape, int ks) {
    const QTileShape &t = QTILE_SHAPES[shape];
    const TileMap tm{(qa.g.R + t.BM - 1) / t.BM, (qa.g.N + t.BN - 1) / t.BN, ks};
    switch (shape) {
        case 0: return launch_shape<64, 64, 2, 2, 2, EPI, 4>(c, qa, tm);
        case 1: return launch_shape<128, 128, 2, 4, 2, EPI, 4>(c, qa, tm);
        case 2: return launch_shape<128, 64, 4, 2, 2, EPI, 4>(c, qa, tm);
        case 3: return launch_shape<64, 128, 2, 4, 2, EPI, 4>(c, qa, tm);
        default: return launch_shape<64, 64, 2, 2, 2, EPI, 4, 2>(c, qa, tm);
    }
}

int launch_qtile(tts_hip_ctx *c, const QTileArgs &qa, int epi, int shape, int ks) {
    if (shape < 0 || shape >= N_QTILE_SHAPES) return set_err("qgemm_tile: shape %d", shape);
    const int kc = qa.g.kchunk ? qa.g.kchunk : qa.g.K;
    if (kc % 128 || qa.g.K % 32 || (ks > 1 && qa.g.K / ks != kc)) return set_err("qgemm_tile: K %d in %d slices", qa.g.K, ks);
    if ((qa.ldr & 3) || (qa.ldw & 3)) return set_err("qgemm_tile: scale strides %d / %d", qa.ldw, qa.ldr);
    if (epi == EPI_CROSS) {
        if (QTILE_SHAPES[shape].BN != 64 || ks != 1) return set_err("qgemm_tile: the cross-attention epilogue needs one head per tile column");
        if (shape == 0) return launch_shape<64, 64, 2, 2, 2, EPI_CROSS, 4>(c, qa, TileMap{(qa.g.R + 63) / 64, qa.g.N / 64, 1});
        if (shape == 4) return launch_shape<64, 64, 2, 2, 2, EPI_CROSS, 4, 2>(c, qa, TileMap{(qa.g.R + 63) / 64, qa.g.N / 64, 1});
        return launch_shape<128, 64, 4, 2, 2, EPI_CROSS, 4>(c, qa, TileMap{(qa.g.R + 127) / 128, qa.g.N / 64, 1});
    }
    if (epi == EPI_STORE) return launch_epi<EPI_STORE>(c, qa, shape, ks);
    if (ks != 1) return set_err("qgemm_tile: k slices need the slab epilogue");
    if (epi == EPI_QKV) return launch_epi<EPI_QKV>(c, qa, shape, ks);
    if (epi == EPI_RESID) return launch_epi<EPI_RESID>(c, qa, shape, ks);
    if (epi == EPI_GELU) return launch_epi<EPI_GELU>(c, qa, shape, ks);
    return set_err("qgemm_tile: no kernel for epilogue %d", epi);
}

int qtile_transpose_scales(tts_hip_ctx *c, const W &w) {
    if (w.type != TTS_HIP_Q8I || !w.stoff) return 0;
    const int nb = (int) (w.K / 32);
    hipLaunchKernelGGL(transpose_scales_kernel, dim3((w.ldw + 255) / 256, nb), dim3(256), 0, c->stream, (const _Float16 *) (c->arena + w.soff), (float *) (c->arena + w.stoff),
                       (int) w.N, nb, w.ldw);
    HIPCHK(hipGetLastError());
    return 0;
}

int qtile_quant_rows(tts_hip_ctx *c, const float *x, int lda, int K, int R) {
    hipLaunchKernelGGL(quant_rows_q8t_kernel, dim3((K / 256 + 3) / 4, R), dim3(256), 0, c->stream, x, lda, K, c->aq, c->adT, c->ldr, R);
    HIPCHK(hipGetLastError());
    return 0;
}

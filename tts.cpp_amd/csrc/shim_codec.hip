// shim_codec.hip — the neural audio codecs: DAC (src/decoder/dac_model.cpp), SNAC (src/decoder/snac_model.cpp) and Kokoro's acoustic path,
// which runs its convolutions on the codec's MFMA conv launchers.
#include "shim_internal.h"

#include "dac_kernels.h"
#include "dac_b3_kernels.h"
#include "kokoro_kernels.h"

// DAC
// ------------------------------------------------------------------------------------------------
// [C][L] out of a device tensor whose rows are LS apart
static int dac_snapshot(tts_hip_ctx *c, int stage, const float *dev, size_t C, size_t L, size_t LS) {
    if (!c->debug) return 0;
    HIPCHK(hipStreamSynchronize(c->stream));
    std::vector<float> &v = c->dac_dbg[stage];
    v.resize(C * L);
    HIPCHK(hipMemcpy2DAsync(v.data(), L * 4, dev, LS * 4, L * 4, C, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// ---- MFMA tile selection (shared by the packer and the launchers) ---------------------------------
// input channels per LDS chunk of the fp32 k=1 kernels: 16 keeps the footprint at 40-45 KB (3-4 workgroups per CU);
// with 32 the 96- and 64-channel tiles needed 80-90 KB = ONE workgroup per CU, and a k=1 conv has only C/32 chunks to
// pipeline over, so its residual loads and stores ran with nothing to overlap (0.88 TB/s)
#define CI32_K1 16
// k=7: 4 channels per chunk (40 KB, 4 workgroups per CU) measured 3.7 % faster than 8 (75 KB, 2 per CU)
#ifndef CI32_T
#define CI32_T 8
#endif
#ifndef CI32_K7
#define CI32_K7 4
#endif
static int conv_tile(int cout, int K, int *CO_T, int *CI_T) {
    if (K != 7 && K != 1) return -1;
    *CI_T = K == 7 ? CI32_K7 : CI32_K1;
    if (cout % 128 == 0) { *CO_T = 128; return 0; }
    if (cout % 96 == 0 && cout % 64 != 0) { *CO_T = 96; return 1; }
    if (cout % 64 == 0) { *CO_T = 64; return 2; }
    return -1;
}
// the k = 1 conv of a residual unit at <= 192 channels goes through conv1x1_direct_kernel (96-channel tiles; fp32 tensors only)
static bool conv1_direct(const tts_hip_ctx *c, int cout, int cin) {
    return c->dac_conv1_direct && !c->dac_f16 && cout == cin && (cin == 96 || cin == 192);
}
static int convt_tile(int cout, int s, int *CO_T) {
    if (s == 8 && cout % 64 == 0) { *CO_T = 64; return 0; }
    if (s == 4 && cout % 64 == 0) { *CO_T = 64; return 1; }
    if (s == 2 && cout % 96 == 0) { *CO_T = 96; return 2; }
    if (s == 2 && cout % 64 == 0) { *CO_T = 64; return 3; }
    return -1;
}

static int pack_one(tts_hip_ctx *c, size_t w_off, int cout, int cin, int KT, int CO_T, int CI_T, bool transposed) {
    const int n_chunks = (cin + CI_T - 1) / CI_T;
    const size_t n = (size_t) ((cout + CO_T - 1) / CO_T) * n_chunks * KT * CI_T * CO_T;
    float *dst = nullptr;
    HIPCHK(hipMalloc((void **) &dst, n * 4));
    hipLaunchKernelGGL(pack_conv_w_kernel, dim3(1024), dim3(256), 0, c->stream, (const float *) (c->arena + w_off), dst, cout, cin, KT,
                       CO_T, CI_T, n_chunks, transposed ? 1 : 0);
    HIPCHK(hipGetLastError());
    c->packed[w_off] = dst;
    return 0;
}

// one-time re-layout of the DAC conv weights into MFMA LDS images (after the arena holds the weights,
// i.e. also after an RCCL broadcast filled it)
static int pack_one16(tts_hip_ctx *c, size_t w_off, int cout, int cin, int KT, int CO_T, int CI_T, bool transposed) {
    const int n_chunks = (cin + CI_T - 1) / CI_T;
    const size_t n = (size_t) ((cout + CO_T - 1) / CO_T) * n_chunks * KT * CI_T * CO_T;
    _Float16 *dst = nullptr;
    HIPCHK(hipMalloc((void **) &dst, n * 2));
    hipLaunchKernelGGL(pack_conv_w16_kernel, dim3(1024), dim3(256), 0, c->stream, (const float *) (c->arena + w_off), dst, cout, cin, KT,
                       CO_T, CI_T, n_chunks, transposed ? 1 : 0);
    HIPCHK(hipGetLastError());
    c->packed16[w_off] = dst;
    return 0;
}
// operand split of the plane kernels (dac_kernels.h): 0 = bf16 x 3 / six products, 1 = fp16 hi + lo / three products (F32 tensors), 2 = fp16 / one product (F16 tensors)
static int dac_scheme(const tts_hip_ctx *c) { return c->dac_f16 ? 2 : (c->dac_split ? 1 : 0); }
// scheme: 0 = three bf16 planes, 1 = fp16 hi + lo (conv1d_mfma_b3_kernel<.., SplitB3 / SplitH2>)
static int b3_scheme(const tts_hip_ctx *c) { return c->has_kokoro ? (c->kk_split ? 1 : 0) : (c->dac_split ? 1 : 0); }
static int pack_one_b3(tts_hip_ctx *c, size_t w_off, int cout, int cin, int CO_T, int KT = 7) {   // 64- or 96-channel tiles, 8 input channels per chunk, tap pairs
    const int n_chunks = (cin + 7) / 8, scheme = b3_scheme(c);
    const size_t n = (size_t) ((cout + CO_T - 1) / CO_T) * n_chunks * split_planes(scheme) * (2 * ((KT + 1) / 2)) * CO_T * 8;
    __bf16 *dst = nullptr;
    HIPCHK(hipMalloc((void **) &dst, n * 2));
    hipLaunchKernelGGL(pack_conv_w_b3_kernel, dim3(1024), dim3(256), 0, c->stream, (const float *) (c->arena + w_off), dst, cout, cin, CO_T, n_chunks, KT, scheme);
    HIPCHK(hipGetLastError());
    c->packed_b3[w_off] = dst;
    return 0;
}
// residual units of 96 / 192 channels as one launch (resunit_b3_kernel): k-steps per stage of the k = 7 / k = 1 part
static bool resunit_shape(int C, int *KS, int *KS2) {
    if (C == 96) { *KS = 4; *KS2 = 3; return true; }
    if (C == 192) { *KS = 2; *KS2 = 4; return true; }
    return false;
}
static int pack_resunit(tts_hip_ctx *c, const DRes &r, int C) {
    int KS = 0, KS2 = 0;
    if (!resunit_shape(C, &KS, &KS2)) return 0;
    __bf16 *dst = nullptr;
    if (c->dac_tap7) {   // resunit_t7_kernel: one tap per k-step, stages of {4, 3} / {2, 2, 2, 1} k-steps per 16-channel chunk
        const int MI = C / 32, SPC = MI == 3 ? 2 : 4, MAXCNT = MI == 3 ? 4 : 2;
        const size_t WST = (size_t) split_planes(dac_scheme(c)) * MAXCNT * 2 * C * 8;
        const size_t n = (size_t) ((C / 16) * SPC + (C / 96) * ((C / 16) / KS2) + 1) * WST;
        HIPCHK(hipMalloc((void **) &dst, n * 2));
        HIPCHK(hipMemsetAsync(dst, 0, n * 2, c->stream));
        hipLaunchKernelGGL(pack_resunit_t7_kernel, dim3(1024), dim3(256), 0, c->stream, (const float *) (c->arena + r.in_w), (const float *) (c->arena + r.out_w), dst, C, KS2, dac_scheme(c));
    } else {
        const ResUnitGeom g = resunit_geom(C, KS, KS2);
        const size_t n = (size_t) (g.n7 + g.n1 + 1) * g.WST;   // + 1: the prefetch of the stage after the last one stays inside the buffer
        HIPCHK(hipMalloc((void **) &dst, n * 2));
        HIPCHK(hipMemsetAsync(dst, 0, n * 2, c->stream));
        hipLaunchKernelGGL(pack_resunit_b3_kernel, dim3(1024), dim3(256), 0, c->stream, (const float *) (c->arena + r.in_w), (const float *) (c->arena + r.out_w), dst, C, KS, KS2);
    }
    HIPCHK(hipGetLastError());
    c->packed_ru[r.in_w] = dst;
    return 0;
}
// transposed convs as bf16 x 3 products (convt_b3_kernel): 32 MI output channels per workgroup
static int convt_b3_tile(int cout, int cin, int s) {
    if (cin % 16) return 0;
    if (s == 8 && cout % 32 == 0) return 32;
    if (s == 4 && cout % 64 == 0) return 64;
    if (s == 2 && cout % 96 == 0) return 96;
    return 0;
}
static bool convt_b3_fits(int cout, int cin, int s) {   // the kernel's LDS request (weight stages + input planes + alpha) inside a CU's 160 KB (three planes: the largest scheme)
    const int t = convt_b3_tile(cout, cin, s);
    return t && (size_t) 6 * (2 * s * 2 * t * 8) * 2 + 6 * 2 * 257 * 8 * 2 + (size_t) cin * 8 <= 160 * 1024;
}
static int pack_convt_b3(tts_hip_ctx *c, const DBlock &b) {
    const int CO_T = convt_b3_tile(b.cout, b.cin, b.stride);
    if (!CO_T) return 0;
    const int n_chunks = b.cin / 16;
    const size_t n = (size_t) (b.cout / CO_T) * n_chunks * split_planes(dac_scheme(c)) * 2 * b.stride * 2 * CO_T * 8;
    __bf16 *dst = nullptr;
    HIPCHK(hipMalloc((void **) &dst, n * 2));
    hipLaunchKernelGGL(pack_convt_w_b3_kernel, dim3(1024), dim3(256), 0, c->stream, (const float *) (c->arena + b.w), dst, b.cout, b.cin, b.stride, CO_T, n_chunks, dac_scheme(c));
    HIPCHK(hipGetLastError());
    c->packed_ct[b.w] = dst;
    return 0;
}
// wide classes on split planes (conv_b3p_kernel): k = 7 in 64-channel tiles (four k-steps per 8-channel chunk), k = 1 in 128-channel
// tiles (one k-step = 16 channels per chunk)
static bool planes_class(const tts_hip_ctx *c, int ch) {
    int ks = 0, ks2 = 0;
    return c->dac_planes && c->dac_b3 && (!c->dac_f16 || c->dac_f16_planes) && ch % 128 == 0 && !(c->dac_fuse && resunit_shape(ch, &ks, &ks2));
}
static int pack_planes(tts_hip_ctx *c, size_t w_off, int cout, int cin, int KT) {
    const bool tapk = KT == 7 && c->dac_tap7 && cin % 16 == 0;
    const int CO_T = KT == 7 ? 64 : (cout % 256 == 0 ? 256 : 128), NS = KT == 7 ? (tapk ? 7 : 4) : 1;
    const int n_chunks = KT == 7 && !tapk ? cin / 8 : cin / 16;
    const size_t n = (size_t) (cout / CO_T) * n_chunks * split_planes(dac_scheme(c)) * NS * 2 * CO_T * 8;
    __bf16 *dst = nullptr;
    HIPCHK(hipMalloc((void **) &dst, n * 2));
    hipLaunchKernelGGL(pack_conv_w_b3p_kernel, dim3(1024), dim3(256), 0, c->stream, (const float *) (c->arena + w_off), dst, cout, cin, KT, CO_T, NS, n_chunks, dac_scheme(c));
    HIPCHK(hipGetLastError());
    c->packed_p[w_off] = dst;
    return 0;
}
#define CI16_K7 16
#define CI16_K1 32
#define CI16_T  16

static int ensure_packed(tts_hip_ctx *c) {
    if (c->dac_packed || (c->d.flags & TTS_HIP_FLAG_VALU_GEMM)) return 0;
    int CO_T = 0, CI_T = 0;
    // F16 codec tensors (quantize --convert-dac-to-f16): since round 6 on the same plane / fused-unit / transposed kernels as F32 tensors, with one fp16
    // plane and one product (SplitH1: ggml's fp16 im2col x fp16 kernel — the activations a conv consumes rounded to fp16, exact products, fp32
    // accumulation); layers those kernels do not cover (other channel counts) keep the fp16 tile kernels of round 2 (packed16)
    const bool f16p = c->dac_f16 && c->dac_f16_planes && c->dac_b3 && c->dac_tap7;
    if (c->dac_f16 && !f16p) {
        if (conv_tile(c->d_c0, 7, &CO_T, &CI_T) >= 0) CHK(pack_one16(c, c->d_initw, c->d_c0, c->d_latent, 7, CO_T, CI16_K7, false));
        for (auto &b : c->dblocks) {
            if (convt_tile(b.cout, b.stride, &CO_T) >= 0) CHK(pack_one16(c, b.w, b.cout, b.cin, 2 * b.stride, CO_T, CI16_T, true));
            for (int r = 0; r < 3; r++) {
                if (conv_tile(b.cout, 7, &CO_T, &CI_T) >= 0) CHK(pack_one16(c, b.res[r].in_w, b.cout, b.cout, 7, CO_T, CI16_K7, false));
                if (conv_tile(b.cout, 1, &CO_T, &CI_T) >= 0) CHK(pack_one16(c, b.res[r].out_w, b.cout, b.cout, 1, CO_T, CI16_K1, false));
            }
        }
        HIPCHK(hipStreamSynchronize(c->stream));
        c->dac_packed = true;
        return 0;
    }
    // Every weight is packed in the ONE image the kernel that will run it stages (the images are as large as the weights: a second format per
    // tensor would be dead device memory in every context).
    if (planes_class(c, c->d_c0) && c->d_latent % 8 == 0) CHK(pack_planes(c, c->d_initw, c->d_c0, c->d_latent, 7));
    else if (f16p) { if (conv_tile(c->d_c0, 7, &CO_T, &CI_T) >= 0) CHK(pack_one16(c, c->d_initw, c->d_c0, c->d_latent, 7, CO_T, CI16_K7, false)); }
    else if (c->dac_b3 && c->d_c0 % 64 == 0) CHK(pack_one_b3(c, c->d_initw, c->d_c0, c->d_latent, 64));
    else if (conv_tile(c->d_c0, 7, &CO_T, &CI_T) >= 0) CHK(pack_one(c, c->d_initw, c->d_c0, c->d_latent, 7, CO_T, CI_T, false));
    for (auto &b : c->dblocks) {
        if (c->dac_convt_b3 && convt_b3_fits(b.cout, b.cin, b.stride)) CHK(pack_convt_b3(c, b));
        else if (f16p) { if (convt_tile(b.cout, b.stride, &CO_T) >= 0) CHK(pack_one16(c, b.w, b.cout, b.cin, 2 * b.stride, CO_T, CI16_T, true)); }
        else if (convt_tile(b.cout, b.stride, &CO_T) >= 0) CHK(pack_one(c, b.w, b.cout, b.cin, 2 * b.stride, CO_T, CI32_T, true));
        int ks = 0, ks2 = 0;
        for (int r = 0; r < 3; r++) {
            if (planes_class(c, b.cout)) {   // split planes: conv_b3p_kernel for both convs of the unit
                CHK(pack_planes(c, b.res[r].in_w, b.cout, b.cout, 7));
                CHK(pack_planes(c, b.res[r].out_w, b.cout, b.cout, 1));
                continue;
            }
            if (c->dac_fuse && resunit_shape(b.cout, &ks, &ks2)) {   // one launch per unit (dilations 1 / 3 / 9 all qualify)
                CHK(pack_resunit(c, b.res[r], b.cout));
                continue;
            }
            if (f16p) {
                if (conv_tile(b.cout, 7, &CO_T, &CI_T) >= 0) CHK(pack_one16(c, b.res[r].in_w, b.cout, b.cout, 7, CO_T, CI16_K7, false));
                if (conv_tile(b.cout, 1, &CO_T, &CI_T) >= 0) CHK(pack_one16(c, b.res[r].out_w, b.cout, b.cout, 1, CO_T, CI16_K1, false));
                continue;
            }
            if (c->dac_b3 && b.cout % 64 == 0) CHK(pack_one_b3(c, b.res[r].in_w, b.cout, b.cout, 64));
            else if (c->dac_b3 >= 2 && b.cout % 96 == 0) CHK(pack_one_b3(c, b.res[r].in_w, b.cout, b.cout, 96));
            else if (conv_tile(b.cout, 7, &CO_T, &CI_T) >= 0) CHK(pack_one(c, b.res[r].in_w, b.cout, b.cout, 7, CO_T, CI_T, false));
            if (conv1_direct(c, b.cout, b.cout)) {   // [cin][cout] for conv1x1_direct_kernel
                CHK(pack_one(c, b.res[r].out_w, b.cout, b.cout, 1, b.cout, CI32_K1, false));
                c->packed_direct.insert(b.res[r].out_w);
            } else if (conv_tile(b.cout, 1, &CO_T, &CI_T) >= 0) CHK(pack_one(c, b.res[r].out_w, b.cout, b.cout, 1, CO_T, CI_T, false));
        }
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    c->dac_packed = true;
    return 0;
}

// LDS request of a codec kernel: the tile buffers plus snake's alpha table — unless (keep_residency) the table would cost a resident workgroup.
// (Leaving LDS free under the codec for another context's decoder workgroups was measured twice and lost: profiles/r02)
static size_t dac_lds_request(const tts_hip_ctx *c, size_t base, size_t table, int *use_table, bool keep_residency = false) {
    (void) c;
    const size_t CU = 160 * 1024;
    *use_table = table ? 1 : 0;
    if (table && keep_residency && CU / (base + table) < CU / base) { *use_table = 0; return base; }
    return base + table;
}

template <int KT, int MI, int NI, int WM, int WN, int CI_T>
static int launch_conv_mfma(tts_hip_ctx *c, const ConvArgs &a_in, int nz) {
    constexpr int CO_T = 32 * MI * WM, T_T = 32 * NI * WN, WCH = KT * CI_T * CO_T;
    const int xw = T_T + (KT - 1) * a_in.dil;
    const int cin_pad = (a_in.cin + CI_T - 1) / CI_T * CI_T;
    ConvArgs a = a_in;
    const size_t lds = dac_lds_request(c, ((size_t) 2 * WCH + 2 * (size_t) ((CI_T * xw + 3) & ~3)) * 4, (a.alpha ? 2 * (size_t) cin_pad : 0) * 4, &a.alpha_tab);
    if (a.dil > 9) return set_err("conv1d_mfma: dilation %d > 9 unsupported", a.dil);
    static std::atomic<uint64_t> attr{0};
    if (attr_needed(attr, c->device)) {
        HIPCHK(hipFuncSetAttribute((const void *) conv1d_mfma_kernel<KT, MI, NI, WM, WN, CI_T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    const dim3 grid((a.L + T_T - 1) / T_T, (a.cout + CO_T - 1) / CO_T, nz);
    hipLaunchKernelGGL((conv1d_mfma_kernel<KT, MI, NI, WM, WN, CI_T>), grid, dim3(64 * WM * WN), lds, c->stream, a);
    HIPCHK(hipGetLastError());
    return 0;
}

// k = 7 conv as six bf16 MFMAs per product (experiment): 64 channels x 256 positions per workgroup of 4 waves,
// or 96 channels x 256 positions per workgroup of 8 waves
template <int MI, int NI, int WM, int WN, int KT, typename SP>
static int launch_conv_b3_s(tts_hip_ctx *c, const ConvArgs &a_in, int nz) {
    constexpr int CO_T = 32 * MI * WM, T_T = 32 * NI * WN, WPL = 2 * ((KT + 1) / 2) * CO_T * 8;
    const int xw = T_T + (KT - 1) * a_in.dil;
    const int cin_pad = (a_in.cin + 7) / 8 * 8;
    ConvArgs a = a_in;
    const size_t lds = dac_lds_request(c, ((size_t) 2 * SP::NPL * WPL + 2 * SP::NPL * (size_t) xw * 8) * 2, (a.alpha ? 2 * (size_t) cin_pad : 0) * 4, &a.alpha_tab, true);
    static std::atomic<uint64_t> attr{0};
    if (attr_needed(attr, c->device)) {
        HIPCHK(hipFuncSetAttribute((const void *) conv1d_mfma_b3_kernel<MI, NI, WM, WN, KT, SP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    const dim3 grid((a.L + T_T - 1) / T_T, (a.cout + CO_T - 1) / CO_T, nz);
    if (lds > 160 * 1024) return set_err("conv1d_mfma_b3: k = %d at dilation %d needs %zu bytes of LDS", KT, a.dil, lds);
    hipLaunchKernelGGL((conv1d_mfma_b3_kernel<MI, NI, WM, WN, KT, SP>), grid, dim3(64 * WM * WN), lds, c->stream, a);
    HIPCHK(hipGetLastError());
    return 0;
}
template <int MI, int NI, int WM, int WN, int KT = 7>
static int launch_conv_b3(tts_hip_ctx *c, const ConvArgs &a, int nz) {
    return b3_scheme(c) ? launch_conv_b3_s<MI, NI, WM, WN, KT, SplitH2>(c, a, nz) : launch_conv_b3_s<MI, NI, WM, WN, KT, SplitB3>(c, a, nz);
}

template <int KT, int MI, int NI, int WM, int WN, int CI_T>
static int launch_conv_mfma16(tts_hip_ctx *c, const ConvArgs &a_in, int nz) {
    constexpr int CO_T = 32 * MI * WM, T_T = 32 * NI * WN, WCH = KT * CI_T * CO_T, XS = CI_T + 8;
    const int xw = T_T + (KT - 1) * a_in.dil;
    const int cin_pad = (a_in.cin + CI_T - 1) / CI_T * CI_T;
    ConvArgs a = a_in;
    const size_t lds = dac_lds_request(c, ((size_t) 2 * WCH + 2 * (size_t) xw * XS) * 2, (a.alpha ? 2 * (size_t) cin_pad : 0) * 4, &a.alpha_tab);
    if (a.dil > 9) return set_err("conv1d_mfma16: dilation %d > 9 unsupported", a.dil);
    static std::atomic<uint64_t> attr{0};
    if (attr_needed(attr, c->device)) {
        HIPCHK(hipFuncSetAttribute((const void *) conv1d_mfma16_kernel<KT, MI, NI, WM, WN, CI_T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    const dim3 grid((a.L + T_T - 1) / T_T, (a.cout + CO_T - 1) / CO_T, nz);
    hipLaunchKernelGGL((conv1d_mfma16_kernel<KT, MI, NI, WM, WN, CI_T>), grid, dim3(64 * WM * WN), lds, c->stream, a);
    HIPCHK(hipGetLastError());
    return 0;
}

template <int S, int MI, int WM, int WN, int CI_T>
static int launch_convt_mfma16(tts_hip_ctx *c, const ConvTArgs &a_in, int nz) {
    constexpr int CO_T = 32 * MI * WM, TI_T = 32 * WN, WCH = CI_T * 2 * S * CO_T, XS = CI_T + 8;
    const int cin_pad = (a_in.cin + CI_T - 1) / CI_T * CI_T;
    ConvTArgs a = a_in;
    const size_t lds = dac_lds_request(c, ((size_t) 2 * WCH + 2 * (size_t) (TI_T + 1) * XS) * 2, (a.alpha ? 2 * (size_t) cin_pad : 0) * 4, &a.alpha_tab, true);
    static std::atomic<uint64_t> attr{0};
    if (attr_needed(attr, c->device)) {
        HIPCHK(hipFuncSetAttribute((const void *) convt1d_mfma16_kernel<S, MI, WM, WN, CI_T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    const dim3 grid((a.L + 1 + TI_T - 1) / TI_T, (a.cout + CO_T - 1) / CO_T, nz);  // ti runs 0..L inclusive
    hipLaunchKernelGGL((convt1d_mfma16_kernel<S, MI, WM, WN, CI_T>), grid, dim3(64 * WM * WN), lds, c->stream, a);
    HIPCHK(hipGetLastError());
    return 0;
}

// Row stride of a codec activation [C][L]: L is frames x 2^k, so rows packed back to back start a multiple of 2 KB (up to 496 KB) apart and the
// C rows a tile touches at the same position fall on few HBM channels.  An odd multiple of 256 B makes consecutive rows walk through
// all of them, whatever power of two the interleave is; the kernels take the stride (a.L / a.Lout) apart from the valid length
// (frames x mult) already.
int dac_row_stride(const tts_hip_ctx *c, int L) {
    (void) c;
    return L;   // rows padded to an odd multiple of 256 B measured no different (profiles/r02/dac_row_stride.log): the address hash already spreads them
}

struct DacBatch {
    int n = 1;                  // utterances (grid.z)
    const uint32_t *frames = nullptr;  // device [n]
    int mult = 1;               // valid length at this stage = frames[z] * mult
    double tot_frames = 0;      // sum of frames (for flop/byte accounting)
};

static int launch_conv(tts_hip_ctx *c, const DacBatch &bt, const float *x, int cin, int L, size_t w, size_t b, size_t alpha, bool has_alpha,
                       int cout, int K, int pad, int dil, const float *resid, bool do_tanh, float *y, size_t alpha_out = 0,
                       bool has_alpha_out = false, bool has_bias = true) {
    ConvArgs a{};
    a.x = x; a.w = (const float *) (c->arena + w); a.b = has_bias ? (const float *) (c->arena + b) : nullptr;
    a.alpha = has_alpha ? (const float *) (c->arena + alpha) : nullptr;
    a.alpha_out = has_alpha_out ? (const float *) (c->arena + alpha_out) : nullptr;
    a.resid = resid; a.y = y; a.cin = cin; a.cout = cout; a.L = L; a.dil = dil; a.pad = pad; a.do_tanh = do_tanh;
    a.frames = bt.frames; a.mult = bt.mult;
    a.x_f16 = c->dac_f16 ? 1 : 0;
    a.prio = K == 7 ? 2 : 0;   // waves raise their issue priority for the staging phase of a chunk (measured 1.2 % faster twice; MFMA phase: slower)
    const double Lv = bt.tot_frames * bt.mult;  // valid positions over the batch
    const double bytes = ((double) cin * Lv + (double) cout * Lv * (resid ? 2 : 1) + (double) cout * cin * K) * 4;
    CHK(prof_begin(c, cout == 1 ? TTS_HIP_K_DAC_FINAL : (K == 7 ? TTS_HIP_K_DAC_CONV7 : TTS_HIP_K_DAC_CONV1), bytes, 2.0 * cout * (double) cin * K * Lv));
    const bool valu = (c->d.flags & TTS_HIP_FLAG_VALU_GEMM) != 0;
    int CO_T = 0, CI_T = 0;
    const int cfg = valu ? -1 : conv_tile(cout, K, &CO_T, &CI_T);
    auto pk = c->packed.find(w);
    auto pk16 = c->packed16.find(w);
    if (!valu && cout == 1 && K == 7) {
        hipLaunchKernelGGL(conv1d_cout1_kernel, dim3((L + C1_T - 1) / C1_T, 1, bt.n), dim3(256), 0, c->stream, a);
        HIPCHK(hipGetLastError());
    } else if (!valu && K == 7 && c->dac_b3 && !c->dac_f16 && c->packed_b3.count(w) && dil <= 9) {
        a.w = (const float *) c->packed_b3[w];   // three bf16 planes (experiment)
        if (cout % 64 == 0) CHK((launch_conv_b3<2, 1, 1, 8>(c, a, bt.n)));   // 64 ch x 256 pos, 8 waves
        else CHK((launch_conv_b3<3, 1, 1, 8>(c, a, bt.n)));                                                  // 96 ch x 256 pos, 8 waves
    } else if (cfg >= 0 && c->dac_f16 && pk16 != c->packed16.end()) {
        a.w = (const float *) pk16->second;  // fp16 LDS images
        if (K == 7 && cfg == 0) CHK((launch_conv_mfma16<7, 2, 2, 2, 2, CI16_K7>(c, a, bt.n)));
        else if (K == 7 && cfg == 1) CHK((launch_conv_mfma16<7, 3, 2, 1, 4, CI16_K7>(c, a, bt.n)));
        else if (K == 7 && cfg == 2) CHK((launch_conv_mfma16<7, 2, 2, 1, 4, CI16_K7>(c, a, bt.n)));
        else if (K == 1 && cfg == 0) CHK((launch_conv_mfma16<1, 2, 2, 2, 2, CI16_K1>(c, a, bt.n)));
        else if (K == 1 && cfg == 1) CHK((launch_conv_mfma16<1, 3, 2, 1, 4, CI16_K1>(c, a, bt.n)));
        else CHK((launch_conv_mfma16<1, 2, 2, 1, 4, CI16_K1>(c, a, bt.n)));
    } else if (K == 1 && pk != c->packed.end() && c->packed_direct.count(w) && !a.alpha && !a.alpha_out && !do_tanh) {
        a.w = pk->second;
        static std::atomic<uint64_t> attr{0};
        if (attr_needed(attr, c->device)) {
            HIPCHK(hipFuncSetAttribute((const void *) conv1x1_direct_kernel<3, 2, 96, 96>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIPCHK(hipFuncSetAttribute((const void *) conv1x1_direct_kernel<6, 1, 192, 48>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
        if (cin == 96) hipLaunchKernelGGL((conv1x1_direct_kernel<3, 2, 96, 96>), dim3((L + 255) / 256, 1, bt.n), dim3(256), (size_t) 96 * 96 * 4, c->stream, a);
        else hipLaunchKernelGGL((conv1x1_direct_kernel<6, 1, 192, 48>), dim3((L + 127) / 128, 1, bt.n), dim3(256), (size_t) 48 * 192 * 4, c->stream, a);
        HIPCHK(hipGetLastError());
    } else if (cfg >= 0 && pk != c->packed.end()) {
        a.w = pk->second;
        // position tiles per channel-tile class as measured best in profiles/r02/dac_variants.log (the 96-channel class on 128-position tiles)
        if (K == 7 && cfg == 0) CHK((launch_conv_mfma<7, 2, 2, 2, 2, CI32_K7>(c, a, bt.n)));
        else if (K == 7 && cfg == 1) CHK((launch_conv_mfma<7, 3, 1, 1, 4, CI32_K7>(c, a, bt.n)));   // 96 ch x 128 pos
        else if (K == 7 && cfg == 2) CHK((launch_conv_mfma<7, 2, 2, 1, 4, CI32_K7>(c, a, bt.n)));
        else if (K == 1 && cfg == 0) CHK((launch_conv_mfma<1, 2, 2, 2, 2, CI32_K1>(c, a, bt.n)));
        else if (K == 1 && cfg == 1) CHK((launch_conv_mfma<1, 3, 2, 1, 4, CI32_K1>(c, a, bt.n)));
        else CHK((launch_conv_mfma<1, 2, 2, 1, 4, CI32_K1>(c, a, bt.n)));
    } else {
        const dim3 grid((L + CV_T - 1) / CV_T, (cout + CV_CO - 1) / CV_CO, bt.n);
        const size_t lds = ((size_t) CV_CI * (CV_T + (K - 1) * dil) + (size_t) CV_CI * K * CV_CO) * 4;
        if (K == 7) hipLaunchKernelGGL(conv1d_kernel<7>, grid, dim3(256), lds, c->stream, a);
        else if (K == 1) hipLaunchKernelGGL(conv1d_kernel<1>, grid, dim3(256), lds, c->stream, a);
        else return set_err("conv1d: kernel size %d unsupported", K);
        HIPCHK(hipGetLastError());
    }
    return prof_end(c);
}

template <int S, int MI, int WM, int WN, int CI_T>
static int launch_convt_mfma(tts_hip_ctx *c, const ConvTArgs &a_in, int nz) {
    constexpr int CO_T = 32 * MI * WM, TI_T = 32 * WN, WCH = CI_T * 2 * S * CO_T;
    const int cin_pad = (a_in.cin + CI_T - 1) / CI_T * CI_T;
    ConvTArgs a = a_in;
    const size_t lds = dac_lds_request(c, ((size_t) 2 * WCH + 2 * (size_t) ((CI_T * (TI_T + 1) + 3) & ~3)) * 4, (a.alpha ? 2 * (size_t) cin_pad : 0) * 4, &a.alpha_tab, true);   // two resident workgroups matter more to the transposed convs than the table
    static std::atomic<uint64_t> attr{0};
    if (attr_needed(attr, c->device)) {
        HIPCHK(hipFuncSetAttribute((const void *) convt1d_mfma_kernel<S, MI, WM, WN, CI_T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    const dim3 grid((a.L + 1 + TI_T - 1) / TI_T, (a.cout + CO_T - 1) / CO_T, nz);  // ti runs 0..L inclusive
    hipLaunchKernelGGL((convt1d_mfma_kernel<S, MI, WM, WN, CI_T>), grid, dim3(64 * WM * WN), lds, c->stream, a);
    HIPCHK(hipGetLastError());
    return 0;
}

template <int S, int MI, bool PL, typename SP>
static int launch_convt_b3_t(tts_hip_ctx *c, const ConvTArgs &a, int nz) {
    constexpr int CO_T = 32 * MI, WPL = 2 * S * 2 * CO_T * 8, xpl = 2 * 257 * 8;
    const size_t lds = (size_t) 2 * SP::NPL * WPL * 2 + (size_t) 2 * SP::NPL * xpl * 2 + (size_t) a.cin * 8;
    if (lds > 160 * 1024) return set_err("convt_b3: %d input channels need %zu bytes of LDS", a.cin, lds);
    static std::atomic<uint64_t> attr{0};
    if (attr_needed(attr, c->device)) {
        HIPCHK(hipFuncSetAttribute((const void *) convt_b3_kernel<S, MI, PL, SP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    ConvTArgs b = a;
    b.npos = (a.L + 1 + 255) / 256; b.nz = nz;   // ti runs 0..L inclusive
    b.nco = xcd_order(a.cout / CO_T, (double) a.cin * a.L * nz * 4, (double) a.cout * a.cin * 2 * S * 6);
    hipLaunchKernelGGL((convt_b3_kernel<S, MI, PL, SP>), dim3(xcd_grid(b.npos, a.cout / CO_T, b.nz)), dim3(512), lds, c->stream, b);
    HIPCHK(hipGetLastError());
    return 0;
}
template <int S, int MI, typename SP>
static int launch_convt_b3_s(tts_hip_ctx *c, const ConvTArgs &a, int nz) {
    return a.xp ? launch_convt_b3_t<S, MI, true, SP>(c, a, nz) : launch_convt_b3_t<S, MI, false, SP>(c, a, nz);
}
template <int S, int MI>
static int launch_convt_b3(tts_hip_ctx *c, const ConvTArgs &a, int nz) {
    const int sc = dac_scheme(c);
    return sc == 0 ? launch_convt_b3_s<S, MI, SplitB3>(c, a, nz) : sc == 1 ? launch_convt_b3_s<S, MI, SplitH2>(c, a, nz) : launch_convt_b3_s<S, MI, SplitH1>(c, a, nz);
}

static int launch_convt(tts_hip_ctx *c, ConvTArgs ta, size_t w_off, int nz) {
    const bool valu = (c->d.flags & TTS_HIP_FLAG_VALU_GEMM) != 0;
    const int s = ta.stride;
    auto pct = c->packed_ct.find(w_off);
    if (!valu && c->dac_convt_b3 && pct != c->packed_ct.end()) {
        ta.w = (const float *) pct->second;
        ta.x_f16 = 0;
        if (s == 8) return launch_convt_b3<8, 1>(c, ta, nz);
        if (s == 4) return launch_convt_b3<4, 2>(c, ta, nz);
        return launch_convt_b3<2, 3>(c, ta, nz);
    }
    int CO_T = 0;
    const int cfg = valu ? -1 : convt_tile(ta.cout, s, &CO_T);
    ta.x_f16 = c->dac_f16 ? 1 : 0;
    auto pk16 = c->packed16.find(w_off);
    if (cfg >= 0 && c->dac_f16 && pk16 != c->packed16.end()) {
        ta.w = (const float *) pk16->second;
        if (cfg == 0) return launch_convt_mfma16<8, 1, 2, 2, CI16_T>(c, ta, nz);
        if (cfg == 1) return launch_convt_mfma16<4, 2, 1, 4, CI16_T>(c, ta, nz);
        if (cfg == 2) return launch_convt_mfma16<2, 3, 1, 4, CI16_T>(c, ta, nz);
        return launch_convt_mfma16<2, 2, 1, 4, CI16_T>(c, ta, nz);
    }
    auto pk = c->packed.find(w_off);
    if (cfg >= 0 && pk != c->packed.end()) {
        ta.w = pk->second;
        if (cfg == 0) return launch_convt_mfma<8, 1, 2, 2, CI32_T>(c, ta, nz);
        if (cfg == 1) return launch_convt_mfma<4, 2, 1, 4, CI32_T>(c, ta, nz);
        if (cfg == 2) return launch_convt_mfma<2, 3, 1, 4, CI32_T>(c, ta, nz);
        return launch_convt_mfma<2, 2, 1, 4, CI32_T>(c, ta, nz);
    }
    const dim3 grid((ta.Lout + CV_T - 1) / CV_T, (ta.cout + CV_CO - 1) / CV_CO, nz);
    const size_t lds = ((size_t) CT_CI * ((CV_T + s - 1) / s + 2) + (size_t) CT_CI * 2 * s * CV_CO) * 4;
    hipLaunchKernelGGL(convt1d_kernel, grid, dim3(256), lds, c->stream, ta);
    HIPCHK(hipGetLastError());
    return 0;
}

// ---- wide classes on split planes --------------------------------------------------------------------------------------------------
static int launch_split(tts_hip_ctx *c, const DacBatch &bt, const float *x, int C, int LS, size_t alpha, bool has_alpha, __bf16 *yp) {
    SplitArgs a{};
    a.x = x; a.alpha = has_alpha ? (const float *) (c->arena + alpha) : nullptr; a.yp = yp; a.C = C; a.L = LS; a.frames = bt.frames; a.mult = bt.mult;
    const double Lv = bt.tot_frames * bt.mult;
    CHK(prof_begin(c, TTS_HIP_K_DAC_CONV1, (double) C * Lv * 10, 0));
    const int sc = dac_scheme(c);
    if (sc == 0) hipLaunchKernelGGL(snake_split_kernel<SplitB3>, dim3((LS + 255) / 256, C / 8, bt.n), dim3(256), 0, c->stream, a);
    else if (sc == 1) hipLaunchKernelGGL(snake_split_kernel<SplitH2>, dim3((LS + 255) / 256, C / 8, bt.n), dim3(256), 0, c->stream, a);
    else hipLaunchKernelGGL(snake_split_kernel<SplitH1>, dim3((LS + 255) / 256, C / 8, bt.n), dim3(256), 0, c->stream, a);
    HIPCHK(hipGetLastError());
    return prof_end(c);
}
template <int KT, int MI, int NI, int WM, int WN, int NS, int MINW, int NB, typename SP>
static int launch_conv_b3p_s(tts_hip_ctx *c, const PConvArgs &a, int nz) {
    constexpr int CO_T = 32 * MI * WM, T_T = 32 * NI * WN, WPL = NS * 2 * CO_T * 8, NCG = KT == 7 ? (NS == 7 ? 2 : 1) : 2 * NS;
    const int xw = T_T + (KT - 1) * a.dil;
    const size_t lds = (size_t) NB * (SP::NPL * WPL * 2 + (size_t) SP::NPL * NCG * xw * 8 * 2);
    static std::atomic<uint64_t> attr{0};
    if (attr_needed(attr, c->device)) {
        HIPCHK(hipFuncSetAttribute((const void *) conv_b3p_kernel<KT, MI, NI, WM, WN, NS, MINW, NB, SP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    PConvArgs b = a;
    b.npos = (a.L + T_T - 1) / T_T; b.nz = nz;
    b.nco = xcd_order(a.cout / CO_T, (double) a.cin * a.L * nz * 6, (double) a.cout * a.cin * KT * 6);
    hipLaunchKernelGGL((conv_b3p_kernel<KT, MI, NI, WM, WN, NS, MINW, NB, SP>), dim3(xcd_grid(b.npos, a.cout / CO_T, b.nz)), dim3(64 * WM * WN), lds, c->stream, b);
    HIPCHK(hipGetLastError());
    return 0;
}
template <int KT, int MI, int NI, int WM, int WN, int NS, int MINW, int NB = 2>
static int launch_conv_b3p_t(tts_hip_ctx *c, const PConvArgs &a, int nz) {
    const int sc = dac_scheme(c);
    if (sc == 0) return launch_conv_b3p_s<KT, MI, NI, WM, WN, NS, MINW, NB, SplitB3>(c, a, nz);
    if (sc == 1) return launch_conv_b3p_s<KT, MI, NI, WM, WN, NS, MINW, NB, SplitH2>(c, a, nz);
    return launch_conv_b3p_s<KT, MI, NI, WM, WN, NS, MINW, NB, SplitH1>(c, a, nz);
}
// conv on planes: y (fp32, may be NULL) and / or yp (planes with the consumer's snake, may be NULL)
static int launch_conv_planes(tts_hip_ctx *c, const DacBatch &bt, const __bf16 *xp, int cin, int LS, size_t w, size_t b, int cout, int K, int dil, const float *resid,
                              float *y, __bf16 *yp, size_t alpha_out, bool has_alpha_out) {
    PConvArgs a{};
    a.xp = xp; a.w = c->packed_p.at(w); a.b = (const float *) (c->arena + b); a.resid = resid; a.y = y; a.yp = yp;
    a.alpha_out = has_alpha_out ? (const float *) (c->arena + alpha_out) : nullptr;
    a.cin = cin; a.cout = cout; a.L = LS; a.dil = dil; a.pad = K == 7 ? 3 * dil : 0; a.frames = bt.frames; a.mult = bt.mult;
    const double Lv = bt.tot_frames * bt.mult;
    const double bytes = ((double) cin * Lv * 6 + (double) cout * Lv * ((resid ? 4 : 0) + (y ? 4 : 0) + (yp ? 6 : 0)) + (double) cout * cin * K * 6);
    CHK(prof_begin(c, K == 7 ? TTS_HIP_K_DAC_CONV7 : TTS_HIP_K_DAC_CONV1, bytes, 2.0 * cout * (double) cin * K * Lv));
    if (K == 7) {
        const bool tapk = c->dac_tap7 && cin % 16 == 0;
        // 64 ch x 256 pos, 4 waves.  One tap per k-step: one LDS buffer (73 KB, two workgroups per CU).  Measured and dropped (64-utterance pass,
        // k = 7 family, profiles/r03/tap7_call16.txt): 8 waves 49.9 ms (128 registers, spills), two LDS buffers 53.5 / 48.6 ms against 45.7.
        // Round 6, fp16 hi + lo planes (48.5 KB of LDS: a third workgroup per CU would fit): compiled for three waves per SIMD (<= 168 registers from 198) the
        // kernel spills 120 bytes per lane and the family takes 30.6 ms against 26.1 (profiles/CALLS_r06.md 107).
        if (tapk) CHK((launch_conv_b3p_t<7, 2, 2, 1, 4, 7, 2, 1>(c, a, bt.n)));
        else CHK((launch_conv_b3p_t<7, 2, 2, 1, 4, 4, 2>(c, a, bt.n)));                             // tap pairs (dac_tap7 = 0)
    } else if (cout % 256 == 0) {
        CHK((launch_conv_b3p_t<1, 4, 2, 2, 4, 1, 2>(c, a, bt.n)));                                  // 256 ch x 256 pos, 8 waves
    } else {
        CHK((launch_conv_b3p_t<1, 2, 4, 2, 4, 1, 2>(c, a, bt.n)));                                  // 128 ch x 512 pos, 8 waves (128 x 256: 21.9 against 20.4 ms per pass)
    }
    return prof_end(c);
}

// one residual unit (gnac.cpp:133-149) as one launch
template <int MI, int KS, int KS2>
static int launch_resunit_t(tts_hip_ctx *c, const ResUnitArgs &a, int nz) {
    constexpr int C = 32 * MI;
    const ResUnitGeom g = resunit_geom(C, KS, KS2);
    const int xw = 256 + 6 * a.dil;
    const size_t lds = (size_t) 2 * g.WST * 2 + (size_t) 6 * xw * 8 * 2 + (size_t) C * 24;
    static std::atomic<uint64_t> attr{0};
    if (attr_needed(attr, c->device)) {
        HIPCHK(hipFuncSetAttribute((const void *) resunit_b3_kernel<MI, KS, KS2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    hipLaunchKernelGGL((resunit_b3_kernel<MI, KS, KS2>), dim3((a.L + 255) / 256, 1, nz), dim3(512), lds, c->stream, a);
    HIPCHK(hipGetLastError());
    return 0;
}
template <int MI, int KS2, typename SP, bool WDMA, int VAR, int NI = 1>
static int launch_resunit_t7_s(tts_hip_ctx *c, const ResUnitArgs &a, int nz) {
    constexpr int C = 32 * MI, T_T = 256 * NI;
    const int xw = T_T + 6 * a.dil;
    const size_t WST = (size_t) SP::NPL * ResT7<MI>::MAXCNT * 2 * C * 8;
    const size_t lds = 2 * WST * 2 + (size_t) 2 * SP::NPL * 2 * xw * 8 * 2 + (size_t) C * 24;
    static std::atomic<uint64_t> attr{0};
    if (attr_needed(attr, c->device)) {
        HIPCHK(hipFuncSetAttribute((const void *) resunit_t7_kernel<MI, KS2, SP, WDMA, VAR, NI>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    hipLaunchKernelGGL((resunit_t7_kernel<MI, KS2, SP, WDMA, VAR, NI>), dim3((a.L + T_T - 1) / T_T, 1, nz), dim3(512), lds, c->stream, a);
    HIPCHK(hipGetLastError());
    return 0;
}
// round 6 default: weight stages by global_load_lds, the k = 1 conv's B operand made once (dac_wdma = 1, VAR 2), and at 96 channels two position tiles per
// wave (NI 2: 4.72 -> 4.47 ms per launch; at 192 channels the accumulators of two tiles do not fit); tune("dac_wdma") = 0: the round-5 form
template <int MI, int KS2>
static int launch_resunit_t7(tts_hip_ctx *c, const ResUnitArgs &a, int nz) {
    const int sc = dac_scheme(c);
    constexpr int NI = MI == 3 ? 2 : 1;
    if (c->dac_wdma)
        return sc == 0 ? launch_resunit_t7_s<MI, KS2, SplitB3, true, 0>(c, a, nz) : sc == 1 ? launch_resunit_t7_s<MI, KS2, SplitH2, true, 2, NI>(c, a, nz) : launch_resunit_t7_s<MI, KS2, SplitH1, true, 2, NI>(c, a, nz);
    return sc == 0 ? launch_resunit_t7_s<MI, KS2, SplitB3, false, 0>(c, a, nz) : sc == 1 ? launch_resunit_t7_s<MI, KS2, SplitH2, false, 0>(c, a, nz) : launch_resunit_t7_s<MI, KS2, SplitH1, false, 0>(c, a, nz);
}
static bool resunit_fused(const tts_hip_ctx *c, const DRes &r, int dil) {
    return c->dac_fuse && !(c->d.flags & TTS_HIP_FLAG_VALU_GEMM) && dil <= 9 && c->packed_ru.count(r.in_w);
}
static int launch_resunit(tts_hip_ctx *c, const DacBatch &bt, const DRes &r, int C, int LS, int dil, const float *x, float *y) {
    ResUnitArgs a{};
    a.x = x; a.y = y; a.w = c->packed_ru.at(r.in_w);
    a.b7 = (const float *) (c->arena + r.in_b); a.b1 = (const float *) (c->arena + r.out_b);
    a.alpha_in = (const float *) (c->arena + r.in_alpha); a.alpha_mid = (const float *) (c->arena + r.out_alpha);
    a.L = LS; a.dil = dil; a.pad = 3 * dil; a.frames = bt.frames; a.mult = bt.mult;
    const double Lv = bt.tot_frames * bt.mult;
    CHK(prof_begin(c, TTS_HIP_K_DAC_RESUNIT, (2.0 * C * Lv + 8.0 * C * C) * 4, 2.0 * C * (double) C * 8 * Lv));
    if (c->dac_tap7) {
        if (C == 96) CHK((launch_resunit_t7<3, 3>(c, a, bt.n)));
        else CHK((launch_resunit_t7<6, 4>(c, a, bt.n)));
    } else if (C == 96) CHK((launch_resunit_t<3, 4, 3>(c, a, bt.n)));
    else CHK((launch_resunit_t<6, 2, 4>(c, a, bt.n)));
    return prof_end(c);
}

// dac_runner::run for n utterances at once (grid.z = utterance, per-utterance lengths): the early blocks have
// few positions per utterance, so batching is what fills the 256 CUs there.
static int dac_decode_batch_on(tts_hip_ctx *c, const uint32_t *codes, const uint32_t *frames, uint32_t n, float *pcm_out);
// One codec pass carries at most `dac_group` utterances (TTS_HIP_DAC_GROUP, default 64): the activation buffers are sized for a group
// (3 x 197 KB per frame: 9.4 GB for 64 x 248 frames instead of 56 GB for a 384-utterance batch), and passes of different contexts on
// one device take turns (a per-device mutex): a pass fills the chip with compute-bound convolutions, two of them interleaved only
// stretch each other, while another context's latency-bound decoder loop does fit next to one.
static int dac_decode_batch(tts_hip_ctx *c, const uint32_t *codes, const uint32_t *frames, uint32_t n, float *pcm_out) {
    if (!c || !c->finalized || !c->has_dac) return set_err("tts_hip_dac_decode: context has no finalized DAC");
    if (!frames || !codes || !pcm_out) return set_err("tts_hip_dac_decode: null argument");
    const uint32_t G = (uint32_t) std::max(1, c->dac_group);
    size_t code_off = 0, pcm_off = 0;
    for (uint32_t g0 = 0; g0 < n; g0 += G) {
        const uint32_t m = std::min(G, n - g0);
        size_t fr = 0;
        for (uint32_t i = 0; i < m; i++) fr += frames[g0 + i];
        int rc;
        {
            std::lock_guard<std::mutex> lock(g_dac_pass_mutex[(unsigned) c->device % 64]);
            if (!c->dac_stream) {
                rc = dac_decode_batch_on(c, codes + code_off * c->d_ncb, frames + g0, m, pcm_out + pcm_off);
            } else {
                // the decoder stream is idle here (every decoder entry point synchronises before it returns)
                hipStream_t ar = c->stream;
                c->stream = c->dac_stream;
                rc = dac_decode_batch_on(c, codes + code_off * c->d_ncb, frames + g0, m, pcm_out + pcm_off);
                if (rc) (void) hipStreamSynchronize(c->stream);
                c->stream = ar;
            }
            // an error return must not hand the device's shared buffers to the next context while launches of this pass are still in flight
            if (rc) (void) hipStreamSynchronize(c->stream);
        }
        if (rc) return rc;
        code_off += fr;
        pcm_off += fr * (size_t) c->d_up;
    }
    return 0;
}

static int dac_decode_batch_on(tts_hip_ctx *c, const uint32_t *codes, const uint32_t *frames, uint32_t n, float *pcm_out) {
    if (!c->weights_present) return set_err("tts_hip_dac_decode: weights not present");
    if (!codes || !pcm_out || !frames) return set_err("tts_hip_dac_decode: null argument");
    HIPCHK(hipSetDevice(c->device));
    uint32_t Fmax = 0;
    size_t tot = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (frames[i] > c->d.dac_max_frames) return set_err("tts_hip_dac_decode: %u frames > max %u", frames[i], c->d.dac_max_frames);
        Fmax = std::max(Fmax, frames[i]);
        tot += frames[i];
    }
    if (Fmax == 0) return 0;  // empty response (cli.cpp:87-90 treats n_outputs==0 as the soft failure)
    for (size_t i = 0; i < tot * c->d_ncb; i++)
        if (codes[i] >= (uint32_t) c->d_cbsize) return set_err("tts_hip_dac_decode: code %u >= codebook size %d", codes[i], c->d_cbsize);
    c->dac_dbg.clear();
    CHK(ensure_packed(c));
    // buffers: n utterances, rows at the padded stride of the stage
    const size_t need_frames = (size_t) n * Fmax;
    size_t need_elems = 0, pcm_elems = 0;
    {
        int Lq = (int) Fmax;
        need_elems = (size_t) std::max(c->d_latent, c->d_c0) * dac_row_stride(c, Lq);
        // split planes (6 bytes per element) of the wide classes live in these buffers too
        if (c->packed_p.count(c->d_initw)) need_elems = std::max(need_elems, ((size_t) c->d_latent * dac_row_stride(c, Lq) * 3 + 1) / 2);
        for (auto &b : c->dblocks) {
            need_elems = std::max(need_elems, (size_t) b.cin * dac_row_stride(c, Lq));
            Lq = (Lq - 1) * b.stride - 2 * b.padding + 2 * b.stride;
            need_elems = std::max(need_elems, (size_t) b.cout * dac_row_stride(c, Lq));
            if (c->packed_p.count(b.res[0].in_w)) need_elems = std::max(need_elems, ((size_t) b.cout * dac_row_stride(c, Lq) * 3 + 1) / 2);
        }
        need_elems *= n;
        pcm_elems = (size_t) n * dac_row_stride(c, Lq);
    }
    // the device's codec buffers (the caller holds the device's pass lock)
    DacBuffers &B = g_dac_buffers[(unsigned) c->device % 64];
    if (!c->dac_buf_user) { c->dac_buf_user = true; B.users++; }
    if (need_frames * c->d_ncb > B.cap_codes || need_elems > B.dbuf_elems || pcm_elems > B.h_pcm_elems || (!c->packed_p.empty() && !B.dplanes)) {
        HIPCHK(hipStreamSynchronize(c->stream));
        const size_t want_elems = std::max(need_elems, B.dbuf_elems), want_codes = std::max(need_frames * c->d_ncb, B.cap_codes), want_pcm = std::max(pcm_elems, B.h_pcm_elems);
        auto drop = [&]() {
            for (int i = 0; i < 3; i++) { free_dev(B.dbuf[i]); B.dbuf[i] = nullptr; }
            free_dev(B.dplanes); B.dplanes = nullptr;
            free_dev(B.d_codes); B.d_codes = nullptr;
            if (B.h_pcm) { (void) hipHostFree(B.h_pcm); B.h_pcm = nullptr; }
            B.dbuf_elems = B.cap_codes = B.h_pcm_elems = 0;   // nothing is held: the next pass of any context of this device allocates again
        };
        drop();
        hipError_t e = hipSuccess;
        for (int i = 0; i < 3 && e == hipSuccess; i++) e = hipMalloc((void **) &B.dbuf[i], want_elems * 4);
        if (e == hipSuccess && !c->packed_p.empty()) e = hipMalloc((void **) &B.dplanes, want_elems * 4);   // 6 bytes per element of the widest planes class <= a dbuf
        if (e == hipSuccess) e = hipMalloc((void **) &B.d_codes, want_codes * 4);
        if (e == hipSuccess) e = hipHostMalloc((void **) &B.h_pcm, want_pcm * 4);
        if (e != hipSuccess) {
            drop();
            return set_err("tts_hip_dac_decode: codec buffers of %zu floats x %d + %zu ids: %s", want_elems, c->packed_p.empty() ? 3 : 4, want_codes, hipGetErrorString(e));
        }
        B.dbuf_elems = want_elems; B.cap_codes = want_codes; B.h_pcm_elems = want_pcm;   // capacities are recorded only once every buffer exists
    }
    if (n > c->d_frames_cap) {
        free_dev(c->d_frames);
        HIPCHK(hipMalloc((void **) &c->d_frames, (size_t) n * 4));
        c->d_frames_cap = n;
    }
    HIPCHK(hipMemcpyAsync(c->d_frames, frames, (size_t) n * 4, hipMemcpyHostToDevice, c->stream));
    {   // codes padded to [n][Fmax][n_cb]
        size_t off = 0;
        for (uint32_t i = 0; i < n; i++) {
            if (frames[i]) HIPCHK(hipMemcpyAsync(B.d_codes + (size_t) i * Fmax * c->d_ncb, codes + off * c->d_ncb, (size_t) frames[i] * c->d_ncb * 4,
                                                 hipMemcpyHostToDevice, c->stream));
            off += frames[i];
        }
    }
    DacBatch bt;
    bt.n = (int) n; bt.frames = c->d_frames; bt.mult = 1; bt.tot_frames = (double) tot;
    int L = (int) Fmax;                      // longest utterance at this stage
    int LS = dac_row_stride(c, L);           // row stride of this stage's activations
    float *cur = B.dbuf[0], *t1 = B.dbuf[1], *t2 = B.dbuf[2];

    DacEmbedArgs ea{};
    ea.frames = c->d_frames;
    ea.codes = B.d_codes; ea.codebook = (const float *) (c->arena + c->d_codebook); ea.proj_w = (const float *) (c->arena + c->d_projw);
    ea.proj_b = (const float *) (c->arena + c->d_projb); ea.n_cb = c->d_ncb; ea.cb_size = c->d_cbsize; ea.cb_dim = c->d_cbdim;
    ea.latent = c->d_latent; ea.T = L; ea.Tout = LS; ea.out = cur; ea.x_f16 = c->dac_f16 ? 1 : 0;
    CHK(prof_begin(c, TTS_HIP_K_DAC_EMBED, (double) c->d_latent * tot * 4, 2.0 * c->d_latent * tot * c->d_ncb * c->d_cbdim));
    if (c->d_ncb == 9 && c->d_cbdim == 8)
        hipLaunchKernelGGL((dac_embed_tile_kernel<9, 8>), dim3((L + 63) / 64, (c->d_latent + 4 * EMB_CH - 1) / (4 * EMB_CH), n), dim3(256), 0, c->stream, ea);
    else
        hipLaunchKernelGGL(dac_embed_kernel, dim3((L + 63) / 64, c->d_latent, n), dim3(64), 0, c->stream, ea);
    HIPCHK(hipGetLastError());
    CHK(prof_end(c));
    if (n == 1) CHK(dac_snapshot(c, 0, cur, (size_t) c->d_latent, (size_t) L, (size_t) LS));

    // A transposed conv on the bf16 x 3 kernel takes its input as split planes when the producer is a planes conv: the producer's epilogue applies
    // this layer's snake and the split once per element, where the fp32 form redoes both in every one of the cout / 32 channel-tile workgroups.
    auto convt_takes_planes = [&](size_t bi) {
        if (bi >= c->dblocks.size() || !c->dac_convt_planes || (c->d.flags & TTS_HIP_FLAG_VALU_GEMM)) return false;
        const DBlock &nb = c->dblocks[bi];
        return c->dac_convt_b3 && c->packed_ct.count(nb.w) != 0 && nb.cin % 16 == 0;
    };
    const __bf16 *convt_in = nullptr;    // planes of the next transposed conv's input, when its producer wrote them
    if (c->packed_p.count(c->d_initw) && !(c->d.flags & TTS_HIP_FLAG_VALU_GEMM)) {
        // quantizer output -> split planes (no snake in front of the first conv) -> k = 7 conv on planes -> the first transposed conv's planes
        // (snaked with ITS alpha), or fp32 for it
        CHK(launch_split(c, bt, cur, c->d_latent, LS, 0, false, (__bf16 *) t2));
        const bool pl = convt_takes_planes(0);
        CHK(launch_conv_planes(c, bt, (const __bf16 *) t2, c->d_latent, LS, c->d_initw, c->d_initb, c->d_c0, 7, 1, nullptr, (!pl || (c->debug && n == 1)) ? t1 : nullptr,
                               pl ? (__bf16 *) B.dplanes : nullptr, pl ? c->dblocks[0].alpha : 0, pl));
        if (pl) convt_in = (const __bf16 *) B.dplanes;
    } else
    CHK(launch_conv(c, bt, cur, c->d_latent, LS, c->d_initw, c->d_initb, 0, false, c->d_c0, 7, 3, 1, nullptr, false, t1));
    std::swap(cur, t1);
    if (n == 1) CHK(dac_snapshot(c, 1, cur, (size_t) c->d_c0, (size_t) L, (size_t) LS));

    int C = c->d_c0;
    for (size_t bi = 0; bi < c->dblocks.size(); bi++) {
        const DBlock &b = c->dblocks[bi];
        const int Lout = (L - 1) * b.stride - 2 * b.padding + 2 * b.stride, LSout = dac_row_stride(c, Lout);
        ConvTArgs ta{};
        ta.x = cur; ta.w = (const float *) (c->arena + b.w); ta.b = (const float *) (c->arena + b.b);
        ta.alpha = (const float *) (c->arena + b.alpha); ta.y = t1; ta.cin = b.cin; ta.cout = b.cout; ta.L = LS;
        ta.Lout = LSout; ta.stride = b.stride; ta.pad = b.padding;
        ta.frames = c->d_frames; ta.mult = bt.mult;
        ta.xp = convt_in;
        convt_in = nullptr;
        const double Lov = bt.tot_frames * bt.mult * b.stride;
        CHK(prof_begin(c, TTS_HIP_K_DAC_CONVT, ((double) b.cin * bt.tot_frames * bt.mult + (double) b.cout * Lov + (double) b.cin * b.cout * 2 * b.stride) * 4,
                       2.0 * b.cin * (double) b.cout * 2 * Lov));
        CHK(launch_convt(c, ta, b.w, (int) n));
        CHK(prof_end(c));
        std::swap(cur, t1);
        L = Lout; LS = LSout; C = b.cout;
        bt.mult *= b.stride;
        if (c->packed_p.count(b.res[0].in_w) && !(c->d.flags & TTS_HIP_FLAG_VALU_GEMM)) {
            // a wide class on split planes: the activation a conv consumes is written by its producer already snaked (the consumer's alpha)
            // and split; the fp32 tensor exists only where the residual add and the next transposed conv need it.
            //   X (cur, fp32) --split(snake in_alpha 0)--> PA ;  k7(PA) -> PB = split(snake out_alpha) ;  k1(PB) + X -> X' (fp32) [+ PA for the next unit]
            __bf16 *PA = (__bf16 *) t2, *PB = (__bf16 *) B.dplanes;
            CHK(launch_split(c, bt, cur, C, LS, b.res[0].in_alpha, true, PA));
            for (int r = 0; r < 3; r++) {
                int dil = 1;
                for (int e = 0; e < r; e++) dil *= 3;
                CHK(launch_conv_planes(c, bt, PA, C, LS, b.res[r].in_w, b.res[r].in_b, C, 7, dil, nullptr, nullptr, PB, b.res[r].out_alpha, true));
                // the unit's output: fp32 for the residual stream, plus the planes of whoever consumes it next — the next unit's k = 7 conv, or
                // (last unit) the next block's transposed conv; then the fp32 tensor is written only when the debug snapshot wants it
                const bool to_convt = r == 2 && convt_takes_planes(bi + 1);
                const bool want_y = r < 2 || !to_convt || (c->debug && n == 1);
                CHK(launch_conv_planes(c, bt, PB, C, LS, b.res[r].out_w, b.res[r].out_b, C, 1, 1, cur, want_y ? t1 : nullptr, (r < 2 || to_convt) ? PA : nullptr,
                                       r < 2 ? b.res[r + 1].in_alpha : (to_convt ? c->dblocks[bi + 1].alpha : 0), r < 2 || to_convt));
                if (to_convt) convt_in = PA;
                if (want_y) std::swap(cur, t1);
            }
            if (n == 1) CHK(dac_snapshot(c, 2 + (int) bi, cur, (size_t) C, (size_t) L, (size_t) LS));
            continue;
        }
        for (int r = 0; r < 3; r++) {  // build_residual_unit: dilation 3^r, padding 3^(r+1) (gnac.h:44-48)
            int dil = 1;
            for (int e = 0; e < r; e++) dil *= 3;
            // snake(out_alpha) of the k=1 conv's input is applied in the k=7 conv's epilogue (same arithmetic, once
            // per element instead of once per output-channel tile)
            if (resunit_fused(c, b.res[r], dil)) {
                CHK(launch_resunit(c, bt, b.res[r], C, LS, dil, cur, t1));
                std::swap(cur, t1);
                continue;
            }
            CHK(launch_conv(c, bt, cur, C, LS, b.res[r].in_w, b.res[r].in_b, b.res[r].in_alpha, true, C, 7, 3 * dil, dil, nullptr, false, t1,
                            b.res[r].out_alpha, true));
            CHK(launch_conv(c, bt, t1, C, LS, b.res[r].out_w, b.res[r].out_b, 0, false, C, 1, 0, 1, cur, false, t2));
            std::swap(cur, t2);
        }
        if (n == 1) CHK(dac_snapshot(c, 2 + (int) bi, cur, (size_t) C, (size_t) L, (size_t) LS));
    }
    CHK(launch_conv(c, bt, cur, C, LS, c->d_fw, c->d_fb, c->d_falpha, true, 1, 7, 3, 1, nullptr, true, t1));
    HIPCHK(hipMemcpyAsync(B.h_pcm, t1, (size_t) n * LS * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    size_t off = 0;
    for (uint32_t i = 0; i < n; i++) {
        memcpy(pcm_out + off, B.h_pcm + (size_t) i * LS, (size_t) frames[i] * c->d_up * 4);
        off += (size_t) frames[i] * c->d_up;
    }
    return 0;
}

extern "C" int tts_hip_dac_decode(tts_hip_ctx *c, const uint32_t *codes, uint32_t frames, float *pcm_out) {
    return dac_decode_batch(c, codes, &frames, 1, pcm_out);
}

extern "C" int tts_hip_dac_decode_batch(tts_hip_ctx *c, const uint32_t *codes, const uint32_t *frames, uint32_t n, float *pcm_out) {
    if (n == 0) return 0;
    return dac_decode_batch(c, codes, frames, n, pcm_out);
}

// ------------------------------------------------------------------------------------------------
// Kokoro (src/models/kokoro/model.cpp:938-1047, 1141-1242, 195-244); first version, see kokoro_kernels.h
// ------------------------------------------------------------------------------------------------
extern "C" tts_hip_ctx *tts_hip_kokoro_create(int device, const tts_hip_kokoro_desc *kd) {
    if (!kd || kd->struct_size != sizeof(tts_hip_kokoro_desc)) { set_err("tts_hip_kokoro_create: bad desc (struct_size mismatch)"); return nullptr; }
    if (kd->n_upsamples == 0 || kd->n_upsamples > 4 || kd->n_kernels == 0 || kd->n_upsamples * kd->n_kernels > 16 || kd->n_fft < 2 || kd->hop == 0 || kd->max_ctx < 3) {
        set_err("tts_hip_kokoro_create: generator geometry out of range");
        return nullptr;
    }
    tts_hip_desc d{};
    d.struct_size = sizeof(d);
    d.max_seqs = 1;
    d.flags = TTS_HIP_FLAG_NO_PARLER | TTS_HIP_FLAG_NO_DAC;
    tts_hip_ctx *c = tts_hip_create(device, &d);
    if (!c) return nullptr;
    c->has_kokoro = true;
    c->ko = *kd;
    if (hipMalloc((void **) &c->kk_stuck, 4) != hipSuccess || hipMemset(c->kk_stuck, 0, 4) != hipSuccess) { set_err("tts_hip_kokoro_create: hipMalloc failed"); tts_hip_destroy(c); return nullptr; }
    return c;
}

namespace {
// scratch for one call: pieces of a context-owned pool that grows to the largest call seen (a synthesis asks for ~100 buffers: one
// hipMalloc / hipFree pair each cost more host time than the kernels they fed); what does not fit yet is allocated for this call alone
struct KScratch {
    tts_hip_ctx *c;
    size_t off = 0, want = 0;
    std::vector<void *> extra;
    bool failed = false;
    explicit KScratch(tts_hip_ctx *c_) : c(c_) {
        if (c->kk_pool_next > c->kk_pool_cap) {
            (void) hipStreamSynchronize(c->stream);
            if (c->kk_pool) (void) hipFree(c->kk_pool);
            c->kk_pool = nullptr; c->kk_pool_cap = 0;
            const size_t cap = c->kk_pool_next + c->kk_pool_next / 4;
            if (hipMalloc((void **) &c->kk_pool, cap) == hipSuccess) c->kk_pool_cap = cap;
        }
    }
    ~KScratch() {
        if (!extra.empty()) (void) hipStreamSynchronize(c->stream);
        for (void *p : extra) (void) hipFree(p);
        if (want > c->kk_pool_cap) c->kk_pool_next = std::max(c->kk_pool_next, want);
    }
    float *f(size_t n) {
        const size_t bytes = ((n ? n : 1) * sizeof(float) + 255) & ~(size_t) 255;
        want += bytes;
        if (c->kk_pool && off + bytes <= c->kk_pool_cap) {
            float *p = (float *) (c->kk_pool + off);
            off += bytes;
            return p;
        }
        void *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) { failed = true; return nullptr; }
        extra.push_back(p);
        return (float *) p;
    }
};
inline dim3 kgrid(int64_t n, int bs = 256) { return dim3((unsigned) ((n + bs - 1) / bs)); }

struct KRun {
    tts_hip_ctx *c;
    KScratch &s;
    std::string err;
    hipStream_t st;
    KRun(tts_hip_ctx *c_, KScratch &s_) : c(c_), s(s_), st(c_->stream) {}
    bool has(const std::string &n) const { return c->k_tensors.count("kokoro." + n) != 0; }
    const float *w(const std::string &n, int64_t *ne = nullptr) {
        auto it = c->k_tensors.find("kokoro." + n);
        if (it == c->k_tensors.end()) { if (err.empty()) err = "missing tensor 'kokoro." + n + "'"; return nullptr; }
        if (ne) memcpy(ne, it->second.ne, sizeof(int64_t) * 4);
        return (const float *) (c->arena + it->second.off);
    }
    bool ok() {
        if (!err.empty()) return false;
        if (s.failed) { err = "device scratch allocation failed"; return false; }
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) { err = std::string("kernel launch failed: ") + hipGetErrorString(e); return false; }
        return true;
    }
    void linear(const float *W, const float *b, const float *x, int ldx, int R, int K, int N, float *y, int ldy, int acc = 0) {
        if (!W || !x || !y) return;
        if (c->kk_mfma && R >= 32 && K % 16 == 0 && ldx % 4 == 0 && ((uintptr_t) W & 15) == 0 && ((uintptr_t) x & 15) == 0) {
            // many rows (ALBERT and the predictor over the whole sequence): 64 x 64 tiles on the exact-fp32 matrix pipe
            hipLaunchKernelGGL(kk_linear_mfma_kernel, dim3((unsigned) ((N + 63) / 64), (unsigned) ((R + 63) / 64)), dim3(256), 0, st, W, b, x, ldx, R, K, N, y, ldy, acc);
            return;
        }
        hipLaunchKernelGGL(kk_linear_kernel, dim3((unsigned) (((int64_t) R * N + 3) / 4)), dim3(256), 0, st, W, b, x, ldx, R, K, N, y, ldy, acc);
    }
    void norm_rows(const float *x, int ldx, int R, int H, float eps, const float *w_, const float *b, int mode, float *y, int ldy) {
        if (!x || !y) return;
        hipLaunchKernelGGL(kk_norm_rows_kernel, dim3(R), dim3(64), 0, st, x, ldx, H, eps, w_, b, mode, y, ldy);
    }
    void copy(float *dst, const float *src, size_t n) { if (dst && src) (void) hipMemcpyAsync(dst, src, n * 4, hipMemcpyDeviceToDevice, st); }
    // build_lstm (:35-51): one bidirectional cell; x [L][in] -> out [L][2 hid]
    void bilstm(const std::string &base, const float *x, int L, int in, int hid, float *out) {
        const int cp = hid / 4;
        const bool split = c->kk_lstm_split && hid % 16 == 0 && (cp == 4 || cp == 8 || cp == 16 || cp == 32 || cp == 64 || cp == 128) && L > 1;
        float *pre = s.f((size_t) 8 * L * hid);   // both directions' input pre-activations
        if (split) {
            // the recurrence of both directions in one launch over hid/16 workgroups each (kk_lstm_split_kernel)
            LstmArgs la{};
            for (int dir = 0; dir < 2; dir++) {
                const std::string wn = dir ? ".0.reverse_weights." : ".0.weights.", bn = dir ? ".0.reverse_biases." : ".0.biases.";
                float *pd = pre ? pre + (size_t) dir * 4 * L * hid : nullptr;
                for (int g = 0; g < 4; g++) {
                    linear(w(base + wn + std::to_string(2 * g)), w(base + bn + std::to_string(2 * g)), x, in, L, in, hid, pd ? pd + (size_t) g * L * hid : nullptr, hid);
                    la.whh[dir][g] = w(base + wn + std::to_string(2 * g + 1));
                    la.bhh[dir][g] = w(base + bn + std::to_string(2 * g + 1));
                }
                la.pre[dir] = pd;
            }
            float *xch = s.f((size_t) 2 * 2 * hid * 2);   // 8-byte granules
            if (!err.empty() || !pre || !xch) return;
            (void) hipMemsetAsync(xch, 0, (size_t) 2 * 2 * hid * 8, st);
            (void) hipMemsetAsync(c->kk_stuck, 0, 4, st);
            la.xch = (unsigned long long *) xch; la.out = out; la.L = L; la.hid = hid; la.out_stride = 2 * hid; la.stuck = c->kk_stuck;
            const dim3 grid(hid / 16, 2);
            const size_t lds = (size_t) hid * 4;
            switch (cp) {
                case 4: hipLaunchKernelGGL(kk_lstm_split_kernel<4>, grid, dim3(256), lds, st, la); break;
                case 8: hipLaunchKernelGGL(kk_lstm_split_kernel<8>, grid, dim3(256), lds, st, la); break;
                case 16: hipLaunchKernelGGL(kk_lstm_split_kernel<16>, grid, dim3(256), lds, st, la); break;
                case 32: hipLaunchKernelGGL(kk_lstm_split_kernel<32>, grid, dim3(256), lds, st, la); break;
                case 64: hipLaunchKernelGGL(kk_lstm_split_kernel<64>, grid, dim3(256), lds, st, la); break;
                default: hipLaunchKernelGGL(kk_lstm_split_kernel<128>, grid, dim3(256), lds, st, la); break;
            }
            return;
        }
        for (int dir = 0; dir < 2; dir++) {
            const std::string wn = dir ? ".0.reverse_weights." : ".0.weights.", bn = dir ? ".0.reverse_biases." : ".0.biases.";
            const float *whh[4], *bhh[4];
            for (int g = 0; g < 4; g++) {
                linear(w(base + wn + std::to_string(2 * g)), w(base + bn + std::to_string(2 * g)), x, in, L, in, hid, pre + (size_t) g * L * hid, hid);
                whh[g] = w(base + wn + std::to_string(2 * g + 1));
                bhh[g] = w(base + bn + std::to_string(2 * g + 1));
            }
            if (!err.empty() || !pre) return;
            const int threads = std::max(64, (hid + 63) / 64 * 64);
            hipLaunchKernelGGL(kk_lstm_kernel, dim3(1), dim3(threads), (size_t) hid * 4, st, (const float *) pre, whh[0], whh[1], whh[2], whh[3], bhh[0], bhh[1], bhh[2], bhh[3],
                               L, hid, dir, out, 2 * hid, dir * hid);
        }
    }
    // gamma / beta = W style + b, then the fused instance norm (:93-101)
    void adain(float *x, int C, int64_t L, const float *style, int S, const std::string &gw, const std::string &gb, const std::string &bw, const std::string &bb, int act,
               float slope, const float *alpha) {
        float *gamma = s.f(C), *beta = s.f(C);
        linear(w(gw), w(gb), style, S, 1, S, C, gamma, C);
        linear(w(bw), w(bb), style, S, 1, S, C, beta, C);
        if (!err.empty() || !gamma || !beta) return;
        if (c->kk_mfma && L >= 8192) {   // long rows: slices over workgroups, three phases (kk_adain_split_kernel)
            const int S = (int) std::min<int64_t>(32, std::max<int64_t>(2, (int64_t) 2048 / C));
            float *part = s.f((size_t) 2 * C * S);
            if (!part) return;
            for (int phase = 0; phase < 3; phase++)
                hipLaunchKernelGGL(kk_adain_split_kernel, dim3(C, S), dim3(256), 0, st, x, L, (const float *) gamma, (const float *) beta, act, slope, alpha, part, phase);
            return;
        }
        hipLaunchKernelGGL(kk_adain_kernel, dim3(C), dim3(256), 0, st, x, L, (const float *) gamma, (const float *) beta, act, slope, alpha);
    }
    // Stride-1 "same" convolutions with enough channels (the generator's residual blocks k = 3 / 7 / 11 with dilations 1 / 3 / 5, the
    // AdaIN residual blocks k = 3, the text encoder k = 5, conv_post k = 7: 59 % of the model's kernel time through the one-thread-per-
    // output kernel, profiles/r02/kernel_stats_kokoro_82m.csv) go through the codec's exact-fp32 MFMA conv kernel (conv1d_mfma_kernel: input
    // channels staged through LDS, weights pre-packed once per tensor into its LDS image); `acc` becomes its residual input (y += conv).
    // Everything else (stride 2, nearest-2x input, one output channel, the scaled shortcut) stays on kk_conv1d_kernel.
    // The same convolutions as bf16 x 3 split products on the bf16 matrix pipe (conv1d_mfma_b3_kernel<.., KT>: fp32-level error, 2-3x the rate of the
    // exact-fp32 MFMA at these sizes): k = 3 / 5 / 7 / 11, 64-channel tiles x 256 positions, weights packed once per tensor as three bf16 planes.
    template <int KT>
    bool conv_b3(const float *x, int cin, int64_t L, const float *wt, const float *b, int cout, int pad, int dil, float *y, int acc) {
        const size_t w_off = (size_t) ((const char *) wt - c->arena);
        if (c->packed_b3.find(w_off) == c->packed_b3.end() && pack_one_b3(c, w_off, cout, cin, 64, KT) != 0) { err = tts_hip_last_error(); return true; }
        ConvArgs a{};
        a.x = x; a.w = (const float *) c->packed_b3[w_off]; a.b = b; a.alpha = nullptr; a.alpha_out = nullptr; a.resid = acc ? y : nullptr; a.y = y;
        a.cin = cin; a.cout = cout; a.L = (int) L; a.dil = dil; a.pad = pad; a.do_tanh = 0; a.frames = nullptr; a.mult = 1; a.x_f16 = 0;
        if (prof_begin(c, TTS_HIP_K_KOKORO_CONV, ((double) cin * L + (double) cout * L * (acc ? 2 : 1) + (double) cout * cin * KT) * 4, 2.0 * cout * (double) cin * KT * L) != 0) { err = tts_hip_last_error(); return true; }
        const int rc = launch_conv_b3<2, 1, 1, 8, KT>(c, a, 1);
        if (rc != 0 || prof_end(c) != 0) err = tts_hip_last_error();
        return true;
    }
    template <int KT, int CI_T>
    bool conv_mfma(const float *x, int cin, int64_t L, const float *wt, const float *b, int cout, int pad, int dil, float *y, int acc) {
        const size_t w_off = (size_t) ((const char *) wt - c->arena);
        const int CO_T = cout % 128 == 0 ? 128 : 64;   // other widths (conv_post: 22 channels): 64-channel tiles, zero-padded weights, stores masked
        if (c->packed.find(w_off) == c->packed.end() && pack_one(c, w_off, cout, cin, KT, CO_T, CI_T, false) != 0) { err = tts_hip_last_error(); return true; }
        ConvArgs a{};
        a.x = x; a.w = c->packed[w_off]; a.b = b; a.alpha = nullptr; a.alpha_out = nullptr; a.resid = acc ? y : nullptr; a.y = y;
        a.cin = cin; a.cout = cout; a.L = (int) L; a.dil = dil; a.pad = pad; a.do_tanh = 0; a.frames = nullptr; a.mult = 1; a.x_f16 = 0;
        if (prof_begin(c, TTS_HIP_K_KOKORO_CONV, ((double) cin * L + (double) cout * L * (acc ? 2 : 1) + (double) cout * cin * KT) * 4, 2.0 * cout * (double) cin * KT * L) != 0) { err = tts_hip_last_error(); return true; }
        const int rc = CO_T == 128 ? launch_conv_mfma<KT, 2, 2, 2, 2, CI_T>(c, a, 1) : launch_conv_mfma<KT, 2, 2, 1, 4, CI_T>(c, a, 1);
        if (rc != 0 || prof_end(c) != 0) err = tts_hip_last_error();
        return true;
    }
    void conv1d(const float *x, int cin, int64_t L, const float *wt, const float *b, int cout, int K, int stride, int pad, int dil, int in_shift, float *y, int64_t Lout,
                int acc, float post) {
        if (!x || !wt || !y) return;
        const bool same = stride == 1 && !in_shift && Lout == L && pad * 2 == dil * (K - 1) && post == 1.0f && dil <= 9;
        if (same && c->kk_mfma && cout >= 16 && cin >= 16 && L < (1 << 30) && (const char *) wt >= c->arena && (const char *) wt < c->arena + c->arena_bytes) {
            if (K == 1 && (L & 3) == 0 && (((uintptr_t) x | (uintptr_t) y) & 15) == 0 && conv_mfma<1, 16>(x, cin, L, wt, b, cout, pad, dil, y, acc)) return;   // k = 1 stages 16-byte pieces
            if (c->kk_b3 && dil <= 5) {   // bf16 x 3 split products (tune("kokoro_b3", 0): the exact-fp32 MFMA kernels below)
                if (K == 3 && conv_b3<3>(x, cin, L, wt, b, cout, pad, dil, y, acc)) return;
                if (K == 5 && conv_b3<5>(x, cin, L, wt, b, cout, pad, dil, y, acc)) return;
                if (K == 7 && conv_b3<7>(x, cin, L, wt, b, cout, pad, dil, y, acc)) return;
                if (K == 11 && conv_b3<11>(x, cin, L, wt, b, cout, pad, dil, y, acc)) return;
            }
            if (K == 3 && conv_mfma<3, 8>(x, cin, L, wt, b, cout, pad, dil, y, acc)) return;
            if (K == 5 && conv_mfma<5, 4>(x, cin, L, wt, b, cout, pad, dil, y, acc)) return;
            if (K == 7 && conv_mfma<7, 4>(x, cin, L, wt, b, cout, pad, dil, y, acc)) return;
            if (K == 11 && conv_mfma<11, 4>(x, cin, L, wt, b, cout, pad, dil, y, acc)) return;
        }
        if (c->kk_mfma && K == 1 && stride == 1 && pad == 0 && dil == 1 && cin >= 16 && cout >= 16 && Lout == (in_shift ? 2 * L : L)) {
            // the k = 1 shortcuts (any channel count, nearest-2x input, accumulate + 1/sqrt 2): 64 x 64 tiles on the exact-fp32 matrix pipe
            hipLaunchKernelGGL(kk_conv1x1_mfma_kernel, dim3((unsigned) ((Lout + 63) / 64), (unsigned) ((cout + 63) / 64)), dim3(256), 0, st, x, cin, L, wt, b, cout, in_shift, y, Lout, acc, post);
            return;
        }
        hipLaunchKernelGGL(kk_conv1d_kernel, kgrid((int64_t) cout * Lout), dim3(256), 0, st, x, cin, L, wt, b, cout, K, stride, pad, dil, in_shift, y, Lout, acc, post);
    }
    // build_ada_residual_conv (:88-134): x [cin][L] -> [cout][L or 2L]
    float *ada_block(const std::string &base, const float *x, int64_t L, const float *style, int S, int &C, int64_t &Lout) {
        int64_t ne[4];
        const float *conv1 = w(base + ".conv1_weight", ne);
        if (!conv1) return nullptr;
        const int cin = (int) ne[1], cout = (int) ne[2];
        if (cin != C) { err = base + ": channel count mismatch"; return nullptr; }
        float *cur = s.f((size_t) cin * L);
        copy(cur, x, (size_t) cin * L);
        adain(cur, cin, L, style, S, base + ".norm1_gamma_weight", base + ".norm1_gamma_bias", base + ".norm1_beta_weight", base + ".norm1_beta_bias", 1, 0.2f, nullptr);
        const bool pool = has(base + ".pool_weight");
        int64_t Lc = L;
        if (pool) {
            float *up = s.f((size_t) cin * 2 * L);
            if (up) hipLaunchKernelGGL(kk_pool_convt_kernel, kgrid((int64_t) cin * 2 * L), dim3(256), 0, st, (const float *) cur, cin, L, w(base + ".pool_weight"), w(base + ".pool_bias"), up);
            cur = up;
            Lc = 2 * L;
        }
        float *y = s.f((size_t) cout * Lc);
        conv1d(cur, cin, Lc, conv1, w(base + ".conv1_bias"), cout, 3, 1, 1, 1, 0, y, Lc, 0, 1.0f);
        adain(y, cout, Lc, style, S, base + ".norm2_gamma_weight", base + ".norm2_gamma_bias", base + ".norm2_beta_weight", base + ".norm2_beta_bias", 1, 0.2f, nullptr);
        float *res = s.f((size_t) cout * Lc);
        conv1d(y, cout, Lc, w(base + ".conv2_weight"), w(base + ".conv2_bias"), cout, 3, 1, 1, 1, 0, res, Lc, 0, 1.0f);
        const float inv = 1.0f / sqrtf(2.0f);
        if (has(base + ".conv1x1_weight")) {
            conv1d(x, cin, L, w(base + ".conv1x1_weight"), nullptr, cout, 1, 1, 0, 1, pool ? 1 : 0, res, Lc, 1, inv);
        } else if (res) {
            hipLaunchKernelGGL(kk_add_kernel, kgrid((int64_t) cout * Lc), dim3(256), 0, st, (const float *) res, x, res, (int64_t) cout * Lc, inv);
        }
        C = cout;
        Lout = Lc;
        return res;
    }
    // build_kokoro_generator_res_block (:136-165), in place on x [C][L]
    void gen_res(const std::string &base, float *x, int C, int64_t L, const float *style, int S, const uint32_t *pads, const uint32_t *dils) {
        for (int i = 0; i < 3; i++) {
            const std::string b = base + "." + std::to_string(i) + ".";
            float *cur = s.f((size_t) C * L), *y = s.f((size_t) C * L);
            copy(cur, x, (size_t) C * L);
            adain(cur, C, L, style, S, b + "gamma1_weight", b + "gamma1_bias", b + "beta1_weight", b + "beta1_bias", 2, 0.0f, w(b + "alpha1"));
            int64_t ne[4];
            const float *w1 = w(b + "convs1_weight", ne);
            if (!w1) return;
            conv1d(cur, C, L, w1, w(b + "convs1_bias"), C, (int) ne[0], 1, (int) pads[i], (int) dils[i], 0, y, L, 0, 1.0f);
            adain(y, C, L, style, S, b + "gamma2_weight", b + "gamma2_bias", b + "beta2_weight", b + "beta2_bias", 2, 0.0f, w(b + "alpha2"));
            const float *w2 = w(b + "convs2_weight", ne);
            if (!w2) return;
            conv1d(y, C, L, w2, w(b + "convs2_bias"), C, (int) ne[0], 1, (int) pads[0], 1, 0, x, L, 1, 1.0f);   // x += conv (:160-161)
        }
    }
};

int kokoro_dims(tts_hip_ctx *c, KRun &k, int &D, int &S) {
    int64_t ne[4];
    if (!k.w("duration_predictor.encode", ne)) return -1;
    D = (int) ne[1];
    if (!k.w("duration_predictor.layers.1.gamma_weight", ne)) return -1;
    S = (int) ne[0];
    return 0;
}
const float *kokoro_voice(tts_hip_ctx *c, KRun &k, const char *voice, uint32_t n, int S, bool second_half) {
    int64_t ne[4];
    const float *v = k.w(std::string("voice_tensors.") + (voice ? voice : ""), ne);
    if (!v) return nullptr;
    if ((int) ne[0] != 2 * S || (int64_t) n - 3 >= ne[1]) { k.err = "voice tensor shape does not cover this token count"; return nullptr; }
    return v + (size_t) (n - 3) * 2 * S + (second_half ? S : 0);   // row n_tokens - 3 (:1012, :1149, :1220)
}
}  // namespace

// after a synchronised Kokoro call: did a bounded spin of kk_lstm_split_kernel give up?
static int kokoro_check_stuck(tts_hip_ctx *c, const char *who) {
    int stuck = 0;
    HIPCHK(hipMemcpy(&stuck, c->kk_stuck, 4, hipMemcpyDeviceToHost));
    if (stuck) {
        (void) hipMemset(c->kk_stuck, 0, 4);
        return set_err("%s: the workgroups of a split LSTM recurrence never saw each other's hidden state (device oversubscribed?); "
                       "TTS_HIP_KOKORO_LSTM_SPLIT=0 selects the single-workgroup kernel", who);
    }
    return 0;
}

extern "C" int tts_hip_kokoro_durations(tts_hip_ctx *c, const uint32_t *tokens, uint32_t n, const char *voice, float *lens_out, float *hidden_out) {
    if (!c || !c->has_kokoro) return set_err("tts_hip_kokoro_durations: not a Kokoro context (tts_hip_kokoro_create)");
    if (!c->finalized || !c->weights_present) return set_err("tts_hip_kokoro_durations: context not finalized");
    if (!tokens || !lens_out) return set_err("tts_hip_kokoro_durations: null argument");
    if (n < 3 || n > c->ko.max_ctx) return set_err("tts_hip_kokoro_durations: %u tokens outside 3..%u", n, c->ko.max_ctx);
    HIPCHK(hipSetDevice(c->device));
    KScratch s(c);
    KRun k(c, s);
    int64_t ne[4];
    const float *tok_embd = k.w("albert.token_embd", ne);
    if (!tok_embd) return set_err("tts_hip_kokoro_durations: %s", k.err.c_str());
    const int E = (int) ne[0], vocab = (int) ne[1];
    for (uint32_t i = 0; i < n; i++)
        if (tokens[i] >= (uint32_t) vocab) return set_err("tts_hip_kokoro_durations: token %u >= vocabulary %d", tokens[i], vocab);
    const float *embd = k.w("albert.embd", ne);
    const int H = embd ? (int) ne[1] : 0, NH = (int) c->ko.n_attn_heads, hs = NH ? H / NH : 0;
    const float *ffn_w = k.w("albert.layer.0.ffn", ne);
    const int F = ffn_w ? (int) ne[1] : 0;
    int D = 0, S = 0;
    if (kokoro_dims(c, k, D, S) != 0 || !embd || !ffn_w || NH == 0 || H % NH) return set_err("tts_hip_kokoro_durations: %s", k.err.empty() ? "bad ALBERT shapes" : k.err.c_str());
    const float *style = kokoro_voice(c, k, voice, n, S, true);
    if (!style) return set_err("tts_hip_kokoro_durations: %s", k.err.c_str());
    const int N = (int) n, Wd = D + S;
    uint32_t *d_tok = (uint32_t *) s.f(n);
    float *x0 = s.f((size_t) N * E), *x = s.f((size_t) N * H), *q = s.f((size_t) N * H), *kk = s.f((size_t) N * H), *v = s.f((size_t) N * H), *att = s.f((size_t) N * H);
    float *o = s.f((size_t) N * H), *ff = s.f((size_t) N * F), *cur = s.f((size_t) N * Wd), *ls = s.f((size_t) N * D), *gamma = s.f(D), *beta = s.f(D);
    if (s.failed) return set_err("tts_hip_kokoro_durations: device scratch allocation failed");
    HIPCHK(hipMemcpyAsync(d_tok, tokens, (size_t) n * 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(kk_albert_embed_kernel, kgrid((int64_t) N * E), dim3(256), 0, c->stream, tok_embd, k.w("albert.position_embd"), k.w("albert.token_type_embd"),
                       (const uint32_t *) d_tok, N, E, x0);
    k.norm_rows(x0, E, N, E, 1e-12f, k.w("albert.norm"), k.w("albert.norm_bias"), 0, x0, E);
    k.linear(embd, k.w("albert.embd_bias"), x0, E, N, E, H, x, H);
    const std::string L0 = "albert.layer.0.";
    for (uint32_t r = 0; r < c->ko.n_recurrence; r++) {
        k.linear(k.w(L0 + "q"), k.w(L0 + "q_bias"), x, H, N, H, H, q, H);
        k.linear(k.w(L0 + "k"), k.w(L0 + "k_bias"), x, H, N, H, H, kk, H);
        k.linear(k.w(L0 + "v"), k.w(L0 + "v_bias"), x, H, N, H, H, v, H);
        if (hs == 64 && c->kk_attn_lds && (size_t) (64 * 65 + 4 * 64 + 4 * N) * 4 <= 64 * 1024)   // keys staged through LDS, four rows per workgroup (tune("kokoro_attn_lds") = 0: one wave per row)
            hipLaunchKernelGGL(kk_albert_attn64_kernel, dim3(NH, (N + 3) / 4), dim3(256), (size_t) (64 * 65 + 4 * 64 + 4 * N) * 4, c->stream, (const float *) q, (const float *) kk,
                               (const float *) v, N, H, c->ko.attn_scale, att);
        else
            hipLaunchKernelGGL(kk_albert_attn_kernel, dim3(NH, N), dim3(64), (size_t) N * 4, c->stream, (const float *) q, (const float *) kk, (const float *) v, N, H, hs,
                               c->ko.attn_scale, att);
        k.linear(k.w(L0 + "o"), k.w(L0 + "o_bias"), att, H, N, H, H, o, H);
        hipLaunchKernelGGL(kk_add_kernel, kgrid((int64_t) N * H), dim3(256), 0, c->stream, (const float *) o, (const float *) x, o, (int64_t) N * H, 1.0f);
        k.norm_rows(o, H, N, H, 1e-12f, k.w(L0 + "ffn_norm"), k.w(L0 + "ffn_norm_bias"), 0, x, H);
        k.linear(ffn_w, k.w(L0 + "ffn_bias"), x, H, N, H, F, ff, F);
        hipLaunchKernelGGL(kk_gelu_kernel, kgrid((int64_t) N * F), dim3(256), 0, c->stream, ff, (int64_t) N * F);
        k.linear(k.w(L0 + "ffn_out"), k.w(L0 + "ffn_out_bias"), ff, F, N, F, H, o, H);
        hipLaunchKernelGGL(kk_add_kernel, kgrid((int64_t) N * H), dim3(256), 0, c->stream, (const float *) o, (const float *) x, o, (int64_t) N * H, 1.0f);
        k.norm_rows(o, H, N, H, 1e-12f, k.w(L0 + "attn_norm"), k.w(L0 + "attn_norm_bias"), 0, x, H);
        if (!k.ok()) return set_err("tts_hip_kokoro_durations: %s", k.err.c_str());
    }
    const std::string dp = "duration_predictor.";
    k.linear(k.w(dp + "encode"), k.w(dp + "encode_bias"), x, H, N, H, D, cur, Wd);
    hipLaunchKernelGGL(kk_fill_cols_kernel, kgrid((int64_t) N * S), dim3(256), 0, c->stream, cur, N, Wd, D, style, S);
    for (uint32_t l = 0; l < c->ko.n_dp_layers; l++) {
        const std::string lb = dp + "layers." + std::to_string(2 * l + 1) + ".";
        k.bilstm(dp + "layers." + std::to_string(2 * l) + ".lstm", cur, N, Wd, D / 2, ls);
        k.linear(k.w(lb + "gamma_weight"), k.w(lb + "gamma_bias"), style, S, 1, S, D, gamma, D);
        k.linear(k.w(lb + "beta_weight"), k.w(lb + "beta_bias"), style, S, 1, S, D, beta, D);
        k.norm_rows(ls, D, N, D, 1e-5f, gamma, beta, 1, cur, Wd);   // the style columns of cur stay in place
        if (!k.ok()) return set_err("tts_hip_kokoro_durations: %s", k.err.c_str());
    }
    if (hidden_out) HIPCHK(hipMemcpyAsync(hidden_out, cur, (size_t) N * Wd * 4, hipMemcpyDeviceToHost, c->stream));
    k.bilstm(dp + "duration_lstm", cur, N, Wd, D / 2, ls);
    const float *dpw = k.w(dp + "duration_proj", ne);
    const int ND = dpw ? (int) ne[1] : 0;
    float *dur = s.f((size_t) N * ND), *lens = s.f(n);
    k.linear(dpw, k.w(dp + "duration_proj_bias"), ls, D, N, D, ND, dur, ND);
    if (dur && lens) hipLaunchKernelGGL(kk_duration_kernel, kgrid(N, 64), dim3(64), 0, c->stream, (const float *) dur, N, ND, lens);
    if (!k.ok()) return set_err("tts_hip_kokoro_durations: %s", k.err.c_str());
    HIPCHK(hipMemcpyAsync(lens_out, lens, (size_t) n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return kokoro_check_stuck(c, "tts_hip_kokoro_durations");
}

extern "C" int tts_hip_kokoro_generate(tts_hip_ctx *c, const uint32_t *tokens, uint32_t n, const float *lens, const float *hidden, const char *voice, const float *noise,
                                       float *pcm_out, float *hsrc_out, const float *hsrc_in) {
    if (!c || !c->has_kokoro) return set_err("tts_hip_kokoro_generate: not a Kokoro context (tts_hip_kokoro_create)");
    if (!c->finalized || !c->weights_present) return set_err("tts_hip_kokoro_generate: context not finalized");
    if (!tokens || !lens || !hidden || !noise || !pcm_out) return set_err("tts_hip_kokoro_generate: null argument");
    if (n < 3 || n > c->ko.max_ctx) return set_err("tts_hip_kokoro_generate: %u tokens outside 3..%u", n, c->ko.max_ctx);
    HIPCHK(hipSetDevice(c->device));
    const tts_hip_kokoro_desc &kd = c->ko;
    KScratch s(c);
    KRun k(c, s);
    int D = 0, S = 0;
    if (kokoro_dims(c, k, D, S) != 0) return set_err("tts_hip_kokoro_generate: %s", k.err.c_str());
    const int N = (int) n, Wd = D + S;
    std::vector<int> tok_of;
    for (int i = 0; i < N; i++) {
        if (!(lens[i] >= 1.0f) || lens[i] > 50.0f || lens[i] != floorf(lens[i])) return set_err("tts_hip_kokoro_generate: length %g of token %d is not a whole number in 1..50", lens[i], i);
        for (int r = 0; r < (int) lens[i]; r++) tok_of.push_back(i);
    }
    const int64_t T = (int64_t) tok_of.size();
    const float *style_p = kokoro_voice(c, k, voice, n, S, true), *style_d = kokoro_voice(c, k, voice, n, S, false);
    if (!style_p || !style_d) return set_err("tts_hip_kokoro_generate: %s", k.err.c_str());
    int64_t ne[4];
    const float *te = k.w("text_encoder.embedding_weight", ne);
    if (!te) return set_err("tts_hip_kokoro_generate: %s", k.err.c_str());
    const int C = (int) ne[0], vocab = (int) ne[1];
    for (uint32_t i = 0; i < n; i++)
        if (tokens[i] >= (uint32_t) vocab) return set_err("tts_hip_kokoro_generate: token %u >= vocabulary %d", tokens[i], vocab);
    const int NHm = (int) kd.harmonic_num + 1, up = (int) kd.upsample_scale;
    const int64_t L2 = 2 * T, LS = L2 * up, out_len = T * kd.up_sampling_factor;
    const int Nf = (int) kd.n_fft, hop = (int) kd.hop, nbins = Nf / 2 + 1;
    const int64_t F = LS / hop + 1;

    uint32_t *d_tok = (uint32_t *) s.f(n);
    int *d_idx = (int *) s.f((size_t) T);
    float *d_hidden = s.f((size_t) N * Wd), *d_noise = s.f((size_t) NHm * LS);
    if (s.failed) return set_err("tts_hip_kokoro_generate: device scratch allocation failed");
    HIPCHK(hipMemcpyAsync(d_tok, tokens, (size_t) n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_idx, tok_of.data(), (size_t) T * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_hidden, hidden, (size_t) N * Wd * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_noise, noise, (size_t) NHm * LS * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));   // tok_of is a local

    // alignment + shared LSTM (:1157-1166)
    float *en = s.f((size_t) T * Wd), *sh = s.f((size_t) T * D), *shc = s.f((size_t) D * T);
    if (s.failed) return set_err("tts_hip_kokoro_generate: device scratch allocation failed");
    hipLaunchKernelGGL(kk_gather_rows_kernel, kgrid(T * Wd), dim3(256), 0, c->stream, (const float *) d_hidden, (const int *) d_idx, (int) T, Wd, en);
    k.bilstm("duration_predictor.shared_lstm", en, (int) T, Wd, D / 2, sh);
    hipLaunchKernelGGL(kk_transpose_kernel, kgrid(T * D), dim3(256), 0, c->stream, (const float *) sh, (int) T, D, shc);
    // F0 / N branches (:1169-1192)
    float *curves[2] = {nullptr, nullptr};
    const char *branch[2] = {"f0", "n"};
    for (int b = 0; b < 2; b++) {
        float *cur = shc;
        int Cb = D;
        int64_t L = T;
        for (uint32_t i = 0; i < kd.f0_n_blocks; i++) {
            cur = k.ada_block(std::string("duration_predictor.") + branch[b] + "_blocks." + std::to_string(i), cur, L, style_p, S, Cb, L);
            if (!cur || !k.ok()) return set_err("tts_hip_kokoro_generate: %s", k.err.c_str());
        }
        if (L != L2) return set_err("tts_hip_kokoro_generate: the %s branch does not double the frame count", branch[b]);
        curves[b] = s.f((size_t) L2);
        k.conv1d(cur, Cb, L, k.w(std::string("duration_predictor.") + branch[b] + "_proj_kernel"), k.w(std::string("duration_predictor.") + branch[b] + "_proj_bias"), 1, 1, 1,
                 0, 1, 0, curves[b], L2, 0, 1.0f);
    }
    // text encoder (:1196-1210)
    float *tx = s.f((size_t) C * N);
    hipLaunchKernelGGL(kk_embed_cols_kernel, kgrid((int64_t) N * C), dim3(256), 0, c->stream, te, (const uint32_t *) d_tok, N, C, tx);
    for (uint32_t l = 0; l < kd.n_conv_layers; l++) {
        const std::string lb = "text_encoder.layers." + std::to_string(l) + ".";
        const float *cw = k.w(lb + "weight", ne);
        if (!cw) return set_err("tts_hip_kokoro_generate: %s", k.err.c_str());
        float *y = s.f((size_t) C * N);
        k.conv1d(tx, C, N, cw, k.w(lb + "bias"), C, (int) ne[0], 1, 2, 1, 0, y, N, 0, 1.0f);
        if (y) hipLaunchKernelGGL(kk_chan_norm_kernel, dim3(N), dim3(64), 0, c->stream, y, C, (int64_t) N, k.w(lb + "gamma"), k.w(lb + "beta"), 0.2f);
        tx = y;
    }
    float *txr = s.f((size_t) N * C), *tl = s.f((size_t) N * C), *asr = s.f((size_t) C * T);
    if (s.failed) return set_err("tts_hip_kokoro_generate: device scratch allocation failed");
    hipLaunchKernelGGL(kk_transpose_kernel, kgrid((int64_t) C * N), dim3(256), 0, c->stream, (const float *) tx, C, N, txr);
    k.bilstm("text_encoder.lstm", txr, N, C, C / 2, tl);
    hipLaunchKernelGGL(kk_gather_cols_kernel, kgrid((int64_t) C * T), dim3(256), 0, c->stream, (const float *) tl, (const int *) d_idx, (int) T, C, asr);
    if (!k.ok()) return set_err("tts_hip_kokoro_generate: %s", k.err.c_str());
    // decoder (:1222-1241)
    float *f0d = s.f((size_t) T), *nd = s.f((size_t) T);
    k.conv1d(curves[0], 1, L2, k.w("decoder.f0_conv_weight"), k.w("decoder.f0_conv_bias"), 1, 3, 2, 1, 1, 0, f0d, T, 0, 1.0f);
    k.conv1d(curves[1], 1, L2, k.w("decoder.n_conv_weight"), k.w("decoder.n_conv_bias"), 1, 3, 2, 1, 1, 0, nd, T, 0, 1.0f);
    int Cc = C + 2;
    float *cat0 = s.f((size_t) Cc * T);
    k.copy(cat0, asr, (size_t) C * T);
    k.copy(cat0 + (size_t) C * T, f0d, (size_t) T);
    k.copy(cat0 + (size_t) (C + 1) * T, nd, (size_t) T);
    int64_t Lc = T;
    float *cur = k.ada_block("decoder.encoder_block", cat0, T, style_d, S, Cc, Lc);
    const float *aw = k.w("decoder.asr_conv_weight", ne);
    if (!cur || !aw) return set_err("tts_hip_kokoro_generate: %s", k.err.c_str());
    const int CA = (int) ne[2];
    float *asr_res = s.f((size_t) CA * T);
    k.conv1d(asr, C, T, aw, k.w("decoder.asr_conv_bias"), CA, 1, 1, 0, 1, 0, asr_res, T, 0, 1.0f);
    for (uint32_t i = 0; i < kd.n_decoder_blocks; i++) {
        int Cin = Cc + CA + 2;
        float *cat = s.f((size_t) Cin * T);
        k.copy(cat, cur, (size_t) Cc * T);
        k.copy(cat + (size_t) Cc * T, asr_res, (size_t) CA * T);
        k.copy(cat + (size_t) (Cc + CA) * T, f0d, (size_t) T);
        k.copy(cat + (size_t) (Cc + CA + 1) * T, nd, (size_t) T);
        cur = k.ada_block("decoder.decoder_blocks." + std::to_string(i), cat, T, style_d, S, Cin, Lc);
        if (!cur || !k.ok()) return set_err("tts_hip_kokoro_generate: %s", k.err.c_str());
        Cc = Cin;
    }
    if (Lc != L2) return set_err("tts_hip_kokoro_generate: the decoder does not end at twice the frame count");
    // harmonic source + STFT conditioning (:173-206)
    float *phase = s.f((size_t) NHm * L2), *sine = s.f((size_t) NHm * LS), *har = s.f((size_t) LS), *win = s.f(Nf), *hs = s.f((size_t) 2 * nbins * F);
    if (s.failed) return set_err("tts_hip_kokoro_generate: device scratch allocation failed");
    {
        std::vector<float> hw((size_t) Nf);
        for (int i = 0; i < Nf; i++) hw[(size_t) i] = (float) pow(sin(M_PI * (double) i / (double) Nf), 2.0);   // hann_window, util.cpp:134-139
        HIPCHK(hipMemcpyAsync(win, hw.data(), (size_t) Nf * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    hipLaunchKernelGGL(kk_sine_phase_kernel, kgrid(NHm, 64), dim3(64), 0, c->stream, (const float *) curves[0], L2, NHm, kd.sample_rate, kd.upsample_scale * 2.0f * (float) M_PI, phase);
    hipLaunchKernelGGL(kk_sine_source_kernel, kgrid((int64_t) NHm * LS), dim3(256), 0, c->stream, (const float *) phase, (const float *) curves[0], L2, NHm, up, kd.voice_threshold,
                       kd.sin_amp, kd.noise_std, (const float *) d_noise, sine);
    hipLaunchKernelGGL(kk_source_merge_kernel, kgrid(LS), dim3(256), 0, c->stream, (const float *) sine, NHm, LS, k.w("decoder.generator.m_source_weight"),
                       k.w("decoder.generator.m_source_bias"), har);
    hipLaunchKernelGGL(kk_stft_kernel, kgrid((int64_t) nbins * F), dim3(256), 0, c->stream, (const float *) har, LS, (const float *) win, Nf, hop, F, hs);
    if (!k.ok()) return set_err("tts_hip_kokoro_generate: %s", k.err.c_str());
    if (hsrc_out) HIPCHK(hipMemcpyAsync(hsrc_out, hs, (size_t) 2 * nbins * F * 4, hipMemcpyDeviceToHost, c->stream));
    if (hsrc_in) {
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipMemcpyAsync(hs, hsrc_in, (size_t) 2 * nbins * F * 4, hipMemcpyHostToDevice, c->stream));
    }
    // generator (:208-241)
    float *g = cur;
    int Cg = Cc;
    int64_t Lg = Lc;
    const std::string gb = "decoder.generator.";
    for (uint32_t i = 0; i < kd.n_upsamples; i++) {
        hipLaunchKernelGGL(kk_leaky_kernel, kgrid((int64_t) Cg * Lg), dim3(256), 0, c->stream, g, (int64_t) Cg * Lg, 0.1f);
        const float *uw = k.w(gb + "ups." + std::to_string(i) + ".weight", ne);
        if (!uw) return set_err("tts_hip_kokoro_generate: %s", k.err.c_str());
        const int K = (int) ne[0], Co = (int) ne[1];
        const int64_t Lo = (Lg - 1) * kd.up_stride[i] - 2 * (int64_t) kd.up_padding[i] + K;
        float *y = s.f((size_t) Co * Lo);
        const int S_ = (int) kd.up_stride[i];
        const bool mfma_up = c->kk_mfma && K == 2 * S_ && (S_ == 10 || S_ == 6) && Co % 64 == 0 && 2 * (int) kd.up_padding[i] == K - S_ && Lo == Lg * S_;
        if (y && mfma_up) {
            // the generator's ConvTranspose1d (stride 10 / 6, kernel = 2 x stride: every output touches two taps) on the codec's
            // phase-decomposed MFMA kernel; the one-thread-per-output kernel spent 18 ms per launch here
            const size_t w_off = (size_t) ((const char *) uw - c->arena);
            if (c->packed.find(w_off) == c->packed.end()) CHK(pack_one(c, w_off, Co, Cg, K, 64, CI32_T, true));
            ConvTArgs ta{};
            ta.x = g; ta.w = c->packed[w_off]; ta.b = k.w(gb + "ups." + std::to_string(i) + ".bias"); ta.alpha = nullptr; ta.y = y;
            ta.cin = Cg; ta.cout = Co; ta.L = (int) Lg; ta.Lout = (int) Lo; ta.stride = S_; ta.pad = (int) kd.up_padding[i]; ta.frames = nullptr; ta.mult = 1;
            if (S_ == 10) CHK((launch_convt_mfma<10, 1, 2, 2, CI32_T>(c, ta, 1)));
            else CHK((launch_convt_mfma<6, 1, 2, 2, CI32_T>(c, ta, 1)));
        } else
        if (y) hipLaunchKernelGGL(kk_convt1d_kernel, kgrid((int64_t) Co * Lo), dim3(256), 0, c->stream, (const float *) g, Cg, Lg, uw, k.w(gb + "ups." + std::to_string(i) + ".bias"), Co, K,
                                  (int) kd.up_stride[i], (int) kd.up_padding[i], y, Lo);
        g = y; Cg = Co; Lg = Lo;
        if (i == kd.n_upsamples - 1) {
            float *p = s.f((size_t) Cg * (Lg + 1));
            if (p) hipLaunchKernelGGL(kk_pad_front_kernel, kgrid((int64_t) Cg * (Lg + 1)), dim3(256), 0, c->stream, (const float *) g, Cg, Lg, p);
            g = p; Lg += 1;
        }
        const std::string nbk = gb + "noise_blocks." + std::to_string(i) + ".";
        const float *nw = k.w(nbk + "conv_weight", ne);
        if (!nw) return set_err("tts_hip_kokoro_generate: %s", k.err.c_str());
        const int NK = (int) ne[0];
        const int64_t Ls = (F + 2 * (int64_t) kd.noise_padding[i] - (NK - 1) - 1) / kd.noise_stride[i] + 1;
        if (Ls != Lg) return set_err("tts_hip_kokoro_generate: source length %lld != %lld at stage %u", (long long) Ls, (long long) Lg, i);
        float *xs = s.f((size_t) Cg * Lg);
        k.conv1d(hs, 2 * nbins, F, nw, k.w(nbk + "conv_bias"), Cg, NK, (int) kd.noise_stride[i], (int) kd.noise_padding[i], 1, 0, xs, Lg, 0, 1.0f);
        k.gen_res(nbk + "resblock", xs, Cg, Lg, style_d, S, kd.noise_res_padding[i], kd.noise_res_dilation[i]);
        if (g && xs) hipLaunchKernelGGL(kk_add_kernel, kgrid((int64_t) Cg * Lg), dim3(256), 0, c->stream, (const float *) g, (const float *) xs, g, (int64_t) Cg * Lg, 1.0f);
        float *sum = s.f((size_t) Cg * Lg), *br = s.f((size_t) Cg * Lg);
        for (uint32_t ii = 0; ii < kd.n_kernels; ii++) {
            float *dst = ii == 0 ? sum : br;
            k.copy(dst, g, (size_t) Cg * Lg);
            const uint32_t ri = i * kd.n_kernels + ii;
            k.gen_res(gb + "resblocks." + std::to_string(ri), dst, Cg, Lg, style_d, S, kd.res_padding[ri], kd.res_dilation[ri]);
            if (ii > 0 && sum && br)
                hipLaunchKernelGGL(kk_add_kernel, kgrid((int64_t) Cg * Lg), dim3(256), 0, c->stream, (const float *) sum, (const float *) br, sum, (int64_t) Cg * Lg,
                                   ii == kd.n_kernels - 1 ? 1.0f / (float) kd.n_kernels : 1.0f);
        }
        g = sum;
        if (!k.ok()) return set_err("tts_hip_kokoro_generate: %s", k.err.c_str());
    }
    hipLaunchKernelGGL(kk_leaky_kernel, kgrid((int64_t) Cg * Lg), dim3(256), 0, c->stream, g, (int64_t) Cg * Lg, 0.01f);
    const float *pw = k.w(gb + "conv_post_weight", ne);
    if (!pw) return set_err("tts_hip_kokoro_generate: %s", k.err.c_str());
    if (Lg != F) return set_err("tts_hip_kokoro_generate: generator length %lld != STFT frames %lld", (long long) Lg, (long long) F);
    float *post = s.f((size_t) 2 * nbins * Lg), *pcm = s.f((size_t) out_len);
    k.conv1d(g, Cg, Lg, pw, k.w(gb + "conv_post_bias"), 2 * nbins, (int) ne[0], 1, (int) kd.out_conv_padding, 1, 0, post, Lg, 0, 1.0f);
    if (post && pcm) {
        hipLaunchKernelGGL(kk_spec_phase_kernel, kgrid((int64_t) 2 * nbins * Lg), dim3(256), 0, c->stream, post, nbins, Lg);
        hipLaunchKernelGGL(kk_istft_kernel, kgrid(out_len), dim3(256), 0, c->stream, (const float *) post, Lg, (const float *) win, Nf, hop, pcm, out_len);
    }
    if (!k.ok()) return set_err("tts_hip_kokoro_generate: %s", k.err.c_str());
    HIPCHK(hipMemcpyAsync(pcm_out, pcm, (size_t) out_len * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return kokoro_check_stuck(c, "tts_hip_kokoro_generate");
}

// ------------------------------------------------------------------------------------------------
// SNAC codec (src/decoder/snac_model.cpp:110-208)
// ------------------------------------------------------------------------------------------------
extern "C" tts_hip_ctx *tts_hip_snac_create(int device, const tts_hip_snac_desc *sd) {
    if (!sd || sd->struct_size != sizeof(tts_hip_snac_desc)) { set_err("tts_hip_snac_create: bad desc (struct_size mismatch)"); return nullptr; }
    if (sd->n_blocks == 0 || sd->n_blocks > TTS_HIP_MAX_DAC_BLOCKS || sd->n_codebooks == 0 || sd->n_codebooks > 4) { set_err("tts_hip_snac_create: n_blocks / n_codebooks out of range"); return nullptr; }
    tts_hip_desc d{};
    d.struct_size = sizeof(d);
    d.max_seqs = 1;
    d.flags = (sd->flags & TTS_HIP_FLAG_VALU_GEMM) | TTS_HIP_FLAG_NO_PARLER | TTS_HIP_FLAG_NO_DAC;
    tts_hip_ctx *c = tts_hip_create(device, &d);
    if (!c) return nullptr;
    c->has_snac = true;
    c->snac = *sd;
    for (uint32_t i = 0; i < sd->n_codebooks; i++)
        if (c->snac.repeats[i] == 0) c->snac.repeats[i] = 1;
    return c;
}

static int ensure_packed_snac(tts_hip_ctx *c) {
    if (c->snac_packed || (c->d.flags & TTS_HIP_FLAG_VALU_GEMM)) return 0;
    int CO_T = 0, CI_T = 0;
    if (conv_tile(c->s_c0, 1, &CO_T, &CI_T) >= 0) CHK(pack_one(c, c->s_upw, c->s_c0, c->s_latent, 1, CO_T, CI_T, false));
    for (auto &b : c->sblocks) {
        if (convt_tile(b.cout, b.stride, &CO_T) >= 0) CHK(pack_one(c, b.w, b.cout, b.cin, 2 * b.stride, CO_T, CI32_T, true));
        if (conv_tile(b.cout, 1, &CO_T, &CI_T) >= 0) {
            CHK(pack_one(c, b.noise_w, b.cout, b.cout, 1, CO_T, CI_T, false));
            for (int r = 0; r < 3; r++) CHK(pack_one(c, b.res[r].out_w, b.cout, b.cout, 1, CO_T, CI_T, false));
        }
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    c->snac_packed = true;
    return 0;
}

extern "C" int tts_hip_snac_decode(tts_hip_ctx *c, const uint32_t *codes, uint32_t T_, const float *noise, float *pcm_out) {
    if (!c || !c->has_snac) return set_err("tts_hip_snac_decode: not a SNAC context (tts_hip_snac_create)");
    if (!c->finalized || !c->weights_present) return set_err("tts_hip_snac_decode: context not finalized");
    if (!codes || !pcm_out) return set_err("tts_hip_snac_decode: null argument");
    if (T_ == 0) return 0;
    const tts_hip_snac_desc &sd = c->snac;
    if (sd.max_frames && T_ > sd.max_frames) return set_err("tts_hip_snac_decode: %u tokens exceed snac.max_generation_size %u", T_, sd.max_frames);
    size_t n_codes = 0;
    for (uint32_t i = 0; i < sd.n_codebooks; i++) {
        if (T_ % sd.repeats[i]) return set_err("tts_hip_snac_decode: T=%u is not a multiple of the level-%u repeat %u", T_, i, sd.repeats[i]);
        n_codes += T_ / sd.repeats[i];
    }
    for (size_t i = 0; i < n_codes; i++)
        if (codes[i] >= (uint32_t) c->s_cbsize) return set_err("tts_hip_snac_decode: code %u >= codebook size %d", codes[i], c->s_cbsize);
    HIPCHK(hipSetDevice(c->device));
    CHK(ensure_packed_snac(c));
    const int T = (int) T_;
    // buffers sized for max_frames (or this call): largest activation = max over stages of C * L
    const size_t Tcap = std::max<size_t>(sd.max_frames, T_);
    if (!c->sbuf[0] || Tcap > c->dac_cap_frames) {
        size_t mx = (size_t) std::max(c->s_latent, c->s_c0), up = 1, noise_len = 0;
        for (auto &b : c->sblocks) { mx = std::max(mx, (size_t) b.cin * up); up *= b.stride; mx = std::max(mx, (size_t) b.cout * up); noise_len += up; }
        HIPCHK(hipStreamSynchronize(c->stream));
        for (int i = 0; i < 3; i++) { free_dev(c->sbuf[i]); c->sbuf[i] = nullptr; HIPCHK(hipMalloc((void **) &c->sbuf[i], mx * Tcap * 4)); }
        free_dev(c->s_noise); c->s_noise = nullptr;
        HIPCHK(hipMalloc((void **) &c->s_noise, noise_len * Tcap * 4));
        free_dev(c->s_codes); c->s_codes = nullptr;
        HIPCHK(hipMalloc((void **) &c->s_codes, Tcap * sd.n_codebooks * 4));
        c->dac_cap_frames = Tcap;
    }
    auto f32 = [&](size_t off) { return (const float *) (c->arena + off); };
    HIPCHK(hipMemcpyAsync(c->s_codes, codes, n_codes * 4, hipMemcpyHostToDevice, c->stream));
    size_t noise_len = 0;
    { size_t up = 1; for (auto &b : c->sblocks) { up *= b.stride; noise_len += up * (size_t) T; } }
    if (noise) HIPCHK(hipMemcpyAsync(c->s_noise, noise, noise_len * 4, hipMemcpyHostToDevice, c->stream));

    float *cur = c->sbuf[0], *t1 = c->sbuf[1], *t2 = c->sbuf[2];
    int L = T;
    SnacEmbedArgs ea{};
    ea.codes = c->s_codes; ea.codebook = f32(c->s_codebook); ea.proj_w = f32(c->s_projw); ea.proj_b = f32(c->s_projb);
    ea.n_cb = (int) sd.n_codebooks; ea.cb_size = c->s_cbsize; ea.cb_dim = c->s_cbdim; ea.latent = c->s_latent; ea.T = T; ea.out = cur;
    for (uint32_t i = 0; i < 4; i++) ea.rep[i] = i < sd.n_codebooks ? (int) sd.repeats[i] : 1;
    hipLaunchKernelGGL(snac_embed_kernel, dim3((T + 63) / 64, c->s_latent), dim3(64), 0, c->stream, ea);
    HIPCHK(hipGetLastError());
    auto dw = [&](const float *x, size_t w, size_t b, const float *ain, const float *aout, float *y, int C, int Ln, int pad, int dil) {
        hipLaunchKernelGGL(dwconv7_kernel, dim3((Ln + 255) / 256, C), dim3(256), 0, c->stream, x, f32(w), f32(b), ain, aout, y, C, Ln, pad, dil);
        return hipGetLastError() == hipSuccess ? 0 : set_err("dwconv7_kernel launch failed");
    };
    DacBatch bt;
    bt.n = 1; bt.frames = nullptr; bt.mult = 1; bt.tot_frames = (double) T;
    CHK(dw(cur, c->s_inw, c->s_inb, nullptr, nullptr, t1, c->s_latent, L, 3, 1));                                 // :141-142
    CHK(launch_conv(c, bt, t1, c->s_latent, L, c->s_upw, c->s_upb, 0, false, c->s_c0, 1, 0, 1, nullptr, false, cur));   // :143-144
    int C = c->s_c0;
    size_t noise_off = 0;
    for (auto &b : c->sblocks) {                                                                                  // build_layer, gnac.cpp:151-164
        ConvTArgs ta{};
        ta.x = cur; ta.w = f32(b.w); ta.b = f32(b.b); ta.alpha = f32(b.alpha); ta.y = t1; ta.cin = b.cin; ta.cout = b.cout; ta.L = L;
        ta.Lout = (L - 1) * b.stride - 2 * b.padding + 2 * b.stride; ta.stride = b.stride; ta.pad = b.padding;
        ta.frames = nullptr; ta.mult = 1;
        CHK(prof_begin(c, TTS_HIP_K_DAC_CONVT, 0, 2.0 * b.cin * (double) b.cout * 2 * ta.Lout));
        CHK(launch_convt(c, ta, b.w, 1));
        CHK(prof_end(c));
        std::swap(cur, t1);
        L = ta.Lout; C = b.cout;
        bt.tot_frames = (double) L;   // launch_conv's accounting: valid positions = tot_frames * mult
        if (noise) {                                                                                              // gnac.cpp:155-159
            CHK(launch_conv(c, bt, cur, C, L, b.noise_w, 0, 0, false, C, 1, 0, 1, nullptr, false, t1, 0, false, false));
            hipLaunchKernelGGL(noise_fma_kernel, dim3((unsigned) (((size_t) C * L + 255) / 256)), dim3(256), 0, c->stream, cur, (const float *) t1,
                               (const float *) (c->s_noise + noise_off), C, L);
            HIPCHK(hipGetLastError());
        }
        noise_off += (size_t) L;
        for (int r = 0; r < 3; r++) {                                                                             // build_residual_unit, groups > 1
            int dil = 1;
            for (int e = 0; e < r; e++) dil *= 3;
            // snake(in_alpha) on the way in, depthwise k7, bias, and the pointwise conv's snake(out_alpha) on the way out
            CHK(dw(cur, b.res[r].in_w, b.res[r].in_b, f32(b.res[r].in_alpha), f32(b.res[r].out_alpha), t1, C, L, 3 * dil, dil));
            CHK(launch_conv(c, bt, t1, C, L, b.res[r].out_w, b.res[r].out_b, 0, false, C, 1, 0, 1, cur, false, t2));
            std::swap(cur, t2);
        }
    }
    CHK(launch_conv(c, bt, cur, C, L, c->s_fw, c->s_fb, c->s_falpha, true, 1, 7, 3, 1, nullptr, true, t1));        // :152-155
    HIPCHK(hipMemcpyAsync(pcm_out, t1, (size_t) L * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// ------------------------------------------------------------------------------------------------

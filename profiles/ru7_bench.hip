// Stand-alone timing of the fused residual unit (resunit_t7_kernel<MI, KS2, SplitH2, WDMA, VAR>) at the DAC-44k sizes, variant against variant.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 profiles/ru7_bench.hip -o profiles/ru7_bench ; profiles/ru7_bench [n_utt] [frames]
// Prints per (channels, variant): ms per launch, issued fp16 TFLOP/s (three products), max |y - y(variant 0)|, and the cycle stamps of one launch.
#include "../tts.cpp_amd/csrc/dac_kernels.h"
#include "../tts.cpp_amd/csrc/dac_b3_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

static float frand(uint32_t &s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }

template <int MI, int KS2, bool WDMA, int VAR, int NI = 1>
static float run(const ResUnitArgs &a, int n, int reps) {
    using SP = SplitH2;
    constexpr int C = 32 * MI;
    const int xw = 256 * NI + 6 * a.dil;
    const size_t WST = (size_t) SP::NPL * ResT7<MI>::MAXCNT * 2 * C * 8;
    const size_t lds = 2 * WST * 2 + (size_t) 2 * SP::NPL * 2 * xw * 8 * 2 + (size_t) C * 24;
    CK(hipFuncSetAttribute((const void *) resunit_t7_kernel<MI, KS2, SP, WDMA, VAR, NI>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 grid((a.L + 256 * NI - 1) / (256 * NI), 1, n);
    hipLaunchKernelGGL((resunit_t7_kernel<MI, KS2, SP, WDMA, VAR, NI>), grid, dim3(512), lds, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((resunit_t7_kernel<MI, KS2, SP, WDMA, VAR, NI>), grid, dim3(512), lds, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

template <int MI, int KS2>
static void bench(int n, int frames, int mult, int dil) {
    constexpr int C = 32 * MI;
    const int L = frames * mult;
    const size_t ne = (size_t) n * C * L;
    uint32_t seed = 1234 + C;
    std::vector<float> hx(ne), hw7((size_t) C * C * 7), hw1((size_t) C * C), hb(4 * C);
    for (auto &v : hx) v = frand(seed);
    for (auto &v : hw7) v = frand(seed) * 0.05f;
    for (auto &v : hw1) v = frand(seed) * 0.1f;
    for (int i = 0; i < C; i++) { hb[i] = frand(seed) * 0.1f; hb[C + i] = frand(seed) * 0.1f; hb[2 * C + i] = 1.0f + 0.5f * frand(seed); hb[3 * C + i] = 1.0f + 0.5f * frand(seed); }
    float *x, *y, *y0, *w7, *w1, *b;
    CK(hipMalloc(&x, ne * 4)); CK(hipMalloc(&y, ne * 4)); CK(hipMalloc(&y0, ne * 4));
    CK(hipMalloc(&w7, hw7.size() * 4)); CK(hipMalloc(&w1, hw1.size() * 4)); CK(hipMalloc(&b, hb.size() * 4));
    CK(hipMemcpy(x, hx.data(), ne * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w7, hw7.data(), hw7.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w1, hw1.data(), hw1.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    using G = ResT7<MI>;
    const size_t WST = (size_t) 2 * G::MAXCNT * 2 * C * 8;
    const size_t nst = (size_t) (C / 16) * G::SPC + (size_t) (MI / 3) * ((C / 16) / KS2) + 1;
    __bf16 *wp;
    CK(hipMalloc(&wp, nst * WST * 2)); CK(hipMemset(wp, 0, nst * WST * 2));
    hipLaunchKernelGGL(pack_resunit_t7_kernel, dim3(1024), dim3(256), 0, 0, w7, w1, wp, C, KS2, 1);
    ResUnitArgs a{};
    a.x = x; a.w = wp; a.b7 = b; a.b1 = b + C; a.alpha_in = b + 2 * C; a.alpha_mid = b + 3 * C;
    a.L = L; a.dil = dil; a.pad = 3 * dil; a.frames = nullptr; a.mult = 1;
    const double flops = 2.0 * C * C * 8 * (double) L * n, issued = flops * 3.0;
    std::vector<float> h0(ne), h1(ne);
    const size_t nwg = (size_t) ((L + 255) / 256) * n;
    long long *st; CK(hipMalloc(&st, nwg * 5 * 8));
    auto report = [&](const char *name, float ms, bool first) {
        CK(hipMemcpy(first ? h0.data() : h1.data(), first ? y0 : y, ne * 4, hipMemcpyDeviceToHost));
        double md = 0;
        if (!first) for (size_t i = 0; i < ne; i += 7) md = std::max(md, (double) std::fabs(h0[i] - h1[i]));
        printf("C=%d dil=%d %-22s %8.3f ms  %7.1f TF fp32-equiv  %7.1f TF issued fp16  maxdiff %.2e\n", C, dil, name, ms, flops / ms / 1e9, issued / ms / 1e9, md);
        fflush(stdout);
    };
    auto stamps = [&]() {
        std::vector<long long> hs(nwg * 5);
        CK(hipMemcpy(hs.data(), st, nwg * 5 * 8, hipMemcpyDeviceToHost));
        double d[4] = {0, 0, 0, 0};
        for (size_t w = 0; w < nwg; w++) for (int k = 0; k < 4; k++) d[k] += (double) (hs[w * 5 + k + 1] - hs[w * 5 + k]);
        printf("   stamps (cycle counter ticks per workgroup, wave 0): prologue %.0f  k7 %.0f  mid-transform %.0f  k1 + epilogue %.0f\n", d[0] / nwg, d[1] / nwg, d[2] / nwg, d[3] / nwg);
    };
#define VARIANT(name, WD, V, first) VARIANT_NI(name, WD, V, 1, first)
#define VARIANT_NI(name, WD, V, NIv, first) do { a.y = first ? y0 : y; a.stamps = nullptr; float ms = run<MI, KS2, WD, V, NIv>(a, n, 3); report(name, ms, first); \
        CK(hipMemset(st, 0, nwg * 5 * 8)); a.stamps = st; run<MI, KS2, WD, V, NIv>(a, n, 1); a.stamps = nullptr; stamps(); } while (0)
    VARIANT("regs", false, 0, true);
    VARIANT("wdma", true, 0, false);
    VARIANT("wdma once", true, 2, false);
    VARIANT_NI("wdma once ni2", true, 2, 2, false);
    CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(y0)); CK(hipFree(w7)); CK(hipFree(w1)); CK(hipFree(b)); CK(hipFree(wp)); CK(hipFree(st));
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 64, frames = argc > 2 ? atoi(argv[2]) : 248;
    const int dil = argc > 3 ? atoi(argv[3]) : 1;
    bench<3, 3>(n, frames, 512, dil);
    bench<6, 4>(n, frames, 256, dil);
    return 0;
}

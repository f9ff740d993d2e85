// Stand-alone timing of the fused residual unit kernel variants (resunit_b3_kernel<.., SCHED>) at the DAC-44k sizes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 profiles/ru_bench.hip -o profiles/ru_bench ; profiles/ru_bench [n_utt] [frames]
// Prints per (channels, variant): ms per launch, fp32-equivalent TFLOP/s, issued bf16 TFLOP/s, max |y - y(variant 0)|.
#include "../tts.cpp_amd/csrc/dac_kernels.h"
#include "../tts.cpp_amd/csrc/dac_b3_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

static float frand(uint32_t &s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }

template <int MI, int KS, int KS2, int SCHED, int ABL = 0>
static float run(const ResUnitArgs &a, int n, int reps) {
    constexpr int C = 32 * MI;
    const ResUnitGeom g = resunit_geom(C, KS, KS2);
    const int xw = 256 + 6 * a.dil;
    const size_t lds = (size_t) 2 * g.WST * 2 + (size_t) 6 * xw * 8 * 2 + (size_t) C * 24;
    CK(hipFuncSetAttribute((const void *) resunit_b3_kernel<MI, KS, KS2, SCHED, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 grid((a.L + 255) / 256, 1, n);
    hipLaunchKernelGGL((resunit_b3_kernel<MI, KS, KS2, SCHED, ABL>), grid, dim3(512), lds, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((resunit_b3_kernel<MI, KS, KS2, SCHED, ABL>), grid, dim3(512), lds, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

template <int MI, int KS, int KS2>
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
    const ResUnitGeom g = resunit_geom(C, KS, KS2);
    __bf16 *wp;
    const size_t nw = (size_t) (g.n7 + g.n1 + 1) * g.WST;
    CK(hipMalloc(&wp, nw * 2)); CK(hipMemset(wp, 0, nw * 2));
    hipLaunchKernelGGL(pack_resunit_b3_kernel, dim3(1024), dim3(256), 0, 0, w7, w1, wp, C, KS, KS2);
    ResUnitArgs a{};
    a.x = x; a.w = wp; a.b7 = b; a.b1 = b + C; a.alpha_in = b + 2 * C; a.alpha_mid = b + 3 * C;
    a.L = L; a.dil = dil; a.pad = 3 * dil; a.frames = nullptr; a.mult = 1;
    const double flops = 2.0 * C * C * 8 * (double) L * n, issued = flops * (6.0 * 8 / 7 * 7 + 6.0) / 8;   // k = 7 padded to 8 taps, six products each
    std::vector<float> h0(ne), h1(ne);
    auto report = [&](const char *name, float ms, bool first) {
        CK(hipMemcpy(first ? h0.data() : h1.data(), first ? y0 : y, ne * 4, hipMemcpyDeviceToHost));
        double md = 0;
        if (!first) for (size_t i = 0; i < ne; i += 7) md = std::max(md, (double) std::fabs(h0[i] - h1[i]));
        printf("C=%d dil=%d %-10s %8.3f ms  %7.1f TF fp32-equiv  %7.1f TF issued bf16  maxdiff %.2e\n", C, dil, name, ms, flops / ms / 1e9, issued / ms / 1e9, md);
        fflush(stdout);
    };
    a.y = y0; report("sched0", run<MI, KS, KS2, 0>(a, n, 3), true);
    a.y = y;  report("sched1", run<MI, KS, KS2, 1>(a, n, 3), false);
    {   // cycle stamps of one launch: prologue / k = 7 stages / k = 1 passes / epilogue, mean over workgroups
        const size_t nwg = (size_t) ((L + 255) / 256) * n;
        long long *st; CK(hipMalloc(&st, nwg * 5 * 8)); CK(hipMemset(st, 0, nwg * 5 * 8));
        a.stamps = st; a.y = y; run<MI, KS, KS2, 1>(a, n, 1); a.stamps = nullptr;
        std::vector<long long> hs(nwg * 5);
        CK(hipMemcpy(hs.data(), st, nwg * 5 * 8, hipMemcpyDeviceToHost));
        double d[4] = {0, 0, 0, 0};
        for (size_t w = 0; w < nwg; w++) for (int k = 0; k < 4; k++) d[k] += (double) (hs[w * 5 + k + 1] - hs[w * 5 + k]);
        printf("   stamps (cycle counter ticks per workgroup): prologue %.0f  k7 %.0f  k1 %.0f  epilogue %.0f\n", d[0] / nwg, d[1] / nwg, d[2] / nwg, d[3] / nwg);
        CK(hipFree(st));
    }
    a.y = y;  report("abl15", run<MI, KS, KS2, 1, 15>(a, n, 3), false);
    a.y = y;  report("abl31 +nox", run<MI, KS, KS2, 1, 31>(a, n, 3), false);
    a.y = y;  report("abl47 +nok1", run<MI, KS, KS2, 1, 47>(a, n, 3), false);
    a.y = y;  report("abl63 mfma only", run<MI, KS, KS2, 1, 63>(a, n, 3), false);
    a.y = y;  report("abl32 nok1", run<MI, KS, KS2, 1, 32>(a, n, 3), false);
    a.y = y;  report("abl16 nox", run<MI, KS, KS2, 1, 16>(a, n, 3), false);
    CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(y0)); CK(hipFree(w7)); CK(hipFree(w1)); CK(hipFree(b)); CK(hipFree(wp));
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 64, frames = argc > 2 ? atoi(argv[2]) : 248;
    bench<3, 4, 3>(n, frames, 512, 1);

    bench<6, 2, 4>(n, frames, 256, 1);

    return 0;
}

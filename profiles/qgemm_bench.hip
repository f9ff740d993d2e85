// qgemm_bench.hip — standalone check + micro-benchmark of qgemm_tile_kernel (no torch, starts in a second).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o qgemm_bench profiles/qgemm_bench.hip && ./qgemm_bench [rows ...]
// Every (shape, rows, tile, k-slices) combination is checked against a per-element device reference (block dots in integers, the block
// terms added in k order in double) and timed over launches that cycle through NBUF weight copies, as gemm_bench.hip does.
#include "../tts.cpp_amd/csrc/qgemm_tile_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash_u(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__global__ void fill_i8_kernel(int8_t *p, size_t n, unsigned seed, int lo, int hi) {
    size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    if (i >= n) return;
    p[i] = (int8_t) (lo + (int) (hash_u((unsigned) i * 2654435761u + seed) % (unsigned) (hi - lo + 1)));
}
__global__ void fill_h_kernel(_Float16 *p, size_t n, unsigned seed, float scale) {
    size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    if (i >= n) return;
    p[i] = (_Float16) ((0.25f + (float) (hash_u((unsigned) i * 2654435761u + seed) & 0xFFFF) / 65536.0f) * scale);
}
__global__ void fill_f_kernel(float *p, size_t n, unsigned seed, float scale) {
    size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    if (i >= n) return;
    p[i] = ((float) (hash_u((unsigned) i * 2654435761u + seed) & 0xFFFF) / 32768.0f - 1.0f) * scale;
}
// reference: y[r][n] = sum_b (float) sumi * (wd * ad), terms added in k order (double accumulator like the oracle)
__global__ void ref_kernel(const int8_t *W, const _Float16 *wd, const int8_t *aq, const float *adT, int ldr, float *out, int R, int N, int K) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (n >= N || r >= R) return;
    double acc = 0.0;
    const int nb = K / 32;
    for (int b = 0; b < nb; b++) {
        int sumi = 0;
        for (int j = 0; j < 32; j++) sumi += (int) W[(size_t) n * K + b * 32 + j] * (int) aq[(size_t) r * K + b * 32 + j];
        acc += (double) ((float) sumi * ((float) wd[(size_t) n * nb + b] * adT[(size_t) b * ldr + r]));
    }
    out[(size_t) r * N + n] = (float) acc;
}
__global__ void fold_kernel(const float *slabs, float *out, size_t n, int ks, size_t stride) {
    size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int z = 0; z < ks; z++) s += slabs[z * stride + i];
    out[i] = s;
}

typedef void (*kern_t)(QTileArgs, TileMap);
#ifndef BENCH_DBG
#define BENCH_DBG 0   // -DBENCH_DBG=1: the kernels carry the QGEMM_BENCH_DBG / QGEMM_BENCH_STAMPS switches
#endif
struct Cfg { const char *name; int BM, BN, threads, S; size_t lds; kern_t k; };
#define CFGK(BM, BN, WM, WN, S, WPE, KG) { #BM "x" #BN "/" #WM "x" #WN "/s" #S "w" #WPE "kg" #KG, BM, BN, WM * WN * KG * 64, S, std::max((size_t) S * (BM + BN) * 144, (size_t) (KG - 1) * (BM / 32) * (BN / 32) * 4096), qgemm_tile_kernel<BM, BN, WM, WN, S, EPI_STORE, WPE, false, false, KG> }
#define CFG(BM, BN, WM, WN, S, WPE, PIPE) { #BM "x" #BN "/" #WM "x" #WN "/s" #S "w" #WPE "p" #PIPE, BM, BN, WM * WN * 64, S, (size_t) S * (BM + BN) * 144, BENCH_DBG ? (kern_t) qgemm_tile_kernel<BM, BN, WM, WN, S, EPI_STORE, WPE, PIPE, true> : (kern_t) qgemm_tile_kernel<BM, BN, WM, WN, S, EPI_STORE, WPE, PIPE, false> }
static Cfg cfgs[] = {
    CFG(64, 64, 2, 2, 2, 4, false), CFG(64, 64, 2, 2, 3, 3, true), CFG(64, 64, 2, 2, 4, 2, true), CFG(128, 64, 4, 2, 2, 4, false), CFG(128, 128, 4, 4, 2, 4, true), CFG(128, 128, 2, 4, 2, 3, true),
    CFG(128, 128, 2, 4, 3, 2, true), CFG(128, 128, 2, 4, 4, 2, true), CFG(256, 128, 4, 4, 2, 2, true),
    CFGK(64, 64, 2, 2, 2, 4, 2), CFGK(64, 64, 2, 2, 2, 4, 4), CFGK(128, 128, 2, 4, 2, 4, 2),
};




int main(int argc, char **argv) {
    const int NBUF = getenv("GEMM_BENCH_NBUF") ? atoi(getenv("GEMM_BENCH_NBUF")) : 40;
    struct Shape { const char *name; int N, K; bool splitk; } shapes[] = {
        {"qkv", 3072, 1024, false}, {"proj", 1024, 1024, true}, {"fc1", 4096, 1024, false}, {"fc2", 1024, 4096, true}, {"heads", 9792, 1024, false}};
    std::vector<int> Rs = {1024};
    if (argc > 1) { Rs.clear(); for (int i = 1; i < argc; i++) Rs.push_back(atoi(argv[i])); }
    const int RMAXB = 1024, LDR = 1024;
    const size_t wmax = (size_t) 9792 * 1024;
    const int ldw_max = 9984;
    int8_t *W, *aq; _Float16 *wd; float *wdT, *adT, *X, *out, *ref, *fold;
    CK(hipMalloc(&W, wmax * NBUF));
    CK(hipMalloc(&wd, wmax / 32 * 2 * NBUF));
    CK(hipMalloc(&wdT, (size_t) ldw_max * 128 * 4 * NBUF));
    CK(hipMalloc(&X, (size_t) RMAXB * 4096 * 4));
    CK(hipMalloc(&aq, (size_t) RMAXB * 4096));
    CK(hipMalloc(&adT, (size_t) LDR * 128 * 4));
    CK(hipMalloc(&out, (size_t) 8 * RMAXB * 9792 * 4));
    CK(hipMalloc(&ref, (size_t) RMAXB * 9792 * 4));
    CK(hipMalloc(&fold, (size_t) RMAXB * 9792 * 4));
    fill_i8_kernel<<<(wmax * NBUF + 255) / 256, 256>>>(W, wmax * NBUF, 12345u, -127, 127);
    fill_h_kernel<<<(wmax / 32 * NBUF + 255) / 256, 256>>>(wd, wmax / 32 * NBUF, 99u, 0.01f);
    fill_f_kernel<<<((size_t) RMAXB * 4096 + 255) / 256, 256>>>(X, (size_t) RMAXB * 4096, 777u, 3.0f);
    CK(hipMemset(adT, 0, (size_t) LDR * 128 * 4));
    CK(hipDeviceSynchronize());
    for (auto &c : cfgs) CK(hipFuncSetAttribute((const void *) c.k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    long long *stamps; CK(hipMalloc(&stamps, 8192 * 4 * 8)); std::vector<long long> h_st(8192 * 4);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> h_out((size_t) RMAXB * 9792), h_ref((size_t) RMAXB * 9792);
    const char *only = getenv("QGEMM_BENCH_ONLY");

    for (auto &sh : shapes) {
        if (getenv("QGEMM_BENCH_SHAPE") && strcmp(getenv("QGEMM_BENCH_SHAPE"), sh.name)) continue;
        const int nb = sh.K / 32, ldw = (sh.N + 255) & ~255;
        for (int i = 0; i < NBUF; i++)
            transpose_scales_kernel<<<dim3((ldw + 255) / 256, nb), 256>>>(wd + (size_t) i * sh.N * nb, wdT + (size_t) i * ldw * nb, sh.N, nb, ldw);
        for (int R : Rs) {
            quant_rows_q8t_kernel<<<dim3((sh.K / 256 + 3) / 4, R), 256>>>(X, sh.K, sh.K, aq, adT, LDR, R);
            ref_kernel<<<dim3((sh.N + 255) / 256, R), 256>>>(W, wd, aq, adT, LDR, ref, R, sh.N, sh.K);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h_ref.data(), ref, (size_t) R * sh.N * 4, hipMemcpyDeviceToHost));
            double refmax = 0;
            for (size_t i = 0; i < (size_t) R * sh.N; i++) refmax = std::max(refmax, (double) fabsf(h_ref[i]));
            double best = 1e9; char bestname[64] = "";
            for (auto &c : cfgs) {
                if (only && !strstr(only, c.name)) continue;
                if (c.lds > 160 * 1024) continue;
                for (int ks : {1, 2, 4}) {
                    if (ks > 1 && !sh.splitk) continue;
                    if (sh.K % (ks * 256)) continue;
                    if (sh.K / ks > 1024 && !getenv("QGEMM_BENCH_LONGK") && sh.splitk && ks < 4 && sh.K > 1024) continue;
                    QTileArgs qa{};
                    qa.g.K = sh.K; qa.g.N = sh.N; qa.g.R = R; qa.g.out = out; qa.g.ldo = sh.N; qa.g.H = 1024;
                    qa.g.kchunk = ks > 1 ? sh.K / ks : 0; qa.g.slab_stride = (int64_t) R * sh.N;
                    qa.aq = aq; qa.adT = adT; qa.ldr = LDR; qa.ldw = ldw; qa.dbg = getenv("QGEMM_BENCH_DBG") ? atoi(getenv("QGEMM_BENCH_DBG")) : 0;
                    TileMap tm{(R + c.BM - 1) / c.BM, (sh.N + c.BN - 1) / c.BN, ks};
                    const int total = tm.m_tiles * tm.n_tiles * ks, grid = (total + 7) / 8 * 8;
                    auto launch = [&](int i) {
                        qa.g.W = W + (size_t) (i % NBUF) * sh.N * sh.K;
                        qa.wdT = wdT + (size_t) (i % NBUF) * ldw * nb;
                        hipLaunchKernelGGL(c.k, dim3(grid), dim3(c.threads), c.lds, 0, qa, tm);
                    };
                    CK(hipMemset(out, 0xFF, (size_t) ks * R * sh.N * 4));
                    launch(0);
                    CK(hipGetLastError());
                    const float *res = out;
                    if (ks > 1) { fold_kernel<<<((size_t) R * sh.N + 255) / 256, 256>>>(out, fold, (size_t) R * sh.N, ks, (size_t) R * sh.N); res = fold; }
                    CK(hipDeviceSynchronize());
                    CK(hipMemcpy(h_out.data(), res, (size_t) R * sh.N * 4, hipMemcpyDeviceToHost));
                    double err = 0;
                    for (size_t i = 0; i < (size_t) R * sh.N; i++) {
                        const double d = fabs((double) h_out[i] - (double) h_ref[i]);
                        if (!(d <= err)) err = d;   // NaN-propagating
                    }
                    for (int i = 0; i < 5; i++) launch(i);
                    CK(hipDeviceSynchronize());
                    const int iters = 60;
                    CK(hipEventRecord(e0));
                    for (int i = 0; i < iters; i++) launch(i);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    const double us = ms * 1000.0 / iters, tops = 2.0 * R * sh.N * (double) sh.K / us * 1e-6;
                    const bool ok = err <= 2e-6 * refmax;
                    if (getenv("QGEMM_BENCH_STAMPS")) {
                        qa.stamps = stamps;
                        launch(1); CK(hipDeviceSynchronize());
                        CK(hipMemcpy(h_st.data(), stamps, (size_t) total * 32, hipMemcpyDeviceToHost));
                        double w = 0, st = 0, cp = 0;
                        for (int i = 0; i < total; i++) { w += h_st[i * 4]; st += h_st[i * 4 + 1]; cp += h_st[i * 4 + 2]; }
                        printf("   wave 0 of a workgroup, mean shader clocks: wait+barrier %.0f, stage issue %.0f, compute %.0f (per k-tile: %.0f / %.0f / %.0f)\n", w / total, st / total, cp / total,
                               w / total / (sh.K / ks / 128), st / total / (sh.K / ks / 128), cp / total / (sh.K / ks / 128));
                        qa.stamps = nullptr;
                    }
                    printf("%-6s R=%4d %-18s ks=%d wgs=%5d  %8.2f us  %7.1f Top/s  err %.2e / %.2e %s\n", sh.name, R, c.name, ks, total, us, tops, err, refmax, ok ? "ok" : "MISMATCH");
                    if (ok && us < best) { best = us; snprintf(bestname, sizeof bestname, "%s ks=%d", c.name, ks); }
                }
            }
            printf("BEST %-6s R=%4d %s %.2f us\n", sh.name, R, bestname, best);
        }
    }
    return 0;
}

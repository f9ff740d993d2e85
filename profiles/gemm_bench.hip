// gemm_bench.hip — standalone micro-benchmark + check of gemm_tile_kernel (no torch, starts in a second).
//   hipcc --offload-arch=gfx950 -O3 -o gemm_bench profiles/gemm_bench.hip && ./gemm_bench
// Every (shape, rows, tile, k-slices) combination is checked against a plain per-element device reference and
// timed over launches that cycle through NBUF weight copies (> the 256 MB infinity cache in total), which is
// what a decoder step sees: each matrix is touched once per step.
#include "../tts.cpp_amd/csrc/gemm_tile_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void ref_kernel(const _Float16 *W, const _Float16 *A, float *out, int R, int N, int K) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (n >= N || r >= R) return;
    float acc = 0.f;
    for (int k = 0; k < K; k++) acc += (float) W[(size_t) n * K + k] * (float) A[(size_t) r * K + k];
    out[(size_t) r * N + n] = acc;
}
__global__ void fill_kernel(_Float16 *p, size_t n, unsigned seed, float scale) {
    size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned) i * 2654435761u + seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    p[i] = (_Float16) (((float) (x & 0xFFFF) / 32768.0f - 1.0f) * scale);
}
__global__ void fold_kernel(const float *slabs, float *out, size_t n, int ks, size_t stride) {
    size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int z = 0; z < ks; z++) s += slabs[z * stride + i];
    out[i] = s;
}

typedef void (*kern_t)(GemmArgs, TileMap);
struct Cfg { const char *name; int BM, BN, threads, BK, S; kern_t k; };
#define CFG(BM, BN, WM, WN, BK, S) { #BM "x" #BN "/" #WM "x" #WN "/k" #BK "s" #S, BM, BN, WM * WN * 64, BK, S, gemm_tile_kernel<BM, BN, WM, WN, BK, S, EPI_STORE> }
#define CFGS(BM, BN, WM, WN) CFG(BM, BN, WM, WN, 64, 4), CFG(BM, BN, WM, WN, 128, 3), CFG(BM, BN, WM, WN, 128, 4)
static Cfg cfgs[] = {
    CFGS(32, 32, 1, 2), CFG(32, 32, 1, 2, 64, 8), CFG(32, 32, 1, 2, 128, 6), CFGS(32, 64, 1, 4), CFGS(64, 32, 2, 2), CFGS(64, 64, 2, 2), CFG(64, 64, 2, 2, 64, 6),
    CFGS(64, 64, 2, 4), CFGS(64, 128, 2, 2), CFGS(128, 64, 2, 2), CFGS(128, 64, 4, 2), CFGS(128, 128, 2, 4), CFG(128, 128, 2, 4, 64, 2),
    // round 4: 64 x 64 wave tiles (4 x 4 fragments: 8 LDS reads per 16 MFMAs instead of 6 per 8) — four waves, one per SIMD
    CFG(128, 128, 2, 2, 64, 4), CFG(128, 128, 2, 2, 64, 3), CFG(128, 128, 2, 2, 128, 2), CFG(128, 64, 2, 1, 64, 4), CFG(64, 128, 1, 2, 64, 4), CFG(256, 128, 4, 2, 64, 3), CFG(128, 256, 2, 4, 64, 3),
};

int main(int argc, char **argv) {
    const int NBUF = getenv("GEMM_BENCH_NBUF") ? atoi(getenv("GEMM_BENCH_NBUF")) : 40;   // 1: the same weight copy every launch (L2 / MALL hot): separates the L2 -> CU path from HBM
    struct Shape { const char *name; int N, K; bool splitk; } shapes[] = {
        {"qkv", 3072, 1024, false}, {"proj", 1024, 1024, true}, {"fc1", 4096, 1024, false}, {"fc2", 1024, 4096, true}, {"heads", 9792, 1024, false}};
    std::vector<int> Rs = {64, 128, 192, 256, 384, 512};
    if (argc > 1) { Rs.clear(); for (int i = 1; i < argc; i++) Rs.push_back(atoi(argv[i])); }
    const int RMAXB = 1024;
    _Float16 *W, *A; float *out, *ref, *fold;
    const size_t wmax = (size_t) 9792 * 1024;  // largest matrix (elements)
    CK(hipMalloc(&W, wmax * 2 * NBUF));
    CK(hipMalloc(&A, (size_t) RMAXB * 4096 * 2));
    CK(hipMalloc(&out, (size_t) 8 * RMAXB * 9792 * 4));
    CK(hipMalloc(&ref, (size_t) RMAXB * 9792 * 4));
    CK(hipMalloc(&fold, (size_t) RMAXB * 9792 * 4));
    fill_kernel<<<(wmax * NBUF + 255) / 256, 256>>>(W, wmax * NBUF, 12345u, 0.05f);
    fill_kernel<<<((size_t) RMAXB * 4096 + 255) / 256, 256>>>(A, (size_t) RMAXB * 4096, 777u, 1.0f);
    CK(hipDeviceSynchronize());
    for (auto &c : cfgs) CK(hipFuncSetAttribute((const void *) c.k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> h_out((size_t) RMAXB * 9792), h_ref((size_t) RMAXB * 9792);

    for (auto &sh : shapes) {
        for (int R : Rs) {
            ref_kernel<<<dim3((sh.N + 255) / 256, R), 256>>>(W, A, ref, R, sh.N, sh.K);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h_ref.data(), ref, (size_t) R * sh.N * 4, hipMemcpyDeviceToHost));
            double best = 1e9; char bestname[64] = "";
            for (auto &c : cfgs) {
                // GEMM_BENCH_ONLY=product: the two instances the decoder runs at 1024 rows (64x64/2x4/k128s3 and 128x128/2x4/k64s4), for counter passes
                if (getenv("GEMM_BENCH_ONLY") && strcmp(c.name, "64x64/2x4/k128s3") && strcmp(c.name, "128x128/2x4/k64s4")) continue;
                for (int ks : {1, 2, 4, 8}) {
                    if (ks > 1 && !sh.splitk) continue;
                    if (sh.K / ks < 256 || (sh.K / ks) % c.BK) continue;
                    GemmArgs g{};
                    g.K = sh.K; g.N = sh.N; g.R = R; g.A = A; g.lda = sh.K; g.out = out; g.ldo = sh.N;
                    g.kchunk = ks > 1 ? sh.K / ks : 0; g.slab_stride = (int64_t) RMAXB * sh.N;
                    TileMap tm{(R + c.BM - 1) / c.BM, (sh.N + c.BN - 1) / c.BN, ks};
                    const int total = tm.m_tiles * tm.n_tiles * tm.k_slices;
                    const int grid = (total + 7) / 8 * 8;
                    const size_t lds = (size_t) c.S * (c.BM + c.BN) * c.BK * 2;
                    if (lds > 160 * 1024) continue;   // does not fit a CU's LDS: the launch would fail and the check would read the previous configuration's output
                    // check (weight copy 0)
                    g.W = W;
                    CK(hipMemset(out, 0xFF, (size_t) ks * RMAXB * sh.N * 4));
                    hipLaunchKernelGGL(c.k, dim3(grid), dim3(c.threads), lds, 0, g, tm);
                    if (hipGetLastError() != hipSuccess) { printf("%-5s R=%3d %-12s ks=%d launch failed\n", sh.name, R, c.name, ks); continue; }
                    fold_kernel<<<((size_t) R * sh.N + 255) / 256, 256>>>(out, fold, (size_t) R * sh.N, ks, (size_t) RMAXB * sh.N);
                    CK(hipDeviceSynchronize());
                    CK(hipMemcpy(h_out.data(), fold, (size_t) R * sh.N * 4, hipMemcpyDeviceToHost));
                    double maxerr = 0, maxref = 0;
                    for (size_t i = 0; i < (size_t) R * sh.N; i++) { maxerr = fmax(maxerr, fabs(h_out[i] - h_ref[i])); maxref = fmax(maxref, fabs(h_ref[i])); }
                    const bool ok = maxerr <= 2e-4 * maxref && maxref > 0;
                    // time
                    const int iters = 80;
                    for (int i = 0; i < 8; i++) { g.W = W + (size_t) (i % NBUF) * wmax; hipLaunchKernelGGL(c.k, dim3(grid), dim3(c.threads), lds, 0, g, tm); }
                    CK(hipEventRecord(e0));
                    for (int i = 0; i < iters; i++) { g.W = W + (size_t) (i % NBUF) * wmax; hipLaunchKernelGGL(c.k, dim3(grid), dim3(c.threads), lds, 0, g, tm); }
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    const double us = ms * 1e3 / iters;
                    const double tf = 2.0 * R * sh.N * sh.K / us * 1e-6;
                    const double gbs = ((double) sh.N * sh.K * 2 + (double) R * sh.K * 2 + (double) R * sh.N * 4 * ks) / us * 1e-3;
                    printf("%-5s R=%3d %-12s ks=%d blocks=%4d  %7.2f us  %6.1f TFLOP/s  %6.0f GB/s  %s (err %.1e)\n", sh.name, R, c.name, ks, total, us, tf, gbs, ok ? "ok" : "MISMATCH", maxerr / (maxref + 1e-30));
                    if (ok && us < best) { best = us; snprintf(bestname, sizeof bestname, "%s ks=%d", c.name, ks); }
                }
            }
            printf("BEST %-5s R=%3d %-18s %7.2f us  %6.1f TFLOP/s\n", sh.name, R, bestname, best, 2.0 * R * sh.N * sh.K / best * 1e-6);
            fflush(stdout);
        }
    }
    return 0;
}

// gemv_bench.hip — standalone micro-benchmark + check of gemv_stream_kernel against gemm16_kernel (no torch).
//   hipcc --offload-arch=gfx950 -O3 -o gemv_bench profiles/gemv_bench.hip && ./gemv_bench [rows ...]
// Launches cycle through weight copies that together exceed the 256 MB infinity cache: a decoder step touches
// every matrix once.
#include "../tts.cpp_amd/csrc/gemv_stream_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void ref_kernel(const _Float16 *W, const float *A, float *out, int R, int N, int K) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (n >= N || r >= R) return;
    float acc = 0.f;
    for (int k = 0; k < K; k++) acc += (float) W[(size_t) n * K + k] * (float) (_Float16) A[(size_t) r * K + k];
    out[(size_t) r * N + n] = acc;
}
__global__ void fill_kernel(_Float16 *p, size_t n, unsigned seed, float scale) {
    size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned) i * 2654435761u + seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    p[i] = (_Float16) (((float) (x & 0xFFFF) / 32768.0f - 1.0f) * scale);
}
__global__ void fillf_kernel(float *p, size_t n, unsigned seed) {
    size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned) i * 2654435761u + seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    p[i] = (float) (x & 0xFFFF) / 32768.0f - 1.0f;
}
__global__ void fold_kernel(const float *slabs, float *out, size_t n, int ks, size_t stride) {
    size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int z = 0; z < ks; z++) s += slabs[z * stride + i];
    out[i] = s;
}

typedef void (*skern_t)(GemmArgs, StreamMap);
struct SCfg { const char *name; int nwv; skern_t k; };
static SCfg scfgs[] = {
    {"ws4", 4, gemv_stream_kernel<4, PRO_F32, EPI_STORE>},
    {"ws8", 8, gemv_stream_kernel<8, PRO_F32, EPI_STORE>},
    {"ws16", 16, gemv_stream_kernel<16, PRO_F32, EPI_STORE>},
};

int main(int argc, char **argv) {
    struct Shape { const char *name; int N, K, base_ks; } shapes[] = {
        {"dia_gateup", 16384, 2048, 1}, {"dia_down", 2048, 8192, 2}, {"dia_qkv", 3072, 2048, 1}, {"dia_o", 2048, 2048, 1},
        {"p_qkv", 3072, 1024, 1}, {"p_proj", 1024, 1024, 1}, {"p_fc1", 4096, 1024, 1}, {"p_fc2", 1024, 4096, 2}, {"p_heads", 9792, 1024, 1}};
    std::vector<int> Rs = {8};
    if (argc > 1) { Rs.clear(); for (int i = 1; i < argc; i++) Rs.push_back(atoi(argv[i])); }
    const int RMAXB = 16;
    const size_t wmax = (size_t) 16384 * 2048;  // largest matrix (elements)
    const int NBUF = 10;                         // 10 x 67 MB
    _Float16 *W; float *A, *out, *ref, *fold;
    CK(hipMalloc(&W, wmax * 2 * NBUF));
    CK(hipMalloc(&A, (size_t) RMAXB * 8192 * 4));
    CK(hipMalloc(&out, (size_t) 32 * RMAXB * 16384 * 4));
    CK(hipMalloc(&ref, (size_t) RMAXB * 16384 * 4));
    CK(hipMalloc(&fold, (size_t) RMAXB * 16384 * 4));
    fill_kernel<<<(wmax * NBUF + 255) / 256, 256>>>(W, wmax * NBUF, 12345u, 0.05f);
    fillf_kernel<<<((size_t) RMAXB * 8192 + 255) / 256, 256>>>(A, (size_t) RMAXB * 8192, 777u);
    CK(hipDeviceSynchronize());
    for (auto &c : scfgs) CK(hipFuncSetAttribute((const void *) c.k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> h_out((size_t) RMAXB * 16384), h_ref((size_t) RMAXB * 16384);

    for (auto &sh : shapes) {
        const size_t wstride = (size_t) sh.N * sh.K;
        const int nbuf = (int) std::min<size_t>(40, wmax * NBUF / wstride);
        for (int R : Rs) {
            ref_kernel<<<dim3((sh.N + 255) / 256, R), 256>>>(W, A, ref, R, sh.N, sh.K);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h_ref.data(), ref, (size_t) R * sh.N * 4, hipMemcpyDeviceToHost));
            auto check = [&](int ks) {
                fold_kernel<<<((size_t) R * sh.N + 255) / 256, 256>>>(out, fold, (size_t) R * sh.N, ks, (size_t) RMAXB * sh.N);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(h_out.data(), fold, (size_t) R * sh.N * 4, hipMemcpyDeviceToHost));
                double maxerr = 0, maxref = 0;
                for (size_t i = 0; i < (size_t) R * sh.N; i++) { maxerr = fmax(maxerr, fabs(h_out[i] - h_ref[i])); maxref = fmax(maxref, fabs(h_ref[i])); }
                return maxerr / (maxref + 1e-30);
            };
            const double wbytes = (double) sh.N * sh.K * 2;
            // ---- baseline: gemm16_kernel as the product launches it --------------------------------
            {
                GemmArgs g{};
                g.K = sh.K; g.N = sh.N; g.R = R; g.A = A; g.lda = sh.K; g.out = out; g.ldo = sh.N;
                const int ks = sh.base_ks;
                g.kchunk = ks > 1 ? sh.K / ks : 0; g.slab_stride = (int64_t) RMAXB * sh.N;
                const int nw = (sh.K / ks) / 256;
                const size_t lds = nw > 1 ? (size_t) nw * 4 * 64 * 4 : 0;
                auto launch = [&](int i) { g.W = W + (size_t) (i % nbuf) * wstride; hipLaunchKernelGGL((gemm16_kernel<1, PRO_F32, EPI_STORE, 1>), dim3(sh.N / 16, ks), dim3(nw * 64), lds, 0, g); };
                CK(hipMemset(out, 0xFF, (size_t) ks * RMAXB * sh.N * 4));
                launch(0);
                const double err = check(ks);
                const int iters = 60;
                for (int i = 0; i < 6; i++) launch(i);
                CK(hipEventRecord(e0));
                for (int i = 0; i < iters; i++) launch(i);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double us = ms * 1e3 / iters;
                printf("%-10s R=%2d gemm16 ks=%d wgs=%5d            %7.2f us  %6.0f GB/s  (err %.1e)\n", sh.name, R, ks, sh.N / 16 * ks, us, wbytes / us * 1e-3, err);
            }
            double best = 1e9; char bestname[96] = "";
            for (auto &c : scfgs) {
                for (int ks : {1, 2, 4, 8, 16, 32}) {
                    const int kslice = sh.K / ks;
                    if (kslice < 256 || kslice % 256) continue;
                    const int RS = R <= 8 ? 8 : 16;
                    const size_t lds = (size_t) RS * (kslice + 32) * 2;
                    if (lds > 160 * 1024) continue;
                    const int tiles = sh.N / 16;
                    for (int per_cu : {1, 2, 4}) {
                        if (per_cu * lds > 160 * 1024 || per_cu * c.nwv > 32) continue;
                        int grid = 256 * per_cu / ks * ks;
                        // no more workgroups than there is work for
                        const int need = (tiles + c.nwv - 1) / c.nwv * ks;
                        if (need < grid) { if (per_cu > 1) continue; grid = need; }
                        GemmArgs g{};
                        g.K = sh.K; g.N = sh.N; g.R = R; g.A = A; g.lda = sh.K; g.out = out; g.ldo = sh.N;
                        g.slab_stride = (int64_t) RMAXB * sh.N;
                        StreamMap sm{ks, kslice};
                        auto launch = [&](int i) { g.W = W + (size_t) (i % nbuf) * wstride; hipLaunchKernelGGL(c.k, dim3(grid), dim3(c.nwv * 64), lds, 0, g, sm); };
                        CK(hipMemset(out, 0xFF, (size_t) ks * RMAXB * sh.N * 4));
                        launch(0);
                        const double err = check(ks);
                        const int iters = 60;
                        for (int i = 0; i < 6; i++) launch(i);
                        CK(hipEventRecord(e0));
                        for (int i = 0; i < iters; i++) launch(i);
                        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                        const double us = ms * 1e3 / iters;
                        const bool ok = err < 2e-4;
                        printf("%-10s R=%2d %-5s ks=%2d grid=%4d (%d/CU) lds=%3zuK %7.2f us  %6.0f GB/s  %s (err %.1e)\n", sh.name, R, c.name, ks, grid, per_cu, lds >> 10, us, wbytes / us * 1e-3, ok ? "ok" : "MISMATCH", err);
                        if (ok && us < best) { best = us; snprintf(bestname, sizeof bestname, "%s ks=%d grid=%d", c.name, ks, grid); }
                    }
                }
            }
            printf("BEST %-10s R=%2d %-24s %7.2f us  %6.0f GB/s\n", sh.name, R, bestname, best, wbytes / best * 1e-3);
            fflush(stdout);
        }
    }
    return 0;
}

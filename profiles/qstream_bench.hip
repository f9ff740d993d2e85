// qstream_bench.hip — standalone check + micro-benchmark of qgemv_stream_kernel (5 .. 16 rows on a Q8_0-expanded matrix) at Orpheus-3B's shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o qstream_bench profiles/qstream_bench.hip && ./qstream_bench [rows]
// Every (shape, waves, depth, k-slices) combination is checked against a per-element device reference and timed over launches that cycle through
// NBUF weight copies (so the weights come from HBM, not from the memory-side cache); `read` is a plain 16-byte-per-lane stream of the same bytes.
#include "../tts.cpp_amd/csrc/gemv_stream_kernels.h"
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
    p[i] = (0.25f + (float) (hash_u((unsigned) i * 2654435761u + seed) & 0xFFFF) / 65536.0f) * scale;
}
__global__ void ref_kernel(const int8_t *W, const _Float16 *wd, const int8_t *aq, const float *ad, float *out, int R, int N, int K) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (n >= N || r >= R) return;
    double acc = 0.0;
    const int nb = K / 32;
    for (int b = 0; b < nb; b++) {
        int sumi = 0;
        for (int j = 0; j < 32; j++) sumi += (int) W[(size_t) n * K + b * 32 + j] * (int) aq[(size_t) r * K + b * 32 + j];
        acc += (double) ((float) sumi * ((float) wd[(size_t) n * nb + b] * ad[(size_t) r * nb + b]));
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
__global__ __launch_bounds__(256) void read_kernel(const int4v *p, size_t n16, int *sink) {
    int4v acc = {0, 0, 0, 0};
    const size_t stride = (size_t) gridDim.x * 256;
    size_t i = blockIdx.x * (size_t) 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const int4v a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + stride), c = __builtin_nontemporal_load(p + i + 2 * stride), d = __builtin_nontemporal_load(p + i + 3 * stride);
        acc ^= a ^ b ^ c ^ d;
    }
    for (; i < n16; i += stride) acc ^= __builtin_nontemporal_load(p + i);
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678) *sink = 1;
}

typedef void (*kern_t)(QGemmArgs, StreamMap);
struct Cfg { const char *name; int nwv, depth, rt, wl; kern_t k; };
#define CFG(NWV, D, RT) { "w" #NWV "d" #D "rt" #RT, NWV, D, RT, 0, qgemv_stream_kernel<NWV, D, RT> }
#define CFGL(NWV, D, RT) { "w" #NWV "d" #D "rt" #RT "lds", NWV, D, RT, 1, qgemv_stream_kernel<NWV, D, RT, true> }
static const Cfg CFGS[] = { CFG(4, 2, 1), CFG(8, 2, 1), CFGL(4, 2, 1), CFGL(8, 2, 1), CFG(4, 2, 2), CFG(8, 2, 2), CFGL(8, 2, 2), CFG(4, 2, 4), CFG(8, 2, 4) };
struct Shape { const char *name; int K, N; };
static const Shape SHAPES[] = { {"qkv", 3072, 5120}, {"o", 3072, 3072}, {"gate|up", 3072, 16384}, {"down", 8192, 3072}, {"head", 3072, 156940} };

int main(int argc, char **argv) {
    const int R = argc > 1 ? atoi(argv[1]) : 8;
    const int ITER = 40;
    int *sink; CK(hipMalloc(&sink, 4));
    for (const Cfg &c : CFGS) CK(hipFuncSetAttribute((const void *) c.k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *only_shape = getenv("QSTREAM_SHAPE"), *only_cfg = getenv("QSTREAM_CFG");   // e.g. QSTREAM_SHAPE=head QSTREAM_CFG=w8d2rt1 QSTREAM_KS=1: one line (counter runs)
    const int only_ks = getenv("QSTREAM_KS") ? atoi(getenv("QSTREAM_KS")) : 0;
    for (const Shape &sh : SHAPES) {
        if (only_shape && strcmp(only_shape, sh.name)) continue;
        const int K = sh.K, N = sh.N, nb = K / 32;
        const size_t wbytes = (size_t) N * K, sbytes = (size_t) N * nb * 2;
        const int NBUF = (int) std::max<size_t>(2, std::min<size_t>(24, (size_t) 1200e6 / (wbytes + sbytes)));
        int8_t *W; _Float16 *wd; int8_t *aq; float *ad, *out, *ref, *folded;
        CK(hipMalloc(&W, wbytes * NBUF)); CK(hipMalloc(&wd, sbytes * NBUF));
        CK(hipMalloc(&aq, (size_t) R * K)); CK(hipMalloc(&ad, (size_t) R * nb * 4));
        CK(hipMalloc(&out, (size_t) 16 * 64 * N * 4)); CK(hipMalloc(&ref, (size_t) R * N * 4)); CK(hipMalloc(&folded, (size_t) R * N * 4));
        for (int b = 0; b < NBUF; b++) {
            fill_i8_kernel<<<(unsigned) ((wbytes + 255) / 256), 256>>>(W + b * wbytes, wbytes, 17u, -127, 127);
            fill_h_kernel<<<(unsigned) (((size_t) N * nb + 255) / 256), 256>>>(wd + (size_t) b * N * nb, (size_t) N * nb, 29u, 0.01f);
        }
        fill_i8_kernel<<<(unsigned) (((size_t) R * K + 255) / 256), 256>>>(aq, (size_t) R * K, 5u, -127, 127);
        fill_f_kernel<<<(unsigned) (((size_t) R * nb + 255) / 256), 256>>>(ad, (size_t) R * nb, 7u, 0.02f);
        ref_kernel<<<dim3((N + 255) / 256, R), 256>>>(W, wd, aq, ad, ref, R, N, K);
        CK(hipDeviceSynchronize());
        std::vector<float> href((size_t) R * N), hout((size_t) R * N);
        CK(hipMemcpy(href.data(), ref, href.size() * 4, hipMemcpyDeviceToHost));
        double refmax = 0; for (float v : href) refmax = std::max(refmax, (double) fabsf(v));
        const double mb = (wbytes + sbytes) / 1e6;
        {   // the plain stream
            const size_t n16 = wbytes / 16;
            for (int it = 0; it < 3; it++) read_kernel<<<2048, 256>>>((const int4v *) (W + (it % NBUF) * wbytes), n16, sink);
            CK(hipEventRecord(e0));
            for (int it = 0; it < ITER; it++) read_kernel<<<2048, 256>>>((const int4v *) (W + (it % NBUF) * wbytes), n16, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-8s K %5d N %6d  %6.1f MB  plain read %7.2f us  %5.2f TB/s\n", sh.name, K, N, mb, ms * 1e3 / ITER, wbytes / (ms * 1e-3 / ITER) / 1e12);
        }
        const int tiles = (N + 15) / 16;
        for (int ks : {1, 2, 3, 4, 6, 8, 12, 16}) {
            if (K % (ks * 256) || (only_ks && ks != only_ks)) continue;
            const int KS = K / ks;
            const int srows = R <= 16 ? 16 : R <= 32 ? 32 : 64, RS = R <= 8 ? 8 : srows;
            const size_t lds = (size_t) (RS + 1) * KS + (size_t) RS * (KS / 32) * 4;
            if (lds > 64 * 1024) continue;
            for (const Cfg &c : CFGS) {
                if (c.rt * 16 != srows || (only_cfg && strcmp(only_cfg, c.name))) continue;
                const int items = tiles * ks;
                int grid = ((items + c.nwv - 1) / c.nwv + ks - 1) / ks * ks;
                const int maxwg = 256 * std::max(1, 16 / c.nwv) * 2;   // persistent beyond that: waves walk several tiles
                if (grid > maxwg) grid = maxwg / ks * ks;
                QGemmArgs qa{};
                qa.g.K = K; qa.g.N = N; qa.g.R = R; qa.g.out = out; qa.g.ldo = N; qa.g.slab_stride = (int64_t) 64 * N;
                qa.aq = aq; qa.ad = ad;
                const StreamMap sm{ks, KS};
                auto launch = [&](int b) {
                    qa.g.W = W + (size_t) b * wbytes; qa.wd = wd + (size_t) b * N * nb;
                    hipLaunchKernelGGL(c.k, dim3(grid), dim3(c.nwv * 64), ((lds + 15) & ~(size_t) 15) + (c.wl ? (size_t) c.nwv * c.depth * 4096 : 0), 0, qa, sm);
                };
                CK(hipMemset(out, 0xff, (size_t) 16 * 64 * N * 4));
                launch(0);
                fold_kernel<<<(unsigned) (((size_t) R * N + 255) / 256), 256>>>(out, folded, (size_t) R * N, ks, (size_t) 64 * N);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(hout.data(), folded, hout.size() * 4, hipMemcpyDeviceToHost));
                double err = 0; for (size_t i = 0; i < hout.size(); i++) { double d = fabs((double) hout[i] - href[i]); if (!(d <= err)) err = d; }
                for (int it = 0; it < 3; it++) launch(it % NBUF);
                CK(hipEventRecord(e0));
                for (int it = 0; it < ITER; it++) launch(it % NBUF);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double us = ms * 1e3 / ITER;
                printf("  ks %d %-6s wgs %5d  %7.2f us  %5.2f TB/s  err %.1e%s\n", ks, c.name, grid, us, (wbytes + sbytes) / (us * 1e-6) / 1e12, err / refmax, err / refmax < 1e-5 ? "" : "  WRONG");
            }
        }
        CK(hipFree(W)); CK(hipFree(wd)); CK(hipFree(aq)); CK(hipFree(ad)); CK(hipFree(out)); CK(hipFree(ref)); CK(hipFree(folded));
    }
    return 0;
}

// What does v_mfma_f32_32x32x16_bf16 sustain in the accumulation patterns the codec kernels use?  (MI355X_MICROARCH.md: 32 cycles per SIMD back to back.)
//   NACC accumulators, term-major (every accumulator every NACC-th MFMA), 512-thread workgroups, LDS request pins the workgroups per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef float float16d __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8d __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NACC, int REUSE>
__global__ __launch_bounds__(512, 2) void k(float *out, int iters, float seed) {
    extern __shared__ char smem[];
    float16d acc[NACC];
    for (int i = 0; i < NACC; i++) for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
    bf16x8d a[3], b[3];
    uint32_t h = (threadIdx.x + 977u * blockIdx.x) * 2654435761u + (uint32_t) seed;
    for (int p = 0; p < 3; p++) for (int e = 0; e < 8; e++) {
        if (seed > 1.5f) {   // random mantissas: the data-dependent power draw of a real workload
            h = h * 1664525u + 1013904223u; a[p][e] = (__bf16) (((h >> 8) & 0xffff) / 32768.0f - 1.0f);
            h = h * 1664525u + 1013904223u; b[p][e] = (__bf16) (((h >> 8) & 0xffff) / 32768.0f - 1.0f);
        } else { a[p][e] = (__bf16) (seed + threadIdx.x + p); b[p][e] = (__bf16) (seed * e + p); }
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REUSE; r++)
#pragma unroll
            for (int tm = 0; tm < 6; tm++)
#pragma unroll
                for (int i = 0; i < NACC; i++)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm % 3], b[(tm + i) % 3], acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; i++) for (int e = 0; e < 16; e++) s += acc[i][e];
    if (s == 12345.f) out[threadIdx.x] = s;
}

template <int NACC, int REUSE>
static void run(const char *name, int lds, float seed = 1.0f) {
    float *out; CK(hipMalloc(&out, 4096));
    CK(hipFuncSetAttribute((const void *) k<NACC, REUSE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int iters = 2000, grid = 256 * 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<NACC, REUSE>), dim3(grid), dim3(512), lds, 0, out, iters, seed);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<NACC, REUSE>), dim3(grid), dim3(512), lds, 0, out, iters, seed);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double n = (double) grid * 8 * iters * REUSE * 6 * NACC;
    printf("%-28s lds %6d  %8.3f ms  %7.1f TF  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", name, lds, ms, n * 32768 / ms / 1e9, ms * 1e-3 * 2.4e9 * 1024 / n);
}
int main() {
    run<3, 4>("3 acc, 1 WG/CU", 104 * 1024);
    run<3, 4>("3 acc, 2 WG/CU", 60 * 1024);
    run<6, 2>("6 acc, 1 WG/CU", 104 * 1024);
    run<6, 2>("6 acc, 2 WG/CU", 60 * 1024);
    run<1, 12>("1 acc (dependent chain)", 104 * 1024);
    run<2, 6>("2 acc", 104 * 1024);
    run<3, 4>("3 acc, 1 WG/CU, random data", 104 * 1024, 2.0f);
    run<3, 4>("3 acc, 2 WG/CU, random data", 60 * 1024, 2.0f);
    run<6, 2>("6 acc, 1 WG/CU, random data", 104 * 1024, 2.0f);
    return 0;
}

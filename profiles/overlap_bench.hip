// Can an HBM-bound kernel (the decoder's self-attention: 55 VGPRs, 4 KB of LDS per 256-thread workgroup) make progress UNDER an MFMA-bound one
// (the codec's convolutions) when the two are resident on the same CUs — and what does it cost the MFMA kernel?  
//   M<NT, MINW>: back-to-back v_mfma_f32_32x32x16_bf16 on random operands re-read from LDS, 12 accumulators (192 registers) per wave:
//                  M8 = 512 threads, 2 waves per SIMD, 150 KB of LDS  -> the footprint of today's codec workgroups (one per CU, nothing else fits)
//                  M4 = 256 threads, 1 wave per SIMD,  81 KB of LDS   -> the "slim" footprint: one per CU, half the registers and LDS stay free
//   A: every 256-thread workgroup streams a contiguous 256 KB slice of a 8 GB buffer with 8 x 16-byte loads in flight per lane (attn_kernel's
//      access pattern: whole 256-byte lines), ~40 registers, 4 KB of LDS.
// Each is timed alone, then both are launched on two streams (A on the high-priority one) with the same total work; wall = max of the two ends.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef float float16d __attribute__((ext_vector_type(16)));
typedef float float4d __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8d __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NT, int MINW>
__global__ __launch_bounds__(NT, MINW) void mfma_k(float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NACC = 12;
    bf16x8d *frag = (bf16x8d *) smem;                 // 1024 fragments of 16 bytes: random operands
    uint32_t h = (threadIdx.x + 977u * blockIdx.x) * 2654435761u + 12345u;
    for (int i = threadIdx.x; i < 1024; i += NT) {
        bf16x8d v;
        for (int e = 0; e < 8; e++) { h = h * 1664525u + 1013904223u; v[e] = (__bf16) (((h >> 8) & 0xffff) / 32768.0f - 1.0f); }
        frag[i] = v;
    }
    __syncthreads();
    float16d acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; it++) {
        // one "k-step group": 3 A fragments + 3 B fragments from LDS (conflict-free 16-byte reads), 6 x NACC / 3 MFMAs on them
        bf16x8d a[3], b[3];
#pragma unroll
        for (int p = 0; p < 3; p++) { a[p] = frag[((it * 6 + p) * 64 + lane) & 1023]; b[p] = frag[((it * 6 + 3 + p) * 64 + lane) & 1023]; }
#pragma unroll
        for (int tm = 0; tm < 6; tm++)
#pragma unroll
            for (int i = 0; i < NACC; i++)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm % 3], b[(tm + i) % 3], acc[i], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++)
#pragma unroll
        for (int e = 0; e < 16; e++) s += acc[i][e];
    if (s == 12345.f) out[threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void stream_k(const float4d *buf, float *out, int64_t vec_per_wg) {
    __shared__ float red[1024];
    const float4d *p = buf + (int64_t) blockIdx.x * vec_per_wg;
    float4d s = {0, 0, 0, 0};
    for (int64_t i = threadIdx.x; i < vec_per_wg; i += 256 * 8) {
        float4d v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = __builtin_nontemporal_load(p + i + j * 256);
#pragma unroll
        for (int j = 0; j < 8; j++) s += v[j];
    }
    red[threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    __syncthreads();
    if (threadIdx.x == 0 && red[5] == 1234.5f) out[blockIdx.x] = red[7];
}

int main() {
    float *out; CK(hipMalloc(&out, 1 << 20));
    const size_t bytes = (size_t) 8 << 30;
    float4d *buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
    int least = 0, greatest = 0; CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t sm, sa; CK(hipStreamCreateWithPriority(&sm, hipStreamNonBlocking, least)); CK(hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, greatest));
    CK(hipFuncSetAttribute((const void *) mfma_k<512, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *) mfma_k<256, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t m0, m1, a0, a1; CK(hipEventCreate(&m0)); CK(hipEventCreate(&m1)); CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1));
    const int64_t vec_per_wg = (256 * 1024) / 16;
    const int a_grid = (int) (bytes / (256 * 1024));          // 32768 workgroups, 8 GB per launch
    const int A_REP = 6;                                      // 48 GB of reads ~ 8.5 ms alone
    auto launch_a = [&]() { for (int r = 0; r < A_REP; r++) hipLaunchKernelGGL(stream_k, dim3(a_grid), dim3(256), 0, sa, buf, out, vec_per_wg); };
    struct Cfg { const char *name; int nt, minw, lds, wg_per_cu_grid; int iters; };
    // total MFMAs equal: grid 256 * 8 workgroups of 8 waves x iters == 256 * 8 workgroups of 4 waves x 2 iters
    const int IT = 1500;
    Cfg cfgs[] = {
        {"M8: 8 waves, 2 per SIMD, 150 KB LDS (today's codec footprint)", 512, 2, 150 * 1024, 8, IT},
        {"M4: 4 waves, 1 per SIMD (218 registers), 81 KB LDS (slim: half the CU stays free)", 256, 1, 81 * 1024, 8, 2 * IT},
        {"M4x2: 4 waves, 2 workgroups per CU, 75 KB LDS (control: same occupancy as M8)", 256, 2, 75 * 1024, 16, IT},
    };
    auto launch_m = [&](const Cfg &c) {
        if (c.nt == 512) hipLaunchKernelGGL((mfma_k<512, 2>), dim3(256 * c.wg_per_cu_grid), dim3(512), c.lds, sm, out, c.iters);
        else hipLaunchKernelGGL((mfma_k<256, 2>), dim3(256 * c.wg_per_cu_grid), dim3(256), c.lds, sm, out, c.iters);
    };
    // A alone
    launch_a(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a0, sa)); launch_a(); CK(hipEventRecord(a1, sa)); CK(hipDeviceSynchronize());
    float ta; CK(hipEventElapsedTime(&ta, a0, a1));
    printf("A alone: %.2f ms for %.0f GB = %.2f TB/s\n", ta, A_REP * bytes / 1e9, A_REP * bytes / ta / 1e9);
    for (const Cfg &c : cfgs) {
        launch_m(c); CK(hipDeviceSynchronize());
        CK(hipEventRecord(m0, sm)); launch_m(c); CK(hipEventRecord(m1, sm)); CK(hipDeviceSynchronize());
        float tm; CK(hipEventElapsedTime(&tm, m0, m1));
        const double n = 256.0 * c.wg_per_cu_grid * (c.nt / 64) * c.iters * 6 * 12;
        // together: M first (it fills the CUs), then A on the high-priority stream
        hipEvent_t w0, w1; CK(hipEventCreate(&w0)); CK(hipEventCreate(&w1));
        CK(hipEventRecord(w0, sm));
        CK(hipStreamWaitEvent(sa, w0, 0));
        CK(hipEventRecord(m0, sm)); launch_m(c); CK(hipEventRecord(m1, sm));
        CK(hipEventRecord(a0, sa)); launch_a(); CK(hipEventRecord(a1, sa));
        CK(hipStreamWaitEvent(sm, a1, 0)); CK(hipEventRecord(w1, sm));
        CK(hipDeviceSynchronize());
        float tm2, ta2, wall; CK(hipEventElapsedTime(&tm2, m0, m1)); CK(hipEventElapsedTime(&ta2, a0, a1)); CK(hipEventElapsedTime(&wall, w0, w1));
        printf("%s\n   alone %.2f ms = %.0f TF | together: M %.2f ms, A %.2f ms, wall %.2f ms vs %.2f serial -> %.2fx (1.0 = time-shared, %.2f = perfect overlap)\n",
               c.name, tm, n * 32768 / tm / 1e9, tm2, ta2, wall, tm + ta, (tm + ta) / wall, (tm + ta) / (tm > ta ? tm : ta));
    }
    return 0;
}

// stride_read_bench.hip — round 5: is the layout of Dia's cross K / V cache ([positions][heads x 128] fp32 per row-sequence: a (head, key-slice) workgroup
// reads 512-byte pieces 8 KB apart) what keeps its attention launch at 4.6 TB/s?  The same 134 MB per launch read by 1024 workgroups x 256 threads
//   A  as the attention reads them: per workgroup 128 keys x 512 B at stride 8 KB, from a K and a V tensor, 64 keys per round trip;
//   B  the same bytes per workgroup as two contiguous 64 KB pieces (a head-major cache);
//   C  B with all 128 KB of a workgroup requested at once (32 x 16 bytes per lane).
// 18 "layers" of 134 MB each are walked per timed pass (2.4 GB: nothing survives in the 256 MB memory-side cache).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/stride_read_bench profiles/stride_read_bench.hip && /tmp/stride_read_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float float4v __attribute__((ext_vector_type(4)));
constexpr int S = 1024, NH = 16, HD = 128, ROWS = 8, NZ = 8, A = NH * HD;

// MODE 0: strided pieces (interleaved heads); 1: contiguous per (row, head, slice), two round trips; 2: contiguous, one round trip
template <int MODE>
__global__ __launch_bounds__(256) void read_kernel(const float *k, const float *v, float *out) {
    const int z = blockIdx.x, r = blockIdx.y, h = blockIdx.z, tid = threadIdx.x;
    float4v acc = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 0) {
        const float *kb = k + ((int64_t) r * S + z * 128) * A + h * HD, *vb = v + ((int64_t) r * S + z * 128) * A + h * HD;
        const int grp = tid >> 5, e4 = tid & 31;
        for (int j0 = 0; j0 < 128; j0 += 64) {
            float4v a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int j = j0 + grp + 8 * u;
                a[u] = __builtin_nontemporal_load((const float4v *) (kb + (int64_t) j * A + e4 * 4));
                b[u] = __builtin_nontemporal_load((const float4v *) (vb + (int64_t) j * A + e4 * 4));
            }
#pragma unroll
            for (int u = 0; u < 8; u++) acc += a[u] * b[u];
        }
    } else {
        const float *kb = k + (((int64_t) r * NH + h) * S + z * 128) * HD, *vb = v + (((int64_t) r * NH + h) * S + z * 128) * HD;
        constexpr int NR = MODE == 1 ? 2 : 1, PER = 16 / NR;
        for (int j0 = 0; j0 < NR; j0++) {
            float4v a[PER], b[PER];
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const int off = ((j0 * PER + u) * 256 + tid) * 4;
                a[u] = __builtin_nontemporal_load((const float4v *) (kb + off));
                b[u] = __builtin_nontemporal_load((const float4v *) (vb + off));
            }
#pragma unroll
            for (int u = 0; u < PER; u++) acc += a[u] * b[u];
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.0f;
}
__global__ void fill(float *p, size_t n) {
    size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = (float) (i & 1023) * 1e-3f;
}
template <int MODE>
static void run(const char *name, float *k, float *v, float *out, int layers, size_t per) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int l = 0; l < layers; l++) hipLaunchKernelGGL(read_kernel<MODE>, dim3(NZ, ROWS, NH), dim3(256), 0, 0, k + l * per, v + l * per, out);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int i = 0; i < reps; i++)
        for (int l = 0; l < layers; l++) hipLaunchKernelGGL(read_kernel<MODE>, dim3(NZ, ROWS, NH), dim3(256), 0, 0, k + l * per, v + l * per, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps / layers, bytes = 2.0 * per * 4;
    printf("%-70s %7.2f us per launch  %6.0f GB/s\n", name, us, bytes / us * 1e-3);
}
int main() {
    const int layers = 18;
    const size_t per = (size_t) ROWS * S * A;   // floats of one layer's K (and of its V)
    float *k, *v, *out;
    CK(hipMalloc(&k, per * layers * 4)); CK(hipMalloc(&v, per * layers * 4)); CK(hipMalloc(&out, 4));
    fill<<<(unsigned) ((per * layers + 255) / 256), 256>>>(k, per * layers); fill<<<(unsigned) ((per * layers + 255) / 256), 256>>>(v, per * layers);
    CK(hipDeviceSynchronize());
    printf("%d layers x %.1f MB of K + V per launch, %d workgroups x 256 threads (launches back to back in one stream: ~2 us of launch gap each)\n", layers, 2.0 * per * 4 / 1e6, NZ * ROWS * NH);
    run<0>("A  512-byte pieces 8 KB apart (the cache layout), 64 keys per round trip", k, v, out, layers, per);
    run<1>("B  contiguous 64 KB + 64 KB per workgroup, two round trips", k, v, out, layers, per);
    run<2>("C  contiguous, everything requested at once", k, v, out, layers, per);
    run<0>("A  again", k, v, out, layers, per);
    return 0;
}

// valu_rate.hip — what one SIMD issues per cycle on gfx950 for the instruction mix of qgemm_tile_kernel's scaling step.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 -o valu_rate profiles/valu_rate.hip && ./valu_rate
// Variants (one workgroup per CU, W waves per SIMD, s_memtime around an unrolled loop):
//   0 fma     16 independent v_fma_f32 per round
//   1 mix     the scaling step without its MFMA: 16 x (v_mul_f32, v_add_f32 literal, v_fmac_f32)
//   2 mix+mfma  one v_mfma_i32_32x32x32_i8 per 48 VALU instructions (the kernel's item), result consumed one item later
//   3 pk      the packed form: 8 x (v_pk_mul_f32, v_pk_add_f32, v_pk_fma_f32)
//   4 cvtmix  16 x (v_cvt_f32_i32, v_mul_f32, v_fmac_f32)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int int16v __attribute__((ext_vector_type(16)));
typedef int int4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float2v __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int VAR>
__global__ void rate_kernel(float *out, long long *cycles, int rounds, float s, int4v a, int4v b) {
    float acc[16], wd[16];
    for (int e = 0; e < 16; e++) { acc[e] = threadIdx.x * 0.001f + e; wd[e] = s + e * 0.125f; }
    float ad = s * 3.0f + threadIdx.x;
    int16v magic;
    for (int e = 0; e < 16; e++) magic[e] = 0x4B400000;
    float16v z[2];
    z[0] = __builtin_bit_cast(float16v, magic);
    z[1] = z[0];
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rounds; r++) {
#pragma unroll
        for (int it = 0; it < 4; it++) {
            if (VAR == 0) {
#pragma unroll
                for (int e = 0; e < 16; e++) acc[e] = __builtin_fmaf(acc[e], wd[e], ad);
            } else if (VAR == 1 || VAR == 2) {
                if (VAR == 2) asm volatile("" : "+v"(a));
                if (VAR == 2) z[(it + 1) & 1] = __builtin_bit_cast(float16v, __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, magic, 0, 0, 0));
                const float16v zz = z[it & 1];
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const float u = wd[e] * ad;
                    acc[e] = __builtin_fmaf(zz[e] - 12582912.0f, u, acc[e]);
                }
                if (VAR == 1) { asm volatile("" : "+v"(z[0]), "+v"(z[1])); }
                asm volatile("" : "+v"(ad));
            } else if (VAR == 3) {
                const float16v zz = z[it & 1];
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    float2v zf = (float2v){zz[2 * e], zz[2 * e + 1]};
                    zf -= (float2v){12582912.0f, 12582912.0f};
                    const float2v u = (float2v){wd[2 * e], wd[2 * e + 1]} * (float2v){ad, ad};
                    const float2v c = __builtin_elementwise_fma(zf, u, (float2v){acc[2 * e], acc[2 * e + 1]});
                    acc[2 * e] = c[0]; acc[2 * e + 1] = c[1];
                }
                asm volatile("" : "+v"(z[0]), "+v"(z[1]));
                asm volatile("" : "+v"(ad));
            } else {
                const int16v zi = __builtin_bit_cast(int16v, z[it & 1]);
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const float u = wd[e] * ad;
                    acc[e] = __builtin_fmaf((float) zi[e], u, acc[e]);
                }
                asm volatile("" : "+v"(z[0]), "+v"(z[1]));
                asm volatile("" : "+v"(ad));
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0;
    for (int e = 0; e < 16; e++) sum += acc[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum + z[0][3] + z[1][5];
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int VAR>
static void run(const char *name, int waves_per_simd, int instr_per_item, float *out, long long *cyc) {
    const int rounds = 2000, blocks = 256;
    int4v a = {0x01020304, 0x05060708, 0x01010101, 0x02020202}, b = {0x01010101, 0x01010101, 0x02020202, 0x01010101};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    rate_kernel<VAR><<<blocks, 256 * waves_per_simd>>>(out, cyc, 10, 1.5f, a, b);
    CK(hipEventRecord(e0));
    rate_kernel<VAR><<<blocks, 256 * waves_per_simd>>>(out, cyc, rounds, 1.5f, a, b);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(blocks);
    CK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
    double mean = 0; for (auto v : h) mean += (double) v; mean /= blocks;
    const double items = (double) rounds * 4;
    // s_memtime ticks at 100 MHz: convert through the wall time
    printf("%-9s waves/SIMD %d: %8.3f ms wall, %7.1f ns per item per wave-slot, %6.1f ns per item per SIMD (= %5.1f clk at 2.4 GHz; %d VALU per item -> %4.2f clk each); memtime %.0f ticks\n", name, waves_per_simd, ms,
           ms * 1e6 / items, ms * 1e6 / items / waves_per_simd, ms * 1e6 / items / waves_per_simd * 2.4, instr_per_item, ms * 1e6 / items / waves_per_simd * 2.4 / instr_per_item, mean);
}

int main() {
    float *out; long long *cyc;
    CK(hipMalloc(&out, 256 * 1024 * 4 * 4));
    CK(hipMalloc(&cyc, 256 * 8));
    for (int w : {1, 2, 4}) {
        run<0>("fma", w, 16, out, cyc);
        run<1>("mix", w, 48, out, cyc);
        run<2>("mix+mfma", w, 48, out, cyc);
        run<3>("pk", w, 24, out, cyc);
        run<4>("cvtmix", w, 48, out, cyc);
    }
    return 0;
}

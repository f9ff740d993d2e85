// Bit-for-bit check of the DPP / permlane reductions of parler_kernels.h against the __shfl_xor butterflies they replace.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o profiles/wave_sum_check profiles/wave_sum_check.hip && profiles/wave_sum_check
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../tts.cpp_amd/csrc/parler_kernels.h"

__global__ void check_kernel(const float *x, unsigned *bad, int n_waves) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_waves * 64) return;
    const float v = x[i];
    float s = v, m = v, r = v, rm = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); m = fmaxf(m, __shfl_xor(m, o)); }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { r += __shfl_xor(r, o); rm = fmaxf(rm, __shfl_xor(rm, o)); }
    const float s2 = wave_sum(v), m2 = wave_max(v), r2 = row16_sum(v), rm2 = row16_max(v);
    unsigned b = 0;
    if (__builtin_bit_cast(unsigned, s) != __builtin_bit_cast(unsigned, s2)) b |= 1;
    if (__builtin_bit_cast(unsigned, m) != __builtin_bit_cast(unsigned, m2)) b |= 2;
    if (__builtin_bit_cast(unsigned, r) != __builtin_bit_cast(unsigned, r2)) b |= 4;
    if (__builtin_bit_cast(unsigned, rm) != __builtin_bit_cast(unsigned, rm2)) b |= 8;
    if (b) atomicOr(bad, b);
}

int main() {
    const int n_waves = 1 << 14;
    std::vector<float> h((size_t) n_waves * 64);
    srand(7);
    for (size_t i = 0; i < h.size(); i++) {
        const float u = (float) rand() / RAND_MAX - 0.5f;
        h[i] = u * std::ldexp(1.0f, rand() % 24 - 12);   // mixed magnitudes: the order of additions shows in the last bits
    }
    float *d = nullptr; unsigned *bad = nullptr, hb = 0;
    if (hipMalloc((void **) &d, h.size() * 4) != hipSuccess || hipMalloc((void **) &bad, 4) != hipSuccess) return 2;
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(bad, 0, 4);
    hipLaunchKernelGGL(check_kernel, dim3(n_waves / 4), dim3(256), 0, 0, d, bad, n_waves);
    hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    printf("wave_sum / wave_max / row16_sum / row16_max against the __shfl_xor butterflies over %d waves: %s (mask %u)\n", n_waves, hb ? "DIFFER" : "bit-identical", hb);
    return hb ? 1 : 0;
}

// wave_ops.h — vector typedefs, cross-lane reductions (DPP / permlane, no LDS crossbar) and ggml's GELU: shared by every kernel header.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef int int2v __attribute__((ext_vector_type(2)));

#define LN_EPS 1e-5f  // model.cpp:414 "parler always uses default eps"

// Cross-lane reductions without the LDS crossbar.  __shfl_xor compiles to ds_bpermute_b32 + s_waitcnt lgkmcnt(0): six dependent LDS round
// trips per wave_sum, which is most of a LayerNorm's time at batch 1 (profiles/r03/b1_chain.txt).  The xor butterfly 32, 16, 8, 4, 2, 1 is
// reproduced bit for bit: v_permlane32_swap / v_permlane16_swap of (v, v) leave lane i's value in one result and lane (i ^ 32) / (i ^ 16)'s in the
// other (fp add and max commute); inside a row of 16 lanes the value has period 8 after the xor-8 step, period 4 after the xor-4 step ..., so
// rotating the row by 8, 4, 2, 1 (DPP row_ror) pairs every lane with the same partner value the xor would (profiles/wave_sum_check.hip).
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float v) {   // == v += __shfl_xor(v, 8); ... 4; 2; 1 for aligned groups of 16 lanes
    v += dpp_f<0x128>(v); v += dpp_f<0x124>(v); v += dpp_f<0x122>(v); v += dpp_f<0x121>(v);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_f<0x128>(v)); v = fmaxf(v, dpp_f<0x124>(v)); v = fmaxf(v, dpp_f<0x122>(v)); v = fmaxf(v, dpp_f<0x121>(v));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    v = __builtin_bit_cast(float, (unsigned) a[0]) + __builtin_bit_cast(float, (unsigned) a[1]);
    const unsigned w = __builtin_bit_cast(unsigned, v);
    const auto b = __builtin_amdgcn_permlane16_swap(w, w, false, false);
    v = __builtin_bit_cast(float, (unsigned) b[0]) + __builtin_bit_cast(float, (unsigned) b[1]);
    return row16_sum(v);
}
__device__ __forceinline__ float wave_max(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    v = fmaxf(__builtin_bit_cast(float, (unsigned) a[0]), __builtin_bit_cast(float, (unsigned) a[1]));
    const unsigned w = __builtin_bit_cast(unsigned, v);
    const auto b = __builtin_amdgcn_permlane16_swap(w, w, false, false);
    v = fmaxf(__builtin_bit_cast(float, (unsigned) b[0]), __builtin_bit_cast(float, (unsigned) b[1]));
    return row16_max(v);
}

// max over aligned groups of 32 lanes (a Q8_0 block held one value per lane): the xor-16 step as v_permlane16_swap, the rest inside the rows by DPP.
// A maximum does not depend on the order of the comparisons: the same value as the five-step __shfl_xor butterfly (five ds_bpermute round trips).
__device__ __forceinline__ float lanes32_max(float v) {
    const unsigned w = __builtin_bit_cast(unsigned, v);
    const auto b = __builtin_amdgcn_permlane16_swap(w, w, false, false);
    v = fmaxf(__builtin_bit_cast(float, (unsigned) b[0]), __builtin_bit_cast(float, (unsigned) b[1]));
    return row16_max(v);
}

// ggml_gelu.  mode 1 restates ggml's CPU path, which evaluates GELU through a table indexed by the
// fp16 bits of x and holding fp16 results (upstream ggml_vec_gelu_f32 / GGML_GELU_FP16): a table is
// memoisation, so rounding x to fp16, evaluating in fp32 and rounding the result to fp16 is the
// same function.
__device__ __forceinline__ float gelu_tanh_f32(float x) {
    const float A = 0.044715f, S = 0.79788456080286535587989211986876f;
    return 0.5f * x * (1.0f + tanhf(S * x * (1.0f + A * x * x)));
}
__device__ __forceinline__ float gelu_apply(float x, int mode) {
    if (mode == 0) return gelu_tanh_f32(x);
    if (x <= -10.0f) return 0.0f;
    if (x >= 10.0f) return x;
    const float xr = (float) (_Float16) x;
    return (float) (_Float16) gelu_tanh_f32(xr);
}


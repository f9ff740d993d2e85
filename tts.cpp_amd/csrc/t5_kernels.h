// t5_kernels.h — the pieces of the T5 voice-prompt encoder that are not a GEMM
// (/root/reference/src/models/parler/t5/model.cpp:179-295).  The matmuls go through gemm16_kernel / qgemm16_kernel
// (parler_kernels.h) like the decoder's.
//   t5_embed_kernel     ggml_get_rows(embd, tokens)                                      :229
//   t5_rms_rows_kernel  build_t5_norm: ggml_rms_norm(eps 1e-6) * weight                  :179-185
//   t5_attn_kernel      kq = K q; kq += pos_bias; soft_max_ext(kq, 0-mask, scale 1); V kq  :250-262
//   t5_gated_gelu_kernel  gelu(wi_0 x) * (wi_1 x)                                        :272-274
//   t5_add_bias_kernel  down_proj_bias                                                   :289-291
// One-shot encoder of a few dozen tokens (called per voice change): written for clarity, not for a roofline.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

static __global__ void t5_embed_kernel(const float *table, const uint32_t *ids, int H, float *x) {
    const int t = blockIdx.x;
    const float *row = table + (int64_t) ids[t] * H;
    for (int i = threadIdx.x; i < H; i += blockDim.x) x[(int64_t) t * H + i] = row[i];
}

// one wave per row
static __global__ void t5_rms_rows_kernel(const float *x, int H, const float *w, float *y, int R) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= R) return;
    const float *xr = x + (int64_t) r * H;
    float s = 0.0f;
    for (int i = lane; i < H; i += 64) s += xr[i] * xr[i];
    s = wave_sum(s);
    const float scale = 1.0f / sqrtf(s / (float) H + 1e-6f);
    for (int i = lane; i < H; i += 64) y[(int64_t) r * H + i] = xr[i] * scale * w[i];
}

// One 64-thread workgroup per (head, query): lanes over keys for the scores, lanes over the 64 head dims for the
// output.  qkv [n][3H] (q | k | v), bucket_of_delta[(key - query) + (n_ctx - 1)] (host table with the reference's
// double arithmetic, t5/model.cpp:303-316), rel_bias [n_buckets][n_heads].
static __global__ __launch_bounds__(64) void t5_attn_kernel(const float *qkv, int n, int H, int n_heads, const int *bucket_of_delta, int n_ctx,
                                                     const float *rel_bias, float *out) {
    extern __shared__ float sm[];   // [64] q, then [n] probabilities
    float *qs = sm, *ps = sm + 64;
    const int h = blockIdx.x, qi = blockIdx.y, lane = threadIdx.x;
    const int64_t ld = 3 * (int64_t) H;
    qs[lane] = qkv[(int64_t) qi * ld + h * 64 + lane];
    __syncthreads();
    float mx = -INFINITY;
    for (int ki = lane; ki < n; ki += 64) {
        const float *kr = qkv + (int64_t) ki * ld + H + h * 64;
        float d = 0.0f;
#pragma unroll 8
        for (int e = 0; e < 64; e++) d += qs[e] * kr[e];
        d += rel_bias[bucket_of_delta[ki - qi + n_ctx - 1] * n_heads + h];
        ps[ki] = d;
        mx = fmaxf(mx, d);
    }
    mx = wave_max(mx);
    float sum = 0.0f;
    for (int ki = lane; ki < n; ki += 64) {
        const float p = expf(ps[ki] - mx);
        ps[ki] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    __syncthreads();
    const float inv = 1.0f / sum;
    float acc = 0.0f;
    for (int ki = 0; ki < n; ki++) acc += (ps[ki] * inv) * qkv[(int64_t) ki * ld + 2 * H + h * 64 + lane];
    out[(int64_t) qi * H + h * 64 + lane] = acc;
}

// ug [n][2F] (wi_0 x | wi_1 x) -> g [n][F] = gelu(up) * gate
static __global__ void t5_gated_gelu_kernel(const float *ug, int F, int n, int gelu_mode, float *g) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t) n * F) return;
    const int64_t r = i / F, c = i - r * F;
    g[i] = gelu_apply(ug[r * 2 * F + c], gelu_mode) * ug[r * 2 * F + F + c];
}

static __global__ void t5_add_bias_kernel(float *y, const float *b, int N, int n) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (int64_t) n * N) y[i] += b[i % N];
}

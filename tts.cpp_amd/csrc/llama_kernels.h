// llama_kernels.h — the non-GEMM pieces of the Orpheus decoder (Llama-3 blocks,
// /root/reference/src/models/orpheus/model.cpp:122-125,186-296).  The projections go through gemm16_kernel /
// qgemm16_kernel (parler_kernels.h) — Q4_0 GGUFs (BASELINE config 4) on the integer path.
//   rms_fold_rows_kernel   orpheus_build_layer_norm: ggml_rms_norm(eps 1e-5) * weight :122-125 (+ folds split-K slabs of
//                          the preceding down_proj into the residual stream, ggml_add :283)
//   llama_rope_kv_kernel   ggml_rope_ext(NEOX, frequency factors) on q and k, K/V cache append :186-221,248-251
//   attn_gqa_kernel        mul_mat(k, q) -> soft_max_ext(causal, 1/sqrt(d)) -> mul_mat(kq, v), kv head = q head / rep :228-259
//   silu_mul_kernel        silu(gate x) * (up x) :279
// First version: one wave per row / per (head, row); written for parity, not yet for a roofline.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void rms_fold_rows_kernel(float *x, int H, const float *w, float *y, int R, float eps, const float *parts, int n_parts, int64_t slab_stride) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= R) return;
    float *xr = x + (int64_t) r * H;
    float s = 0.0f;
    for (int i = lane; i < H; i += 64) {
        float v = xr[i];
        if (parts) {
            for (int p = 0; p < n_parts; p++) v += parts[p * slab_stride + (int64_t) r * H + i];
            xr[i] = v;
        }
        s += v * v;
    }
    s = wave_sum(s);
    const float scale = 1.0f / sqrtf(s / (float) H + eps);
    for (int i = lane; i < H; i += 64) y[(int64_t) r * H + i] = xr[i] * scale * w[i];
}

// qkv [R][(NH + 2 NKV) * HD] (q | k | v).  One workgroup per row; thread = (head, pair i).
// theta walks pos, pos*s, (pos*s)*s, ... in fp32 exactly like ggml_rope_cache_init (theta *= theta_scale), theta_scale =
// powf(base, -2/HD) from the host; angle = theta / freq_factor[i]; NEOX pairing (i, i + HD/2).
__global__ __launch_bounds__(256) void llama_rope_kv_kernel(float *qkv, const uint32_t *pos, const float *ff, float theta_scale, int NH, int NKV, int HD,
                                                            float *kcache, float *vcache) {
    const int r = blockIdx.x;
    const int half = HD >> 1;
    const int ld = (NH + 2 * NKV) * HD, kvH = NKV * HD;
    float *row = qkv + (int64_t) r * ld;
    const uint32_t p = pos[r];
    for (int idx = threadIdx.x; idx < (NH + NKV) * half; idx += blockDim.x) {
        const int h = idx / half, i = idx - h * half;
        float theta = (float) p;
        for (int j = 0; j < i; j++) theta *= theta_scale;
        const float ang = theta / (ff ? ff[i] : 1.0f);
        const float cs = cosf(ang), sn = sinf(ang);
        float *v = row + (int64_t) h * HD;      // heads NH.. are the k heads (they follow q in the row)
        const float x0 = v[i], x1 = v[i + half];
        const float y0 = x0 * cs - x1 * sn, y1 = x0 * sn + x1 * cs;
        if (h < NH) { v[i] = y0; v[i + half] = y1; }
        else {
            float *kc = kcache + (int64_t) p * kvH + (h - NH) * HD;
            kc[i] = y0; kc[i + half] = y1;
        }
    }
    const float *vsrc = row + (int64_t) (NH + NKV) * HD;
    for (int i = threadIdx.x; i < kvH; i += blockDim.x) vcache[(int64_t) p * kvH + i] = vsrc[i];
}

// one wave per (q head, row): lanes over keys for the scores, lanes over the head dims for the output
template <int HD>
__global__ __launch_bounds__(64) void attn_gqa_kernel(const float *qkv, int ld, const uint32_t *pos, const float *kcache, const float *vcache, int NH, int NKV,
                                                      float scale, float *out) {
    extern __shared__ float sm[];   // [HD] q, then [T] probabilities
    float *qs = sm, *ps = sm + HD;
    const int h = blockIdx.x, r = blockIdx.y, lane = threadIdx.x;
    const int T = (int) pos[r] + 1;
    const int kvH = NKV * HD, kh = h / (NH / NKV);
    for (int e = lane; e < HD; e += 64) qs[e] = qkv[(int64_t) r * ld + h * HD + e];
    __syncthreads();
    float mx = -INFINITY;
    for (int j = lane; j < T; j += 64) {
        const float *kr = kcache + (int64_t) j * kvH + kh * HD;
        float d = 0.0f;
#pragma unroll 8
        for (int e = 0; e < HD; e++) d += qs[e] * kr[e];
        d *= scale;
        ps[j] = d;
        mx = fmaxf(mx, d);
    }
    mx = wave_max(mx);
    float sum = 0.0f;
    for (int j = lane; j < T; j += 64) {
        const float p = expf(ps[j] - mx);
        ps[j] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    __syncthreads();
    const float inv = 1.0f / sum;
    for (int e = lane; e < HD; e += 64) {
        float acc = 0.0f;
        for (int j = 0; j < T; j++) acc += (ps[j] * inv) * vcache[(int64_t) j * kvH + kh * HD + e];
        out[(int64_t) r * NH * HD + h * HD + e] = acc;
    }
}

// gu [R][2F] (gate | up) -> g [R][F] = silu(gate) * up
__global__ void silu_mul_kernel(const float *gu, int F, int R, float *g) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t) R * F) return;
    const int64_t r = i / F, c = i - r * F;
    const float x = gu[r * 2 * F + c];
    g[i] = (x / (1.0f + expf(-x))) * gu[r * 2 * F + F + c];
}

// dac_kernels.h — gfx950 kernels for the DAC decoder (codes -> PCM).
//
// Replaces the ggml graph of dac_runner::build_dac_graph
// (/root/reference/src/decoder/dac_model.cpp:146-170) and the layer builders in
// /root/reference/src/decoder/general_neural_audio_codec.cpp:133-172.  Kernel ↔ reference:
//   dac_embed_kernel   dac_build_audio_inputs + build_quantize_layer   dac_model.cpp:100-123, gnac.cpp:166-172
//   conv1d_kernel      ggml_conv_1d + ggml_add(bias) [+ snake_1d in front] [+ residual add] [+ tanh]
//                      dac_model.cpp:158-166, gnac.cpp:133-149
//   convt1d_kernel     snake_1d + fork's ggml_conv_transpose_1d + bias               gnac.cpp:151-154
// Activations are [C][L] fp32 with L fastest (ggml ne=[L,C]); weights keep the GGUF/PyTorch memory
// order (Conv1d [Cout][Cin][K], ConvTranspose1d [Cin][Cout][K]).
//
// snake_1d (src/util.cpp:96-101: x + sin(alpha x)^2 / alpha, no epsilon) is never materialised: it
// is applied while the input tile is staged into LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float float4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float snake_f(float x, float alpha, float ralpha) {
    const float s = sinf(x * alpha);
    return x + (s * s) * ralpha;
}

// ------------------------------------------------------------------------------------------------
// quantizer: out[c][t] = sum_i ( b_i[c] + sum_d W_i[c][d] * codebook_i[code[t][i]][d] )
// ------------------------------------------------------------------------------------------------
struct DacEmbedArgs {
    const uint32_t *codes;  // [T][n_cb]
    const float *codebook;  // [n_cb][cb_size][cb_dim]
    const float *proj_w;    // [n_cb][latent][cb_dim]
    const float *proj_b;    // [n_cb][latent]
    int n_cb, cb_size, cb_dim, latent, T;
    float *out;             // [latent][T]
};

__global__ void dac_embed_kernel(DacEmbedArgs a) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (t >= a.T) return;
    float total = 0.0f;
    for (int i = 0; i < a.n_cb; i++) {
        const uint32_t code = a.codes[(int64_t) t * a.n_cb + i];
        const float *cb = a.codebook + ((int64_t) i * a.cb_size + code) * a.cb_dim;
        const float *w = a.proj_w + ((int64_t) i * a.latent + c) * a.cb_dim;
        float acc = 0.0f;
        for (int d = 0; d < a.cb_dim; d++) acc += w[d] * cb[d];
        acc += a.proj_b[i * a.latent + c];
        total = (i == 0) ? acc : (total + acc);
    }
    a.out[(int64_t) c * a.T + t] = total;
}

// ------------------------------------------------------------------------------------------------
// conv1d, stride 1, "same" padding: y[co][t] = b[co] + sum_ci sum_k w[co][ci][k] * f(x[ci][t + k*dil - pad])
//   f = snake (per input channel alpha) or identity;  epilogue: + residual[co][t], tanh.
// Tile: 64 output channels x 64 positions per 256-thread workgroup, 4x4 outputs per thread,
// input channels staged through LDS 8 at a time.
// ------------------------------------------------------------------------------------------------
struct ConvArgs {
    const float *x;      // [cin][L]
    const float *w;      // [cout][cin][K]
    const float *b;      // [cout]
    const float *alpha;  // [cin] snake on the input, or NULL
    const float *resid;  // [cout][L] or NULL
    float *y;            // [cout][L]
    int cin, cout, L, dil, pad, do_tanh;
};

#define CV_CO 64
#define CV_T  64
#define CV_CI 8

template <int KT>
__global__ __launch_bounds__(256) void conv1d_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int halo = (KT - 1) * a.dil;
    const int xw = CV_T + halo;             // staged positions per input channel
    float *xs = (float *) smem;             // [CV_CI][xw]
    float *ws = xs + CV_CI * xw;            // [CV_CI][KT][CV_CO]
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * CV_T, co0 = blockIdx.y * CV_CO;
    const int tt = (tid & 15) * 4, tc = (tid >> 4) * 4;  // thread's 4 positions / 4 channels

    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = 0.0f;

    for (int ci0 = 0; ci0 < a.cin; ci0 += CV_CI) {
        __syncthreads();
        // stage inputs (snake applied here; zero padding outside [0,L))
        for (int i = tid; i < CV_CI * xw; i += 256) {
            const int ci = i / xw, p = i - ci * xw;
            const int t = t0 + p - a.pad;
            const int cig = ci0 + ci;
            float v = 0.0f;
            if (cig < a.cin && t >= 0 && t < a.L) {
                v = a.x[(int64_t) cig * a.L + t];
                if (a.alpha) { const float al = a.alpha[cig]; v = snake_f(v, al, 1.0f / al); }
            }
            xs[i] = v;
        }
        // stage weights as [ci][k][co]
        for (int i = tid; i < CV_CI * KT * CV_CO; i += 256) {
            const int co = i % CV_CO, rest = i / CV_CO;
            const int k = rest % KT, ci = rest / KT;
            const int cog = co0 + co, cig = ci0 + ci;
            float v = 0.0f;
            if (cog < a.cout && cig < a.cin) v = a.w[((int64_t) cog * a.cin + cig) * KT + k];
            ws[i] = v;
        }
        __syncthreads();
#pragma unroll
        for (int ci = 0; ci < CV_CI; ci++) {
#pragma unroll
            for (int k = 0; k < KT; k++) {
                const float4d w4 = *(const float4d *) (ws + (ci * KT + k) * CV_CO + tc);
                const float *xp = xs + ci * xw + tt + k * a.dil;
                const float x0 = xp[0], x1 = xp[1], x2 = xp[2], x3 = xp[3];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    acc[i][0] += w4[i] * x0;
                    acc[i][1] += w4[i] * x1;
                    acc[i][2] += w4[i] * x2;
                    acc[i][3] += w4[i] * x3;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int co = co0 + tc + i;
        if (co >= a.cout) continue;
        const float bias = a.b ? a.b[co] : 0.0f;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int t = t0 + tt + j;
            if (t >= a.L) continue;
            float v = acc[i][j] + bias;
            if (a.resid) v = v + a.resid[(int64_t) co * a.L + t];
            if (a.do_tanh) v = tanhf(v);
            a.y[(int64_t) co * a.L + t] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// ConvTranspose1d(stride s, kernel 2s, padding p):  y[co][to] = b[co] + sum_ci sum_k f(x[ci][ti]) w[ci][co][k],
// to = ti*s + k - p.  Every output touches exactly two taps: k = phi and phi + s with
// phi = (to+p) mod s, ti = (to+p)/s and ti-1.
// Tile: 64 output channels x 64 output positions per workgroup, 4x4 per thread.
// ------------------------------------------------------------------------------------------------
struct ConvTArgs {
    const float *x;      // [cin][L]
    const float *w;      // [cin][cout][2s]
    const float *b;      // [cout]
    const float *alpha;  // [cin] snake on the input
    float *y;            // [cout][Lout]
    int cin, cout, L, Lout, stride, pad;
};

#define CT_CI 8

__global__ __launch_bounds__(256) void convt1d_kernel(ConvTArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int s = a.stride, K = 2 * s;
    const int to0 = blockIdx.x * CV_T, co0 = blockIdx.y * CV_CO;
    const int ti_lo = (to0 + a.pad) / s - 1;            // first input position any output of the tile can touch
    const int xw = (CV_T + s - 1) / s + 2;              // staged input positions
    float *xs = (float *) smem;                         // [CT_CI][xw]
    float *ws = xs + CT_CI * xw;                        // [CT_CI][K][CV_CO]
    const int tid = threadIdx.x;
    const int tt = (tid & 15) * 4, tc = (tid >> 4) * 4;

    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = 0.0f;

    int phi[4], xi[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int q = to0 + tt + j + a.pad;
        phi[j] = q % s;
        xi[j] = q / s - ti_lo;  // index of ti in the staged window; ti-1 is xi-1 >= 0
    }

    for (int ci0 = 0; ci0 < a.cin; ci0 += CT_CI) {
        __syncthreads();
        for (int i = tid; i < CT_CI * xw; i += 256) {
            const int ci = i / xw, p = i - ci * xw;
            const int ti = ti_lo + p, cig = ci0 + ci;
            float v = 0.0f;
            if (cig < a.cin && ti >= 0 && ti < a.L) {
                v = a.x[(int64_t) cig * a.L + ti];
                if (a.alpha) { const float al = a.alpha[cig]; v = snake_f(v, al, 1.0f / al); }
            }
            xs[i] = v;
        }
        for (int i = tid; i < CT_CI * K * CV_CO; i += 256) {
            const int co = i % CV_CO, rest = i / CV_CO;
            const int k = rest % K, ci = rest / K;
            const int cog = co0 + co, cig = ci0 + ci;
            float v = 0.0f;
            if (cog < a.cout && cig < a.cin) v = a.w[((int64_t) cig * a.cout + cog) * K + k];
            ws[i] = v;
        }
        __syncthreads();
#pragma unroll
        for (int ci = 0; ci < CT_CI; ci++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float xa = xs[ci * xw + xi[j]];       // tap k = phi      (ti)
                const float xb = xs[ci * xw + xi[j] - 1];   // tap k = phi + s  (ti - 1)
                const float4d wa = *(const float4d *) (ws + (ci * K + phi[j]) * CV_CO + tc);
                const float4d wb = *(const float4d *) (ws + (ci * K + phi[j] + s) * CV_CO + tc);
#pragma unroll
                for (int i = 0; i < 4; i++) acc[i][j] += xa * wa[i] + xb * wb[i];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int co = co0 + tc + i;
        if (co >= a.cout) continue;
        const float bias = a.b ? a.b[co] : 0.0f;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int to = to0 + tt + j;
            if (to >= a.Lout) continue;
            a.y[(int64_t) co * a.Lout + to] = acc[i][j] + bias;
        }
    }
}

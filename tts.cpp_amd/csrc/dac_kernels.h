// dac_kernels.h — gfx950 kernels for the DAC decoder (codes -> PCM).
//
// Replaces the ggml graph of dac_runner::build_dac_graph
// (/root/reference/src/decoder/dac_model.cpp:146-170) and the layer builders in
// /root/reference/src/decoder/general_neural_audio_codec.cpp:133-172.  Kernel ↔ reference:
//   dac_embed_tile_kernel / dac_embed_kernel   dac_build_audio_inputs + build_quantize_layer   dac_model.cpp:100-123, gnac.cpp:166-172
//   conv1d_mfma_kernel<KT,...>   ggml_conv_1d + ggml_add(bias) [+ snake_1d in front] [+ residual add] [+ tanh], exact fp32 on
//                      v_mfma_f32_32x32x2_f32                                        dac_model.cpp:158-166, gnac.cpp:133-149
//   conv1x1_direct_kernel        the k = 1 conv + bias + residual of a residual unit at 96 / 192 channels        gnac.cpp:147-149
//   convt1d_mfma_kernel<S,...>   snake_1d + fork's ggml_conv_transpose_1d + bias, phase-decomposed             gnac.cpp:151-154
//   conv1d_cout1_kernel          final snake + conv (1 output channel) + tanh                                  dac_model.cpp:163-166
//   conv1d_mfma16_kernel / convt1d_mfma16_kernel   the same layers with F16 tensors (fp16 im2col, v_mfma_f32_32x32x16_f16)
//   conv1d_mfma_b3_kernel        experiment, off by default: k = 7 conv with fp32 operands as three bf16 terms (v_mfma_f32_32x32x16_bf16)
//   conv1d_kernel / convt1d_kernel   the same layers without the matrix pipe (TTS_HIP_FLAG_VALU_GEMM: cross-check in the tests)
//   pack_conv_w_kernel / pack_conv_w16_kernel / pack_conv_w_b3_kernel   one-time re-layout of the weights into the LDS images
//   snac_embed_kernel, dwconv7_kernel, noise_fma_kernel   SNAC's quantizer, depthwise conv and noise block (snac_model.cpp)
// Activations are [C][L] fp32 with L fastest (ggml ne=[L,C]); weights keep the GGUF/PyTorch memory
// order (Conv1d [Cout][Cin][K], ConvTranspose1d [Cin][Cout][K]).
//
// snake_1d (src/util.cpp:96-101: x + sin(alpha x)^2 / alpha, no epsilon) is never materialised: it
// is applied while the input tile is staged into LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float float4d __attribute__((ext_vector_type(4)));

// sin for snake's argument range.  libm's sinf spends ~250 instructions (branches, a Payne-Hanek path for huge arguments) and the
// staging of every conv input tile evaluates it once per element: 7.5 % of the codec's time (profiles/r02/dac_sin_experiment.log).
// |x| < 125: 4-term Cody-Waite reduction by pi (every product exact, so the reduced argument is correct to < 1 ulp) and the degree-9
// odd minimax polynomial on [-pi/2, pi/2].  Checked against float64 sin on 12 M random arguments (tests/test_oracle_cpu.py restates it in
// numpy): max abs error 1.3e-7, <= 2 ulp — libm's own float result is within 1.4 ulp on the same points; larger arguments go to sinf.
__device__ __forceinline__ float snake_sin_poly(float x) {   // |x| < 125
    const float q = rintf(x * 0.318309886183790671537767526745028724f);
    float r = fmaf(q, -3.140625f, x);
    r = fmaf(q, -0.0009670257568359375f, r);
    r = fmaf(q, -6.2771141529083251953e-07f, r);
    r = fmaf(q, -1.2154201256553420762e-10f, r);
    const float s = r * r;
    if (((int) q) & 1) r = -r;
    float u = 2.6083159809786593541503e-06f;
    u = fmaf(u, s, -0.0001981069071916863322258f);
    u = fmaf(u, s, 0.00833307858556509017944336f);
    u = fmaf(u, s, -0.166666597127914428710938f);
    return fmaf(s, u * r, r);
}
__device__ __forceinline__ float snake_sin(float x) {
    if (!(fabsf(x) < 125.0f)) return sinf(x);
    return snake_sin_poly(x);
}

__device__ __forceinline__ float snake_f(float x, float alpha, float ralpha) {
    const float s = snake_sin(x * alpha);
    return x + (s * s) * ralpha;
}
// N values at once.  snake_f's range test is a divergent branch per element: N calls in a row are N separate control-flow regions, i.e. N
// dependent chains of ~20 operations one after the other (measured: ~190 cycles per element, profiles/r03/ru_bench_3.txt — the staging
// and conversion phases of the codec kernels were bound by it).  Here the test is made once for the group and wave-uniformly: the
// common case is N interleavable straight-line chains; if any lane holds an argument beyond the polynomial's range the whole wave takes
// the per-element form.  Same function per element either way (bit-identical to snake_f).
template <int N>
__device__ __forceinline__ void snake_vec(float (&v)[N], const float (&al)[N], const float (&ral)[N]) {
    float arg[N];
    bool big = false;
#pragma unroll
    for (int e = 0; e < N; e++) { arg[e] = v[e] * al[e]; big = big || !(fabsf(arg[e]) < 125.0f); }
    if (__builtin_amdgcn_ballot_w64(big) != 0) {
#pragma unroll
        for (int e = 0; e < N; e++) { const float s = snake_sin(arg[e]); v[e] = v[e] + (s * s) * ral[e]; }
    } else {
#pragma unroll
        for (int e = 0; e < N; e++) { const float s = snake_sin_poly(arg[e]); v[e] = v[e] + (s * s) * ral[e]; }
    }
}
// snake with alpha from the LDS table (alpha, then 1/alpha, cin_pad entries each) or straight from memory; same arithmetic
__device__ __forceinline__ float snake_ch(float x, int cig, int cin, int cin_pad, const float *als, const float *alpha, int tab) {
    if (tab) return snake_f(x, als[cig], als[cin_pad + cig]);
    const float al = cig < cin ? alpha[cig] : 1.0f;
    return snake_f(x, al, 1.0f / al);
}

// ------------------------------------------------------------------------------------------------
// quantizer: out[c][t] = sum_i ( b_i[c] + sum_d W_i[c][d] * codebook_i[code[t][i]][d] )
// ------------------------------------------------------------------------------------------------
struct DacEmbedArgs {
    const uint32_t *frames; // [n] valid frames per utterance (grid.z), or NULL
    const uint32_t *codes;  // [n][T][n_cb]
    const float *codebook;  // [n_cb][cb_size][cb_dim]
    const float *proj_w;    // [n_cb][latent][cb_dim]
    const float *proj_b;    // [n_cb][latent]
    int n_cb, cb_size, cb_dim, latent, T;
    int Tout;               // row stride of out (>= T)
    float *out;             // [latent][Tout]
    int x_f16;              // F16 conv kernels: the conv input (codebook row) goes through an fp16 im2col
};

static __global__ void dac_embed_kernel(DacEmbedArgs a) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y, z = blockIdx.z;
    const int Tz = a.frames ? (int) a.frames[z] : a.T;
    if (t >= Tz) return;
    float total = 0.0f;
    for (int i = 0; i < a.n_cb; i++) {
        const uint32_t code = a.codes[((int64_t) z * a.T + t) * a.n_cb + i];
        const float *cb = a.codebook + ((int64_t) i * a.cb_size + code) * a.cb_dim;
        const float *w = a.proj_w + ((int64_t) i * a.latent + c) * a.cb_dim;
        float acc = 0.0f;
        for (int d = 0; d < a.cb_dim; d++) acc += w[d] * (a.x_f16 ? (float) (_Float16) cb[d] : cb[d]);
        acc += a.proj_b[i * a.latent + c];
        total = (i == 0) ? acc : (total + acc);
    }
    a.out[((int64_t) z * a.latent + c) * a.Tout + t] = total;
}

// The same sum with the work shaped for the chip: one lane per frame holds the NCB gathered codebook rows (NCB x D
// registers) and each wave walks EMB_CH output channels, whose projection rows are wave-uniform (scalar loads). The
// per-thread form above gathers the rows again for every one of the 1024 channels and runs 262 144 single-wave
// workgroups of two dependent loads each (2.9 ms per 64-utterance pass against ~65 MB written).
// Same operation order per output: D products summed in order, + bias, codebooks summed in order.
#define EMB_CH 32
template <int NCB, int D>
__global__ __launch_bounds__(256) void dac_embed_tile_kernel(DacEmbedArgs a) {
    const int t = blockIdx.x * 64 + (threadIdx.x & 63);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), z = blockIdx.z;
    const int Tz = a.frames ? (int) a.frames[z] : a.T;
    if (t >= Tz) return;
    float v[NCB][D];
#pragma unroll
    for (int i = 0; i < NCB; i++) {
        const uint32_t code = a.codes[((int64_t) z * a.T + t) * NCB + i];
        const float4 *cb = (const float4 *) (a.codebook + ((int64_t) i * a.cb_size + code) * D);
#pragma unroll
        for (int q = 0; q < D / 4; q++) {
            const float4 r = cb[q];
            v[i][q * 4] = r.x; v[i][q * 4 + 1] = r.y; v[i][q * 4 + 2] = r.z; v[i][q * 4 + 3] = r.w;
        }
        if (a.x_f16) {
#pragma unroll
            for (int d = 0; d < D; d++) v[i][d] = (float) (_Float16) v[i][d];
        }
    }
    const int c0 = (blockIdx.y * 4 + wv) * EMB_CH;
#pragma unroll 2
    for (int cc = 0; cc < EMB_CH; cc++) {
        const int c = c0 + cc;
        if (c >= a.latent) break;
        float total = 0.0f;
#pragma unroll
        for (int i = 0; i < NCB; i++) {
            const float *w = a.proj_w + ((int64_t) i * a.latent + c) * D;
            float acc = 0.0f;
#pragma unroll
            for (int d = 0; d < D; d++) acc += w[d] * v[i][d];
            acc += a.proj_b[i * a.latent + c];
            total = (i == 0) ? acc : (total + acc);
        }
        a.out[((int64_t) z * a.latent + c) * a.Tout + t] = total;
    }
}

// ------------------------------------------------------------------------------------------------
// conv1d, stride 1, "same" padding: y[co][t] = b[co] + sum_ci sum_k w[co][ci][k] * f(x[ci][t + k*dil - pad])
//   f = snake (per input channel alpha) or identity;  epilogue: + residual[co][t], tanh.
// Tile: 64 output channels x 64 positions per 256-thread workgroup, 4x4 outputs per thread,
// input channels staged through LDS 8 at a time.
// ------------------------------------------------------------------------------------------------
struct ConvArgs {
    const float *x;      // [n][cin][L]
    const float *w;      // [cout][cin][K]
    const float *b;      // [cout]
    const float *alpha;  // [cin] snake on the input, or NULL
    const float *resid;  // [n][cout][L] or NULL
    float *y;            // [n][cout][L]
    int cin, cout, L, dil, pad, do_tanh;   // L = row stride (longest utterance of the batch at this stage)
    const uint32_t *frames;  // [n] valid frames per utterance (grid.z) or NULL; valid length = frames[z] * mult
    int mult;
    const float *alpha_out;  // [cout] snake applied to the OUTPUT (the next layer's snake_1d, fused here), or NULL
    int x_f16;               // F16 conv kernel: inputs are rounded to fp16 on the way in (ggml's fp16 im2col)
    int alpha_tab;           // MFMA kernels: 1 = alpha and 1/alpha of every input channel staged in LDS, 0 = read from memory
    int prio;                // conv1d_mfma_kernel: 1 = waves raise their issue priority for the MFMA phase of a chunk, 2 = for the staging phase
};

__device__ __forceinline__ int valid_len(const uint32_t *frames, int mult, int L) {
    return frames ? (int) frames[blockIdx.z] * mult : L;
}

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
    const int LS = a.L, L = valid_len(a.frames, a.mult, a.L);
    if (t0 >= L) return;
    const float *xg = a.x + (int64_t) blockIdx.z * a.cin * LS;
    float *yg = a.y + (int64_t) blockIdx.z * a.cout * LS;
    const float *rg = a.resid ? a.resid + (int64_t) blockIdx.z * a.cout * LS : nullptr;

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
            if (cig < a.cin && t >= 0 && t < L) {
                v = xg[(int64_t) cig * LS + t];
                if (a.alpha) { const float al = a.alpha[cig]; v = snake_f(v, al, 1.0f / al); }
                if (a.x_f16) v = (float) (_Float16) v;
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
            if (t >= L) continue;
            float v = acc[i][j] + bias;
            if (rg) v = v + rg[(int64_t) co * LS + t];
            if (a.alpha_out) { const float al = a.alpha_out[co]; v = snake_f(v, al, 1.0f / al); }
            if (a.do_tanh) v = tanhf(v);
            yg[(int64_t) co * LS + t] = v;
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
    const float *x;      // [n][cin][L]
    const float *w;      // [cin][cout][2s]
    const float *b;      // [cout]
    const float *alpha;  // [cin] snake on the input
    float *y;            // [n][cout][Lout]
    int cin, cout, L, Lout, stride, pad;   // L / Lout = row strides
    const uint32_t *frames;  // per-utterance frames (grid.z) or NULL; valid input length = frames[z] * mult
    int mult;
    int x_f16;               // F16 kernel: inputs rounded to fp16 (ggml_compute_forward_conv_transpose_1d_f16_f32)
    int alpha_tab;           // as ConvArgs
    int npos, nco, nz;       // convt_b3_kernel: tiles along ti / output channels, utterances (xcd_tile in dac_b3_kernels.h)
    const __bf16 *xp;        // convt_b3_kernel<.., true>: the input as split planes [n][3][cin/8][L][8] — already snaked with `alpha` and split by its
                             // producer (conv_b3p_kernel's epilogue), staged as a straight copy; x / alpha are not read then
};

#define CT_CI 8

static __global__ __launch_bounds__(256) void convt1d_kernel(ConvTArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int s = a.stride, K = 2 * s;
    const int to0 = blockIdx.x * CV_T, co0 = blockIdx.y * CV_CO;
    const int ti_lo = (to0 + a.pad) / s - 1;            // first input position any output of the tile can touch
    const int xw = (CV_T + s - 1) / s + 2;              // staged input positions
    float *xs = (float *) smem;                         // [CT_CI][xw]
    float *ws = xs + CT_CI * xw;                        // [CT_CI][K][CV_CO]
    const int tid = threadIdx.x;
    const int tt = (tid & 15) * 4, tc = (tid >> 4) * 4;
    const int LS = a.L, L = valid_len(a.frames, a.mult, a.L);
    const int LoS = a.Lout, Lout = a.frames ? (L - 1) * s - 2 * a.pad + K : a.Lout;
    if (to0 >= Lout) return;
    const float *xg = a.x + (int64_t) blockIdx.z * a.cin * LS;
    float *yg = a.y + (int64_t) blockIdx.z * a.cout * LoS;

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
            if (cig < a.cin && ti >= 0 && ti < L) {
                v = xg[(int64_t) cig * LS + ti];
                if (a.alpha) { const float al = a.alpha[cig]; v = snake_f(v, al, 1.0f / al); }
                if (a.x_f16) v = (float) (_Float16) v;
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
            if (to >= Lout) continue;
            yg[(int64_t) co * LoS + to] = acc[i][j] + bias;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// SNAC (src/decoder/snac_model.cpp): the pieces the DAC kernels do not cover.
//   snac_embed_kernel   snac_build_audio_inputs :86-108 — codebook levels at 1/4, 1/2, 1x the latent rate,
//                       quantize layer per level (gnac.cpp:166-172), repeat_interleave, sum
//   dwconv7_kernel      snake_1d + ggml_conv_1d_dw (groups = channels) + bias [+ the next snake]  :141-142, gnac.cpp:135-145
//   noise_fma_kernel    x + noise * conv1x1(x)   gnac.cpp:155-159 (the 1x1 conv is a conv1d launch)
// ------------------------------------------------------------------------------------------------
struct SnacEmbedArgs {
    const uint32_t *codes;   // level-major: T/rep[0] ids, then T/rep[1], ... (snac_runner::set_inputs :161-178)
    const float *codebook;   // [n_cb][cb_size][cb_dim]
    const float *proj_w;     // [n_cb][latent][cb_dim]
    const float *proj_b;     // [n_cb][latent]
    int n_cb, cb_size, cb_dim, latent, T;
    int rep[4];
    float *out;              // [latent][T]
};

static __global__ void snac_embed_kernel(SnacEmbedArgs a) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (t >= a.T) return;
    float total = 0.0f;
    int off = 0;
    for (int i = 0; i < a.n_cb; i++) {
        const uint32_t code = a.codes[off + t / a.rep[i]];
        off += a.T / a.rep[i];
        const float *cb = a.codebook + ((int64_t) i * a.cb_size + code) * a.cb_dim;
        const float *w = a.proj_w + ((int64_t) i * a.latent + c) * a.cb_dim;
        float acc = 0.0f;
        for (int d = 0; d < a.cb_dim; d++) acc += w[d] * cb[d];
        acc += a.proj_b[i * a.latent + c];
        total = (i == 0) ? acc : (total + acc);
    }
    a.out[(int64_t) c * a.T + t] = total;
}

// depthwise k = 7: one output per thread (memory-shaped: 1 read + 1 write per element, 7 taps from L1/L2)
static __global__ __launch_bounds__(256) void dwconv7_kernel(const float *x, const float *w, const float *b, const float *alpha_in,
                                                      const float *alpha_out, float *y, int C, int L, int pad, int dil) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (t >= L) return;
    const float *xr = x + (int64_t) c * L;
    const float al = alpha_in ? alpha_in[c] : 1.0f, ral = 1.0f / al;
    float acc = b ? b[c] : 0.0f;
#pragma unroll
    for (int k = 0; k < 7; k++) {
        const int ti = t + k * dil - pad;
        if (ti >= 0 && ti < L) {
            float v = xr[ti];
            if (alpha_in) v = snake_f(v, al, ral);
            acc += w[c * 7 + k] * v;
        }
    }
    if (alpha_out) { const float ao = alpha_out[c]; acc = snake_f(acc, ao, 1.0f / ao); }
    y[(int64_t) c * L + t] = acc;
}

static __global__ void noise_fma_kernel(float *x, const float *h, const float *noise, int C, int L) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t) C * L) return;
    x[i] = x[i] + h[i] * noise[i % L];
}

// ================================================================================================
// MFMA paths.  v_mfma_f32_32x32x2_f32 is exact fp32 (bitwise an fmaf chain in k order) at the fp32
// vector rate, so the codec keeps the reference's fp32 numerics while the contraction runs on the
// matrix pipe.  Fragment maps (cdna_hip_programming.md §3): A[i = lane&31][k = lane>>5],
// B[k = lane>>5][j = lane&31], D col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
// ================================================================================================
typedef float float16d __attribute__((ext_vector_type(16)));

// Weight pre-packing (once, after the weights are resident): each (channel tile, input-channel chunk)
// of a conv weight is rewritten as the exact LDS image the MFMA kernels consume, zero padded, so
// staging is a straight contiguous float4 copy.
//   conv1d   src [cout][cin][KT]  -> dst [co_tile][chunk][(k*CI_T + ci)][CO_T]
//   convT1d  src [cin][cout][K2]  -> dst [co_tile][chunk][ci][k][CO_T]
static __global__ void pack_conv_w_kernel(const float *src, float *dst, int cout, int cin, int KT, int CO_T, int CI_T, int n_chunks,
                                   int transposed_src) {
    const int64_t total = (int64_t) ((cout + CO_T - 1) / CO_T) * n_chunks * KT * CI_T * CO_T;
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t) gridDim.x * blockDim.x) {
        const int col = (int) (i % CO_T);
        int64_t r = i / CO_T;
        int k, cil;
        if (!transposed_src) { cil = (int) (r % CI_T); r /= CI_T; k = (int) (r % KT); r /= KT; }
        else                 { k = (int) (r % KT); r /= KT; cil = (int) (r % CI_T); r /= CI_T; }
        const int ch = (int) (r % n_chunks);
        const int ct = (int) (r / n_chunks);
        const int co = ct * CO_T + col, ci = ch * CI_T + cil;
        float v = 0.0f;
        if (co < cout && ci < cin)
            v = transposed_src ? src[((int64_t) ci * cout + co) * KT + k] : src[((int64_t) co * cin + ci) * KT + k];
        dst[i] = v;
    }
}

// conv1d as an implicit GEMM: M = cout, N = positions, K = (tap, ci).  Workgroup tile
// (32*MI*WM) channels x (32*NI*WN) positions.  Input channels go through LDS CI_T at a time with
// snake applied on the way in; a.w is the PACKED weight (see pack_conv_w_kernel).  Two LDS buffers:
// the global loads of chunk c+1 are issued before the MFMA loop of chunk c and land in registers
// while the matrix pipe works; one barrier per chunk.
#ifndef CONV7_MIN_WAVES
#define CONV7_MIN_WAVES 1   // 4 (= 128 registers per wave, accumulators out of the AGPRs, 11-21 spills) measured 12 % slower: profiles/r02/dac_conv7_occupancy_ab.log
#endif
template <int KT, int MI, int NI, int WM, int WN, int CI_T>
__global__ __launch_bounds__(64 * WM * WN, (KT == 7 && MI * NI <= 4) ? CONV7_MIN_WAVES : 1) void conv1d_mfma_kernel(ConvArgs a) {
    constexpr int CO_T = 32 * MI * WM, T_T = 32 * NI * WN, NT = 64 * WM * WN;
    constexpr int WCH = KT * CI_T * CO_T;                    // floats per packed weight chunk
    constexpr int WV = (WCH / 4 + NT - 1) / NT;              // float4 per thread per chunk
    constexpr int XMAX = CI_T * (T_T + (KT - 1) * 9);        // dilation <= 9 (3^2)
    constexpr int XV = (XMAX + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int halo = (KT - 1) * a.dil;
    const int xw = T_T + halo;
    const int xsz = (CI_T * xw + 3) & ~3;
    float *wsb = (float *) smem;                // [2][WCH]
    float *xsb = wsb + 2 * WCH;                 // [2][xsz]
    float *als = xsb + 2 * xsz;                 // [cin_pad] alpha, then [cin_pad] 1/alpha
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv / WN, wn = wv % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    const int t0 = blockIdx.x * T_T, co0 = blockIdx.y * CO_T;
    const int n_chunks = (a.cin + CI_T - 1) / CI_T;
    const int cin_pad = n_chunks * CI_T;
    const int LS = a.L, L = valid_len(a.frames, a.mult, a.L);
    if (t0 >= L) return;  // whole workgroup past the end of this (shorter) utterance
    const float *xg = a.x + (int64_t) blockIdx.z * a.cin * LS;
    float *yg = a.y + (int64_t) blockIdx.z * a.cout * LS;
    const float *rg = a.resid ? a.resid + (int64_t) blockIdx.z * a.cout * LS : nullptr;
    const float4d *wg = (const float4d *) (a.w + (int64_t) blockIdx.y * n_chunks * WCH);
    if (a.prio >= 3) {   // experiment: workgroups sharing a SIMD at different static priorities, so that their MFMA phases do not line up
        const unsigned bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned pr = (a.prio == 3 ? bid / 256u : bid) % 3u;
        if (pr == 1) __builtin_amdgcn_s_setprio(1);
        else if (pr == 2) __builtin_amdgcn_s_setprio(2);
    }

    if (a.alpha && a.alpha_tab) {
        for (int i = tid; i < cin_pad; i += NT) {
            const float al = i < a.cin ? a.alpha[i] : 1.0f;
            als[i] = al;
            als[cin_pad + i] = 1.0f / al;
        }
    }

    float16d acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NI; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

    float4d wreg[WV];
    float xreg[XV];
    auto prefetch = [&](int c) {
        const float4d *wp = wg + (int64_t) c * (WCH / 4);
#pragma unroll
        for (int j = 0; j < WV; j++) {
            const int i = tid + j * NT;
            if (i < WCH / 4) wreg[j] = wp[i];
        }
        if constexpr (KT == 1) {
            // k = 1: no halo, tiles and rows are 16-byte aligned (T_T, L and the row stride are multiples of 4):
            // stage the input with 16-byte loads
#pragma unroll
            for (int j = 0; j < XV / 4; j++) {
                const int i4 = tid + j * NT;  // float4 index in the [CI_T][T_T] tile
                const int ci = (i4 * 4) / T_T, p = (i4 * 4) - ci * T_T;
                const int t = t0 + p, cig = c * CI_T + ci;
                float4d v = {0.f, 0.f, 0.f, 0.f};
                if (cig < a.cin && t < L) v = *(const float4d *) (xg + (int64_t) cig * LS + t);
                xreg[4 * j] = v[0]; xreg[4 * j + 1] = v[1]; xreg[4 * j + 2] = v[2]; xreg[4 * j + 3] = v[3];
            }
        } else {
#pragma unroll
            for (int j = 0; j < XV; j++) {
                const int i = tid + j * NT;
                float v = 0.0f;
                if (i < CI_T * xw) {
                    const int ci = i / xw, p = i - ci * xw;
                    const int t = t0 + p - a.pad, cig = c * CI_T + ci;
                    if (cig < a.cin && t >= 0 && t < L) v = xg[(int64_t) cig * LS + t];
                }
                xreg[j] = v;
            }
        }
    };
    auto commit = [&](int c, int buf) {
        float4d *wd = (float4d *) (wsb + buf * WCH);
#pragma unroll
        for (int j = 0; j < WV; j++) {
            const int i = tid + j * NT;
            if (i < WCH / 4) wd[i] = wreg[j];
        }
        float *xd = xsb + buf * xsz;
        // snake on all of a thread's staged values at once (snake_vec: one wave-uniform range test instead of a divergent branch per value)
        if (a.alpha) {
            float al[XV], ral[XV];
#pragma unroll
            for (int j = 0; j < XV; j++) {
                int cig;
                if constexpr (KT == 1) cig = c * CI_T + ((tid + (j / 4) * NT) * 4) / T_T;
                else { const int i = tid + j * NT; cig = c * CI_T + (i < CI_T * xw ? i / xw : 0); }
                if (a.alpha_tab) { al[j] = als[cig]; ral[j] = als[cin_pad + cig]; }
                else { al[j] = cig < a.cin ? a.alpha[cig] : 1.0f; ral[j] = 1.0f / al[j]; }
            }
            snake_vec<XV>(xreg, al, ral);   // snake(0) == 0: zero padding is preserved
        }
        if constexpr (KT == 1) {
#pragma unroll
            for (int j = 0; j < XV / 4; j++) {
                const int i4 = tid + j * NT;
                float4d v = {xreg[4 * j], xreg[4 * j + 1], xreg[4 * j + 2], xreg[4 * j + 3]};
                *(float4d *) (xd + i4 * 4) = v;
            }
        } else {
#pragma unroll
            for (int j = 0; j < XV; j++) {
                const int i = tid + j * NT;
                if (i < CI_T * xw) xd[i] = xreg[j];
            }
        }
    };

    prefetch(0);
    __syncthreads();  // alpha table visible
    commit(0, 0);
    __syncthreads();
    for (int c = 0; c < n_chunks; c++) {
        const int buf = c & 1;
        if (c + 1 < n_chunks) prefetch(c + 1);
        const float *ws = wsb + buf * WCH;
        const float *xs = xsb + buf * xsz;
        if (a.prio == 1) __builtin_amdgcn_s_setprio(2);
        else if (a.prio == 2) __builtin_amdgcn_s_setprio(0);
#pragma unroll 4
        for (int kk = 0; kk < KT * CI_T; kk += 2) {
            const int kq = kk + hi;              // this half-wave's k index
            const int tap = kq / CI_T, ci = kq - tap * CI_T;
            float af[MI], bf[NI];
#pragma unroll
            for (int i = 0; i < MI; i++) af[i] = ws[kq * CO_T + (wm * MI + i) * 32 + l31];
#pragma unroll
            for (int j = 0; j < NI; j++) bf[j] = xs[ci * xw + (wn * NI + j) * 32 + l31 + tap * a.dil];
#pragma unroll
            for (int i = 0; i < MI; i++)
#pragma unroll
                for (int j = 0; j < NI; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (a.prio == 1) __builtin_amdgcn_s_setprio(0);
        else if (a.prio == 2) __builtin_amdgcn_s_setprio(2);
        if (c + 1 < n_chunks) commit(c + 1, buf ^ 1);
        __syncthreads();
    }
    if (a.prio == 1 || a.prio == 2) __builtin_amdgcn_s_setprio(0);
    // Epilogue in phases per (32-channel block, channel quad): the phase's loads first (bias, the consumer's alpha, the residual values — clamped
    // indices, the run-time switches tested once per phase, not per value), then its arithmetic, then its predicated stores; a scheduling barrier
    // keeps the compiler from pulling every phase's loads to the front (that cost 100+ registers and two of the three resident waves).  The
    // per-value form (`a.b ? a.b[co] : 0`, `if (rg)`, `if (t >= L) continue` around single loads) compiled to load -> s_waitcnt -> store for each
    // of the 16 MI (x NI) values: 100-160 dependent round trips at the end of every launch (profiles/tools/isa_serial_loads.py) — most of what
    // Kokoro's short convolutions were.  4 MI round trips now; same arithmetic per value.
#pragma unroll
    for (int i = 0; i < MI; i++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float bias[4], alo[4], res[NI][4];
            int cos[4];
#pragma unroll
            for (int m = 0; m < 4; m++) cos[m] = min(co0 + (wm * MI + i) * 32 + m + 8 * q + 4 * hi, a.cout - 1);
            if (a.b) {
#pragma unroll
                for (int m = 0; m < 4; m++) bias[m] = a.b[cos[m]];
            } else {
#pragma unroll
                for (int m = 0; m < 4; m++) bias[m] = 0.0f;
            }
            if (a.alpha_out) {
#pragma unroll
                for (int m = 0; m < 4; m++) alo[m] = a.alpha_out[cos[m]];
            } else {
#pragma unroll
                for (int m = 0; m < 4; m++) alo[m] = 1.0f;
            }
            if (rg) {
#pragma unroll
                for (int j = 0; j < NI; j++) {
                    const int t = min(t0 + (wn * NI + j) * 32 + l31, L - 1);
#pragma unroll
                    for (int m = 0; m < 4; m++) res[j][m] = rg[(int64_t) cos[m] * LS + t];
                }
            }
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const int e = 4 * q + m;
                const int co = co0 + (wm * MI + i) * 32 + m + 8 * q + 4 * hi;
                const float ral_o = 1.0f / alo[m];
#pragma unroll
                for (int j = 0; j < NI; j++) {
                    const int t = t0 + (wn * NI + j) * 32 + l31;
                    float v = acc[i][j][e] + bias[m];
                    if (rg) v = v + res[j][m];
                    if (a.alpha_out) v = snake_f(v, alo[m], ral_o);
                    if (a.do_tanh) v = tanhf(v);
                    if (co < a.cout && t < L) yg[(int64_t) co * LS + t] = v;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// k = 1 conv + bias + residual of a residual unit (gnac.cpp:147-149) for CIN <= 192 input channels, shaped for HBM instead of for the
// matrix pipe: at 96 / 192 channels the op moves 12 B per 2 * C flops, and conv1d_mfma_kernel<1,...> — 16-channel chunks through LDS, one
// barrier per chunk, one chunk of loads in flight — ran it at 1.7 TB/s with exactly the algorithmic traffic (PMC: profiles/r02/pmc_*).
// Here nothing but the weights goes through LDS:
//   * the MFMA B operand of v_mfma_f32_32x32x2_f32 is one float per lane, B[k = lane >> 5][j = lane & 31] = x[ci = 2 s + (lane >> 5)]
//     [t = t_wave + lane & 31]: each lane loads its own operands straight from global memory (a wave instruction = two 128-byte row
//     segments), ALL of them issued before the first MFMA — up to 63 loads in flight per wave, ~100 KB per CU, no staging, no barrier;
//   * the weight tile [CIN][CO_T] (the packed image of pack_conv_w_kernel; CO_T = all output channels, so x is read once) goes
//     through LDS KH input channels (36 KB) at a time;
//   * accumulation order over the input channels is the old kernel's (ascending, two per MFMA), so the results are bit-identical.
// Workgroup = 4 waves side by side in t: (32 * MI) output channels x (4 * 32 * NI) positions: 96 x 256 or 192 x 128.
template <int MI, int NI, int CIN, int KH>
__global__ __launch_bounds__(256) void conv1x1_direct_kernel(ConvArgs a) {
    constexpr int CO_T = 32 * MI, T_W = 32 * NI, T_T = 4 * T_W, NS = CIN / 2;
    constexpr int NH = CIN / KH;                                // the weight tile goes through LDS KH input channels (36 KB) at a time
    constexpr int W4 = KH * CO_T / 4, WV = (W4 + 255) / 256;
    static_assert(CIN % KH == 0 && KH % 2 == 0, "input channels in LDS phases of KH");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *ws = (float *) smem;   // [KH][CO_T]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int t0 = blockIdx.x * T_T, co0 = blockIdx.y * CO_T;
    const int LS = a.L, L = valid_len(a.frames, a.mult, a.L);
    if (t0 >= L) return;
    const float *xg = a.x + (int64_t) blockIdx.z * a.cin * LS;
    float *yg = a.y + (int64_t) blockIdx.z * a.cout * LS;
    const float *rg = a.resid ? a.resid + (int64_t) blockIdx.z * a.cout * LS : nullptr;
    const float4d *wg = (const float4d *) (a.w + (int64_t) blockIdx.y * CIN * CO_T);

    float4d wreg[WV];
#pragma unroll
    for (int j = 0; j < WV; j++) {
        const int i = tid + j * 256;
        if (i < W4) wreg[j] = wg[i];
    }
    float xb[NI][NS];
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const float *xr = xg + (int64_t) (2 * s + hi) * LS;
#pragma unroll
        for (int j = 0; j < NI; j++) {
            const int t = t0 + wv * T_W + j * 32 + l31;
            xb[j][s] = t < L ? xr[t] : 0.0f;
        }
    }

    float16d acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NI; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;
#pragma unroll
    for (int h = 0; h < NH; h++) {
        if (h > 0) __syncthreads();   // every wave is done with the previous phase of the weights
#pragma unroll
        for (int j = 0; j < WV; j++) {
            const int i = tid + j * 256;
            if (i < W4) ((float4d *) ws)[i] = wreg[j];
        }
        if (h + 1 < NH) {             // the next phase's weights travel while this phase's MFMAs run
#pragma unroll
            for (int j = 0; j < WV; j++) {
                const int i = tid + j * 256;
                if (i < W4) wreg[j] = wg[(h + 1) * W4 + i];
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < KH / 2; s++) {
            float af[MI];
#pragma unroll
            for (int i = 0; i < MI; i++) af[i] = ws[(2 * s + hi) * CO_T + i * 32 + l31];
#pragma unroll
            for (int i = 0; i < MI; i++)
#pragma unroll
                for (int j = 0; j < NI; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], xb[j][h * (KH / 2) + s], acc[i][j], 0, 0, 0);
        }
    }

    // epilogue per 32-channel block: its residual values are requested together, then stored (requesting the whole tile at once costs the
    // second wave per SIMD: 200 + 96 registers)
#pragma unroll
    for (int i = 0; i < MI; i++) {
        float rv[16][NI];
        if (rg) {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int co = co0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
#pragma unroll
                for (int j = 0; j < NI; j++) {
                    const int t = t0 + wv * T_W + j * 32 + l31;
                    rv[e][j] = (co < a.cout && t < L) ? rg[(int64_t) co * LS + t] : 0.0f;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int co = co0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
            if (co >= a.cout) continue;
            const float bias = a.b ? a.b[co] : 0.0f;
#pragma unroll
            for (int j = 0; j < NI; j++) {
                const int t = t0 + wv * T_W + j * 32 + l31;
                if (t >= L) continue;
                float v = acc[i][j][e] + bias;
                if (rg) v = v + rv[e][j];
                yg[(int64_t) co * LS + t] = v;
            }
        }
    }
}

// ConvTranspose1d(stride S, kernel 2S) on the matrix pipe: for output phase phi = (to+p) mod S,
//   y[co][ti*S + phi - p] = b[co] + sum_ci ( f(x[ci][ti]) w[ci][co][phi] + f(x[ci][ti-1]) w[ci][co][phi+S] )
// i.e. S small GEMMs (M = cout, N = ti, K = 2*cin) that share one B operand: per input channel one
// MFMA k-step whose two k slots are the taps (ti, ti-1).  A wave owns 32*MI channels x 32 ti x S phases.
// a.w is the PACKED weight; same two-buffer / register-prefetch structure as conv1d_mfma_kernel.
#ifndef CONVT_MIN_WAVES
#define CONVT_MIN_WAVES 2   // the stride-8 and stride-4 instantiations came out at 264 / 265 registers: 8 over what lets a second wave onto the SIMD
#endif
template <int S, int MI, int WM, int WN, int CI_T>
__global__ __launch_bounds__(64 * WM * WN, CONVT_MIN_WAVES) void convt1d_mfma_kernel(ConvTArgs a) {
    constexpr int CO_T = 32 * MI * WM, TI_T = 32 * WN, NT = 64 * WM * WN, K2 = 2 * S;
    constexpr int WCH = CI_T * K2 * CO_T;
    constexpr int WV = (WCH / 4 + NT - 1) / NT;
    constexpr int xw = TI_T + 1;                   // positions ti0-1 .. ti0+TI_T-1
    constexpr int xsz = (CI_T * xw + 3) & ~3;
    constexpr int XV = (CI_T * xw + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *wsb = (float *) smem;                   // [2][WCH]
    float *xsb = wsb + 2 * WCH;                    // [2][xsz]
    float *als = xsb + 2 * xsz;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv / WN, wn = wv % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    const int ti0 = blockIdx.x * TI_T, co0 = blockIdx.y * CO_T;
    const int n_chunks = (a.cin + CI_T - 1) / CI_T;
    const int cin_pad = n_chunks * CI_T;
    const int LS = a.L, L = valid_len(a.frames, a.mult, a.L);
    const int LoS = a.Lout, Lout = a.frames ? (L - 1) * S - 2 * a.pad + K2 : a.Lout;
    if (ti0 > L) return;  // ti runs 0..L inclusive
    const float *xg = a.x + (int64_t) blockIdx.z * a.cin * LS;
    float *yg = a.y + (int64_t) blockIdx.z * a.cout * LoS;
    const float4d *wg = (const float4d *) (a.w + (int64_t) blockIdx.y * n_chunks * WCH);

    if (a.alpha && a.alpha_tab) {
        for (int i = tid; i < cin_pad; i += NT) {
            const float al = i < a.cin ? a.alpha[i] : 1.0f;
            als[i] = al;
            als[cin_pad + i] = 1.0f / al;
        }
    }

    float16d acc[MI][S];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int ph = 0; ph < S; ph++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][ph][e] = 0.0f;

    float4d wreg[WV];
    float xreg[XV];
    auto prefetch = [&](int c) {
        const float4d *wp = wg + (int64_t) c * (WCH / 4);
#pragma unroll
        for (int j = 0; j < WV; j++) {
            const int i = tid + j * NT;
            if (i < WCH / 4) wreg[j] = wp[i];
        }
#pragma unroll
        for (int j = 0; j < XV; j++) {
            const int i = tid + j * NT;
            float v = 0.0f;
            if (i < CI_T * xw) {
                const int ci = i / xw, p = i - ci * xw;
                const int ti = ti0 - 1 + p, cig = c * CI_T + ci;
                if (cig < a.cin && ti >= 0 && ti < L) v = xg[(int64_t) cig * LS + ti];
            }
            xreg[j] = v;
        }
    };
    auto commit = [&](int c, int buf) {
        float4d *wd = (float4d *) (wsb + buf * WCH);
#pragma unroll
        for (int j = 0; j < WV; j++) {
            const int i = tid + j * NT;
            if (i < WCH / 4) wd[i] = wreg[j];
        }
        float *xd = xsb + buf * xsz;
        if (a.alpha) {
            float al[XV], ral[XV];
#pragma unroll
            for (int j = 0; j < XV; j++) {
                const int i = tid + j * NT;
                const int cig = c * CI_T + (i < CI_T * xw ? i / xw : 0);
                if (a.alpha_tab) { al[j] = als[cig]; ral[j] = als[cin_pad + cig]; }
                else { al[j] = cig < a.cin ? a.alpha[cig] : 1.0f; ral[j] = 1.0f / al[j]; }
            }
            snake_vec<XV>(xreg, al, ral);
        }
#pragma unroll
        for (int j = 0; j < XV; j++) {
            const int i = tid + j * NT;
            if (i < CI_T * xw) xd[i] = xreg[j];
        }
    };

    prefetch(0);
    __syncthreads();
    commit(0, 0);
    __syncthreads();
    for (int c = 0; c < n_chunks; c++) {
        const int buf = c & 1;
        if (c + 1 < n_chunks) prefetch(c + 1);
        const float *ws = wsb + buf * WCH;
        const float *xs = xsb + buf * xsz;
#pragma unroll 2
        for (int ci = 0; ci < CI_T; ci++) {
            // k slot 0 (lanes 0-31): x[ti];  k slot 1 (lanes 32-63): x[ti-1]
            const float bf = xs[ci * xw + wn * 32 + l31 + 1 - hi];
#pragma unroll
            for (int i = 0; i < MI; i++)
#pragma unroll
                for (int ph = 0; ph < S; ph++) {
                    const float af = ws[(ci * K2 + ph + hi * S) * CO_T + (wm * MI + i) * 32 + l31];
                    acc[i][ph] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[i][ph], 0, 0, 0);
                }
        }
        if (c + 1 < n_chunks) commit(c + 1, buf ^ 1);
        __syncthreads();
    }
    const int ti = ti0 + wn * 32 + l31;
#pragma unroll
    for (int i = 0; i < MI; i++) {
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int co = co0 + (wm * MI + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
            if (co >= a.cout) continue;
            const float bias = a.b ? a.b[co] : 0.0f;
#pragma unroll
            for (int ph = 0; ph < S; ph++) {
                const int to = ti * S + ph - a.pad;
                if (to >= 0 && to < Lout) yg[(int64_t) co * LoS + to] = acc[i][ph][e] + bias;
            }
        }
    }
}

// final conv: Cout = 1, k = 7 (dac_model.cpp:163-166: snake -> conv -> + bias -> tanh).  One output per
// thread; a 1-channel output gives the matrix pipe nothing to do, so this is a staged dot product.
#define C1_T 256
#define C1_CI 16
static __global__ __launch_bounds__(256, 4) void conv1d_cout1_kernel(ConvArgs a) {
    __shared__ float xs[C1_CI][C1_T + 8];
    __shared__ float wsm[C1_CI][8];
    const int tid = threadIdx.x, t0 = blockIdx.x * C1_T;
    const int LS = a.L, L = valid_len(a.frames, a.mult, a.L);
    if (t0 >= L) return;
    const float *xg = a.x + (int64_t) blockIdx.z * a.cin * LS;
    float *yg = a.y + (int64_t) blockIdx.z * LS;
    // A chunk = 16 input channels x (256 + 6) positions.  Every load of a chunk is requested before the first is used (clamped addresses, validity
    // applied afterwards; snake on the 16 values at once: one wave-uniform range test) and the NEXT chunk's loads fly under the current chunk's dot
    // products: the staging loop of rounds 1-4 (`for i: load, snake, store`) was 17 dependent round trips per chunk, 2.06 ms per launch at 0.19 of the
    // HBM peak for a kernel that reads its input once (profiles/r05/bench_full_call3.json, dac_final).  Four waves per SIMD stay resident (a first
    // version with two register sets took 348 registers, one workgroup per CU, and was 2.5 x slower: profiles/r05/dac_bench_call8.txt).
    // Same products in the same order: bit-identical output.
    const int tm = t0 + tid - a.pad, te = t0 + C1_T + tid - a.pad;   // this thread's position and (tid < 6) the halo position behind the tile
    const int tmc = min(max(tm, 0), L - 1), tec = min(max(te, 0), L - 1);
    const bool okm = tm >= 0 && tm < L, oke = te >= 0 && te < L;
    float v[C1_CI], e[C1_CI];
    auto request = [&](int ci0) __attribute__((always_inline)) {
#pragma unroll
        for (int ci = 0; ci < C1_CI; ci++) v[ci] = xg[(int64_t) min(ci0 + ci, a.cin - 1) * LS + tmc];
        if (tid < 6) {
#pragma unroll
            for (int ci = 0; ci < C1_CI; ci++) e[ci] = xg[(int64_t) min(ci0 + ci, a.cin - 1) * LS + tec];
        }
    };
    auto stage = [&](int ci0, float (&x)[C1_CI], bool ok, int col) __attribute__((always_inline)) {
        if (a.alpha) {
            float al[C1_CI], ral[C1_CI];
#pragma unroll
            for (int ci = 0; ci < C1_CI; ci++) { al[ci] = a.alpha[min(ci0 + ci, a.cin - 1)]; ral[ci] = 1.0f / al[ci]; }
            snake_vec<C1_CI>(x, al, ral);
        }
#pragma unroll
        for (int ci = 0; ci < C1_CI; ci++) {
            float y = (ok && ci0 + ci < a.cin) ? x[ci] : 0.0f;
            if (a.x_f16) y = (float) (_Float16) y;
            xs[ci][col] = y;
        }
    };
    request(0);
    float acc = 0.0f;
    for (int ci0 = 0; ci0 < a.cin; ci0 += C1_CI) {
        __syncthreads();   // the previous chunk's tile has been consumed
        stage(ci0, v, okm, tid);
        if (tid < 6) stage(ci0, e, oke, C1_T + tid);
        if (tid < C1_CI * 7) {
            const int ci = tid / 7, k = tid - ci * 7;
            wsm[ci][k] = (ci0 + ci < a.cin) ? a.w[(int64_t) (ci0 + ci) * 7 + k] : 0.0f;
        }
        if (ci0 + C1_CI < a.cin) request(ci0 + C1_CI);
        __syncthreads();
#pragma unroll
        for (int ci = 0; ci < C1_CI; ci++)
#pragma unroll
            for (int k = 0; k < 7; k++) acc += wsm[ci][k] * xs[ci][tid + k];
    }
    const int t = t0 + tid;
    if (t < L) {
        float vv = acc + (a.b ? a.b[0] : 0.0f);
        if (a.do_tanh) vv = tanhf(vv);
        yg[t] = vv;
    }
}


// ================================================================================================
// fp16 MFMA paths for F16 codec weights (quantize --convert-dac-to-f16, quantize_impl.cpp:264-266).
// ggml lowers conv_1d to im2col(F16) x kernel(F16) and conv_transpose_1d_f16_f32 converts its source to fp16
// (upstream ggml): inputs are rounded to fp16 AFTER snake, products accumulate in fp32.
// v_mfma_f32_32x32x16_f16 does exactly that: fp16 x fp16 products are exact in fp32.
// Fragment maps: A[i = lane&31][k = 8*(lane>>5) + 0..7], B[k = 8*(lane>>5) + 0..7][j = lane&31], D as above.
// LDS images:  weights [k-step][lane>>5][channel][8 k]  (pre-packed, straight 16-byte copies; conflict-free reads)
//              inputs  [position][CI_T + 8] fp16        (channel-contiguous: one 16-byte read per B fragment)
// ================================================================================================
typedef _Float16 half8d __attribute__((ext_vector_type(8)));

//   conv1d   src [cout][cin][KT]  -> dst [co_tile][chunk][k][cg][hi][CO_T][8]   (ci = chunk*CI_T + cg*16 + hi*8 + j)
//   convT1d  src [cin][cout][K2]  -> same with k = 0..K2-1
static __global__ void pack_conv_w16_kernel(const float *src, _Float16 *dst, int cout, int cin, int KT, int CO_T, int CI_T, int n_chunks,
                                     int transposed_src) {
    const int NCG = CI_T / 16;
    const int64_t total = (int64_t) ((cout + CO_T - 1) / CO_T) * n_chunks * KT * CI_T * CO_T;
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t) gridDim.x * blockDim.x) {
        const int j = (int) (i % 8);
        int64_t r = i / 8;
        const int col = (int) (r % CO_T); r /= CO_T;
        const int hi = (int) (r % 2); r /= 2;
        const int cg = (int) (r % NCG); r /= NCG;
        const int k = (int) (r % KT); r /= KT;
        const int ch = (int) (r % n_chunks);
        const int ct = (int) (r / n_chunks);
        const int co = ct * CO_T + col, ci = ch * CI_T + cg * 16 + hi * 8 + j;
        float v = 0.0f;
        if (co < cout && ci < cin)
            v = transposed_src ? src[((int64_t) ci * cout + co) * KT + k] : src[((int64_t) co * cin + ci) * KT + k];
        dst[i] = (_Float16) v;  // exact: the tensor was stored as F16
    }
}

// ================================================================================================
// EXPERIMENT (off by default, TTS_HIP_DAC_BF16X3=1): the k = 7 conv of F32 tensors with every fp32 operand split into
// three bf16 terms, x = x1 + x2 + x3 (x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2): 24 mantissa bits, the
// subtractions are exact), and the six products of weight >= 2^-16 on v_mfma_f32_32x32x16_bf16 with fp32 accumulation:
//   w*x ~= w1*x1 + w1*x2 + w2*x1 + w1*x3 + w2*x2 + w3*x1          (dropped: 2^-24 and below)
// bf16 x bf16 products are exact in fp32, so the error is the dropped terms (measured 6e-9 relative on K = 1344 sums,
// tests/test_oracle_cpu.py) under the fp32 accumulation's own rounding (7.5e-7 against 8.9e-7 for the 32x32x2 fp32 chain).
// 6 MFMAs at 16x the fp32 rate = a 2.7x higher matrix ceiling.
// One k step of the MFMA = 16 values of the reduction: the half-wave `hi` takes tap 2s + hi, 8 input channels each;
// the eighth tap (s = 3, hi = 1) has zero weights and reads tap 6's rows.  Chunk = 8 input channels x 7 taps.
// LDS images per plane:  weights [s][hi][channel co][8 ci] (16-byte rows, pre-packed: straight copies)
//                        inputs  [position][8 ci]          (16-byte rows: one read per B fragment, conflict-free)
// ================================================================================================
typedef __bf16 bf16x8d __attribute__((ext_vector_type(8)));
typedef unsigned int uint4d __attribute__((ext_vector_type(4)));

// x = h1 + h2 + h3 with bf16 terms (round-to-nearest-even conversions; the remainders are exact in fp32)
__device__ __forceinline__ void split_bf16x3(float x, __bf16 &h1, __bf16 &h2, __bf16 &h3) {
    h1 = (__bf16) x;
    const float r1 = x - (float) h1;
    h2 = (__bf16) r1;
    h3 = (__bf16) (r1 - (float) h2);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Operand splits of the codec's MFMA convolutions (dac_b3_kernels.h): how an fp32 operand is carried as 16-bit planes and which
// partial products are formed.  The 16-bit planes are stored as __bf16 containers whatever the scheme (16-byte rows are copied untyped).
//   SplitB3  x = b1 + b2 + b3 exactly, bf16 terms; six partial products of weight >= 2^-16 on v_mfma_f32_32x32x16_bf16 (rounds 3 - 5)
//   SplitH2  x = h + l, fp16 terms: h = fp16(x), l = fp16(x - h) (x - h is exact in fp32 and at most 2^-11 |x|; below 2^-14 it is an fp16
//            SUBNORMAL, which v_mfma_f32_32x32x16_f16 multiplies exactly — profiles/mfma_denorm.hip); three partial products
//            h h' + h l' + l h' (fp16 x fp16 is exact in fp32), the dropped l l' and the split remainders are <= 2^-22 of a product:
//            half the matrix instructions of SplitB3 for ~4 x its (fp32-level) error (tests/test_oracle_cpu.py)
//   SplitH1  x ~ fp16(x): one plane, one product — F16 codec tensors (ggml's fp16 im2col x fp16 kernel: exact products, the reference's
//            own arithmetic for `quantize --convert-dac-to-f16`)
// ------------------------------------------------------------------------------------------------------------------------------------
typedef _Float16 half8d __attribute__((ext_vector_type(8)));
struct SplitB3 {
    static constexpr int NPL = 3, NT = 6, ID = 0;
    // the six partial products, smallest first; term-major over the accumulators: consecutive MFMAs write different registers
    static __host__ __device__ constexpr int ta(int tm) { return tm == 0 ? 2 : tm == 1 ? 0 : tm == 2 ? 1 : tm == 3 ? 1 : 0; }
    static __host__ __device__ constexpr int tb(int tm) { return tm == 0 ? 0 : tm == 1 ? 2 : tm == 2 ? 1 : tm == 3 ? 0 : tm == 4 ? 1 : 0; }
    static __device__ __forceinline__ void split(float x, __bf16 (&p)[3]) { split_bf16x3(x, p[0], p[1], p[2]); }
    static __device__ __forceinline__ float16d mfma(bf16x8d a, bf16x8d b, float16d c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
struct SplitH2 {
    static constexpr int NPL = 2, NT = 3, ID = 1;
    static __host__ __device__ constexpr int ta(int tm) { return tm == 0 ? 1 : 0; }   // l h', h l', h h'
    static __host__ __device__ constexpr int tb(int tm) { return tm == 1 ? 1 : 0; }
    static __device__ __forceinline__ void split(float x, __bf16 (&p)[2]) {
        const _Float16 h = (_Float16) __builtin_fminf(__builtin_fmaxf(x, -65504.0f), 65504.0f);   // beyond fp16's range the high part saturates instead of becoming inf - inf
        const _Float16 l = (_Float16) (x - (float) h);
        p[0] = __builtin_bit_cast(__bf16, h);
        p[1] = __builtin_bit_cast(__bf16, l);
    }
    static __device__ __forceinline__ float16d mfma(bf16x8d a, bf16x8d b, float16d c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8d, a), __builtin_bit_cast(half8d, b), c, 0, 0, 0);
    }
};
struct SplitH1 {
    static constexpr int NPL = 1, NT = 1, ID = 2;
    static __host__ __device__ constexpr int ta(int) { return 0; }
    static __host__ __device__ constexpr int tb(int) { return 0; }
    static __device__ __forceinline__ void split(float x, __bf16 (&p)[1]) { p[0] = __builtin_bit_cast(__bf16, (_Float16) x); }
    static __device__ __forceinline__ float16d mfma(bf16x8d a, bf16x8d b, float16d c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8d, a), __builtin_bit_cast(half8d, b), c, 0, 0, 0);
    }
};
// a value into its planes at dst[base + pl * plane_sz] (weight packers: the scheme is a run-time argument there)
__device__ __forceinline__ void split_store(int scheme, float v, __bf16 *dst, int64_t base, int64_t plane_sz) {
    if (scheme == 0) { __bf16 p[3]; SplitB3::split(v, p); dst[base] = p[0]; dst[base + plane_sz] = p[1]; dst[base + 2 * plane_sz] = p[2]; }
    else if (scheme == 1) { __bf16 p[2]; SplitH2::split(v, p); dst[base] = p[0]; dst[base + plane_sz] = p[1]; }
    else { __bf16 p[1]; SplitH1::split(v, p); dst[base] = p[0]; }
}
__host__ __device__ inline int split_planes(int scheme) { return scheme == 0 ? 3 : scheme == 1 ? 2 : 1; }

//   conv1d  src [cout][cin][7]  ->  dst [co_tile][chunk][plane][s][hi][CO_T][8]   (ci = chunk*8 + j, tap = 2s + hi, tap 7 = 0)
static __global__ void pack_conv_w_b3_kernel(const float *src, __bf16 *dst, int cout, int cin, int CO_T, int n_chunks, int KT = 7, int scheme = 0) {
    const int NST = (KT + 1) / 2;                                                     // k-steps per chunk: tap pairs (an odd tap count leaves one zero slot)
    const int64_t plane_sz = (int64_t) 2 * NST * CO_T * 8;                            // NST steps x 2 halves x CO_T x 8
    const int64_t total = (int64_t) ((cout + CO_T - 1) / CO_T) * n_chunks * plane_sz;  // one thread per (element, all planes)
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t) gridDim.x * blockDim.x) {
        const int j = (int) (i % 8);
        int64_t r = i / 8;
        const int col = (int) (r % CO_T); r /= CO_T;
        const int hi = (int) (r % 2); r /= 2;
        const int st = (int) (r % NST); r /= NST;
        const int ch = (int) (r % n_chunks);
        const int ct = (int) (r / n_chunks);
        const int co = ct * CO_T + col, ci = ch * 8 + j, tap = 2 * st + hi;
        float v = 0.0f;
        if (co < cout && ci < cin && tap < KT) v = src[((int64_t) co * cin + ci) * KT + tap];
        const int64_t base = ((int64_t) ct * n_chunks + ch) * split_planes(scheme) * plane_sz + (i % plane_sz);
        split_store(scheme, v, dst, base, plane_sz);
    }
}

// KT (round 4): any odd tap count — (KT + 1) / 2 k-steps of tap pairs per 8-channel chunk, the odd slot on zero weights (Kokoro's k = 3 / 5 / 7 / 11
// same-convolutions with dilations 1 / 3 / 5; the DAC's fallback k = 7).
// SP (round 6): the operand split — SplitB3 (three bf16 planes, six products) or SplitH2 (fp16 hi + lo, three products), as in dac_b3_kernels.h.
template <int MI, int NI, int WM, int WN, int KT = 7, typename SP = SplitB3>
__global__ __launch_bounds__(64 * WM * WN, WM * WN >= 8 ? 1 : 2) void conv1d_mfma_b3_kernel(ConvArgs a) {   // 2 waves per SIMD either way
    constexpr int CI_T = 8, NST = (KT + 1) / 2, NPL = SP::NPL;
    constexpr int CO_T = 32 * MI * WM, T_T = 32 * NI * WN, NT = 64 * WM * WN;
    constexpr int WPL = 2 * NST * CO_T * 8;                  // bf16 per weight plane of a chunk
    constexpr int WV = (NPL * WPL / 8 + NT - 1) / NT;        // 16-byte vectors per thread per chunk (all planes)
    constexpr int XU = (T_T + (KT - 1) * 9 + NT - 1) / NT;   // positions per thread per chunk (8 channels each), dilation <= 9
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int halo = (KT - 1) * a.dil;
    const int xw = T_T + halo;
    const int xpl = xw * 8;                                  // bf16 per input plane
    __bf16 *wsb = (__bf16 *) smem;                           // [2][NPL][WPL]
    __bf16 *xsb = wsb + 2 * NPL * WPL;                       // [2][NPL][xpl]
    float *als = (float *) (xsb + 2 * NPL * xpl);            // [cin_pad] alpha, then [cin_pad] 1/alpha   (xpl * 2 B is a multiple of 16)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv / WN, wn = wv % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    const int t0 = blockIdx.x * T_T, co0 = blockIdx.y * CO_T;
    const int n_chunks = (a.cin + CI_T - 1) / CI_T;
    const int cin_pad = n_chunks * CI_T;
    const int LS = a.L, L = valid_len(a.frames, a.mult, a.L);
    if (t0 >= L) return;
    const float *xg = a.x + (int64_t) blockIdx.z * a.cin * LS;
    float *yg = a.y + (int64_t) blockIdx.z * a.cout * LS;
    const float *rg = a.resid ? a.resid + (int64_t) blockIdx.z * a.cout * LS : nullptr;
    const uint4d *wg = (const uint4d *) ((const __bf16 *) a.w + (int64_t) blockIdx.y * n_chunks * NPL * WPL);

    if (a.alpha && a.alpha_tab) {
        for (int i = tid; i < cin_pad; i += NT) {
            const float al = i < a.cin ? a.alpha[i] : 1.0f;
            als[i] = al;
            als[cin_pad + i] = 1.0f / al;
        }
    }

    float16d acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NI; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

    uint4d wreg[WV];
    float xreg[XU][8];
    auto prefetch = [&](int c) __attribute__((always_inline)) {
        const uint4d *wp = wg + (int64_t) c * (NPL * WPL / 8);
#pragma unroll
        for (int j = 0; j < WV; j++) {
            const int i = tid + j * NT;
            if (i < NPL * WPL / 8) wreg[j] = wp[i];
        }
#pragma unroll
        for (int j = 0; j < XU; j++) {
            const int p = tid + j * NT;            // lanes run along positions: coalesced rows
            const int t = t0 + p - a.pad;
            const bool ok = p < xw && t >= 0 && t < L;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int cig = c * CI_T + e;
                xreg[j][e] = (ok && cig < a.cin) ? xg[(int64_t) cig * LS + t] : 0.0f;
            }
        }
    };
    auto commit = [&](int c, int buf) __attribute__((always_inline)) {
        uint4d *wd = (uint4d *) (wsb + buf * NPL * WPL);
#pragma unroll
        for (int j = 0; j < WV; j++) {
            const int i = tid + j * NT;
            if (i < NPL * WPL / 8) wd[i] = wreg[j];
        }
        __bf16 *xd = xsb + buf * NPL * xpl;
#pragma unroll
        for (int j = 0; j < XU; j++) {
            const int p = tid + j * NT;
            if (p < xw) {
                bf16x8d hp[NPL];
                if (a.alpha) {
                    float al[8], ral[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const int cig = c * CI_T + e;
                        if (a.alpha_tab) { al[e] = als[cig]; ral[e] = als[cin_pad + cig]; }
                        else { al[e] = cig < a.cin ? a.alpha[cig] : 1.0f; ral[e] = 1.0f / al[e]; }
                    }
                    snake_vec<8>(xreg[j], al, ral);  // snake(0) == 0: zero padding is preserved
                }
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    __bf16 bp[NPL];
                    SP::split(xreg[j][e], bp);
#pragma unroll
                    for (int pl = 0; pl < NPL; pl++) hp[pl][e] = bp[pl];
                }
#pragma unroll
                for (int pl = 0; pl < NPL; pl++) *(bf16x8d *) (xd + pl * xpl + p * 8) = hp[pl];
            }
        }
    };

    prefetch(0);
    __syncthreads();  // alpha table visible
    commit(0, 0);
    __syncthreads();
    for (int c = 0; c < n_chunks; c++) {
        const int buf = c & 1;
        if (c + 1 < n_chunks) prefetch(c + 1);
        const __bf16 *ws = wsb + buf * NPL * WPL;
        const __bf16 *xs = xsb + buf * NPL * xpl;
#pragma unroll
        for (int st = 0; st < NST; st++) {
            const int tap = (2 * st + hi < KT) ? 2 * st + hi : KT - 1;   // the slot past the last tap: zero weights, any valid rows
            bf16x8d af[NPL][MI], bf[NPL][NI];
#pragma unroll
            for (int pl = 0; pl < NPL; pl++) {
#pragma unroll
                for (int i = 0; i < MI; i++)
                    af[pl][i] = *(const bf16x8d *) (ws + pl * WPL + (((st * 2 + hi) * CO_T) + (wm * MI + i) * 32 + l31) * 8);
#pragma unroll
                for (int j = 0; j < NI; j++)
                    bf[pl][j] = *(const bf16x8d *) (xs + pl * xpl + ((wn * NI + j) * 32 + l31 + tap * a.dil) * 8);
            }
            // the partial products, smallest first; term-major so that consecutive MFMAs write different accumulators
#pragma unroll
            for (int tm = 0; tm < SP::NT; tm++)
#pragma unroll
                for (int i = 0; i < MI; i++)
#pragma unroll
                    for (int j = 0; j < NI; j++)
                        acc[i][j] = SP::mfma(af[SP::ta(tm)][i], bf[SP::tb(tm)][j], acc[i][j]);
        }
        if (c + 1 < n_chunks) commit(c + 1, buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < MI; i++) {
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int co = co0 + (wm * MI + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
            if (co >= a.cout) continue;
            const float bias = a.b ? a.b[co] : 0.0f;
            const float al_o = a.alpha_out ? a.alpha_out[co] : 1.0f, ral_o = 1.0f / al_o;
#pragma unroll
            for (int j = 0; j < NI; j++) {
                const int t = t0 + (wn * NI + j) * 32 + l31;
                if (t >= L) continue;
                float v = acc[i][j][e] + bias;
                if (rg) v = v + rg[(int64_t) co * LS + t];
                if (a.alpha_out) v = snake_f(v, al_o, ral_o);
                if (a.do_tanh) v = tanhf(v);
                yg[(int64_t) co * LS + t] = v;
            }
        }
    }
}

template <int KT, int MI, int NI, int WM, int WN, int CI_T>
__global__ __launch_bounds__(64 * WM * WN, (KT == 1 && MI == 3) ? 2 : 1) void conv1d_mfma16_kernel(ConvArgs a) {   // the 96-channel k = 1 tile came out at 264 registers: 8 over what a second wave per SIMD allows
    constexpr int CO_T = 32 * MI * WM, T_T = 32 * NI * WN, NT = 64 * WM * WN;
    constexpr int NCG = CI_T / 16, QG = CI_T / 8, XS = CI_T + 8;
    constexpr int WCH = KT * CI_T * CO_T;                     // halves per packed weight chunk
    constexpr int WV = (WCH / 8 + NT - 1) / NT;               // 16-byte vectors per thread per chunk
    constexpr int XU = ((T_T + (KT - 1) * 9) * QG + NT - 1) / NT;  // (position, 8-channel group) units per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int halo = (KT - 1) * a.dil;
    const int xw = T_T + halo;
    const int xsz = xw * XS;                                  // halves (XS is a multiple of 8)
    _Float16 *wsb = (_Float16 *) smem;                        // [2][WCH]
    _Float16 *xsb = wsb + 2 * WCH;                            // [2][xsz]
    float *als = (float *) (xsb + 2 * xsz);                   // [cin_pad] alpha, then [cin_pad] 1/alpha
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv / WN, wn = wv % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    const int t0 = blockIdx.x * T_T, co0 = blockIdx.y * CO_T;
    const int n_chunks = (a.cin + CI_T - 1) / CI_T;
    const int cin_pad = n_chunks * CI_T;
    const int LS = a.L, L = valid_len(a.frames, a.mult, a.L);
    if (t0 >= L) return;
    const float *xg = a.x + (int64_t) blockIdx.z * a.cin * LS;
    float *yg = a.y + (int64_t) blockIdx.z * a.cout * LS;
    const float *rg = a.resid ? a.resid + (int64_t) blockIdx.z * a.cout * LS : nullptr;
    const half8d *wg = (const half8d *) ((const _Float16 *) a.w + (int64_t) blockIdx.y * n_chunks * WCH);

    if (a.alpha && a.alpha_tab) {
        for (int i = tid; i < cin_pad; i += NT) {
            const float al = i < a.cin ? a.alpha[i] : 1.0f;
            als[i] = al;
            als[cin_pad + i] = 1.0f / al;
        }
    }

    float16d acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NI; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

    half8d wreg[WV];
    float xreg[XU][8];
    auto prefetch = [&](int c) __attribute__((always_inline)) {
        const half8d *wp = wg + (int64_t) c * (WCH / 8);
#pragma unroll
        for (int j = 0; j < WV; j++) {
            const int i = tid + j * NT;
            if (i < WCH / 8) wreg[j] = wp[i];
        }
#pragma unroll
        for (int j = 0; j < XU; j++) {
            const int u = tid + j * NT;
            const int q = u / xw, p = u - q * xw;   // lanes run along positions: coalesced rows
            const int t = t0 + p - a.pad;
            const bool ok = q < QG && t >= 0 && t < L;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int cig = c * CI_T + q * 8 + e;
                xreg[j][e] = (ok && cig < a.cin) ? xg[(int64_t) cig * LS + t] : 0.0f;
            }
        }
    };
    auto commit = [&](int c, int buf) __attribute__((always_inline)) {   // left out of line by the inliner in two k = 1 instantiations: every captured array went through scratch (28.9 TFLOP/s)
        half8d *wd = (half8d *) (wsb + buf * WCH);
#pragma unroll
        for (int j = 0; j < WV; j++) {
            const int i = tid + j * NT;
            if (i < WCH / 8) wd[i] = wreg[j];
        }
        _Float16 *xd = xsb + buf * xsz;
#pragma unroll
        for (int j = 0; j < XU; j++) {
            const int u = tid + j * NT;
            const int q = u / xw, p = u - q * xw;
            if (q < QG) {
                half8d h;
                if (a.alpha) {
                    float al[8], ral[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const int cig = c * CI_T + q * 8 + e;
                        if (a.alpha_tab) { al[e] = als[cig]; ral[e] = als[cin_pad + cig]; }
                        else { al[e] = cig < a.cin ? a.alpha[cig] : 1.0f; ral[e] = 1.0f / al[e]; }
                    }
                    snake_vec<8>(xreg[j], al, ral);  // snake(0) == 0: zero padding is preserved
                }
#pragma unroll
                for (int e = 0; e < 8; e++) h[e] = (_Float16) xreg[j][e];   // the fp16 im2col
                *(half8d *) (xd + p * XS + q * 8) = h;
            }
        }
    };

    prefetch(0);
    __syncthreads();  // alpha table visible
    commit(0, 0);
    __syncthreads();
    for (int c = 0; c < n_chunks; c++) {
        const int buf = c & 1;
        if (c + 1 < n_chunks) prefetch(c + 1);
        const _Float16 *ws = wsb + buf * WCH;
        const _Float16 *xs = xsb + buf * xsz;
#pragma unroll
        for (int tap = 0; tap < KT; tap++) {
#pragma unroll
            for (int cg = 0; cg < NCG; cg++) {
                half8d af[MI], bf[NI];
#pragma unroll
                for (int i = 0; i < MI; i++)
                    af[i] = *(const half8d *) (ws + ((((tap * NCG + cg) * 2 + hi) * CO_T) + (wm * MI + i) * 32 + l31) * 8);
#pragma unroll
                for (int j = 0; j < NI; j++)
                    bf[j] = *(const half8d *) (xs + ((wn * NI + j) * 32 + l31 + tap * a.dil) * XS + cg * 16 + hi * 8);
#pragma unroll
                for (int i = 0; i < MI; i++)
#pragma unroll
                    for (int j = 0; j < NI; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
        if (c + 1 < n_chunks) commit(c + 1, buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < MI; i++) {
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int co = co0 + (wm * MI + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
            if (co >= a.cout) continue;
            const float bias = a.b ? a.b[co] : 0.0f;
            const float al_o = a.alpha_out ? a.alpha_out[co] : 1.0f, ral_o = 1.0f / al_o;
#pragma unroll
            for (int j = 0; j < NI; j++) {
                const int t = t0 + (wn * NI + j) * 32 + l31;
                if (t >= L) continue;
                float v = acc[i][j][e] + bias;
                if (rg) v = v + rg[(int64_t) co * LS + t];
                if (a.alpha_out) v = snake_f(v, al_o, ral_o);
                if (a.do_tanh) v = tanhf(v);
                yg[(int64_t) co * LS + t] = v;
            }
        }
    }
}

// ConvTranspose1d, phase-decomposed as convt1d_mfma_kernel; one k-step = 16 input channels of one tap slot
// (slot 0: x[ti] with w[..][phi], slot 1: x[ti-1] with w[..][phi+S]).
template <int S, int MI, int WM, int WN, int CI_T>
__global__ __launch_bounds__(64 * WM * WN, CONVT_MIN_WAVES) void convt1d_mfma16_kernel(ConvTArgs a) {
    constexpr int CO_T = 32 * MI * WM, TI_T = 32 * WN, NT = 64 * WM * WN, K2 = 2 * S;
    constexpr int NCG = CI_T / 16, QG = CI_T / 8, XS = CI_T + 8;
    constexpr int WCH = CI_T * K2 * CO_T;
    constexpr int WV = (WCH / 8 + NT - 1) / NT;
    constexpr int xw = TI_T + 1;                   // positions ti0-1 .. ti0+TI_T-1
    constexpr int xsz = xw * XS;
    constexpr int XU = (xw * QG + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16 *wsb = (_Float16 *) smem;             // [2][WCH]
    _Float16 *xsb = wsb + 2 * WCH;                 // [2][xsz]
    float *als = (float *) (xsb + 2 * xsz);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv / WN, wn = wv % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    const int ti0 = blockIdx.x * TI_T, co0 = blockIdx.y * CO_T;
    const int n_chunks = (a.cin + CI_T - 1) / CI_T;
    const int cin_pad = n_chunks * CI_T;
    const int LS = a.L, L = valid_len(a.frames, a.mult, a.L);
    const int LoS = a.Lout, Lout = a.frames ? (L - 1) * S - 2 * a.pad + K2 : a.Lout;
    if (ti0 > L) return;  // ti runs 0..L inclusive
    const float *xg = a.x + (int64_t) blockIdx.z * a.cin * LS;
    float *yg = a.y + (int64_t) blockIdx.z * a.cout * LoS;
    const half8d *wg = (const half8d *) ((const _Float16 *) a.w + (int64_t) blockIdx.y * n_chunks * WCH);

    if (a.alpha && a.alpha_tab) {
        for (int i = tid; i < cin_pad; i += NT) {
            const float al = i < a.cin ? a.alpha[i] : 1.0f;
            als[i] = al;
            als[cin_pad + i] = 1.0f / al;
        }
    }

    float16d acc[MI][S];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int ph = 0; ph < S; ph++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][ph][e] = 0.0f;

    half8d wreg[WV];
    float xreg[XU][8];
    auto prefetch = [&](int c) {
        const half8d *wp = wg + (int64_t) c * (WCH / 8);
#pragma unroll
        for (int j = 0; j < WV; j++) {
            const int i = tid + j * NT;
            if (i < WCH / 8) wreg[j] = wp[i];
        }
#pragma unroll
        for (int j = 0; j < XU; j++) {
            const int u = tid + j * NT;
            const int q = u / xw, p = u - q * xw;
            const int ti = ti0 - 1 + p;
            const bool ok = q < QG && ti >= 0 && ti < L;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int cig = c * CI_T + q * 8 + e;
                xreg[j][e] = (ok && cig < a.cin) ? xg[(int64_t) cig * LS + ti] : 0.0f;
            }
        }
    };
    auto commit = [&](int c, int buf) {
        half8d *wd = (half8d *) (wsb + buf * WCH);
#pragma unroll
        for (int j = 0; j < WV; j++) {
            const int i = tid + j * NT;
            if (i < WCH / 8) wd[i] = wreg[j];
        }
        _Float16 *xd = xsb + buf * xsz;
#pragma unroll
        for (int j = 0; j < XU; j++) {
            const int u = tid + j * NT;
            const int q = u / xw, p = u - q * xw;
            if (q < QG) {
                half8d h;
                if (a.alpha) {
                    float al[8], ral[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const int cig = c * CI_T + q * 8 + e;
                        if (a.alpha_tab) { al[e] = als[cig]; ral[e] = als[cin_pad + cig]; }
                        else { al[e] = cig < a.cin ? a.alpha[cig] : 1.0f; ral[e] = 1.0f / al[e]; }
                    }
                    snake_vec<8>(xreg[j], al, ral);
                }
#pragma unroll
                for (int e = 0; e < 8; e++) h[e] = (_Float16) xreg[j][e];
                *(half8d *) (xd + p * XS + q * 8) = h;
            }
        }
    };

    prefetch(0);
    __syncthreads();
    commit(0, 0);
    __syncthreads();
    for (int c = 0; c < n_chunks; c++) {
        const int buf = c & 1;
        if (c + 1 < n_chunks) prefetch(c + 1);
        const _Float16 *ws = wsb + buf * WCH;
        const _Float16 *xs = xsb + buf * xsz;
#pragma unroll
        for (int ts = 0; ts < 2; ts++) {
#pragma unroll
            for (int cg = 0; cg < NCG; cg++) {
                const half8d bf = *(const half8d *) (xs + (wn * 32 + l31 + 1 - ts) * XS + cg * 16 + hi * 8);
#pragma unroll
                for (int i = 0; i < MI; i++)
#pragma unroll
                    for (int ph = 0; ph < S; ph++) {
                        const half8d af = *(const half8d *) (ws + (((((ph + ts * S) * NCG + cg) * 2 + hi) * CO_T) + (wm * MI + i) * 32 + l31) * 8);
                        acc[i][ph] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[i][ph], 0, 0, 0);
                    }
            }
        }
        if (c + 1 < n_chunks) commit(c + 1, buf ^ 1);
        __syncthreads();
    }
    const int ti = ti0 + wn * 32 + l31;
#pragma unroll
    for (int i = 0; i < MI; i++) {
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int co = co0 + (wm * MI + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
            if (co >= a.cout) continue;
            const float bias = a.b ? a.b[co] : 0.0f;
#pragma unroll
            for (int ph = 0; ph < S; ph++) {
                const int to = ti * S + ph - a.pad;
                if (to >= 0 && to < Lout) yg[(int64_t) co * LoS + to] = acc[i][ph][e] + bias;
            }
        }
    }
}

// kokoro_kernels.h — first device version of the Kokoro graphs (/root/reference/src/models/kokoro/model.cpp), written for
// parity: plain fp32 kernels, one thread or one wave per output, no tiling.  Layouts: sequences [L][C] (row = position) in the
// transformer / LSTM parts, [C][L] (row = channel) in the convolutional parts, as in oracle/kokoro_oracle.c.
//   kk_linear_kernel        mul_mat + bias for R rows (ALBERT projections :969-1003, LSTM input gates :54-57, AdaIN gamma / beta :93-94)
//   kk_norm_rows_kernel     ggml_norm over the features of each row, optional affine / AdaLayerNorm (1 + gamma) x + beta (:25-31, :1021-1026)
//   kk_albert_attn_kernel   softmax(scale q.k) v per (head, row), no mask (:975-990)
//   kk_lstm_kernel          one direction of build_lstm_run (:53-86), sequential over the positions, one workgroup
//   kk_adain_kernel         instance norm over L per channel, (1 + gamma) x + beta, optional leaky relu / snake (:96-101, :142-150)
//   kk_conv1d_kernel        ggml_conv_1d with stride / padding / dilation (+ bias, optional nearest 2x input, accumulate)
//   kk_convt1d_kernel       ggml_conv_transpose_1d (stride, padding), dense (:211) and depthwise with output padding (:104)
//   kk_sine_* / kk_stft / kk_istft   harmonic source (:173-193), STFT conditioning (:199-206), iSTFT head (:240-241)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "wave_ops.h"

// y[r][n] = sum_k W[n][k] x[r][k] + b[n]; one wave per output
static __global__ __launch_bounds__(256) void kk_linear_kernel(const float *W, const float *b, const float *x, int ldx, int R, int K, int N, float *y, int ldy, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int64_t o = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= (int64_t) R * N) return;
    const int r = (int) (o / N), n = (int) (o - (int64_t) r * N);
    float acc = 0.0f;
    for (int k = lane; k < K; k += 64) acc += W[(int64_t) n * K + k] * x[(int64_t) r * ldx + k];
    acc = wave_sum(acc);
    if (lane == 0) {
        float *p = y + (int64_t) r * ldy + n;
        const float v = acc + (b ? b[n] : 0.0f);
        *p = accumulate ? *p + v : v;
    }
}

// The same product for many rows (ALBERT over the whole phoneme sequence: 402 rows x 768 -> 3072) on the exact-fp32 matrix pipe:
// kk_linear_kernel gives a wave to every output and re-reads a weight row and an activation row from L2 for each (7 GB of L2 reads for
// one FFN projection).  Here a workgroup owns 64 rows x 64 features; 16-column slices of both operands are transposed into LDS
// ([k][row], [k][feature]) so that v_mfma_f32_32x32x2_f32 reads one float per lane (A[i = row][k], B[k][j = feature]); the next slice's
// global loads travel under the MFMAs of the current one.  fp32 products, accumulation over k ascending as an fma chain.
// Requires K % 16 == 0, ldx % 4 == 0 and 16-byte aligned x / W (the caller checks); b may be NULL.
static __global__ __launch_bounds__(256) void kk_linear_mfma_kernel(const float *W, const float *b, const float *x, int ldx, int R, int K, int N, float *y, int ldy, int accumulate) {
    typedef float f16v __attribute__((ext_vector_type(16)));
    typedef float f4v __attribute__((ext_vector_type(4)));
    constexpr int KC = 16, LD = 68;
    __shared__ float xs[2][KC][LD], ws[2][KC][LD];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5, wm = wv >> 1, wn = wv & 1;
    const int r0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int si = tid >> 2, k4 = (tid & 3) * 4;   // staging: row / feature si, columns k4..k4+3 of the slice
    const bool xok = r0 + si < R, wok = n0 + si < N;
    const float *xp = x + (int64_t) (r0 + si) * ldx + k4, *wp = W + (int64_t) (n0 + si) * K + k4;
    f4v xa = {0.f, 0.f, 0.f, 0.f}, wa = {0.f, 0.f, 0.f, 0.f};
    if (xok) xa = *(const f4v *) xp;
    if (wok) wa = *(const f4v *) wp;
    f16v acc;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.0f;
    const int nc = K / KC;
    for (int c = 0; c < nc; c++) {
        const int buf = c & 1;
#pragma unroll
        for (int e = 0; e < 4; e++) { xs[buf][k4 + e][si] = xa[e]; ws[buf][k4 + e][si] = wa[e]; }
        if (c + 1 < nc) {
            if (xok) xa = *(const f4v *) (xp + (c + 1) * KC);
            if (wok) wa = *(const f4v *) (wp + (c + 1) * KC);
        }
        __syncthreads();   // slice c is in LDS; the other buffer was last read before the previous barrier
#pragma unroll
        for (int kk = 0; kk < KC; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[buf][kk + hi][wm * 32 + l31], ws[buf][kk + hi][wn * 32 + l31], acc, 0, 0, 0);
    }
    const int n = n0 + wn * 32 + l31;
    if (n >= N) return;
    const float bias = b ? b[n] : 0.0f;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const int r = r0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
        if (r >= R) continue;
        float *p = y + (int64_t) r * ldy + n;
        const float v = acc[e] + bias;
        *p = accumulate ? *p + v : v;
    }
}

// mode 0: y = norm(x) * w + b (w may be NULL: plain norm); mode 1 (AdaLayerNorm): y = n + n * gamma + beta with gamma = w, beta = b
static __global__ __launch_bounds__(64) void kk_norm_rows_kernel(const float *x, int ldx, int H, float eps, const float *w, const float *b, int mode, float *y, int ldy) {
    const int r = blockIdx.x, lane = threadIdx.x;
    const float *xr = x + (int64_t) r * ldx;
    float s = 0.0f;
    for (int i = lane; i < H; i += 64) s += xr[i];
    const float mean = wave_sum(s) / (float) H;
    float v2 = 0.0f;
    for (int i = lane; i < H; i += 64) { const float d = xr[i] - mean; v2 += d * d; }
    const float scale = 1.0f / sqrtf(wave_sum(v2) / (float) H + eps);
    for (int i = lane; i < H; i += 64) {
        const float n = (xr[i] - mean) * scale;
        float o = n;
        if (mode == 1) o = (n + n * w[i]) + b[i];
        else if (w) o = n * w[i] + (b ? b[i] : 0.0f);
        y[(int64_t) r * ldy + i] = o;
    }
}

// q, k, v [n][H] -> out [n][H]; grid (heads, n), 64 threads; dynamic LDS n floats
static __global__ __launch_bounds__(64) void kk_albert_attn_kernel(const float *q, const float *k, const float *v, int n, int H, int hs, float scale, float *out) {
    extern __shared__ float kk_sc[];
    const int h = blockIdx.x, t = blockIdx.y, lane = threadIdx.x;
    const float *qr = q + (int64_t) t * H + h * hs;
    float mx = -INFINITY;
    for (int j = lane; j < n; j += 64) {
        const float *kr = k + (int64_t) j * H + h * hs;
        float d = 0.0f;
        for (int e = 0; e < hs; e++) d += qr[e] * kr[e];
        d *= scale;
        kk_sc[j] = d;
        mx = fmaxf(mx, d);
    }
    mx = wave_max(mx);
    float sum = 0.0f;
    for (int j = lane; j < n; j += 64) { const float p = expf(kk_sc[j] - mx); kk_sc[j] = p; sum += p; }
    sum = wave_sum(sum);
    __syncthreads();
    const float inv = 1.0f / sum;
    for (int e = lane; e < hs; e += 64) {
        float a = 0.0f;
        for (int j = 0; j < n; j++) a += (kk_sc[j] * inv) * v[(int64_t) j * H + h * hs + e];
        out[(int64_t) t * H + h * hs + e] = a;
    }
}

// The same attention for head size 64 with the keys staged through LDS (round 6).  kk_albert_attn_kernel's lanes walk a K row each (256 bytes per lane, 64
// separate lines per load instruction: 270 us per launch at 402 positions, 9 % of a synthesis); here a workgroup owns four rows of one head, a chunk of 64
// keys is copied to LDS with coalesced loads ([64][65] floats: a lane's key row is conflict-free) and every wave takes its row's dot products from there.
// The arithmetic is kk_albert_attn_kernel's, term for term (sequential 64-term dots, lane-strided max / sum, the value sum in key order): same results.
static __global__ __launch_bounds__(256) void kk_albert_attn64_kernel(const float *q, const float *k, const float *v, int n, int H, float scale, float *out) {
    extern __shared__ float kk_sm[];
    constexpr int HS = 64, LD = 65;
    float *ks = kk_sm;                        // [64][65] the chunk's keys
    float *qs = ks + 64 * LD;                 // [4][64] the rows' queries
    float *sc = qs + 4 * HS;                  // [4][n] scores -> probabilities
    const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = blockIdx.y * 4 + wave;
    const bool live = t < n;
    qs[wave * HS + lane] = live ? q[(int64_t) t * H + h * HS + lane] : 0.0f;
    float *scw = sc + (size_t) wave * n;
    float mx = -INFINITY;
    for (int j0 = 0; j0 < n; j0 += 64) {
        __syncthreads();                      // the previous chunk is consumed (and, first time, the queries are in place)
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int idx = tid + i * 256, key = idx >> 6, e = idx & 63;
            ks[key * LD + e] = k[(int64_t) min(j0 + key, n - 1) * H + h * HS + e];
        }
        __syncthreads();
        if (live && j0 + lane < n) {
            const float *qr = qs + wave * HS, *kr = ks + lane * LD;
            float d = 0.0f;
            for (int e = 0; e < HS; e++) d = __builtin_fmaf(qr[e], kr[e], d);   // explicit: the unrolled form otherwise gets some products as v_pk_mul + v_add (no contraction), kk_albert_attn_kernel's loop is all fma
            d *= scale;
            scw[j0 + lane] = d;
            mx = fmaxf(mx, d);
        }
    }
    if (!live) return;                        // (no barrier below)
    mx = wave_max(mx);
    float sum = 0.0f;
    for (int j = lane; j < n; j += 64) { const float p = expf(scw[j] - mx); scw[j] = p; sum += p; }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    float a = 0.0f;
    for (int j = 0; j < n; j++) a = __builtin_fmaf(scw[j] * inv, v[(int64_t) j * H + h * HS + lane], a);
    out[(int64_t) t * H + h * HS + lane] = a;
}

// ggml_gelu as the reference's CPU path evaluates it (kokoro/model.cpp:1000): tanh form through the table indexed by the fp16 bits of x,
// i.e. x rounded to fp16 and the result rounded to fp16, 0 / x outside (-10, 10) — the same arithmetic as gelu_apply(mode 1), parler_kernels.h
static __global__ void kk_gelu_kernel(float *x, int64_t n) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    float o;
    if (v <= -10.0f) o = 0.0f;
    else if (v >= 10.0f) o = v;
    else {
        const float vr = (float) (_Float16) v;
        o = (float) (_Float16) (0.5f * vr * (1.0f + tanhf(0.79788456080286535587989211986876f * vr * (1.0f + 0.044715f * vr * vr))));
    }
    x[i] = o;
}

// out[i] = (a[i] + b[i]) * scale
static __global__ void kk_add_kernel(const float *a, const float *b, float *out, int64_t n, float scale) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (a[i] + b[i]) * scale;
}

// build_albert_inputs (:10-15): x[t][e] = (token_embd[tok[t]][e] + position_embd[t][e]) + token_type[e]
static __global__ void kk_albert_embed_kernel(const float *tok_embd, const float *pos_embd, const float *type_embd, const uint32_t *tok, int n, int E, float *x) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t) n * E) return;
    const int t = (int) (i / E), e = (int) (i - (int64_t) t * E);
    x[i] = (tok_embd[(int64_t) tok[t] * E + e] + pos_embd[(int64_t) t * E + e]) + type_embd[e];
}

// x[t][off + s] = style[s] for every row (the style half concatenated to the predictor states :1014, :1027)
static __global__ void kk_fill_cols_kernel(float *x, int n, int ld, int off, const float *style, int S) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t) n * S) return;
    const int t = (int) (i / S), s = (int) (i - (int64_t) t * S);
    x[(int64_t) t * ld + off + s] = style[s];
}

// lens[t] = clamp(round(sum_e sigmoid(dur[t][e])), 1, 50) (:1035-1037)
static __global__ void kk_duration_kernel(const float *dur, int n, int ND, float *lens) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    float s = 0.0f;
    for (int e = 0; e < ND; e++) s += 1.0f / (1.0f + expf(-dur[(int64_t) t * ND + e]));
    s = roundf(s);
    lens[t] = s < 1.0f ? 1.0f : (s > 50.0f ? 50.0f : s);
}

// token embedding rows onto channel-major output: y[c][t] = embd[tok[t]][c]
static __global__ void kk_embed_cols_kernel(const float *embd, const uint32_t *tok, int n, int C, float *y) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t) n * C) return;
    const int c = (int) (i / n), t = (int) (i - (int64_t) c * n);
    y[i] = embd[(int64_t) tok[t] * C + c];
}

// One direction of an LSTM cell over L positions.  pre [4][L][hid]: the input part of the gates i, f, g, o for every position
// (W_ih x + b_ih); whh [4][hid][hid], bhh [4][hid].  One workgroup of `hid` threads (hid <= 1024); h lives in LDS.
// The recurrence spread over hid/16 workgroups per direction.  One workgroup re-reads the whole recurrent matrix (4 x hid x hid floats:
// 1 MB at hid = 256) from L2 at every time step, and one CU's load path caps that at ~7.5 us per step (kk_lstm_kernel below: 9 ms per
// call at 1200 steps).  Here a workgroup owns 16 hidden units: thread (unit u, gate g, part p) keeps a quarter (CP = hid/4 columns) of
// row (g, u) of W_hh in registers for the whole sequence, so a step reads nothing but the previous hidden state; the workgroups
// exchange their 16 new h values through 8-byte {value, step tag} granules written with one device-scope store each and polled with
// device-scope loads (a granule is published by a single naturally aligned store, so a reader that sees the tag sees the value: no
// fences, no barrier; MI355X_MICROARCH.md "handoff-1to1").  Spins are bounded: a workgroup that is never scheduled next to its peers
// raises *stuck instead of hanging the device.  blockIdx.y = direction (0 forward, 1 reversed), each with its own pre-activations.
// Same arithmetic as kk_lstm_kernel up to the summation order of the 256-term dot products (4 partial sums of CP terms).
struct LstmArgs {
    const float *pre[2];        // [4][L][hid] input pre-activations (x W_ih^T + b_ih) per direction
    const float *whh[2][4];     // [hid][hid] per gate (i, f, g, o)
    const float *bhh[2][4];     // [hid]
    unsigned long long *xch;    // [2 dirs][2 parities][hid] granules, zeroed before the launch
    float *out;                 // [L][out_stride]
    int L, hid, out_stride;
    int *stuck;
};

template <int CP>
__global__ __launch_bounds__(256) void kk_lstm_split_kernel(LstmArgs a) {
    extern __shared__ float kk_h[];   // [hid] previous hidden state
    const int tid = threadIdx.x, dir = blockIdx.y;
    const int u = tid >> 4, g = (tid >> 2) & 3, p = tid & 3;
    const int unit = blockIdx.x * 16 + u, hid = a.hid, L = a.L;
    float w[CP];
    {
        const float *wr = a.whh[dir][g] + (int64_t) unit * hid + p * CP;
#pragma unroll
        for (int j = 0; j < CP; j++) w[j] = wr[j];
    }
    const float bias = a.bhh[dir][g][unit];
    const float *pre = a.pre[dir] + (int64_t) g * L * hid + unit;
    unsigned long long *xch = a.xch + (int64_t) dir * 2 * hid;
    float c = 0.0f;
    for (int idx = 0; idx < L; idx++) {
        const int t = dir ? L - 1 - idx : idx;
        const float pv = pre[(int64_t) t * hid];            // issued before the wait: the input term does not depend on the exchange
        if (idx == 0) {
            for (int j = tid; j < hid; j += 256) kk_h[j] = 0.0f;
        } else {
            const unsigned long long *slot = xch + (int64_t) ((idx - 1) & 1) * hid;
            for (int j = tid; j < hid; j += 256) {
                unsigned long long v = 0;
                int spins = 0;
                for (;;) {
                    v = __hip_atomic_load(slot + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned) (v >> 32) == (unsigned) idx) break;
                    // a peer that is never co-resident: give up once, for the whole launch — every workgroup sees the flag in its own spin
                    // and leaves, instead of spinning out 2^22 polls per granule and time step (minutes) with a garbage state
                    if (++spins > (1 << 22)) { __hip_atomic_store(a.stuck, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    if ((spins & 1023) == 0 && __hip_atomic_load(a.stuck, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                kk_h[j] = __uint_as_float((unsigned) v);
            }
        }
        if (idx > 0 && __syncthreads_or(__hip_atomic_load(a.stuck, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) return;   // fail fast (the host reports kk_stuck)
        __syncthreads();
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < CP; j++) acc += w[j] * kk_h[p * CP + j];
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);                          // the four parts of row (g, unit)
        const float gate = pv + (acc + bias);
        const int base = (tid & 63) & ~15;                  // lane of (u, gate 0, part 0) inside the wave
        const float gi = __shfl(gate, base), gf = __shfl(gate, base + 4), gg_ = __shfl(gate, base + 8), go = __shfl(gate, base + 12);
        if ((tid & 15) == 0) {
            const float ig = 1.0f / (1.0f + expf(-gi)), fg = 1.0f / (1.0f + expf(-gf)), gg = tanhf(gg_), og = 1.0f / (1.0f + expf(-go));
            c = fg * c + ig * gg;
            const float h = tanhf(c) * og;
            a.out[(int64_t) t * a.out_stride + dir * hid + unit] = h;
            if (idx + 1 < L)
                __hip_atomic_store(xch + (int64_t) (idx & 1) * hid + unit, ((unsigned long long) (unsigned) (idx + 1) << 32) | __float_as_uint(h), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();   // kk_h is rewritten by the next step
    }
}

static __global__ __launch_bounds__(1024) void kk_lstm_kernel(const float *pre, const float *whh0, const float *whh1, const float *whh2, const float *whh3, const float *bhh0,
                                                       const float *bhh1, const float *bhh2, const float *bhh3, int L, int hid, int reversed, float *out, int out_stride,
                                                       int out_off) {
    extern __shared__ float kk_h[];
    const int e = threadIdx.x;
    float c = 0.0f;
    if (e < hid) kk_h[e] = 0.0f;
    __syncthreads();
    const float *whh[4] = {whh0, whh1, whh2, whh3};
    const float *bhh[4] = {bhh0, bhh1, bhh2, bhh3};
    for (int idx = 0; idx < L; idx++) {
        const int t = reversed ? L - 1 - idx : idx;
        float g4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (e < hid) {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const float *w = whh[g] + (int64_t) e * hid;
                float a = 0.0f;
                for (int j = 0; j < hid; j++) a += w[j] * kk_h[j];
                g4[g] = pre[((int64_t) g * L + t) * hid + e] + (a + bhh[g][e]);
            }
        }
        __syncthreads();   // every thread has read h
        if (e < hid) {
            const float ig = 1.0f / (1.0f + expf(-g4[0])), fg = 1.0f / (1.0f + expf(-g4[1])), gg = tanhf(g4[2]), og = 1.0f / (1.0f + expf(-g4[3]));
            c = fg * c + ig * gg;
            const float h = tanhf(c) * og;
            kk_h[e] = h;
            out[(int64_t) t * out_stride + out_off + e] = h;
        }
        __syncthreads();
    }
}

// [R][C] <-> [C][R]
static __global__ void kk_transpose_kernel(const float *x, int R, int C, float *y) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t) R * C) return;
    const int r = (int) (i / C), c = (int) (i - (int64_t) r * C);
    y[(int64_t) c * R + r] = x[i];
}

// rows gathered by index: y[t][:] = x[idx[t]][:]  (the duration mask of set_inputs :1262-1271 as an index)
static __global__ void kk_gather_rows_kernel(const float *x, const int *idx, int T, int W, float *y) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t) T * W) return;
    const int t = (int) (i / W), w = (int) (i - (int64_t) t * W);
    y[i] = x[(int64_t) idx[t] * W + w];
}
// the same onto channel-major output: y[c][t] = x[idx[t]][c]
static __global__ void kk_gather_cols_kernel(const float *x, const int *idx, int T, int C, float *y) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t) T * C) return;
    const int c = (int) (i / T), t = (int) (i - (int64_t) c * T);
    y[i] = x[(int64_t) idx[t] * C + c];
}

// instance norm over L per channel, then n + n * gamma[c] + beta[c]; act 0 none, 1 leaky relu (slope), 2 snake with alpha[c]
static __global__ __launch_bounds__(256) void kk_adain_kernel(float *x, int64_t L, const float *gamma, const float *beta, int act, float slope, const float *alpha) {
    __shared__ float red[4];
    const int c = blockIdx.x, tid = threadIdx.x;
    float *xr = x + (int64_t) c * L;
    float s = 0.0f;
    for (int64_t t = tid; t < L; t += 256) s += xr[t];
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float) L;
    __syncthreads();
    float v2 = 0.0f;
    for (int64_t t = tid; t < L; t += 256) { const float d = xr[t] - mean; v2 += d * d; }
    v2 = wave_sum(v2);
    if ((tid & 63) == 0) red[tid >> 6] = v2;
    __syncthreads();
    const float scale = 1.0f / sqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float) L + 1e-5f);
    const float g = gamma[c], b = beta[c];
    const float al = act == 2 ? alpha[c] : 1.0f;
    for (int64_t t = tid; t < L; t += 256) {
        const float n = (xr[t] - mean) * scale;
        float o = (n + n * g) + b;
        if (act == 1) o = o > 0.0f ? o : o * slope;
        else if (act == 2) { const float sn = sinf(o * al); o = o + (sn * sn) * (1.0f / al); }
        xr[t] = o;
    }
}

// The same instance norm for long rows (the generator's AdaIN blocks: 128 channels x 144 721 positions): one workgroup per channel leaves half
// the chip idle and walks 580 KB three times.  The row is cut into gridDim.y slices; phase 0 writes slice sums, phase 1 the slice sums of
// (x - mean)^2 with the mean folded from phase 0 in slice order, phase 2 folds both and applies — the two-pass statistics of kk_adain_kernel,
// summed slice by slice.  part: [2][C][S].
static __global__ __launch_bounds__(256) void kk_adain_split_kernel(float *x, int64_t L, const float *gamma, const float *beta, int act, float slope, const float *alpha, float *part,
                                                             int phase) {
    __shared__ float red[4];
    const int c = blockIdx.x, sl = blockIdx.y, S = gridDim.y, C = gridDim.x, tid = threadIdx.x;
    const int64_t chunk = ((L + S - 1) / S + 3) & ~(int64_t) 3;
    const int64_t t_lo = (int64_t) sl * chunk, t_hi = t_lo + chunk < L ? t_lo + chunk : L;
    float *xr = x + (int64_t) c * L;
    const float *psum = part + (int64_t) c * S, *pvar = part + (int64_t) C * S + (int64_t) c * S;
    float mean = 0.0f;
    if (phase >= 1) {
        float tot = 0.0f;
        for (int i = 0; i < S; i++) tot += psum[i];
        mean = tot / (float) L;
    }
    if (phase <= 1) {
        float s = 0.0f;
        if (phase == 0) for (int64_t t = t_lo + tid; t < t_hi; t += 256) s += xr[t];
        else for (int64_t t = t_lo + tid; t < t_hi; t += 256) { const float d = xr[t] - mean; s += d * d; }
        s = wave_sum(s);
        if ((tid & 63) == 0) red[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) part[(int64_t) phase * C * S + (int64_t) c * S + sl] = (red[0] + red[1]) + (red[2] + red[3]);
        return;
    }
    float var = 0.0f;
    for (int i = 0; i < S; i++) var += pvar[i];
    const float scale = 1.0f / sqrtf(var / (float) L + 1e-5f);
    const float g = gamma[c], b = beta[c];
    const float al = act == 2 ? alpha[c] : 1.0f;
    for (int64_t t = t_lo + tid; t < t_hi; t += 256) {
        const float n = (xr[t] - mean) * scale;
        float o = (n + n * g) + b;
        if (act == 1) o = o > 0.0f ? o : o * slope;
        else if (act == 2) { const float sn = sinf(o * al); o = o + (sn * sn) * (1.0f / al); }
        xr[t] = o;
    }
}

static __global__ void kk_leaky_kernel(float *x, int64_t n, float slope) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float v = x[i]; x[i] = v > 0.0f ? v : v * slope; }
}

// per-position layer norm over the channels of a [C][L] tensor (text encoder :1201-1203) + leaky relu
static __global__ __launch_bounds__(64) void kk_chan_norm_kernel(float *x, int C, int64_t L, const float *g, const float *b, float slope) {
    const int64_t t = blockIdx.x;
    const int lane = threadIdx.x;
    float s = 0.0f;
    for (int c = lane; c < C; c += 64) s += x[(int64_t) c * L + t];
    const float mean = wave_sum(s) / (float) C;
    float v2 = 0.0f;
    for (int c = lane; c < C; c += 64) { const float d = x[(int64_t) c * L + t] - mean; v2 += d * d; }
    const float scale = 1.0f / sqrtf(wave_sum(v2) / (float) C + 1e-5f);
    for (int c = lane; c < C; c += 64) {
        const float o = (x[(int64_t) c * L + t] - mean) * scale * g[c] + b[c];
        x[(int64_t) c * L + t] = o > 0.0f ? o : o * slope;
    }
}

// y[co][t] (+)= b[co] + sum_ci sum_k w[co][ci][k] * x[ci][(t * stride - pad + k * dil) >> in_shift]
// in_shift 1: the input is read through a nearest-neighbour 2x upsample (the pooled shortcut of build_ada_residual_conv :126-128)
// post_scale: the result (after the optional accumulate) is multiplied by it ((res + shortcut) / sqrt 2 :132, / n_kernels :229)
static __global__ void kk_conv1d_kernel(const float *x, int cin, int64_t L, const float *w, const float *b, int cout, int K, int stride, int pad, int dil, int in_shift,
                                 float *y, int64_t Lout, int accumulate, float post_scale) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t) cout * Lout) return;
    const int co = (int) (i / Lout);
    const int64_t t = i - (int64_t) co * Lout;
    const int64_t Lin = in_shift ? L * 2 : L;
    float acc = b ? b[co] : 0.0f;
    for (int ci = 0; ci < cin; ci++) {
        const float *xr = x + (int64_t) ci * L, *wr = w + ((int64_t) co * cin + ci) * K;
        for (int k = 0; k < K; k++) {
            const int64_t s = t * stride - pad + (int64_t) k * dil;
            if (s >= 0 && s < Lin) acc += wr[k] * xr[in_shift ? s >> 1 : s];
        }
    }
    float v = accumulate ? y[i] + acc : acc;
    y[i] = v * post_scale;
}

// The k = 1 convolutions kk_conv1d_kernel was left with (the shortcuts of the AdaIN residual blocks: 1090 -> 1024 channels, optionally through
// the nearest-neighbour 2x input, accumulated into the block's output and scaled by 1/sqrt 2; row lengths that are no multiple of 4) as a
// 64-channel x 64-position tile GEMM on the exact-fp32 matrix pipe: y[co][t] = ((y[co][t] +) b[co] + sum_ci w[co][ci] x[ci][t >> in_shift]) * post.
// A = w (transposed into LDS as [ci][co]), B = x rows as they lie ([ci][t]); any cin, scalar loads (nothing here is 16-byte aligned).
static __global__ __launch_bounds__(256) void kk_conv1x1_mfma_kernel(const float *x, int cin, int64_t L, const float *w, const float *b, int cout, int in_shift, float *y, int64_t Lout,
                                                              int accumulate, float post_scale) {
    typedef float f16v __attribute__((ext_vector_type(16)));
    constexpr int KC = 16, LD = 68;
    __shared__ float ws[2][KC][LD], xs[2][KC][LD];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5, wm = wv >> 1, wn = wv & 1;
    const int co0 = blockIdx.y * 64;
    const int64_t t0 = (int64_t) blockIdx.x * 64;
    const int wi = tid >> 2, wk = (tid & 3) * 4;      // weights: channel wi, slice columns wk..wk+3
    const int xk = tid >> 4, xt = (tid & 15) * 4;     // input: slice row xk, positions xt..xt+3
    const bool wok = co0 + wi < cout;
    float wa[4], xa[4];
    auto fetch = [&](int c) {
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int ci = c * KC + wk + e;
            wa[e] = (wok && ci < cin) ? w[(int64_t) (co0 + wi) * cin + ci] : 0.0f;
        }
        const int cix = c * KC + xk;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int64_t t = t0 + xt + e;
            xa[e] = (cix < cin && t < Lout) ? x[(int64_t) cix * L + (in_shift ? t >> 1 : t)] : 0.0f;
        }
    };
    fetch(0);
    f16v acc;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.0f;
    const int nc = (cin + KC - 1) / KC;
    for (int c = 0; c < nc; c++) {
        const int buf = c & 1;
#pragma unroll
        for (int e = 0; e < 4; e++) { ws[buf][wk + e][wi] = wa[e]; xs[buf][xk][xt + e] = xa[e]; }
        if (c + 1 < nc) fetch(c + 1);
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KC; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ws[buf][kk + hi][wm * 32 + l31], xs[buf][kk + hi][wn * 32 + l31], acc, 0, 0, 0);
    }
    const int64_t t = t0 + wn * 32 + l31;
    if (t >= Lout) return;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const int co = co0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
        if (co >= cout) continue;
        float *p = y + (int64_t) co * Lout + t;
        float v = acc[e] + (b ? b[co] : 0.0f);
        if (accumulate) v = *p + v;
        *p = v * post_scale;
    }
}

// dense ConvTranspose1d, weight [Cin][Cout][K]: y[co][to] = b[co] + sum over (ci, ti, k) with ti * stride + k - pad == to
static __global__ void kk_convt1d_kernel(const float *x, int cin, int64_t L, const float *w, const float *b, int cout, int K, int stride, int pad, float *y, int64_t Lout) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t) cout * Lout) return;
    const int co = (int) (i / Lout);
    const int64_t to = i - (int64_t) co * Lout;
    float acc = b ? b[co] : 0.0f;
    for (int k = 0; k < K; k++) {
        const int64_t num = to + pad - k;
        if (num < 0 || num % stride) continue;
        const int64_t ti = num / stride;
        if (ti >= L) continue;
        for (int ci = 0; ci < cin; ci++) acc += x[(int64_t) ci * L + ti] * w[((int64_t) ci * cout + co) * K + k];
    }
    y[i] = acc;
}

// depthwise ConvTranspose1d(k 3, stride 2, padding 1, output_padding 1): y[c][o] = b[c] + sum_k x[c][t] w[c][k] with 2 t + k - 1 == o; Lout = 2 L
static __global__ void kk_pool_convt_kernel(const float *x, int C, int64_t L, const float *w, const float *b, float *y) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t) C * 2 * L) return;
    const int c = (int) (i / (2 * L));
    const int64_t o = i - (int64_t) c * 2 * L;
    float acc = b[c];
    for (int k = 0; k < 3; k++) {
        const int64_t num = o + 1 - k;
        if (num < 0 || (num & 1)) continue;
        const int64_t t = num >> 1;
        if (t < L) acc += x[(int64_t) c * L + t] * w[c * 3 + k];
    }
    y[i] = acc;
}

// reflection pad of one sample in front of every channel (:215-220): y[c][0] = x[c][1], y[c][1 + t] = x[c][t]
static __global__ void kk_pad_front_kernel(const float *x, int C, int64_t L, float *y) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t) C * (L + 1)) return;
    const int c = (int) (i / (L + 1));
    const int64_t t = i - (int64_t) c * (L + 1);
    y[i] = x[(int64_t) c * L + (t == 0 ? 1 : t - 1)];
}

// harmonic phases at the frame rate: phase[h][l] = cumsum_l(frac(f0[l] * (h + 1) / sr)) * (upsample_scale * 2 pi); one thread per harmonic
static __global__ void kk_sine_phase_kernel(const float *f0, int64_t L2, int NH, float sample_rate, float factor, float *phase) {
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= NH) return;
    float run = 0.0f;
    for (int64_t l = 0; l < L2; l++) {
        float v = f0[l] * (((float) h + 1.0f) / sample_rate);
        v = v - floorf(v);
        run += v;
        phase[(int64_t) h * L2 + l] = run * factor;
    }
}
// sine[h][j] = sin(linear_upscale(phase)[j]) * uv + noise scale * rand[h][j]   (uv_noise_compute, util.cpp:143-173)
static __global__ void kk_sine_source_kernel(const float *phase, const float *f0, int64_t L2, int NH, int up, float threshold, float sin_amp, float noise_std, const float *noise,
                                      float *sine) {
    const int64_t LS = L2 * up;
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t) NH * LS) return;
    const int h = (int) (i / LS);
    const int64_t j = i - (int64_t) h * LS;
    double src = ((double) j + 0.5) / (double) up - 0.5;   // F.interpolate(linear, align_corners = False)
    if (src < 0) src = 0;
    int64_t i0 = (int64_t) src;
    if (i0 > L2 - 1) i0 = L2 - 1;
    const int64_t i1 = i0 + 1 < L2 ? i0 + 1 : L2 - 1;
    const float fr = (float) (src - (double) i0);
    const float ph = (1.0f - fr) * phase[(int64_t) h * L2 + i0] + fr * phase[(int64_t) h * L2 + i1];
    const bool voiced = f0[j / up] > threshold;
    sine[i] = sinf(ph) * (voiced ? sin_amp : 0.0f) + (voiced ? noise_std : sin_amp / 3.0f) * noise[i];
}
// har[j] = tanh(sum_h mw[h] sine[h][j] + mb)
static __global__ void kk_source_merge_kernel(const float *sine, int NH, int64_t LS, const float *mw, const float *mb, float *har) {
    const int64_t j = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= LS) return;
    float acc = 0.0f;
    for (int h = 0; h < NH; h++) acc += mw[h] * sine[(int64_t) h * LS + j];
    har[j] = tanhf(acc + mb[0]);
}

// torch.stft(center, reflect, onesided): out [2 nb][F] = magnitudes then phases; one thread per (bin, frame)
static __global__ void kk_stft_kernel(const float *x, int64_t L, const float *win, int N, int hop, int64_t F, float *out) {
    const int nb = N / 2 + 1;
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t) nb * F) return;
    const int kb = (int) (i / F);
    const int64_t f = i - (int64_t) kb * F;
    float re = 0.0f, im = 0.0f;
    for (int n = 0; n < N; n++) {
        int64_t s = f * hop + n - N / 2;
        if (s < 0) s = -s;
        if (s >= L) s = 2 * (L - 1) - s;
        const float a = -2.0f * 3.14159265358979323846f * (float) ((kb * n) % N) / (float) N;
        const float xv = win[n] * x[s];
        re += xv * cosf(a);
        im += xv * sinf(a);
    }
    out[i] = sqrtf(re * re + im * im);
    out[(int64_t) nb * F + i] = atan2f(im, re);
}

// post [2 nb][F]: rows < nb -> exp (magnitude), rows >= nb -> sin (phase) (:234-238), in place
static __global__ void kk_spec_phase_kernel(float *post, int nb, int64_t F) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t) 2 * nb * F) return;
    post[i] = i < (int64_t) nb * F ? expf(post[i]) : sinf(post[i]);
}

// inverse STFT from magnitude / phase, overlap-add with the window, trimmed by N / 2, divided by the reference's window envelope
// (compute_window_squared_sum, util.cpp:203-217 for out_len / hop frames); one thread per output sample
static __global__ void kk_istft_kernel(const float *post, int64_t F, const float *win, int N, int hop, float *out, int64_t out_len) {
    const int nb = N / 2 + 1, half = N / 2;
    const int64_t o = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= out_len) return;
    float acc = 0.0f, env = 0.0f;
    const int64_t env_frames = out_len / hop + half / hop;
    // frames f with 0 <= o + half - f * hop < N
    int64_t f_hi = (o + half) / hop;
    int64_t f_lo = (o + half - (N - 1) + hop - 1) / hop;
    if (o + half - (N - 1) < 0) f_lo = 0;
    for (int64_t f = f_lo; f <= f_hi; f++) {
        const int n = (int) (o + half - f * hop);
        if (n < 0 || n >= N) continue;
        if (f < env_frames) env += win[n] * win[n];
        if (f >= F) continue;
        float v = 0.0f;
        for (int kb = 0; kb < nb; kb++) {
            const float mag = post[(int64_t) kb * F + f], ph = post[(int64_t) (nb + kb) * F + f];
            const float a = 2.0f * 3.14159265358979323846f * (float) ((kb * n) % N) / (float) N;
            const float term = mag * cosf(ph) * cosf(a) - mag * sinf(ph) * sinf(a);
            v += (kb == 0 || kb == N / 2) ? term : 2.0f * term;
        }
        acc += v / (float) N * win[n];
    }
    out[o] = acc / env;
}

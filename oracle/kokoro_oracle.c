/* kokoro_oracle.c — CPU restatement of the Kokoro path (TEST INFRASTRUCTURE ONLY, like tts_oracle.c).
 *
 * Follows /root/reference/src/models/kokoro/model.cpp: the duration graph (build_kokoro_duration_graph :938-1047: ALBERT with one
 * shared layer, the prosody predictor's LSTM + AdaLayerNorm stack, the duration head) and the generation graph
 * (build_kokoro_graph :1141-1242: alignment by the duration mask, shared LSTM, F0 / N branches of AdaIN residual blocks, text
 * encoder, decoder blocks, and build_generator :195-244: harmonic source, STFT conditioning, two transposed-conv stages with
 * AdaIN + snake residual blocks, iSTFT head).
 *
 * PARITY UNPINNED.  The reference's arithmetic for this model lives in ops of its absent ggml fork (ggml_stft / ggml_istft,
 * ggml_upscale_linear, ggml_mod, ggml_cumsum, ggml_conv_transpose_1d with groups / output padding, ggml_round); nothing in
 * /root/reference pins their numerics.  The interpretations used here are stated where they apply and are the PyTorch
 * semantics of the module the reference converts from (hexgrad/Kokoro-82M, StyleTTS2 iSTFTNet): torch.stft / istft with
 * center = True and reflect padding, F.interpolate(mode = "linear", align_corners = False), ConvTranspose1d(groups = C,
 * output_padding = 1), nearest-neighbour 2x upsampling, roundf.  tests/golden/tiny_kokoro.npz (float64 torch) pins this file to
 * those definitions, not to ggml.  The one part with an upstream implementation installed here is pinned to it: the ALBERT stage
 * against transformers' AlbertModel (the class kokoro's `bert` is and the converter walks, kokoro_gguf_encoder.py:14-37, :274-287):
 * 1.7e-7 on a 19-token input, tests/golden/upstream_albert.npz, tests/test_upstream_golden.py.
 * Round 5: the stages (bidirectional LSTM, the upsampling AdaIN residual block with its depthwise transposed-conv pool, stft / istft) are also held
 * against ONE PyTorch module each through the orc_kk_stage_* entry points at the end of this file (tests/golden/upstream_kokoro_stages.npz, 2e-5);
 * what stays author-only is the order in which the graph strings them together.
 *
 * Tensors are looked up by their GGUF names (py-gguf/tts_encoders/kokoro_gguf_encoder.py), all fp32. */
#define _GNU_SOURCE
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tts_oracle.h"

typedef struct {
    int32_t n_tensors;
    const char *const *names;
    const float *const *data;
    const int64_t *ne;        /* [n_tensors][4], ne[0] fastest */
    /* kokoro_model defaults (model.h:180-222) that the GGUF may override */
    int32_t n_heads, n_recurrence, n_dp_layers, f0_n_blocks, n_conv_layers, n_decoder_blocks, n_upsamples, n_kernels;
    int32_t n_fft, hop, harmonic_num, up_sampling_factor, out_conv_padding;
    float   attn_scale, upsample_scale, sample_rate, sin_amp, noise_std, voice_threshold;
    /* generator geometry (kokoro.decoder.generator.{up_convs,noise_blocks,res_blocks}.*) */
    int32_t up_stride[4], up_padding[4], noise_stride[4], noise_padding[4];
    int32_t res_padding[16][3], res_dilation[16][3], noise_res_padding[4][3], noise_res_dilation[4][3];
    /* ALBERT's ggml_gelu (model.cpp:1000): 1 = ggml's CPU path, tanh-GELU through its fp16-indexed table (round x to fp16, result
     * rounded to fp16; orc_gelu in tts_oracle.c); 0 = fp32 tanh-GELU, for the comparison with the float64 torch fixture */
    int32_t gelu_mode;
} orc_kokoro_model;

static const float *kt(const orc_kokoro_model *m, const char *name, int64_t *ne) {
    for (int i = 0; i < m->n_tensors; i++)
        if (!strcmp(m->names[i], name)) {
            if (ne) memcpy(ne, m->ne + (size_t) i * 4, 4 * sizeof(int64_t));
            return m->data[i];
        }
    fprintf(stderr, "kokoro oracle: missing tensor '%s'\n", name);
    abort();
}
static const float *ktf(const orc_kokoro_model *m, int64_t *ne, const char *fmt, ...) {
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return kt(m, buf, ne);
}

/* ORC_KOKORO_DUMP=<dir>: raw fp32 dumps of named intermediates (debugging aid for the golden comparison) */
static void kdump(const char *name, const float *p, size_t n) {
    const char *dir = getenv("ORC_KOKORO_DUMP");
    if (!dir) return;
    char path[512];
    snprintf(path, sizeof(path), "%s/%s.bin", dir, name);
    FILE *f = fopen(path, "wb");
    if (!f) return;
    fwrite(p, 4, n, f);
    fclose(f);
}

static float *falloc(size_t n) { return (float *) calloc(n ? n : 1, sizeof(float)); }

/* y[N] = W[N][K] x + b */
static void lin(const float *W, const float *b, const float *x, int K, int N, float *y) {
    for (int n = 0; n < N; n++) {
        double acc = 0.0;
        const float *w = W + (size_t) n * K;
        for (int k = 0; k < K; k++) acc += (double) w[k] * (double) x[k];
        y[n] = (float) acc + (b ? b[n] : 0.0f);
    }
}
static void lin_rows(const float *W, const float *b, const float *x, int R, int K, int N, float *y) {
#pragma omp parallel for schedule(static)
    for (int r = 0; r < R; r++) lin(W, b, x + (size_t) r * K, K, N, y + (size_t) r * N);
}

/* ggml_norm over n values (mean / variance in double like ggml's ggml_float), optional affine */
static void norm_vec(const float *x, int n, float eps, const float *w, const float *b, float *y) {
    double mean = 0.0;
    for (int i = 0; i < n; i++) mean += x[i];
    mean /= n;
    double var = 0.0;
    for (int i = 0; i < n; i++) { const double d = x[i] - mean; var += d * d; }
    var /= n;
    const float scale = 1.0f / sqrtf((float) var + eps);
    for (int i = 0; i < n; i++) {
        const float v = (float) (x[i] - mean) * scale;
        y[i] = w ? v * w[i] + (b ? b[i] : 0.0f) : v;
    }
}

static float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/* build_lstm_run (:53-86): gates i, f, g, o; weights[2j] on the input, weights[2j+1] on the hidden state; h0 = c0 = 0 */
static void lstm_dir(const orc_kokoro_model *m, const char *base, const char *wname, const char *bname, const float *x, int L, int in, int hid, int reversed,
                     float *out, int out_stride, int out_off) {
    const float *W[8], *B[8];
    for (int j = 0; j < 8; j++) {
        W[j] = ktf(m, NULL, "%s.0.%s.%d", base, wname, j);
        B[j] = ktf(m, NULL, "%s.0.%s.%d", base, bname, j);
    }
    float *pre = falloc((size_t) 4 * L * hid);   /* input part of the four gates for every position */
    for (int g = 0; g < 4; g++) lin_rows(W[2 * g], B[2 * g], x, L, in, hid, pre + (size_t) g * L * hid);
    float *h = falloc(hid), *c = falloc(hid), *rec = falloc((size_t) 4 * hid);
    for (int idx = 0; idx < L; idx++) {
        const int t = reversed ? L - 1 - idx : idx;
        for (int g = 0; g < 4; g++) lin(W[2 * g + 1], B[2 * g + 1], h, hid, hid, rec + (size_t) g * hid);
        for (int e = 0; e < hid; e++) {
            const float ig = sigmoidf_(pre[((size_t) 0 * L + t) * hid + e] + rec[e]);
            const float fg = sigmoidf_(pre[((size_t) 1 * L + t) * hid + e] + rec[hid + e]);
            const float gg = tanhf(pre[((size_t) 2 * L + t) * hid + e] + rec[2 * hid + e]);
            const float og = sigmoidf_(pre[((size_t) 3 * L + t) * hid + e] + rec[3 * hid + e]);
            c[e] = fg * c[e] + ig * gg;
            h[e] = tanhf(c[e]) * og;
        }
        memcpy(out + (size_t) t * out_stride + out_off, h, (size_t) hid * 4);
    }
    free(pre); free(h); free(c); free(rec);
}
/* build_lstm (:35-51), one bidirectional cell: forward | reverse concatenated along the features */
static void bilstm(const orc_kokoro_model *m, const char *base, const float *x, int L, int in, int hid, float *out /* [L][2*hid] */) {
    lstm_dir(m, base, "weights", "biases", x, L, in, hid, 0, out, 2 * hid, 0);
    lstm_dir(m, base, "reverse_weights", "reverse_biases", x, L, in, hid, 1, out, 2 * hid, hid);
}

/* conv1d with stride on [C][L] activations, torch weight order [Cout][Cin][K] */
static int64_t conv1d_s(const float *x, int cin, int64_t L, const float *w, const float *b, int cout, int K, int stride, int pad, int dil, float **y_out) {
    const int64_t Lout = (L + 2 * pad - (int64_t) dil * (K - 1) - 1) / stride + 1;
    float *y = falloc((size_t) cout * Lout);
#pragma omp parallel for schedule(static)
    for (int co = 0; co < cout; co++) {
        float *yr = y + (size_t) co * Lout;
        for (int64_t t = 0; t < Lout; t++) {
            double acc = b ? b[co] : 0.0;
            for (int ci = 0; ci < cin; ci++) {
                const float *xr = x + (size_t) ci * L, *wr = w + ((size_t) co * cin + ci) * K;
                for (int k = 0; k < K; k++) {
                    const int64_t s = t * stride - pad + (int64_t) k * dil;
                    if (s >= 0 && s < L) acc += (double) wr[k] * (double) xr[s];
                }
            }
            yr[t] = (float) acc;
        }
    }
    *y_out = y;
    return Lout;
}

/* instance norm over L per channel (ggml_norm on [L, C]) followed by x + x * gamma + beta with gamma / beta = W style + b (:93-101) */
static void adain(const orc_kokoro_model *m, float *x, int C, int64_t L, const float *style, int S, const char *gw, const char *gb, const char *bw, const char *bb) {
    float *gamma = falloc(C), *beta = falloc(C);
    lin(kt(m, gw, NULL), kt(m, gb, NULL), style, S, C, gamma);
    lin(kt(m, bw, NULL), kt(m, bb, NULL), style, S, C, beta);
    for (int c = 0; c < C; c++) {
        float *xr = x + (size_t) c * L;
        norm_vec(xr, (int) L, 1e-5f, NULL, NULL, xr);
        for (int64_t t = 0; t < L; t++) xr[t] = xr[t] + xr[t] * gamma[c] + beta[c];
    }
    free(gamma); free(beta);
}
static void leaky(float *x, size_t n, float slope) {
    for (size_t i = 0; i < n; i++) x[i] = x[i] > 0.0f ? x[i] : x[i] * slope;
}

/* build_ada_residual_conv (:88-134).  x [Cin][L] -> returns [Cout][L or 2L] */
static float *ada_res_block(const orc_kokoro_model *m, const char *base, const float *x, int64_t L, const float *style, int S, int *C_io, int64_t *L_out) {
    char n1[256], n2[256], n3[256], n4[256];
    int64_t ne[4];
    snprintf(n1, sizeof(n1), "%s.conv1_weight", base);
    const float *conv1 = kt(m, n1, ne);
    const int cin = (int) ne[1], cout = (int) ne[2];
    if (cin != *C_io) { fprintf(stderr, "kokoro oracle: %s expects %d channels, got %d\n", base, cin, *C_io); abort(); }
    float *cur = falloc((size_t) cin * L);
    memcpy(cur, x, (size_t) cin * L * 4);
    snprintf(n1, sizeof(n1), "%s.norm1_gamma_weight", base); snprintf(n2, sizeof(n2), "%s.norm1_gamma_bias", base);
    snprintf(n3, sizeof(n3), "%s.norm1_beta_weight", base); snprintf(n4, sizeof(n4), "%s.norm1_beta_bias", base);
    adain(m, cur, cin, L, style, S, n1, n2, n3, n4);
    leaky(cur, (size_t) cin * L, 0.2f);
    int64_t Lc = L;
    int has_pool = 0;
    snprintf(n1, sizeof(n1), "%s.pool_weight", base);
    for (int i = 0; i < m->n_tensors; i++) has_pool |= !strcmp(m->names[i], n1);
    if (has_pool) {
        /* ggml_conv_transpose_1d(pool, cur, 2, 1, 1, 1, C): depthwise ConvTranspose1d(k 3, stride 2, padding 1, output_padding 1) */
        snprintf(n2, sizeof(n2), "%s.pool_bias", base);
        const float *pw = kt(m, n1, NULL), *pb = kt(m, n2, NULL);
        float *up = falloc((size_t) cin * 2 * L);
        for (int c = 0; c < cin; c++) {
            float *ur = up + (size_t) c * 2 * L;
            for (int64_t t = 0; t < 2 * L; t++) ur[t] = pb[c];
            for (int64_t t = 0; t < L; t++)
                for (int k = 0; k < 3; k++) {
                    const int64_t o = 2 * t + k - 1;
                    if (o >= 0 && o < 2 * L) ur[o] += cur[(size_t) c * L + t] * pw[(size_t) c * 3 + k];
                }
        }
        free(cur);
        cur = up;
        Lc = 2 * L;
    }
    snprintf(n1, sizeof(n1), "%s.conv1_bias", base);
    float *y;
    conv1d_s(cur, cin, Lc, conv1, kt(m, n1, NULL), cout, 3, 1, 1, 1, &y);
    free(cur);
    cur = y;
    snprintf(n1, sizeof(n1), "%s.norm2_gamma_weight", base); snprintf(n2, sizeof(n2), "%s.norm2_gamma_bias", base);
    snprintf(n3, sizeof(n3), "%s.norm2_beta_weight", base); snprintf(n4, sizeof(n4), "%s.norm2_beta_bias", base);
    adain(m, cur, cout, Lc, style, S, n1, n2, n3, n4);
    leaky(cur, (size_t) cout * Lc, 0.2f);
    snprintf(n1, sizeof(n1), "%s.conv2_weight", base); snprintf(n2, sizeof(n2), "%s.conv2_bias", base);
    conv1d_s(cur, cout, Lc, kt(m, n1, NULL), kt(m, n2, NULL), cout, 3, 1, 1, 1, &y);
    free(cur);
    float *res = y;
    /* shortcut (:123-131): conv1x1 (no bias is applied) after a nearest-neighbour 2x upsample when the block pools */
    float *sc = falloc((size_t) cout * Lc);
    int has_1x1 = 0;
    snprintf(n1, sizeof(n1), "%s.conv1x1_weight", base);
    for (int i = 0; i < m->n_tensors; i++) has_1x1 |= !strcmp(m->names[i], n1);
    if (has_1x1) {
        const float *w = kt(m, n1, NULL);   /* [Cout][Cin][1] */
#pragma omp parallel for schedule(static)
        for (int co = 0; co < cout; co++)
            for (int64_t t = 0; t < Lc; t++) {
                const int64_t ts = has_pool ? t / 2 : t;
                double acc = 0.0;
                for (int ci = 0; ci < cin; ci++) acc += (double) w[(size_t) co * cin + ci] * (double) x[(size_t) ci * L + ts];
                sc[(size_t) co * Lc + t] = (float) acc;
            }
    } else {
        memcpy(sc, x, (size_t) cin * L * 4);
    }
    const float sq2 = sqrtf(2.0f);
    for (size_t i = 0; i < (size_t) cout * Lc; i++) res[i] = (res[i] + sc[i]) / sq2;
    free(sc);
    *C_io = cout;
    *L_out = Lc;
    return res;
}

/* build_kokoro_generator_res_block (:136-165) on [C][L], in place */
static void gen_res_block(const orc_kokoro_model *m, const char *base, float *x, int C, int64_t L, const float *style, int S, const int32_t *pads, const int32_t *dils) {
    char a[256], b[256], c_[256], d[256];
    for (int i = 0; i < 3; i++) {
        float *cur = falloc((size_t) C * L);
        memcpy(cur, x, (size_t) C * L * 4);
        snprintf(a, sizeof(a), "%s.%d.gamma1_weight", base, i); snprintf(b, sizeof(b), "%s.%d.gamma1_bias", base, i);
        snprintf(c_, sizeof(c_), "%s.%d.beta1_weight", base, i); snprintf(d, sizeof(d), "%s.%d.beta1_bias", base, i);
        adain(m, cur, C, L, style, S, a, b, c_, d);
        snprintf(a, sizeof(a), "%s.%d.alpha1", base, i);
        orc_snake(cur, C, L, kt(m, a, NULL));
        int64_t ne[4];
        snprintf(a, sizeof(a), "%s.%d.convs1_weight", base, i); snprintf(b, sizeof(b), "%s.%d.convs1_bias", base, i);
        const float *w1 = kt(m, a, ne);
        float *y;
        conv1d_s(cur, C, L, w1, kt(m, b, NULL), C, (int) ne[0], 1, pads[i], dils[i], &y);
        free(cur);
        cur = y;
        snprintf(a, sizeof(a), "%s.%d.gamma2_weight", base, i); snprintf(b, sizeof(b), "%s.%d.gamma2_bias", base, i);
        snprintf(c_, sizeof(c_), "%s.%d.beta2_weight", base, i); snprintf(d, sizeof(d), "%s.%d.beta2_bias", base, i);
        adain(m, cur, C, L, style, S, a, b, c_, d);
        snprintf(a, sizeof(a), "%s.%d.alpha2", base, i);
        orc_snake(cur, C, L, kt(m, a, NULL));
        snprintf(a, sizeof(a), "%s.%d.convs2_weight", base, i); snprintf(b, sizeof(b), "%s.%d.convs2_bias", base, i);
        const float *w2 = kt(m, a, ne);
        conv1d_s(cur, C, L, w2, kt(m, b, NULL), C, (int) ne[0], 1, pads[0], 1, &y);   /* padding of the FIRST conv, dilation 1 (:160) */
        free(cur);
        for (size_t j = 0; j < (size_t) C * L; j++) x[j] += y[j];
        free(y);
    }
}

/* ---- duration graph ------------------------------------------------------------------------------------------------------ */
/* tokens [n] (bos ... eos); voice [rows][2S]; lens_out [n]; hidden_out [n][D+S] */
void orc_kokoro_durations(const orc_kokoro_model *m, const uint32_t *tokens, int n, const float *voice, float *lens_out, float *hidden_out) {
    int64_t ne[4];
    const float *tok_embd = kt(m, "kokoro.albert.token_embd", ne);
    const int E = (int) ne[0];
    const float *pos_embd = kt(m, "kokoro.albert.position_embd", NULL), *type_embd = kt(m, "kokoro.albert.token_type_embd", NULL);
    const float *embd = kt(m, "kokoro.albert.embd", ne);
    const int H = (int) ne[1];
    const int NH = m->n_heads, hs = H / NH;
    float *x = falloc((size_t) n * H), *tmp = falloc(E > H ? E : H);
    for (int t = 0; t < n; t++) {   /* build_albert_inputs :10-23 */
        for (int e = 0; e < E; e++) tmp[e] = (tok_embd[(size_t) tokens[t] * E + e] + pos_embd[(size_t) t * E + e]) + type_embd[e];
        norm_vec(tmp, E, 1e-12f, kt(m, "kokoro.albert.norm", NULL), kt(m, "kokoro.albert.norm_bias", NULL), tmp);
        lin(embd, kt(m, "kokoro.albert.embd_bias", NULL), tmp, E, H, x + (size_t) t * H);
    }
    const char *L0 = "kokoro.albert.layer.0.";
    char nm[128], nb[128];
#define AL(w, b) do { snprintf(nm, sizeof(nm), "%s%s", L0, w); snprintf(nb, sizeof(nb), "%s%s", L0, b); } while (0)
    AL("ffn", "ffn_bias");
    const float *ffn_w = kt(m, nm, ne);
    const int F = (int) ne[1];
    float *q = falloc((size_t) n * H), *k = falloc((size_t) n * H), *v = falloc((size_t) n * H), *att = falloc((size_t) n * H), *o = falloc((size_t) n * H);
    float *ff = falloc((size_t) n * F), *sc = falloc(n);
    for (int r = 0; r < m->n_recurrence; r++) {   /* :966-1007: the one layer applied n_recurrence times */
        AL("q", "q_bias"); lin_rows(kt(m, nm, NULL), kt(m, nb, NULL), x, n, H, H, q);
        AL("k", "k_bias"); lin_rows(kt(m, nm, NULL), kt(m, nb, NULL), x, n, H, H, k);
        AL("v", "v_bias"); lin_rows(kt(m, nm, NULL), kt(m, nb, NULL), x, n, H, H, v);
        for (int t = 0; t < n; t++)
            for (int h = 0; h < NH; h++) {
                float mx = -INFINITY;
                for (int j = 0; j < n; j++) {
                    double d = 0.0;
                    for (int e = 0; e < hs; e++) d += (double) q[(size_t) t * H + h * hs + e] * (double) k[(size_t) j * H + h * hs + e];
                    sc[j] = (float) d * m->attn_scale;   /* soft_max_ext(kq, mask = 0, scale) :985 */
                    if (sc[j] > mx) mx = sc[j];
                }
                double sum = 0.0;
                for (int j = 0; j < n; j++) { sc[j] = expf(sc[j] - mx); sum += sc[j]; }
                const float inv = (float) (1.0 / sum);
                for (int e = 0; e < hs; e++) {
                    double a = 0.0;
                    for (int j = 0; j < n; j++) a += (double) (sc[j] * inv) * (double) v[(size_t) j * H + h * hs + e];
                    att[(size_t) t * H + h * hs + e] = (float) a;
                }
            }
        AL("o", "o_bias"); lin_rows(kt(m, nm, NULL), kt(m, nb, NULL), att, n, H, H, o);
        AL("ffn_norm", "ffn_norm_bias");   /* attention.LayerNorm (assign_albert_weight :778-783) */
        for (int t = 0; t < n; t++) {
            for (int e = 0; e < H; e++) o[(size_t) t * H + e] += x[(size_t) t * H + e];
            norm_vec(o + (size_t) t * H, H, 1e-12f, kt(m, nm, NULL), kt(m, nb, NULL), x + (size_t) t * H);
        }
        AL("ffn", "ffn_bias"); lin_rows(ffn_w, kt(m, nb, NULL), x, n, H, F, ff);
        for (size_t i = 0; i < (size_t) n * F; i++) ff[i] = orc_gelu(ff[i], m->gelu_mode);
        AL("ffn_out", "ffn_out_bias"); lin_rows(kt(m, nm, NULL), kt(m, nb, NULL), ff, n, F, H, o);
        AL("attn_norm", "attn_norm_bias");   /* full_layer_layer_norm (:742-747) */
        for (int t = 0; t < n; t++) {
            for (int e = 0; e < H; e++) o[(size_t) t * H + e] += x[(size_t) t * H + e];
            norm_vec(o + (size_t) t * H, H, 1e-12f, kt(m, nm, NULL), kt(m, nb, NULL), x + (size_t) t * H);
        }
    }
#undef AL
    kdump("albert", x, (size_t) n * H);   /* ORC_KOKORO_DUMP: the ALBERT output, compared with transformers' AlbertModel in tests/test_upstream_golden.py */
    /* prosody predictor (:1009-1041) */
    const float *enc_w = kt(m, "kokoro.duration_predictor.encode", ne);
    const int D = (int) ne[1];
    kt(m, "kokoro.duration_predictor.layers.1.gamma_weight", ne);
    const int S = (int) ne[0];
    const float *style = voice + (size_t) (n - 3) * 2 * S + S;   /* second half of row n_tokens - 3 (:1012) */
    const int W = D + S;
    float *cur = falloc((size_t) n * W), *ls = falloc((size_t) n * D), *gamma = falloc(D), *beta = falloc(D);
    for (int t = 0; t < n; t++) {
        lin(enc_w, kt(m, "kokoro.duration_predictor.encode_bias", NULL), x + (size_t) t * H, H, D, cur + (size_t) t * W);
        memcpy(cur + (size_t) t * W + D, style, (size_t) S * 4);
    }
    for (int l = 0; l < m->n_dp_layers; l++) {
        char base[128];
        snprintf(base, sizeof(base), "kokoro.duration_predictor.layers.%d.lstm", 2 * l);
        bilstm(m, base, cur, n, W, D / 2, ls);
        lin(ktf(m, NULL, "kokoro.duration_predictor.layers.%d.gamma_weight", 2 * l + 1), ktf(m, NULL, "kokoro.duration_predictor.layers.%d.gamma_bias", 2 * l + 1), style, S, D, gamma);
        lin(ktf(m, NULL, "kokoro.duration_predictor.layers.%d.beta_weight", 2 * l + 1), ktf(m, NULL, "kokoro.duration_predictor.layers.%d.beta_bias", 2 * l + 1), style, S, D, beta);
        for (int t = 0; t < n; t++) {
            norm_vec(ls + (size_t) t * D, D, 1e-5f, NULL, NULL, ls + (size_t) t * D);
            for (int e = 0; e < D; e++) cur[(size_t) t * W + e] = (ls[(size_t) t * D + e] + ls[(size_t) t * D + e] * gamma[e]) + beta[e];
            memcpy(cur + (size_t) t * W + D, style, (size_t) S * 4);
        }
    }
    if (hidden_out) memcpy(hidden_out, cur, (size_t) n * W * 4);
    bilstm(m, "kokoro.duration_predictor.duration_lstm", cur, n, W, D / 2, ls);
    const float *dp = kt(m, "kokoro.duration_predictor.duration_proj", ne);
    const int ND = (int) ne[1];
    float *dur = falloc(ND);
    for (int t = 0; t < n; t++) {
        lin(dp, kt(m, "kokoro.duration_predictor.duration_proj_bias", NULL), ls + (size_t) t * D, D, ND, dur);
        float s = 0.0f;
        for (int e = 0; e < ND; e++) s += sigmoidf_(dur[e]);
        s = roundf(s);
        lens_out[t] = s < 1.0f ? 1.0f : (s > 50.0f ? 50.0f : s);   /* clamp(round(sum), 1, 50) :1037 */
    }
    free(x); free(tmp); free(q); free(k); free(v); free(att); free(o); free(ff); free(sc); free(cur); free(ls); free(gamma); free(beta); free(dur);
}

/* ---- generation graph ---------------------------------------------------------------------------------------------------- */
/* torch.stft(center = True, reflect, onesided) as magnitude / angle: x [L] -> mag, ph [nb][F], F = L / hop + 1 */
static void stft_mag_phase(const float *x, int64_t L, const float *win, int N, int hop, float *mag, float *ph, int64_t F) {
    const int half = N / 2, nb = N / 2 + 1;
    for (int64_t f = 0; f < F; f++)
        for (int kb = 0; kb < nb; kb++) {
            double re = 0.0, im = 0.0;
            for (int i = 0; i < N; i++) {
                int64_t s = f * hop + i - half;
                if (s < 0) s = -s;
                if (s >= L) s = 2 * (L - 1) - s;
                const double a = -2.0 * M_PI * (double) kb * (double) i / (double) N, xv = (double) win[i] * (double) x[s];
                re += xv * cos(a);
                im += xv * sin(a);
            }
            mag[(size_t) kb * F + f] = (float) sqrt(re * re + im * im);
            ph[(size_t) kb * F + f] = (float) atan2(im, re);
        }
}
/* inverse of the above from magnitude / phase, overlap-add with the window, trimmed by N/2, divided by the reference's window
 * envelope (compute_window_squared_sum, util.cpp:203-217, evaluated for out_len / hop frames as set_inputs does :1260) */
static void istft_mag_phase(const float *mag, const float *ph, int64_t F, const float *win, int N, int hop, float *out, int64_t out_len) {
    const int half = N / 2, nb = N / 2 + 1;
    double *acc = (double *) calloc((size_t) out_len, sizeof(double));
    for (int64_t f = 0; f < F; f++)
        for (int i = 0; i < N; i++) {
            const int64_t o = f * hop + i - half;
            if (o < 0 || o >= out_len) continue;
            double v = 0.0;
            for (int kb = 0; kb < nb; kb++) {
                const double a = 2.0 * M_PI * (double) kb * (double) i / (double) N;
                const double re = (double) mag[(size_t) kb * F + f] * cos((double) ph[(size_t) kb * F + f]);
                const double im = (double) mag[(size_t) kb * F + f] * sin((double) ph[(size_t) kb * F + f]);
                const double term = re * cos(a) - im * sin(a);
                v += (kb == 0 || kb == N / 2) ? term : 2.0 * term;   /* Hermitian half */
            }
            acc[o] += v / N * (double) win[i];
        }
    const int64_t n_frames = out_len / hop;
    float *env = falloc((size_t) out_len);
    for (int64_t i = 0; i < n_frames + half / hop; i++)
        for (int ii = 0; ii < N; ii++) {
            const int64_t idx = ii + i * hop - half;
            if (idx < 0 || idx >= out_len) continue;
            env[idx] += powf(win[ii], 2);
        }
    for (int64_t i = 0; i < out_len; i++) out[i] = (float) acc[i] / env[i];
    free(acc); free(env);
}

/* tokens [n], lens [n] (integers as floats), hidden [n][D+S] from the duration graph, voice [rows][2S],
 * noise [(harmonic_num + 1) * total * up_sampling_factor] uniform draws (set_inputs :1255); pcm_out [total * up_sampling_factor].
 * f0_out / n_out (optional) [2 * total].  Returns the number of samples. */
/* hsrc_out / hsrc_in (optional) [2 * (n_fft / 2 + 1)][F]: the STFT conditioning (magnitudes, then phases) of the harmonic source.
 * The phase channels are atan2 values fed straight into convolutions (:204-211): a bin sitting at +-pi flips by 2 pi under
 * rounding-level differences of the source, so two correct implementations can differ from there on; feeding one
 * implementation's conditioning to the other (hsrc_in) makes everything after it comparable sample for sample. */
int64_t orc_kokoro_generate(const orc_kokoro_model *m, const uint32_t *tokens, int n, const float *lens, const float *hidden, const float *voice,
                            const float *noise, float *pcm_out, float *f0_out, float *n_out, float *hsrc_out, const float *hsrc_in) {
    int64_t ne[4];
    kt(m, "kokoro.duration_predictor.encode", ne);
    const int D = (int) ne[1];
    kt(m, "kokoro.duration_predictor.layers.1.gamma_weight", ne);
    const int S = (int) ne[0], W = D + S;
    int64_t T = 0;
    for (int i = 0; i < n; i++) T += (int64_t) lens[i];
    int *tok_of = (int *) malloc((size_t) T * sizeof(int));   /* the duration mask (:1262-1271) as an index */
    {
        int64_t t = 0;
        for (int i = 0; i < n; i++)
            for (int r = 0; r < (int) lens[i]; r++) tok_of[t++] = i;
    }
    const float *style_p = voice + (size_t) (n - 3) * 2 * S + S;   /* prosody half (:1149) */
    const float *style_d = voice + (size_t) (n - 3) * 2 * S;       /* decoder half (:1220) */
    float *en = falloc((size_t) T * W);
    for (int64_t t = 0; t < T; t++) memcpy(en + (size_t) t * W, hidden + (size_t) tok_of[t] * W, (size_t) W * 4);
    float *sh = falloc((size_t) T * D);
    bilstm(m, "kokoro.duration_predictor.shared_lstm", en, (int) T, W, D / 2, sh);
    float *shc = falloc((size_t) D * T);   /* [D][T] */
    for (int64_t t = 0; t < T; t++)
        for (int e = 0; e < D; e++) shc[(size_t) e * T + t] = sh[(size_t) t * D + e];
    float *curves[2];
    const char *branch[2] = {"f0", "n"};
    int64_t L2 = 0;
    for (int b = 0; b < 2; b++) {   /* :1169-1192 */
        float *cur = falloc((size_t) D * T);
        memcpy(cur, shc, (size_t) D * T * 4);
        int C = D;
        int64_t L = T;
        for (int i = 0; i < m->f0_n_blocks; i++) {
            char base[128];
            snprintf(base, sizeof(base), "kokoro.duration_predictor.%s_blocks.%d", branch[b], i);
            float *nx = ada_res_block(m, base, cur, L, style_p, S, &C, &L);
            free(cur);
            cur = nx;
        }
        const float *pw = ktf(m, NULL, "kokoro.duration_predictor.%s_proj_kernel", branch[b]), *pb = ktf(m, NULL, "kokoro.duration_predictor.%s_proj_bias", branch[b]);
        curves[b] = falloc((size_t) L);
        for (int64_t t = 0; t < L; t++) {
            double acc = 0.0;
            for (int c = 0; c < C; c++) acc += (double) pw[c] * (double) cur[(size_t) c * L + t];
            curves[b][t] = (float) acc + pb[0];
        }
        free(cur);
        L2 = L;
    }
    if (f0_out) memcpy(f0_out, curves[0], (size_t) L2 * 4);
    if (n_out) memcpy(n_out, curves[1], (size_t) L2 * 4);

    /* text encoder (:1196-1210) */
    const float *te = kt(m, "kokoro.text_encoder.embedding_weight", ne);
    const int C = (int) ne[0];
    float *tx = falloc((size_t) C * n);   /* [C][n] */
    for (int t = 0; t < n; t++)
        for (int c = 0; c < C; c++) tx[(size_t) c * n + t] = te[(size_t) tokens[t] * C + c];
    float *col = falloc(C);
    for (int l = 0; l < m->n_conv_layers; l++) {
        const float *cw = ktf(m, ne, "kokoro.text_encoder.layers.%d.weight", l);
        float *y;
        conv1d_s(tx, C, n, cw, ktf(m, NULL, "kokoro.text_encoder.layers.%d.bias", l), C, (int) ne[0], 1, 2, 1, &y);
        free(tx);
        tx = y;
        const float *g = ktf(m, NULL, "kokoro.text_encoder.layers.%d.gamma", l), *bt = ktf(m, NULL, "kokoro.text_encoder.layers.%d.beta", l);
        for (int t = 0; t < n; t++) {   /* layer norm over the channels of each position */
            for (int c = 0; c < C; c++) col[c] = tx[(size_t) c * n + t];
            norm_vec(col, C, 1e-5f, g, bt, col);
            for (int c = 0; c < C; c++) tx[(size_t) c * n + t] = col[c] > 0.0f ? col[c] : col[c] * 0.2f;
        }
    }
    float *txr = falloc((size_t) n * C), *tl = falloc((size_t) n * C);
    for (int t = 0; t < n; t++)
        for (int c = 0; c < C; c++) txr[(size_t) t * C + c] = tx[(size_t) c * n + t];
    bilstm(m, "kokoro.text_encoder.lstm", txr, n, C, C / 2, tl);
    float *asr = falloc((size_t) C * T);   /* [C][T] */
    for (int64_t t = 0; t < T; t++)
        for (int c = 0; c < C; c++) asr[(size_t) c * T + t] = tl[(size_t) tok_of[t] * C + c];

    /* decoder (:1222-1241) */
    float *f0d, *nd;
    conv1d_s(curves[0], 1, L2, kt(m, "kokoro.decoder.f0_conv_weight", NULL), kt(m, "kokoro.decoder.f0_conv_bias", NULL), 1, 3, 2, 1, 1, &f0d);
    conv1d_s(curves[1], 1, L2, kt(m, "kokoro.decoder.n_conv_weight", NULL), kt(m, "kokoro.decoder.n_conv_bias", NULL), 1, 3, 2, 1, 1, &nd);
    int Cc = C + 2;
    float *cur = falloc((size_t) Cc * T);
    memcpy(cur, asr, (size_t) C * T * 4);
    memcpy(cur + (size_t) C * T, f0d, (size_t) T * 4);
    memcpy(cur + (size_t) (C + 1) * T, nd, (size_t) T * 4);
    int64_t Lc = T;
    {
        float *nx = ada_res_block(m, "kokoro.decoder.encoder_block", cur, Lc, style_d, S, &Cc, &Lc);
        free(cur);
        cur = nx;
    }
    const float *aw = kt(m, "kokoro.decoder.asr_conv_weight", ne), *ab = kt(m, "kokoro.decoder.asr_conv_bias", NULL);
    const int CA = (int) ne[2];
    float *asr_res = falloc((size_t) CA * T);
    for (int co = 0; co < CA; co++)
        for (int64_t t = 0; t < T; t++) {
            double acc = 0.0;
            for (int c = 0; c < C; c++) acc += (double) aw[(size_t) co * C + c] * (double) asr[(size_t) c * T + t];
            asr_res[(size_t) co * T + t] = (float) acc + ab[co];
        }
    for (int i = 0; i < m->n_decoder_blocks; i++) {
        int Cin = Cc + CA + 2;
        float *cat = falloc((size_t) Cin * T);
        memcpy(cat, cur, (size_t) Cc * T * 4);
        memcpy(cat + (size_t) Cc * T, asr_res, (size_t) CA * T * 4);
        memcpy(cat + (size_t) (Cc + CA) * T, f0d, (size_t) T * 4);
        memcpy(cat + (size_t) (Cc + CA + 1) * T, nd, (size_t) T * 4);
        char base[128];
        snprintf(base, sizeof(base), "kokoro.decoder.decoder_blocks.%d", i);
        free(cur);
        cur = ada_res_block(m, base, cat, T, style_d, S, &Cin, &Lc);
        free(cat);
        Cc = Cin;
    }
    /* cur [Cc][Lc = 2T] */
    kdump("asr", asr, (size_t) C * T);
    kdump("dec_out", cur, (size_t) Cc * Lc);

    /* generator (:195-244).  harmonic source (:173-193) */
    const int NHm = m->harmonic_num + 1;
    const int64_t up = (int64_t) m->upsample_scale, LS = L2 * up;   /* samples */
    float *sine = falloc((size_t) NHm * LS);
    {
        float *phase = falloc((size_t) L2);
        for (int h = 0; h < NHm; h++) {
            float run = 0.0f;                                /* fp32 throughout, as the graph computes it */
            for (int64_t l = 0; l < L2; l++) {
                float v = curves[0][l] * (((float) h + 1.0f) / m->sample_rate);
                v = v - floorf(v);                           /* ggml_mod(x, 1) */
                run += v;                                    /* ggml_cumsum along the sequence */
                phase[l] = run * (m->upsample_scale * 2.0f * (float) M_PI);
            }
            for (int64_t j = 0; j < LS; j++) {               /* ggml_upscale_linear(x, 300): F.interpolate(linear, align_corners = False) */
                double src = ((double) j + 0.5) / (double) up - 0.5;
                if (src < 0) src = 0;
                int64_t i0 = (int64_t) src;
                if (i0 > L2 - 1) i0 = L2 - 1;
                const int64_t i1 = i0 + 1 < L2 ? i0 + 1 : L2 - 1;
                const float fr = (float) (src - (double) i0);
                const float ph = (1.0f - fr) * phase[i0] + fr * phase[i1];
                const float f0u = curves[0][j / up];         /* nearest upscale of f0 (:177) */
                const int voiced = f0u > m->voice_threshold;
                const float uv = voiced ? m->sin_amp : 0.0f;
                const float nz = (voiced ? m->noise_std : m->sin_amp / 3.0f) * noise[(size_t) h * LS + j];   /* uv_noise_compute, util.cpp:143-173 */
                sine[(size_t) h * LS + j] = sinf(ph) * uv + nz;
            }
        }
        free(phase);
    }
    float *har = falloc((size_t) LS);
    {
        const float *mw = kt(m, "kokoro.decoder.generator.m_source_weight", NULL), *mb = kt(m, "kokoro.decoder.generator.m_source_bias", NULL);
        for (int64_t j = 0; j < LS; j++) {
            double acc = 0.0;
            for (int h = 0; h < NHm; h++) acc += (double) mw[h] * (double) sine[(size_t) h * LS + j];
            har[j] = tanhf((float) acc + mb[0]);
        }
    }
    kdump("sine", sine, (size_t) NHm * LS);
    kdump("har", har, (size_t) LS);
    const int N = m->n_fft, hop = m->hop, nbins = N / 2 + 1;
    float *win = falloc(N);
    for (int i = 0; i < N; i++) win[i] = (float) pow(sin(M_PI * (double) i / (double) N), 2.0);   /* hann_window, util.cpp:134-139 */
    const int64_t F = LS / hop + 1;
    float *hs = falloc((size_t) 2 * nbins * F);   /* magnitude channels then phase channels (:204-206) */
    stft_mag_phase(har, LS, win, N, hop, hs, hs + (size_t) nbins * F, F);
    if (hsrc_out) memcpy(hsrc_out, hs, (size_t) 2 * nbins * F * 4);
    if (hsrc_in) memcpy(hs, hsrc_in, (size_t) 2 * nbins * F * 4);

    kdump("hsrc", hs, (size_t) 2 * nbins * F);
    float *g = cur;
    int Cg = Cc;
    int64_t Lg = Lc;
    for (int i = 0; i < m->n_upsamples; i++) {
        leaky(g, (size_t) Cg * Lg, 0.1f);
        const float *uw = ktf(m, ne, "kokoro.decoder.generator.ups.%d.weight", i);   /* ConvTranspose1d [Cin][Cout][K] */
        const int K = (int) ne[0], Co = (int) ne[1];
        const int64_t Lo = (Lg - 1) * m->up_stride[i] - 2 * (int64_t) m->up_padding[i] + K;
        float *y = falloc((size_t) Co * Lo);
        orc_conv_transpose1d(g, Cg, Lg, uw, ktf(m, NULL, "kokoro.decoder.generator.ups.%d.bias", i), Co, K, m->up_stride[i], m->up_padding[i], y);
        free(g);
        g = y; Cg = Co; Lg = Lo;
        if (i == m->n_upsamples - 1) {   /* reflection pad of one sample in front (:215-220) */
            float *p = falloc((size_t) Cg * (Lg + 1));
            for (int c = 0; c < Cg; c++) {
                p[(size_t) c * (Lg + 1)] = g[(size_t) c * Lg + 1];
                memcpy(p + (size_t) c * (Lg + 1) + 1, g + (size_t) c * Lg, (size_t) Lg * 4);
            }
            free(g);
            g = p; Lg += 1;
        }
        const float *nw = ktf(m, ne, "kokoro.decoder.generator.noise_blocks.%d.conv_weight", i);
        float *xs;
        const int64_t Ls = conv1d_s(hs, 2 * nbins, F, nw, ktf(m, NULL, "kokoro.decoder.generator.noise_blocks.%d.conv_bias", i), Cg, (int) ne[0], m->noise_stride[i],
                                    m->noise_padding[i], 1, &xs);
        if (Ls != Lg) { fprintf(stderr, "kokoro oracle: source length %lld != %lld at stage %d\n", (long long) Ls, (long long) Lg, i); abort(); }
        char base[128];
        snprintf(base, sizeof(base), "kokoro.decoder.generator.noise_blocks.%d.resblock", i);
        gen_res_block(m, base, xs, Cg, Lg, style_d, S, m->noise_res_padding[i], m->noise_res_dilation[i]);
        for (size_t j = 0; j < (size_t) Cg * Lg; j++) g[j] += xs[j];
        free(xs);
        float *sum = falloc((size_t) Cg * Lg), *br = falloc((size_t) Cg * Lg);
        for (int ii = 0; ii < m->n_kernels; ii++) {
            memcpy(br, g, (size_t) Cg * Lg * 4);
            snprintf(base, sizeof(base), "kokoro.decoder.generator.resblocks.%d", i * m->n_kernels + ii);
            gen_res_block(m, base, br, Cg, Lg, style_d, S, m->res_padding[i * m->n_kernels + ii], m->res_dilation[i * m->n_kernels + ii]);
            for (size_t j = 0; j < (size_t) Cg * Lg; j++) sum[j] += br[j];
        }
        for (size_t j = 0; j < (size_t) Cg * Lg; j++) g[j] = sum[j] / (float) m->n_kernels;
        free(sum); free(br);
        { char nm_[32]; snprintf(nm_, sizeof(nm_), "gen_stage%d", i); kdump(nm_, g, (size_t) Cg * Lg); }
    }
    leaky(g, (size_t) Cg * Lg, 0.01f);
    const float *pw = kt(m, "kokoro.decoder.generator.conv_post_weight", ne);
    float *post;
    conv1d_s(g, Cg, Lg, pw, kt(m, "kokoro.decoder.generator.conv_post_bias", NULL), 2 * nbins, (int) ne[0], 1, m->out_conv_padding, 1, &post);
    for (size_t j = 0; j < (size_t) nbins * Lg; j++) {
        post[j] = expf(post[j]);                                          /* spec (:234-238) */
        post[(size_t) nbins * Lg + j] = sinf(post[(size_t) nbins * Lg + j]);   /* phase */
    }
    kdump("post", post, (size_t) 2 * nbins * Lg);
    const int64_t out_len = T * m->up_sampling_factor;
    istft_mag_phase(post, post + (size_t) nbins * Lg, Lg, win, N, hop, pcm_out, out_len);
    free(post); free(g); free(hs); free(win); free(har); free(sine); free(asr_res); free(f0d); free(nd); free(asr); free(tl); free(txr); free(tx); free(col);
    free(curves[0]); free(curves[1]); free(shc); free(sh); free(en); free(tok_of);
    return out_len;
}

/* ---- stage entry points (test infrastructure, round 5) -------------------------------------------------------------------------------
 * One upstream module each, so that tests/test_upstream_golden.py can hold the restatement above against PyTorch's own definitions of the
 * stages the whole-graph fixture (tests/golden/tiny_kokoro.npz, wired by this repository's author) only covers indirectly:
 *   orc_kk_stage_bilstm      build_lstm / build_lstm_run (kokoro/model.cpp:35-86)             <-> torch.nn.LSTM(bidirectional), weights split per
 *                                                                                               gate as kokoro_gguf_encoder.py:289-309 writes them
 *   orc_kk_stage_ada_block   build_ada_residual_conv (:88-134): AdaIN = instance norm + style   <-> F.instance_norm, nn.Linear, F.conv_transpose1d(groups = C,
 *                            affine, leaky relu, depthwise transposed-conv pool, 1x1 shortcut       stride 2, padding 1, output_padding 1), F.conv1d, F.interpolate
 *   orc_kk_stage_stft/istft  stft / istft (util.cpp:111-133) + compute_window_squared_sum       <-> torch.stft / torch.istft (center, reflect, onesided)
 *                            (util.cpp:203-217)
 * The tensors a stage needs are looked up by name in `m` exactly like the full graphs do. */
void orc_kk_stage_bilstm(const orc_kokoro_model *m, const char *base, const float *x, int L, int in, int hid, float *out) {
    bilstm(m, base, x, L, in, hid, out);
}
int64_t orc_kk_stage_ada_block(const orc_kokoro_model *m, const char *base, const float *x, int C, int64_t L, const float *style, int S, float *out, int32_t *C_out) {
    int c = C;
    int64_t lo = 0;
    float *y = ada_res_block(m, base, x, L, style, S, &c, &lo);
    memcpy(out, y, (size_t) c * lo * sizeof(float));
    free(y);
    *C_out = c;
    return lo;
}
void orc_kk_stage_stft(const float *x, int64_t L, const float *win, int N, int hop, float *mag, float *ph) {
    stft_mag_phase(x, L, win, N, hop, mag, ph, L / hop + 1);
}
void orc_kk_stage_istft(const float *mag, const float *ph, int64_t F, const float *win, int N, int hop, float *out, int64_t out_len) {
    istft_mag_phase(mag, ph, F, win, N, hop, out, out_len);
}

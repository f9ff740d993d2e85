/*
 * tts_oracle.c — CPU restatement of the TTS.cpp Parler-TTS decoder step, sampler, delay
 * pattern and DAC decoder.  TEST INFRASTRUCTURE ONLY (see tts_oracle.h header comment):
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * Parity status in tts_oracle.h: sampler pinned to the real reference sampler.cpp (oracle/_ref), graph arithmetic pinned to
 * the upstream models the reference converts from (tests/golden/upstream_*.npz); ggml's kernel-level rounding unpinned
 * (fork absent, no reference golden vectors).
 *
 * Citations are file:line under /root/reference.
 */
#include "tts_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ======================================================================================
 * fp16 <-> fp32 (IEEE binary16, round-to-nearest-even) — what GGML_FP32_TO_FP16 does on
 * every platform (F16C / NEON / the portable bit-twiddling fallback all implement RNE).
 * ==================================================================================== */
float orc_h2f(uint16_t h) {
    uint32_t sign = (uint32_t) (h & 0x8000u) << 16;
    uint32_t exp  = (h >> 10) & 0x1Fu;
    uint32_t man  = h & 0x3FFu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal */
            int e = -1;
            do { e++; man <<= 1; } while ((man & 0x400u) == 0);
            man &= 0x3FFu;
            bits = sign | ((uint32_t) (127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7F800000u | (man << 13);
    } else {
        bits = sign | ((exp + 112u) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

uint16_t orc_f2h(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t absx = x & 0x7FFFFFFFu;
    if (absx >= 0x7F800000u) { /* inf / nan */
        return (uint16_t) (sign | 0x7C00u | ((absx > 0x7F800000u) ? 0x200u : 0));
    }
    if (absx >= 0x477FF000u) { /* rounds to >= 65520 -> inf */
        return (uint16_t) (sign | 0x7C00u);
    }
    if (absx < 0x38800000u) { /* subnormal half or zero */
        if (absx < 0x33000000u) return (uint16_t) sign; /* < 2^-25 -> 0 */
        uint32_t e = absx >> 23;
        uint32_t m = (absx & 0x7FFFFFu) | 0x800000u;
        uint32_t shift = 126 - e; /* 14..24 */
        uint32_t r = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1u);
        uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1u))) r++;
        return (uint16_t) (sign | r);
    }
    uint32_t e = (absx >> 23) - 112u;
    uint32_t m = absx & 0x7FFFFFu;
    uint32_t r = (e << 10) | (m >> 13);
    uint32_t rem = m & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r++;
    return (uint16_t) (sign | r);
}

/* ======================================================================================
 * ggml block formats.  Layouts per SURVEY.md A.3 (upstream ggml-common.h knowledge):
 *   Q4_0: { fp16 d; u8 qs[16] }          Q5_0: { fp16 d; u8 qh[4]; u8 qs[16] }
 *   Q8_0: { fp16 d; i8 qs[32] }
 * Quantizers restate upstream quantize_row_q{4_0,5_0,8_0}_ref (ggml-quants.c).
 * ==================================================================================== */
#define QK 32

size_t orc_row_bytes(int type, int64_t n) {
    switch (type) {
        case ORC_F32:  return (size_t) n * 4;
        case ORC_F16:  return (size_t) n * 2;
        case ORC_Q4_0: return (size_t) (n / QK) * 18;
        case ORC_Q5_0: return (size_t) (n / QK) * 22;
        case ORC_Q8_0: return (size_t) (n / QK) * 34;
        default: return 0;
    }
}

int orc_dequantize(int type, const void *src, float *dst, int64_t n) {
    const uint8_t *p = (const uint8_t *) src;
    switch (type) {
        case ORC_F32:
            memcpy(dst, src, (size_t) n * 4);
            return 0;
        case ORC_F16: {
            const uint16_t *h = (const uint16_t *) src;
            for (int64_t i = 0; i < n; i++) dst[i] = orc_h2f(h[i]);
            return 0;
        }
        case ORC_Q4_0:
            for (int64_t b = 0; b < n / QK; b++, p += 18, dst += QK) {
                uint16_t dh; memcpy(&dh, p, 2);
                float d = orc_h2f(dh);
                const uint8_t *qs = p + 2;
                for (int j = 0; j < 16; j++) {
                    dst[j]      = (float) ((int) (qs[j] & 0x0F) - 8) * d;
                    dst[j + 16] = (float) ((int) (qs[j] >> 4) - 8) * d;
                }
            }
            return 0;
        case ORC_Q5_0:
            for (int64_t b = 0; b < n / QK; b++, p += 22, dst += QK) {
                uint16_t dh; memcpy(&dh, p, 2);
                float d = orc_h2f(dh);
                uint32_t qh; memcpy(&qh, p + 2, 4);
                const uint8_t *qs = p + 6;
                for (int j = 0; j < 16; j++) {
                    uint8_t xh0 = (uint8_t) (((qh >> (j + 0)) << 4) & 0x10);
                    uint8_t xh1 = (uint8_t) ((qh >> (j + 12)) & 0x10);
                    dst[j]      = (float) ((int) ((qs[j] & 0x0F) | xh0) - 16) * d;
                    dst[j + 16] = (float) ((int) ((qs[j] >> 4) | xh1) - 16) * d;
                }
            }
            return 0;
        case ORC_Q8_0:
            for (int64_t b = 0; b < n / QK; b++, p += 34, dst += QK) {
                uint16_t dh; memcpy(&dh, p, 2);
                float d = orc_h2f(dh);
                const int8_t *qs = (const int8_t *) (p + 2);
                for (int j = 0; j < QK; j++) dst[j] = (float) qs[j] * d;
            }
            return 0;
        default:
            return -1;
    }
}

static void quant_q8_0_block(const float *x, uint8_t *p) {
    float amax = 0.0f;
    for (int j = 0; j < QK; j++) { float v = fabsf(x[j]); if (v > amax) amax = v; }
    const float d  = amax / 127.0f;
    const float id = d ? 1.0f / d : 0.0f;
    uint16_t dh = orc_f2h(d);
    memcpy(p, &dh, 2);
    int8_t *qs = (int8_t *) (p + 2);
    for (int j = 0; j < QK; j++) qs[j] = (int8_t) roundf(x[j] * id);
}

int orc_quantize(int type, const float *src, void *dst, int64_t n) {
    uint8_t *p = (uint8_t *) dst;
    switch (type) {
        case ORC_F32:
            memcpy(dst, src, (size_t) n * 4);
            return 0;
        case ORC_F16: {
            uint16_t *h = (uint16_t *) dst;
            for (int64_t i = 0; i < n; i++) h[i] = orc_f2h(src[i]);
            return 0;
        }
        case ORC_Q4_0:
            for (int64_t b = 0; b < n / QK; b++, p += 18, src += QK) {
                float amax = 0.0f, max = 0.0f;
                for (int j = 0; j < QK; j++) { float v = src[j]; if (amax < fabsf(v)) { amax = fabsf(v); max = v; } }
                const float d  = max / -8.0f;
                const float id = d ? 1.0f / d : 0.0f;
                uint16_t dh = orc_f2h(d);
                memcpy(p, &dh, 2);
                uint8_t *qs = p + 2;
                for (int j = 0; j < 16; j++) {
                    float x0 = src[j] * id, x1 = src[j + 16] * id;
                    uint8_t xi0 = (uint8_t) (int8_t) (x0 + 8.5f); if (xi0 > 15) xi0 = 15;
                    uint8_t xi1 = (uint8_t) (int8_t) (x1 + 8.5f); if (xi1 > 15) xi1 = 15;
                    qs[j] = (uint8_t) (xi0 | (xi1 << 4));
                }
            }
            return 0;
        case ORC_Q5_0:
            for (int64_t b = 0; b < n / QK; b++, p += 22, src += QK) {
                float amax = 0.0f, max = 0.0f;
                for (int j = 0; j < QK; j++) { float v = src[j]; if (amax < fabsf(v)) { amax = fabsf(v); max = v; } }
                const float d  = max / -16.0f;
                const float id = d ? 1.0f / d : 0.0f;
                uint16_t dh = orc_f2h(d);
                memcpy(p, &dh, 2);
                uint32_t qh = 0;
                uint8_t *qs = p + 6;
                for (int j = 0; j < 16; j++) {
                    float x0 = src[j] * id, x1 = src[j + 16] * id;
                    uint8_t xi0 = (uint8_t) (int8_t) (x0 + 16.5f); if (xi0 > 31) xi0 = 31;
                    uint8_t xi1 = (uint8_t) (int8_t) (x1 + 16.5f); if (xi1 > 31) xi1 = 31;
                    qs[j] = (uint8_t) ((xi0 & 0x0F) | ((xi1 & 0x0F) << 4));
                    qh |= ((uint32_t) ((xi0 & 0x10u) >> 4)) << (j + 0);
                    qh |= ((uint32_t) ((xi1 & 0x10u) >> 4)) << (j + 16);
                }
                memcpy(p + 2, &qh, 4);
            }
            return 0;
        case ORC_Q8_0:
            for (int64_t b = 0; b < n / QK; b++, p += 34, src += QK) quant_q8_0_block(src, p);
            return 0;
        default:
            return -1;
    }
}

int orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void) n;
    return 1;
#endif
}

/* ======================================================================================
 * mul_mat.  ggml_mul_mat(W, x) with W ne=[K,N] (N rows of K), x ne=[K,R] -> y ne=[N,R].
 * Dot products accumulate in double (ggml accumulates in fp32 SIMD lanes in an
 * implementation-defined order; double is the order-independent statement of the same sum).
 * act_mode 1 restates ggml_compute_forward_mul_mat's conversion of src1 to the weight
 * type's vec_dot_type (upstream ggml knowledge; SURVEY.md §7 "Hard parts").
 * ==================================================================================== */
static double dot_q_q8(int type, const uint8_t *w, const uint8_t *a, int64_t K) {
    /* integer block dot like ggml_vec_dot_q{4_0,5_0,8_0}_q8_0 */
    double sum = 0.0;
    for (int64_t b = 0; b < K / QK; b++) {
        const uint8_t *ab = a + b * 34;
        uint16_t adh; memcpy(&adh, ab, 2);
        const int8_t *aq = (const int8_t *) (ab + 2);
        int32_t sumi = 0;
        float wd;
        if (type == ORC_Q4_0) {
            const uint8_t *wb = w + b * 18;
            uint16_t dh; memcpy(&dh, wb, 2); wd = orc_h2f(dh);
            const uint8_t *qs = wb + 2;
            for (int j = 0; j < 16; j++) {
                sumi += ((int) (qs[j] & 0x0F) - 8) * aq[j] + ((int) (qs[j] >> 4) - 8) * aq[j + 16];
            }
        } else if (type == ORC_Q5_0) {
            const uint8_t *wb = w + b * 22;
            uint16_t dh; memcpy(&dh, wb, 2); wd = orc_h2f(dh);
            uint32_t qh; memcpy(&qh, wb + 2, 4);
            const uint8_t *qs = wb + 6;
            for (int j = 0; j < 16; j++) {
                uint8_t xh0 = (uint8_t) (((qh >> (j + 0)) << 4) & 0x10);
                uint8_t xh1 = (uint8_t) ((qh >> (j + 12)) & 0x10);
                sumi += ((int) ((qs[j] & 0x0F) | xh0) - 16) * aq[j] + ((int) ((qs[j] >> 4) | xh1) - 16) * aq[j + 16];
            }
        } else { /* Q8_0 */
            const uint8_t *wb = w + b * 34;
            uint16_t dh; memcpy(&dh, wb, 2); wd = orc_h2f(dh);
            const int8_t *qs = (const int8_t *) (wb + 2);
            for (int j = 0; j < QK; j++) sumi += (int) qs[j] * aq[j];
        }
        sum += (double) ((float) sumi * (wd * orc_h2f(adh)));
    }
    return sum;
}

void orc_mul_mat(int type, const void *W, int64_t K, int64_t N, const float *x, int64_t R,
                 float *y, int act_mode) {
    const size_t rb = orc_row_bytes(type, K);
    const uint8_t *wp = (const uint8_t *) W;
    const int quant = (type == ORC_Q4_0 || type == ORC_Q5_0 || type == ORC_Q8_0);

    if (quant && act_mode == 1) {
        uint8_t *aq = (uint8_t *) malloc((size_t) R * (size_t) (K / QK) * 34);
        for (int64_t r = 0; r < R; r++)
            for (int64_t b = 0; b < K / QK; b++) quant_q8_0_block(x + r * K + b * QK, aq + (r * (K / QK) + b) * 34);
#pragma omp parallel for schedule(static)
        for (int64_t n = 0; n < N; n++)
            for (int64_t r = 0; r < R; r++)
                y[r * N + n] = (float) dot_q_q8(type, wp + (size_t) n * rb, aq + (size_t) r * (size_t) (K / QK) * 34, K);
        free(aq);
        return;
    }

    float *xa = NULL;
    const float *xs = x;
    if (type == ORC_F16 && act_mode == 1) { /* activations rounded to fp16 (vec_dot_type F16) */
        xa = (float *) malloc((size_t) R * (size_t) K * 4);
        for (int64_t i = 0; i < R * K; i++) xa[i] = orc_h2f(orc_f2h(x[i]));
        xs = xa;
    }
#pragma omp parallel
    {
        float *row = (float *) malloc((size_t) K * 4);
#pragma omp for schedule(static)
        for (int64_t n = 0; n < N; n++) {
            const float *wr;
            if (type == ORC_F32) {
                wr = (const float *) (wp + (size_t) n * rb);
            } else {
                orc_dequantize(type, wp + (size_t) n * rb, row, K);
                wr = row;
            }
            for (int64_t r = 0; r < R; r++) {
                const float *xr = xs + r * K;
                double acc = 0.0;
                for (int64_t k = 0; k < K; k++) acc += (double) wr[k] * (double) xr[k];
                y[r * N + n] = (float) acc;
            }
        }
        free(row);
    }
    free(xa);
}

/* ======================================================================================
 * elementwise pieces
 * ==================================================================================== */
/* parler_build_layer_norm (model.cpp:412-418): ggml_norm eps=1e-5, * weight, + bias.
 * ggml_norm: mean and variance in double (ggml_float), scale = 1/sqrtf(var + eps). */
void orc_layer_norm(const float *x, int H, const float *w, const float *b, float *y) {
    double sum = 0.0;
    for (int i = 0; i < H; i++) sum += (double) x[i];
    float mean = (float) (sum / H);
    double sum2 = 0.0;
    for (int i = 0; i < H; i++) { float v = x[i] - mean; y[i] = v; sum2 += (double) (v * v); }
    float variance = (float) (sum2 / H);
    const float scale = 1.0f / sqrtf(variance + 1e-5f);
    for (int i = 0; i < H; i++) y[i] = y[i] * scale * w[i] + b[i];
}

/* ggml_gelu (model.cpp:602).  mode 0: fp32 tanh approximation.  mode 1: ggml CPU's fp16
 * lookup table (ggml_vec_gelu_f32 with GGML_GELU_FP16; upstream knowledge): the input is
 * rounded to fp16, gelu evaluated in fp32 and the result rounded to fp16 again; |x|>=10 bypass. */
static float gelu_f32(float x) {
    const float GELU_COEF_A = 0.044715f, SQRT_2_OVER_PI = 0.79788456080286535587989211986876f;
    return 0.5f * x * (1.0f + tanhf(SQRT_2_OVER_PI * x * (1.0f + GELU_COEF_A * x * x)));
}
float orc_gelu(float x, int mode) {
    if (mode == 0) return gelu_f32(x);
    if (x <= -10.0f) return 0.0f;
    if (x >= 10.0f) return x;
    return orc_h2f(orc_f2h(gelu_f32(orc_h2f(orc_f2h(x)))));
}

/* ======================================================================================
 * Parler decoder
 * ==================================================================================== */
struct orc_parler_state {
    int H, L, n_ctx, E;
    float *k, *v;   /* [L][n_ctx][H]  (the reference stores V transposed, model.cpp:432-436;
                       layout is free as long as the arithmetic is the same) */
    float *ck, *cv; /* [L][E][H] */
};

orc_parler_state *orc_parler_state_new(const orc_parler_model *m) {
    orc_parler_state *s = (orc_parler_state *) calloc(1, sizeof(*s));
    s->H = m->H; s->L = m->L; s->n_ctx = m->n_ctx; s->E = m->E;
    size_t n = (size_t) m->L * m->n_ctx * m->H;
    s->k = (float *) calloc(n, 4);
    s->v = (float *) calloc(n, 4);
    size_t nc = (size_t) m->L * (m->E > 0 ? m->E : 1) * m->H;
    s->ck = (float *) calloc(nc, 4);
    s->cv = (float *) calloc(nc, 4);
    return s;
}
void orc_parler_state_free(orc_parler_state *s) {
    if (!s) return;
    free(s->k); free(s->v); free(s->ck); free(s->cv); free(s);
}

void orc_parler_prep_cross(const orc_parler_model *m, orc_parler_state *s) {
    if (!m->use_cross) return;
    for (int l = 0; l < m->L; l++) { /* model.cpp:138-140 */
        orc_mul_mat(m->layers[l].ck.type, m->layers[l].ck.data, m->H, m->H, m->text_encoding, m->E,
                    s->ck + (size_t) l * m->E * m->H, m->act_mode);
        orc_mul_mat(m->layers[l].cv.type, m->layers[l].cv.data, m->H, m->H, m->text_encoding, m->E,
                    s->cv + (size_t) l * m->E * m->H, m->act_mode);
    }
}

void orc_parler_get_kv(const orc_parler_state *s, int layer, int n_pos, float *k_out, float *v_out) {
    memcpy(k_out, s->k + (size_t) layer * s->n_ctx * s->H, (size_t) n_pos * s->H * 4);
    memcpy(v_out, s->v + (size_t) layer * s->n_ctx * s->H, (size_t) n_pos * s->H * 4);
}

static void get_row(const orc_w *w, int64_t row, int H, float *dst) {
    const uint8_t *p = (const uint8_t *) w->data + (size_t) row * orc_row_bytes(w->type, H);
    orc_dequantize(w->type, p, dst, H);
}

/* softmax(scale*x [+ mask]) over n (ggml_soft_max_ext, model.cpp:567,589): max, expf(x-max),
 * double sum, scale by 1/sum. */
static void softmax_scaled(float *x, int n, float scale) {
    float mx = -INFINITY;
    for (int i = 0; i < n; i++) { x[i] *= scale; if (x[i] > mx) mx = x[i]; }
    double sum = 0.0;
    for (int i = 0; i < n; i++) { x[i] = expf(x[i] - mx); sum += (double) x[i]; }
    const float inv = (float) (1.0 / sum);
    for (int i = 0; i < n; i++) x[i] *= inv;
}

/* attention of one query row against n_keys rows of K,V ([n][H] position-major), per head */
static void attend(const float *q, const float *K, const float *V, int n_keys, int H, int n_heads,
                   float *out, float *scores) {
    const int d = H / n_heads;
    const float scale = 1.0f / sqrtf((float) d);
    for (int h = 0; h < n_heads; h++) {
        for (int t = 0; t < n_keys; t++) {
            double acc = 0.0;
            const float *kr = K + (size_t) t * H + h * d;
            for (int c = 0; c < d; c++) acc += (double) kr[c] * (double) q[h * d + c];
            scores[t] = (float) acc;
        }
        softmax_scaled(scores, n_keys, scale);
        for (int c = 0; c < d; c++) {
            double acc = 0.0;
            for (int t = 0; t < n_keys; t++) acc += (double) scores[t] * (double) V[(size_t) t * H + h * d + c];
            out[h * d + c] = (float) acc;
        }
    }
}

void orc_parler_decode(const orc_parler_model *m, orc_parler_state *s, int audio,
                       const uint32_t *tokens, int S, uint32_t pos0, float *logits_out,
                       float *hidden_out) {
    const int H = m->H, F = m->F;
    float *x   = (float *) malloc((size_t) S * H * 4);
    float *cur = (float *) malloc((size_t) S * H * 4);
    float *q   = (float *) malloc((size_t) S * H * 4);
    float *kk  = (float *) malloc((size_t) S * H * 4);
    float *vv  = (float *) malloc((size_t) S * H * 4);
    float *att = (float *) malloc((size_t) S * H * 4);
    float *tmp = (float *) malloc((size_t) S * H * 4);
    float *ff  = (float *) malloc((size_t) S * F * 4);
    float *row = (float *) malloc((size_t) H * 4);
    float *scores = (float *) malloc((size_t) (m->n_ctx > m->E ? m->n_ctx : m->E) * 4);

    /* parler_build_inp_embd (model.cpp:387-410) */
    for (int sidx = 0; sidx < S; sidx++) {
        float *xr = x + (size_t) sidx * H;
        if (audio) {
            for (int i = 0; i < m->n_out; i++) {
                get_row(&m->embed_tokens[i], tokens[i], H, row);
                if (i == 0) memcpy(xr, row, (size_t) H * 4);
                else for (int c = 0; c < H; c++) xr[c] = row[c] + xr[c];
            }
        } else {
            get_row(&m->embed_prompts, tokens[sidx], H, xr);
        }
        const float *pe = m->pos_embed + (size_t) (pos0 + sidx) * H;
        for (int c = 0; c < H; c++) xr[c] = xr[c] + pe[c];
    }

    for (int l = 0; l < m->L; l++) {
        const orc_parler_layer *ly = &m->layers[l];
        float *Kl = s->k + (size_t) l * s->n_ctx * H;
        float *Vl = s->v + (size_t) l * s->n_ctx * H;
        /* self attention (model.cpp:538-574) */
        for (int i = 0; i < S; i++) orc_layer_norm(x + (size_t) i * H, H, ly->sa_ln_w, ly->sa_ln_b, cur + (size_t) i * H);
        orc_mul_mat(ly->q.type, ly->q.data, H, H, cur, S, q, m->act_mode);
        orc_mul_mat(ly->k.type, ly->k.data, H, H, cur, S, kk, m->act_mode);
        orc_mul_mat(ly->v.type, ly->v.data, H, H, cur, S, vv, m->act_mode);
        memcpy(Kl + (size_t) pos0 * H, kk, (size_t) S * H * 4); /* parler_build_kv_store :420-439 */
        memcpy(Vl + (size_t) pos0 * H, vv, (size_t) S * H * 4);
        for (int i = 0; i < S; i++) /* causal mask: key <= pos (model.cpp:623-631) */
            attend(q + (size_t) i * H, Kl, Vl, (int) pos0 + i + 1, H, m->n_heads, att + (size_t) i * H, scores);
        orc_mul_mat(ly->o.type, ly->o.data, H, H, att, S, tmp, m->act_mode);
        for (size_t i = 0; i < (size_t) S * H; i++) x[i] = tmp[i] + x[i];

        if (m->use_cross) { /* model.cpp:576-596 */
            for (int i = 0; i < S; i++) orc_layer_norm(x + (size_t) i * H, H, ly->ca_ln_w, ly->ca_ln_b, cur + (size_t) i * H);
            orc_mul_mat(ly->cq.type, ly->cq.data, H, H, cur, S, q, m->act_mode);
            for (int i = 0; i < S; i++)
                attend(q + (size_t) i * H, s->ck + (size_t) l * m->E * H, s->cv + (size_t) l * m->E * H, m->E, H,
                       m->n_heads, att + (size_t) i * H, scores);
            orc_mul_mat(ly->co.type, ly->co.data, H, H, att, S, tmp, m->act_mode);
            for (size_t i = 0; i < (size_t) S * H; i++) x[i] = tmp[i] + x[i];
        }

        /* FFN (model.cpp:598-604) */
        for (int i = 0; i < S; i++) orc_layer_norm(x + (size_t) i * H, H, ly->f_ln_w, ly->f_ln_b, cur + (size_t) i * H);
        orc_mul_mat(ly->fc1.type, ly->fc1.data, H, F, cur, S, ff, m->act_mode);
        for (size_t i = 0; i < (size_t) S * F; i++) ff[i] = orc_gelu(ff[i], m->gelu_mode);
        orc_mul_mat(ly->fc2.type, ly->fc2.data, F, H, ff, S, tmp, m->act_mode);
        for (size_t i = 0; i < (size_t) S * H; i++) x[i] = tmp[i] + x[i];
    }

    /* final norm + heads (model.cpp:608-609, 441-457) */
    for (int i = 0; i < S; i++) orc_layer_norm(x + (size_t) i * H, H, m->ln_w, m->ln_b, cur + (size_t) i * H);
    if (hidden_out) memcpy(hidden_out, cur, (size_t) S * H * 4);
    if (logits_out) {
        for (int i = 0; i < m->n_out; i++)
            orc_mul_mat(m->lm_heads[i].type, m->lm_heads[i].data, H, m->V, cur, S,
                        logits_out + (size_t) i * S * m->V, m->act_mode);
    }
    free(x); free(cur); free(q); free(kk); free(vv); free(att); free(tmp); free(ff); free(row); free(scores);
}

/* model.cpp:778-785: head i is fed BOS until `step > i`, then its last output, EOS once seen.
 * `step` is batch.current_step of the decode that just ran (0 for the text prompt). */
void orc_parler_next_ids(int n_out, int step, const uint32_t *last_outputs, const uint8_t *eos_seen,
                         uint32_t bos, uint32_t eos, uint32_t *next_ids) {
    for (int i = 0; i < n_out; i++)
        next_ids[i] = step > i ? (eos_seen[i] ? eos : last_outputs[i]) : bos;
}

/* ======================================================================================
 * T5 encoder (src/models/parler/t5/model.cpp).  rms norm eps 1e-6 (:181), no attention scale
 * (soft_max_ext(kq, mask, 1.0f, 0) :258 with an all-zero mask :311), relative position bias added to
 * kq before the softmax (:256), gated GELU MLP gelu(wi_0 x) * (wi_1 x) (:272-275).
 * ==================================================================================== */
uint32_t orc_t5_bucket(int i, int ii, int n_buckets_total) {
    /* :303-316.  pos_bucket[i*n + ii] is added to kq[key = ii?]: the tensor is [ne0 = n, ne1 = n] with data index
     * i*n + ii, and build_t5_pos_bias permutes it so that (i, ii) lands on kq's (ne0 = key, ne1 = query) = (i, ii):
     * i is the KEY position, ii the QUERY position, rpos = key - query (HF: memory - context). */
    const int n_buckets = n_buckets_total / 2;
    const int max_exact = n_buckets / 2;
    const float logarithmic_denominator = (float) log(128.0 / max_exact);
    const int ab_rpos = abs(i - ii);
    const int rpos = i - ii;
    int v;
    if (ab_rpos < max_exact) v = ab_rpos;
    else {
        /* log((ab_rpos / max_exact)) with INTEGER division, as written (:314) */
        const int big = max_exact + (int) ((log((double) (ab_rpos / max_exact)) / logarithmic_denominator) * max_exact);
        v = big < n_buckets - 1 ? big : n_buckets - 1;
    }
    return (uint32_t) (rpos > 0 ? n_buckets : 0) + (uint32_t) v;
}

static void t5_rms_norm(const float *x, int H, const float *w, float *y) {
    /* ggml_rms_norm(eps 1e-6) * weight (:179-185): mean of squares in double like ggml's ggml_float */
    double sum = 0.0;
    for (int i = 0; i < H; i++) sum += (double) (x[i] * x[i]);
    const float mean = (float) (sum / H);
    const float scale = 1.0f / sqrtf(mean + 1e-6f);
    for (int i = 0; i < H; i++) y[i] = x[i] * scale * w[i];
}

void orc_t5_encode(const orc_t5_model *m, const uint32_t *ids, int n, float *out) {
    const int H = m->H, F = m->F, NH = m->n_heads, D = H / NH;
    float *x = (float *) malloc((size_t) n * H * 4), *cur = (float *) malloc((size_t) n * H * 4);
    float *q = (float *) malloc((size_t) n * H * 4), *k = (float *) malloc((size_t) n * H * 4), *v = (float *) malloc((size_t) n * H * 4);
    float *att = (float *) malloc((size_t) n * H * 4), *tmp = (float *) malloc((size_t) n * H * 4);
    float *up = (float *) malloc((size_t) n * F * 4), *gate = (float *) malloc((size_t) n * F * 4);
    float *sc = (float *) malloc((size_t) n * 4);
    for (int t = 0; t < n; t++) get_row(&m->embd, ids[t], H, x + (size_t) t * H);
    for (int l = 0; l < m->L; l++) {
        const orc_t5_layer *ly = &m->layers[l];
        for (int t = 0; t < n; t++) t5_rms_norm(x + (size_t) t * H, H, ly->attn_norm, cur + (size_t) t * H);
        orc_mul_mat(ly->q.type, ly->q.data, H, H, cur, n, q, m->act_mode);
        orc_mul_mat(ly->k.type, ly->k.data, H, H, cur, n, k, m->act_mode);
        orc_mul_mat(ly->v.type, ly->v.data, H, H, cur, n, v, m->act_mode);
        for (int h = 0; h < NH; h++) {
            for (int qi = 0; qi < n; qi++) {
                const float *qv = q + (size_t) qi * H + h * D;
                for (int ki = 0; ki < n; ki++) {
                    const float *kv = k + (size_t) ki * H + h * D;
                    double d = 0.0;
                    for (int e = 0; e < D; e++) d += (double) qv[e] * (double) kv[e];
                    sc[ki] = (float) d + m->rel_bias[(size_t) orc_t5_bucket(ki, qi, m->n_buckets) * NH + h];
                }
                softmax_scaled(sc, n, 1.0f);
                float *o = att + (size_t) qi * H + h * D;
                for (int e = 0; e < D; e++) {
                    double a = 0.0;
                    for (int ki = 0; ki < n; ki++) a += (double) sc[ki] * (double) v[(size_t) ki * H + h * D + e];
                    o[e] = (float) a;
                }
            }
        }
        orc_mul_mat(ly->o.type, ly->o.data, H, H, att, n, tmp, m->act_mode);
        for (size_t i = 0; i < (size_t) n * H; i++) x[i] = tmp[i] + x[i];          /* ggml_add(attn_out, residual) :267 */
        for (int t = 0; t < n; t++) t5_rms_norm(x + (size_t) t * H, H, ly->mlp_norm, cur + (size_t) t * H);
        orc_mul_mat(ly->wi_1.type, ly->wi_1.data, H, F, cur, n, gate, m->act_mode);
        orc_mul_mat(ly->wi_0.type, ly->wi_0.data, H, F, cur, n, up, m->act_mode);
        for (size_t i = 0; i < (size_t) n * F; i++) up[i] = orc_gelu(up[i], m->gelu_mode) * gate[i];  /* :274 */
        orc_mul_mat(ly->wo.type, ly->wo.data, F, H, up, n, tmp, m->act_mode);
        for (size_t i = 0; i < (size_t) n * H; i++) x[i] = tmp[i] + x[i];          /* :278 */
    }
    for (int t = 0; t < n; t++) t5_rms_norm(x + (size_t) t * H, H, m->out_norm, cur + (size_t) t * H);
    if (m->down_proj.data) {
        orc_mul_mat(m->down_proj.type, m->down_proj.data, H, m->out_size, cur, n, out, m->act_mode);
        if (m->down_proj_bias)
            for (int t = 0; t < n; t++)
                for (int i = 0; i < m->out_size; i++) out[(size_t) t * m->out_size + i] += m->down_proj_bias[i];
    } else {
        memcpy(out, cur, (size_t) n * H * 4);
    }
    free(x); free(cur); free(q); free(k); free(v); free(att); free(tmp); free(up); free(gate); free(sc);
}

/* adjust_output_tokens (model.cpp:734-760).  Quirks kept: the bound check is
 * `next_index > size` (off by one: next_index == size reads one past the end in the reference;
 * here that element is treated as "out of range -> remove", the only defined behaviour). */
size_t orc_parler_adjust_output_tokens(const uint32_t *tokens, size_t size, int n_out,
                                       uint32_t audio_vocab, uint32_t eos, uint32_t *filtered) {
    size_t w = 0;
    for (size_t i = 0; i < size / (size_t) n_out; i++) {
        int remove = 0;
        for (int ii = 0; ii < n_out; ii++) {
            size_t next_index = i * n_out + (size_t) ii * n_out + ii;
            if (next_index >= size || tokens[next_index] >= audio_vocab) { remove = 1; break; }
        }
        if (!remove) {
            for (int ii = 0; ii < n_out; ii++) {
                size_t next_index = i * n_out + (size_t) ii * n_out + ii;
                filtered[w++] = next_index > size ? eos : tokens[next_index];
            }
        }
    }
    return w;
}

/* ======================================================================================
 * sampler (src/sampler.cpp)
 * ==================================================================================== */
void orc_sampler_init(orc_sampler *s, uint32_t n_heads, uint32_t vocab) {
    memset(s, 0, sizeof(*s));
    s->n_output_heads = n_heads; s->vocab_size = vocab;
    s->temperature = 1.0f; s->top_p = 1.0f; s->repetition_penalty = 1.0f; s->top_k = 0; s->do_sample = 1;
}
void orc_sampler_reset(orc_sampler *s) { /* sampler.cpp:71-80 */
    if (s->repetition_penalty != 1.0f) {
        for (uint32_t i = 0; i < s->n_output_heads; i++) { s->last_token_ids[i] = -1; s->repetition_counts[i] = 0; }
        s->rep_initialised = 1;
    }
}
static float rep_div(const orc_sampler *s, float v, uint32_t head) {
    /* v /= pow(repetition_penalty, repetition_counts[i])  — pow promotes to double */
    return (float) ((double) v / pow((double) s->repetition_penalty, (double) s->repetition_counts[head]));
}
void orc_sampler_max(const orc_sampler *s, const float *logits, uint32_t *out) { /* sampler.cpp:185-204 */
    const int has_rep = s->repetition_penalty != 1.0f;
    for (uint32_t i = 0; i < s->n_output_heads; i++) {
        float mx = -INFINITY; uint32_t id = 0;
        for (uint32_t ii = 0; ii < s->vocab_size; ii++) {
            float v = logits[i * s->vocab_size + ii];
            if (has_rep && s->last_token_ids[i] == (int32_t) ii) v = rep_div(s, v, i);
            if (v > mx) { mx = v; id = ii; }
        }
        out[i] = id;
    }
}

typedef struct { float v; uint32_t idx; } vi_pair;
static int cmp_desc(const void *a, const void *b) {
    const vi_pair *x = (const vi_pair *) a, *y = (const vi_pair *) b;
    if (x->v > y->v) return -1;
    if (x->v < y->v) return 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx); /* ties: std::sort order is unspecified in the reference */
}

/* sampler.cpp:82-116.  picks==NULL -> whole vocab. */
static void sampler_softmax(const orc_sampler *s, float *logits, uint32_t *const *picks, const uint32_t *n_picks,
                            const uint32_t *max_idx) {
    const int has_rep = s->repetition_penalty != 1.0f, has_temp = s->temperature != 1.0f;
    const uint32_t V = s->vocab_size;
    for (uint32_t i = 0; i < s->n_output_heads; i++) {
        float cumsum = 0.0f;
        float max_val = logits[i * V + max_idx[i]];
        if (has_rep && s->last_token_ids[i] == (int32_t) max_idx[i]) max_val = rep_div(s, max_val, i);
        if (has_temp) max_val /= s->temperature;
        const uint32_t n = picks ? n_picks[i] : V;
        for (uint32_t j = 0; j < n; j++) {
            uint32_t ii = picks ? picks[i][j] : j;
            float v = logits[i * V + ii];
            if (has_rep && s->last_token_ids[i] == (int32_t) ii) v = rep_div(s, v, i);
            if (has_temp) v /= s->temperature;
            v = expf(v - max_val);
            cumsum += v;
            logits[i * V + ii] = v;
        }
        for (uint32_t j = 0; j < n; j++) {
            uint32_t ii = picks ? picks[i][j] : j;
            logits[i * V + ii] = logits[i * V + ii] / cumsum;
        }
    }
}

void orc_sampler_sample(orc_sampler *s, float *logits, const float *uniforms, uint32_t *out) {
    const uint32_t V = s->vocab_size, NH = s->n_output_heads;
    if (!s->do_sample) { orc_sampler_max(s, logits, out); return; } /* sampler.cpp:5-7 */
    uint32_t max_vals[ORC_MAX_HEADS];
    float max_head_probs[ORC_MAX_HEADS];
    uint32_t *picks[ORC_MAX_HEADS] = {0};
    uint32_t n_picks[ORC_MAX_HEADS] = {0};
    int have_picks = 0, use_nucleus = 0, performed_softmax = 0;
    const int has_rep = s->repetition_penalty != 1.0f;
    vi_pair *tmp = (vi_pair *) malloc((size_t) V * sizeof(vi_pair));

    orc_sampler_max(s, logits, max_vals);                                   /* :18 */
    if (s->top_p < 1.0f) { sampler_softmax(s, logits, NULL, NULL, max_vals); performed_softmax = 1; } /* :24-29 */
    if (s->top_k > 0 && s->top_k < V) {                                      /* :30-33, topk :152-183 */
        for (uint32_t i = 0; i < NH; i++) {
            for (uint32_t j = 0; j < V; j++) {
                float v = logits[i * V + j];
                if (!performed_softmax && has_rep && s->last_token_ids[i] == (int32_t) j) v = rep_div(s, v, i);
                tmp[j].v = v; tmp[j].idx = j;
            }
            qsort(tmp, V, sizeof(vi_pair), cmp_desc);
            picks[i] = (uint32_t *) malloc((size_t) V * 4);
            n_picks[i] = s->top_k;
            for (uint32_t j = 0; j < s->top_k; j++) picks[i][j] = tmp[j].idx;
        }
        have_picks = 1; use_nucleus = 1;
    }
    if (s->top_p >= 1.0f) { /* :35-38 */
        sampler_softmax(s, logits, have_picks ? picks : NULL, n_picks, max_vals);
        performed_softmax = 1;
    }
    if (s->top_p < 1.0f) { /* topp :118-150 */
        if (!have_picks) {
            for (uint32_t i = 0; i < NH; i++) {
                for (uint32_t j = 0; j < V; j++) { tmp[j].v = logits[i * V + j]; tmp[j].idx = j; }
                qsort(tmp, V, sizeof(vi_pair), cmp_desc);
                picks[i] = (uint32_t *) malloc((size_t) V * 4);
                n_picks[i] = V;
                for (uint32_t j = 0; j < V; j++) picks[i][j] = tmp[j].idx;
            }
            have_picks = 1;
        }
        for (uint32_t i = 0; i < NH; i++) {
            float prob_sum = 0.0f; int trim_to = -1;
            for (uint32_t ii = 0; ii < n_picks[i]; ii++) {
                prob_sum += logits[i * V + picks[i][ii]];
                if (prob_sum >= s->top_p) { trim_to = (int) ii + 1; break; }
            }
            max_head_probs[i] = prob_sum < s->top_p ? prob_sum : s->top_p;
            if (trim_to > 0) n_picks[i] = (uint32_t) trim_to;
        }
        use_nucleus = 1;
    }
    if (has_rep && !s->rep_initialised) orc_sampler_reset(s); /* :43-46 */

    for (uint32_t i = 0; i < NH; i++) { /* :49-68 */
        float assignment = s->top_p < 1.0f ? uniforms[i] * max_head_probs[i] : uniforms[i];
        float cumulative = 0.0f;
        const uint32_t n = use_nucleus ? n_picks[i] : V;
        uint32_t chosen = 0; int found = 0;
        for (uint32_t j = 0; j < n; j++) {
            uint32_t ii = use_nucleus ? picks[i][j] : j;
            cumulative += logits[i * V + ii];
            /* third clause reads picks[i].size() even when picks is empty in the reference (UB when
             * top_k==0 && top_p>=1); restated as "last candidate" which is its evident intent. */
            if (assignment <= cumulative || ii >= V + 1 || j >= n - 1) { chosen = ii; found = 1; break; }
        }
        if (!found) chosen = 0;
        if (has_rep) {
            if (s->last_token_ids[i] != (int32_t) chosen) s->repetition_counts[i] = 0;
            s->last_token_ids[i] = (int32_t) chosen;
            s->repetition_counts[i] += 1;
        }
        out[i] = chosen;
    }
    for (uint32_t i = 0; i < NH; i++) free(picks[i]);
    free(tmp);
}

/* ======================================================================================
 * DAC decoder.  Activations are [C][L] (length fastest), weights in PyTorch order
 * (Conv1d [Cout][Cin][K], ConvTranspose1d [Cin][Cout][K]) which is what the GGUF stores
 * with ne reversed.  fp32 accumulation, input-channel-major.
 * ==================================================================================== */
void orc_conv1d(const float *x, int cin, int64_t L, const float *w, const float *b, int cout, int K,
                int pad, int dil, float *y) {
    /* ggml_conv_1d(kernel, x, stride 1, pad, dil) + bias (dac_model.cpp:158-159) ; L_out = L for "same" pads */
    const int64_t Lout = L + 2 * pad - (int64_t) dil * (K - 1);
#pragma omp parallel for schedule(static)
    for (int co = 0; co < cout; co++) {
        float *yr = y + (size_t) co * Lout;
        for (int64_t t = 0; t < Lout; t++) yr[t] = b ? b[co] : 0.0f;
        for (int ci = 0; ci < cin; ci++) {
            const float *xr = x + (size_t) ci * L;
            for (int k = 0; k < K; k++) {
                const float wv = w[((size_t) co * cin + ci) * K + k];
                const int64_t off = (int64_t) k * dil - pad;
                int64_t t0 = off < 0 ? -off : 0;
                int64_t t1 = Lout;
                if (t1 + off > L) t1 = L - off;
                for (int64_t t = t0; t < t1; t++) yr[t] += wv * xr[t + off];
            }
        }
    }
}

void orc_conv_transpose1d(const float *x, int cin, int64_t L, const float *w, const float *b, int cout,
                          int K, int stride, int pad, float *y) {
    /* fork's ggml_conv_transpose_1d(kernel, x, stride, padding, 1, 0, 1)
     * (general_neural_audio_codec.cpp:153) == torch ConvTranspose1d(stride, padding):
     * y[co][ti*stride + k - pad] += x[ci][ti] * w[ci][co][k];  L_out = (L-1)*stride - 2*pad + K */
    const int64_t Lout = (L - 1) * stride - 2 * (int64_t) pad + K;
#pragma omp parallel for schedule(static)
    for (int co = 0; co < cout; co++) {
        float *yr = y + (size_t) co * Lout;
        for (int64_t t = 0; t < Lout; t++) yr[t] = b ? b[co] : 0.0f;
        for (int ci = 0; ci < cin; ci++) {
            const float *xr = x + (size_t) ci * L;
            const float *wr = w + ((size_t) ci * cout + co) * K;
            for (int64_t ti = 0; ti < L; ti++) {
                const float xv = xr[ti];
                const int64_t base = ti * stride - pad;
                for (int k = 0; k < K; k++) {
                    int64_t to = base + k;
                    if (to >= 0 && to < Lout) yr[to] += xv * wr[k];
                }
            }
        }
    }
}

/* snake_1d (util.cpp:96-101): a + sin(a*alpha)^2 * (1/alpha), no epsilon. */
void orc_snake(float *x, int C, int64_t L, const float *alpha) {
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; c++) {
        const float a = alpha[c];
        const float ra = 1.0f / a;
        float *xr = x + (size_t) c * L;
        for (int64_t t = 0; t < L; t++) {
            float sn = sinf(xr[t] * a);
            xr[t] = xr[t] + (sn * sn) * ra;
        }
    }
}

static void round_buf_f16(float *x, size_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t) n; i++) x[i] = orc_h2f(orc_f2h(x[i]));
}

int64_t orc_dac_decode(const orc_dac_model *m, const uint32_t *codes, int frames, float *pcm_out,
                       int stage, float *stage_out) {
    int64_t L = frames;
    /* f16_conv 0: exact fp32 (F32 codec tensors as this repository reads them); 1: F16 codec tensors — ggml_conv_1d's im2col is F16 and
     * conv_transpose_1d_f16_f32 converts its source, so every conv input is rounded to fp16; 2: the OTHER reading of an F32 model — upstream
     * ggml_conv_1d always builds an F16 im2col (general_neural_audio_codec.cpp:142,146, dac_model.cpp:158,164) and mul_mat then converts the
     * F32 kernel (src1) to the im2col's type: conv inputs AND conv kernels rounded to fp16 (the caller passes kernels already rounded), while
     * ggml_conv_transpose_1d with an F32 kernel stays exact fp32 (ht = 0) */
    const int h = m->f16_conv != 0;
    const int ht = m->f16_conv == 1;
    int C = m->latent;
    /* quantizer: dac_build_audio_inputs (dac_model.cpp:100-123) + build_quantize_layer
     * (general_neural_audio_codec.cpp:166-172): sum_i (out_proj_i * codebook_i[tok] + bias_i) */
    float *cur = (float *) malloc((size_t) C * L * 4);
    for (int i = 0; i < m->n_codebooks; i++) {
#pragma omp parallel for schedule(static)
        for (int c = 0; c < C; c++) {
            for (int64_t t = 0; t < L; t++) {
                const float *cb = m->codebook[i] + (size_t) codes[t * m->n_codebooks + i] * m->codebook_dim;
                float acc = 0.0f;
                for (int d = 0; d < m->codebook_dim; d++)
                    acc += m->out_proj_w[i][(size_t) c * m->codebook_dim + d] * (h ? orc_h2f(orc_f2h(cb[d])) : cb[d]);
                acc += m->out_proj_b[i][c];
                if (i == 0) cur[(size_t) c * L + t] = acc;
                else cur[(size_t) c * L + t] = cur[(size_t) c * L + t] + acc;
            }
        }
    }
    if (stage == 0 && stage_out) memcpy(stage_out, cur, (size_t) C * L * 4);

    /* initial conv k7 pad 3 (dac_model.cpp:158-159) */
    float *nxt = (float *) malloc((size_t) m->c0 * L * 4);
    if (h) round_buf_f16(cur, (size_t) C * L);
    orc_conv1d(cur, C, L, m->init_w, m->init_b, m->c0, 7, 3, 1, nxt);
    free(cur); cur = nxt; C = m->c0;
    if (stage == 1 && stage_out) memcpy(stage_out, cur, (size_t) C * L * 4);

    for (int bi = 0; bi < m->n_blocks; bi++) { /* build_layer (general_neural_audio_codec.cpp:151-164) */
        const orc_dac_block *b = &m->blocks[bi];
        orc_snake(cur, C, L, b->alpha);
        const int K = 2 * b->stride;
        const int64_t L2 = (L - 1) * b->stride - 2 * (int64_t) b->padding + K;
        nxt = (float *) malloc((size_t) b->cout * L2 * 4);
        if (ht) round_buf_f16(cur, (size_t) C * L);
        orc_conv_transpose1d(cur, C, L, b->w, b->b, b->cout, K, b->stride, b->padding, nxt);
        free(cur); cur = nxt; C = b->cout; L = L2;
        float *t1 = (float *) malloc((size_t) C * L * 4);
        float *t2 = (float *) malloc((size_t) C * L * 4);
        for (int r = 0; r < 3; r++) { /* build_residual_unit :133-149; pad=3^(r+1)?? no: pad 3^r*3, dil 3^r (header :44-48) */
            int dil = 1; for (int e = 0; e < r; e++) dil *= 3;
            const int pad = dil * 3;
            memcpy(t1, cur, (size_t) C * L * 4);
            orc_snake(t1, C, L, b->res[r].in_alpha);
            if (h) round_buf_f16(t1, (size_t) C * L);
            orc_conv1d(t1, C, L, b->res[r].in_w, b->res[r].in_b, C, 7, pad, dil, t2);
            orc_snake(t2, C, L, b->res[r].out_alpha);
            if (h) round_buf_f16(t2, (size_t) C * L);
            orc_conv1d(t2, C, L, b->res[r].out_w, b->res[r].out_b, C, 1, 0, 1, t1);
            for (size_t i = 0; i < (size_t) C * L; i++) cur[i] = t1[i] + cur[i];
        }
        free(t1); free(t2);
        if (stage == 2 + bi && stage_out) memcpy(stage_out, cur, (size_t) C * L * 4);
    }
    /* final snake, conv k7 -> 1 channel, tanh (dac_model.cpp:163-166) */
    orc_snake(cur, C, L, m->final_alpha);
    float *pcm = (float *) malloc((size_t) L * 4);
    if (h) round_buf_f16(cur, (size_t) C * L);
    orc_conv1d(cur, C, L, m->final_w, m->final_b, 1, 7, 3, 1, pcm);
    for (int64_t t = 0; t < L; t++) pcm_out[t] = tanhf(pcm[t]);
    free(pcm); free(cur);
    return L;
}


/* ======================================================================================
 * SNAC decoder (src/decoder/snac_model.cpp:86-159).
 * ==================================================================================== */
void orc_conv1d_dw(const float *x, int C, int64_t L, const float *w, const float *b, int K, int pad, int dil, float *y) {
    /* ggml_conv_1d_dw(kernel [K,1,C], x, stride 1, pad, dil) + bias: y[c][t] = b[c] + sum_k w[c][k] x[c][t + k*dil - pad] */
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; c++) {
        for (int64_t t = 0; t < L; t++) {
            float acc = b ? b[c] : 0.0f;
            for (int k = 0; k < K; k++) {
                const int64_t ti = t + (int64_t) k * dil - pad;
                if (ti >= 0 && ti < L) acc += w[(size_t) c * K + k] * x[(size_t) c * L + ti];
            }
            y[(size_t) c * L + t] = acc;
        }
    }
}

int64_t orc_snac_decode(const orc_snac_model *m, const uint32_t *codes, int T, const float *noise, float *pcm_out) {
    int64_t L = T;
    int C = m->latent;
    float *cur = (float *) calloc((size_t) C * L, 4);
    /* snac_build_audio_inputs (:86-108): per level quantize layer on T/repeat ids, repeat_interleave up to T, summed */
    size_t off = 0;
    for (int i = 0; i < m->n_codebooks; i++) {
        const int rep = m->repeats[i], n = T / rep;
        for (int c = 0; c < C; c++) {
            for (int j = 0; j < n; j++) {
                const float *cb = m->codebook[i] + (size_t) codes[off + j] * m->codebook_dim;
                float acc = 0.0f;
                for (int d = 0; d < m->codebook_dim; d++) acc += m->out_proj_w[i][(size_t) c * m->codebook_dim + d] * cb[d];
                acc += m->out_proj_b[i][c];
                for (int rr = 0; rr < rep; rr++) {
                    float *dst = &cur[(size_t) c * L + (size_t) j * rep + rr];
                    *dst = (i == 0) ? acc : (*dst + acc);
                }
            }
        }
        off += (size_t) n;
    }
    float *nxt = (float *) malloc((size_t) C * L * 4);
    orc_conv1d_dw(cur, C, L, m->in_w, m->in_b, 7, 3, 1, nxt);                         /* :141-142 */
    free(cur); cur = nxt;
    nxt = (float *) malloc((size_t) m->c0 * L * 4);
    orc_conv1d(cur, C, L, m->up_w, m->up_b, m->c0, 1, 0, 1, nxt);                     /* :143-144 */
    free(cur); cur = nxt; C = m->c0;
    size_t noise_off = 0;
    for (int bi = 0; bi < m->n_blocks; bi++) {                                        /* build_layer, gnac.cpp:151-164 */
        const orc_snac_block *b = &m->blocks[bi];
        orc_snake(cur, C, L, b->alpha);
        const int K = 2 * b->stride;
        const int64_t L2 = (L - 1) * b->stride - 2 * (int64_t) b->padding + K;
        nxt = (float *) malloc((size_t) b->cout * L2 * 4);
        orc_conv_transpose1d(cur, C, L, b->w, b->b, b->cout, K, b->stride, b->padding, nxt);
        free(cur); cur = nxt; C = b->cout; L = L2;
        float *t1 = (float *) malloc((size_t) C * L * 4), *t2 = (float *) malloc((size_t) C * L * 4);
        if (b->noise_w && noise) {                                                    /* :155-159 */
            orc_conv1d(cur, C, L, b->noise_w, NULL, C, 1, 0, 1, t1);
            for (int c = 0; c < C; c++)
                for (int64_t t = 0; t < L; t++) cur[(size_t) c * L + t] = cur[(size_t) c * L + t] + t1[(size_t) c * L + t] * noise[noise_off + (size_t) t];
        }
        noise_off += (size_t) L;
        for (int r = 0; r < 3; r++) {                                                 /* build_residual_unit :133-149, groups > 1 */
            int dil = 1; for (int e = 0; e < r; e++) dil *= 3;
            memcpy(t1, cur, (size_t) C * L * 4);
            orc_snake(t1, C, L, b->res[r].in_alpha);
            orc_conv1d_dw(t1, C, L, b->res[r].in_w, b->res[r].in_b, 7, 3 * dil, dil, t2);
            orc_snake(t2, C, L, b->res[r].out_alpha);
            orc_conv1d(t2, C, L, b->res[r].out_w, b->res[r].out_b, C, 1, 0, 1, t1);
            for (size_t i = 0; i < (size_t) C * L; i++) cur[i] = t1[i] + cur[i];
        }
        free(t1); free(t2);
    }
    orc_snake(cur, C, L, m->final_alpha);                                             /* :152-155 */
    float *pcm = (float *) malloc((size_t) L * 4);
    orc_conv1d(cur, C, L, m->final_w, m->final_b, 1, 7, 3, 1, pcm);
    for (int64_t t = 0; t < L; t++) pcm_out[t] = tanhf(pcm[t]);
    free(pcm); free(cur);
    return L;
}


/* ======================================================================================
 * Orpheus decoder (Llama-3 blocks, src/models/orpheus/model.cpp:186-296)
 * ==================================================================================== */
struct orc_orpheus_state {
    int L, n_ctx, kvH;
    float *k, *v; /* [L][n_ctx][kvH], un-repeated (the reference's repeat-interleaved copy is the same data 3x) */
};

orc_orpheus_state *orc_orpheus_state_new(const orc_orpheus_model *m) {
    orc_orpheus_state *s = (orc_orpheus_state *) calloc(1, sizeof(*s));
    s->L = m->L; s->n_ctx = m->n_ctx; s->kvH = m->n_kv_heads * m->head_dim;
    s->k = (float *) calloc((size_t) s->L * s->n_ctx * s->kvH, 4);
    s->v = (float *) calloc((size_t) s->L * s->n_ctx * s->kvH, 4);
    return s;
}
void orc_orpheus_state_free(orc_orpheus_state *s) { if (s) { free(s->k); free(s->v); free(s); } }

static void llama_rms_norm(const float *x, int H, const float *w, float *y) { /* :122-125, eps 1e-5 */
    double sum = 0.0;
    for (int i = 0; i < H; i++) sum += (double) (x[i] * x[i]);
    const float scale = 1.0f / sqrtf((float) (sum / H) + 1e-5f);
    for (int i = 0; i < H; i++) y[i] = x[i] * scale * w[i];
}

/* ggml_rope_ext(x, pos, freq_factors, n_dims = head_dim, mode 2 (NEOX pairs i, i + n/2), n_ctx_orig 0, freq_base 500000,
 * freq_scale 1, ext_factor 0, attn_factor 1, beta 0/0): upstream ggml_rope_cache_init walks theta = pos, theta *=
 * powf(base, -2/n) in fp32 and evaluates cosf/sinf of theta / freq_factor. */
static void neox_rope(float *x, int n_heads, int hd, uint32_t pos, const float *ff, float base) {
    const float theta_scale = powf(base, -2.0f / (float) hd);
    for (int h = 0; h < n_heads; h++) {
        float *v = x + (size_t) h * hd;
        float theta = (float) pos;
        for (int i = 0; i < hd / 2; i++) {
            const float ang = theta / (ff ? ff[i] : 1.0f);
            const float c = cosf(ang), sn = sinf(ang);
            const float x0 = v[i], x1 = v[i + hd / 2];
            v[i] = x0 * c - x1 * sn;
            v[i + hd / 2] = x0 * sn + x1 * c;
            theta *= theta_scale;
        }
    }
}
static void llama_rope(float *x, int n_heads, int hd, uint32_t pos, const float *ff) { neox_rope(x, n_heads, hd, pos, ff, 500000.0f); }

void orc_orpheus_decode(const orc_orpheus_model *m, orc_orpheus_state *s, const uint32_t *tokens, int n, uint32_t pos0,
                        float *logits_out, float *hidden_out) {
    const int H = m->H, F = m->F, NH = m->n_heads, NKV = m->n_kv_heads, hd = m->head_dim, kvH = NKV * hd, rep = NH / NKV;
    float *x = (float *) malloc((size_t) n * H * 4), *cur = (float *) malloc((size_t) n * H * 4);
    float *q = (float *) malloc((size_t) n * NH * hd * 4), *k = (float *) malloc((size_t) n * kvH * 4), *v = (float *) malloc((size_t) n * kvH * 4);
    float *att = (float *) malloc((size_t) n * NH * hd * 4), *tmp = (float *) malloc((size_t) n * H * 4);
    float *gate = (float *) malloc((size_t) n * F * 4), *up = (float *) malloc((size_t) n * F * 4);
    float *sc = (float *) malloc((size_t) (pos0 + n) * 4);
    const float scale = 1.0f / sqrtf((float) hd);
    for (int t = 0; t < n; t++) get_row(&m->embd, tokens[t], H, x + (size_t) t * H);
    for (int l = 0; l < m->L; l++) {
        const orc_orpheus_layer *ly = &m->layers[l];
        float *kc = s->k + (size_t) l * s->n_ctx * kvH, *vc = s->v + (size_t) l * s->n_ctx * kvH;
        for (int t = 0; t < n; t++) llama_rms_norm(x + (size_t) t * H, H, ly->input_norm, cur + (size_t) t * H);
        orc_mul_mat(ly->q.type, ly->q.data, H, NH * hd, cur, n, q, m->act_mode);
        orc_mul_mat(ly->k.type, ly->k.data, H, kvH, cur, n, k, m->act_mode);
        orc_mul_mat(ly->v.type, ly->v.data, H, kvH, cur, n, v, m->act_mode);
        for (int t = 0; t < n; t++) {
            llama_rope(q + (size_t) t * NH * hd, NH, hd, pos0 + t, m->rope_freqs);
            llama_rope(k + (size_t) t * kvH, NKV, hd, pos0 + t, m->rope_freqs);
            memcpy(kc + (size_t) (pos0 + t) * kvH, k + (size_t) t * kvH, (size_t) kvH * 4);
            memcpy(vc + (size_t) (pos0 + t) * kvH, v + (size_t) t * kvH, (size_t) kvH * 4);
        }
        for (int t = 0; t < n; t++) {
            const int T = (int) pos0 + t + 1; /* causal mask :329-337 */
            for (int h = 0; h < NH; h++) {
                const float *qv = q + ((size_t) t * NH + h) * hd;
                const int kh = h / rep;
                for (int j = 0; j < T; j++) {
                    const float *kv = kc + (size_t) j * kvH + kh * hd;
                    double d = 0.0;
                    for (int e = 0; e < hd; e++) d += (double) qv[e] * (double) kv[e];
                    sc[j] = (float) d;
                }
                softmax_scaled(sc, T, scale);
                float *o = att + ((size_t) t * NH + h) * hd;
                for (int e = 0; e < hd; e++) {
                    double a = 0.0;
                    for (int j = 0; j < T; j++) a += (double) sc[j] * (double) vc[(size_t) j * kvH + kh * hd + e];
                    o[e] = (float) a;
                }
            }
        }
        orc_mul_mat(ly->o.type, ly->o.data, NH * hd, H, att, n, tmp, m->act_mode);
        for (size_t i = 0; i < (size_t) n * H; i++) x[i] = tmp[i] + x[i];
        for (int t = 0; t < n; t++) llama_rms_norm(x + (size_t) t * H, H, ly->post_norm, cur + (size_t) t * H);
        orc_mul_mat(ly->gate.type, ly->gate.data, H, F, cur, n, gate, m->act_mode);
        orc_mul_mat(ly->up.type, ly->up.data, H, F, cur, n, up, m->act_mode);
        for (size_t i = 0; i < (size_t) n * F; i++) gate[i] = (gate[i] / (1.0f + expf(-gate[i]))) * up[i]; /* ggml_silu, fp32 (:279) */
        orc_mul_mat(ly->down.type, ly->down.data, F, H, gate, n, tmp, m->act_mode);
        for (size_t i = 0; i < (size_t) n * H; i++) x[i] = tmp[i] + x[i];
    }
    for (int t = 0; t < n; t++) llama_rms_norm(x + (size_t) t * H, H, m->out_norm, cur + (size_t) t * H);
    if (hidden_out) memcpy(hidden_out, cur, (size_t) n * H * 4);
    if (logits_out) orc_mul_mat(m->head.type, m->head.data, H, m->V, cur + (size_t) (n - 1) * H, 1, logits_out, m->act_mode);
    free(x); free(cur); free(q); free(k); free(v); free(att); free(tmp); free(gate); free(up); free(sc);
}


/* ======================================================================================
 * Dia (src/models/dia/model.cpp)
 * ==================================================================================== */
struct orc_dia_state {
    int L, max_ctx, max_gen, A, kvH;
    float *ck, *cv; /* cross K/V [L][2][max_ctx][A]; K rows >= prompt_size stay zero (:505-541) */
    float *k, *v;   /* self K/V [L][2][max_gen][kvH], un-repeated (the reference stores each group repeat-interleaved, :454-503) */
};

orc_dia_state *orc_dia_state_new(const orc_dia_model *m) {
    orc_dia_state *s = (orc_dia_state *) calloc(1, sizeof(*s));
    s->L = m->dec_L; s->max_ctx = m->max_ctx; s->max_gen = m->max_gen;
    s->A = m->dec_heads * m->head_dim; s->kvH = m->dec_kv_heads * m->head_dim;
    s->ck = (float *) calloc((size_t) s->L * 2 * s->max_ctx * s->A, 4);
    s->cv = (float *) calloc((size_t) s->L * 2 * s->max_ctx * s->A, 4);
    s->k = (float *) calloc((size_t) s->L * 2 * s->max_gen * s->kvH, 4);
    s->v = (float *) calloc((size_t) s->L * 2 * s->max_gen * s->kvH, 4);
    return s;
}
void orc_dia_state_free(orc_dia_state *s) { if (s) { free(s->ck); free(s->cv); free(s->k); free(s->v); free(s); } }

/* one query against n_keys rows of K/V ([n][ld] position-major), heads of hd, kv head = head / rep, scale 1.0, optional
 * additive mask row (:403, :586, :630) */
static void dia_attend(const float *q, const float *K, const float *V, int n_keys, int ld, int n_heads, int rep, int hd, const float *mask,
                       float *out, float *scores) {
    for (int h = 0; h < n_heads; h++) {
        const int kh = h / rep;
        for (int t = 0; t < n_keys; t++) {
            double acc = 0.0;
            const float *kr = K + (size_t) t * ld + kh * hd;
            for (int c = 0; c < hd; c++) acc += (double) kr[c] * (double) q[h * hd + c];
            scores[t] = (float) acc + (mask ? mask[t] : 0.0f);
        }
        softmax_scaled(scores, n_keys, 1.0f);
        for (int c = 0; c < hd; c++) {
            double acc = 0.0;
            for (int t = 0; t < n_keys; t++) acc += (double) scores[t] * (double) V[(size_t) t * ld + kh * hd + c];
            out[h * hd + c] = (float) acc;
        }
    }
}

static void silu_mul(float *gate, const float *up, size_t n) {
    for (size_t i = 0; i < n; i++) gate[i] = (gate[i] / (1.0f + expf(-gate[i]))) * up[i];
}

void orc_dia_encode(const orc_dia_model *m, orc_dia_state *s, const uint32_t *tokens, int sentence_len, float *enc_out) {
    const int H = m->enc_H, S = m->max_ctx, NH = m->enc_heads, hd = m->head_dim, A = NH * hd, F = m->enc_F, n = 2 * S;
    float *x = (float *) malloc((size_t) n * H * 4), *cur = (float *) malloc((size_t) n * H * 4), *tmp = (float *) malloc((size_t) n * H * 4);
    float *q = (float *) malloc((size_t) n * A * 4), *k = (float *) malloc((size_t) n * A * 4), *v = (float *) malloc((size_t) n * A * 4);
    float *att = (float *) malloc((size_t) n * A * 4);
    float *gate = (float *) malloc((size_t) n * F * 4), *up = (float *) malloc((size_t) n * F * 4);
    float *mask = (float *) malloc((size_t) S * S * 4);
    for (int i = 0; i < S; i++)   /* set_inputs :712-721 */
        for (int j = 0; j < S; j++)
            mask[(size_t) i * S + j] = (i < sentence_len) == (j < sentence_len) ? 0.0f : -INFINITY;
    for (int t = 0; t < S; t++) {
        get_row(&m->enc_embd, tokens[t], H, x + (size_t) t * H);  /* stream 0: the text */
        get_row(&m->enc_embd, 0, H, x + (size_t) (S + t) * H);    /* stream 1: all zeros (:700-703) */
    }
    for (int l = 0; l < m->enc_L; l++) {
        const orc_dia_enc_layer *ly = &m->enc[l];
        for (int t = 0; t < n; t++) llama_rms_norm(x + (size_t) t * H, H, ly->sa_norm, cur + (size_t) t * H);
        orc_mul_mat(ly->q.type, ly->q.data, H, A, cur, n, q, m->act_mode);
        orc_mul_mat(ly->k.type, ly->k.data, H, A, cur, n, k, m->act_mode);
        orc_mul_mat(ly->v.type, ly->v.data, H, A, cur, n, v, m->act_mode);
        for (int t = 0; t < n; t++) {
            neox_rope(q + (size_t) t * A, NH, hd, (uint32_t) (t % S), NULL, 10000.0f);
            neox_rope(k + (size_t) t * A, NH, hd, (uint32_t) (t % S), NULL, 10000.0f);
        }
#pragma omp parallel for schedule(static)
        for (int t = 0; t < n; t++) {
            float *sc = (float *) malloc((size_t) S * 4);
            const int b = t / S;
            dia_attend(q + (size_t) t * A, k + (size_t) b * S * A, v + (size_t) b * S * A, S, A, NH, 1, hd, mask + (size_t) (t % S) * S,
                       att + (size_t) t * A, sc);
            free(sc);
        }
        orc_mul_mat(ly->o.type, ly->o.data, A, H, att, n, tmp, m->act_mode);
        for (size_t i = 0; i < (size_t) n * H; i++) x[i] = tmp[i] + x[i];
        for (int t = 0; t < n; t++) llama_rms_norm(x + (size_t) t * H, H, ly->mlp_norm, cur + (size_t) t * H);
        orc_mul_mat(ly->gate.type, ly->gate.data, H, F, cur, n, gate, m->act_mode);
        orc_mul_mat(ly->up.type, ly->up.data, H, F, cur, n, up, m->act_mode);
        silu_mul(gate, up, (size_t) n * F);
        orc_mul_mat(ly->out.type, ly->out.data, F, H, gate, n, tmp, m->act_mode);
        for (size_t i = 0; i < (size_t) n * H; i++) x[i] = tmp[i] + x[i];
    }
    for (int t = 0; t < n; t++) llama_rms_norm(x + (size_t) t * H, H, m->enc_norm, cur + (size_t) t * H);
    if (enc_out) memcpy(enc_out, cur, (size_t) n * H * 4);
    /* cross K/V for every decoder layer (build_dia_cross_kv_store :505-541) */
    const int DA = s->A;
    float *kk = (float *) malloc((size_t) n * DA * 4);
    memset(s->ck, 0, (size_t) s->L * 2 * S * DA * 4);
    for (int l = 0; l < m->dec_L; l++) {
        const orc_dia_dec_layer *ly = &m->dec[l];
        float *ck = s->ck + (size_t) l * 2 * S * DA, *cv = s->cv + (size_t) l * 2 * S * DA;
        orc_mul_mat(ly->ck.type, ly->ck.data, H, DA, cur, n, kk, m->act_mode);
        for (int b = 0; b < 2; b++)
            for (int t = 0; t < sentence_len; t++) {
                float *row = kk + ((size_t) b * S + t) * DA;
                if (!m->no_cross_rope) neox_rope(row, m->dec_heads, hd, (uint32_t) t, NULL, 10000.0f);
                memcpy(ck + ((size_t) b * S + t) * DA, row, (size_t) DA * 4);
            }
        orc_mul_mat(ly->cv.type, ly->cv.data, H, DA, cur, n, cv, m->act_mode);
    }
    free(kk);
    free(x); free(cur); free(tmp); free(q); free(k); free(v); free(att); free(gate); free(up); free(mask);
}

void orc_dia_step(const orc_dia_model *m, orc_dia_state *s, const uint32_t *ids, uint32_t pos, float *logits_out, float *raw_out) {
    const int H = m->dec_H, NH = m->dec_heads, NKV = m->dec_kv_heads, hd = m->head_dim, A = NH * hd, kvH = NKV * hd, rep = NH / NKV;
    const int F = m->dec_F, S = m->max_ctx, G = m->max_gen, V = m->V, NO = m->n_out, n = 2;
    float *x = (float *) calloc((size_t) n * H, 4), *cur = (float *) malloc((size_t) n * H * 4), *tmp = (float *) malloc((size_t) n * H * 4);
    float *row = (float *) malloc((size_t) H * 4);
    float *q = (float *) malloc((size_t) n * A * 4), *k = (float *) malloc((size_t) n * kvH * 4), *v = (float *) malloc((size_t) n * kvH * 4);
    float *att = (float *) malloc((size_t) n * A * 4);
    float *gate = (float *) malloc((size_t) n * F * 4), *up = (float *) malloc((size_t) n * F * 4);
    float *sc = (float *) malloc((size_t) (S > (int) pos + 1 ? S : (int) pos + 1) * 4);
    float *raw = (float *) malloc((size_t) n * NO * V * 4), *hl = (float *) malloc((size_t) n * V * 4);
    /* build_dia_decoder_inp_embd :337-350: embds[0] first, then embds[i] + running; both streams get the same ids (:724-726) */
    for (int i = 0; i < NO; i++) {
        get_row(&m->dec_embd[i], ids[i], H, row);
        for (int e = 0; e < H; e++) x[e] = i == 0 ? row[e] : row[e] + x[e];
    }
    memcpy(x + H, x, (size_t) H * 4);
    for (int l = 0; l < m->dec_L; l++) {
        const orc_dia_dec_layer *ly = &m->dec[l];
        for (int t = 0; t < n; t++) llama_rms_norm(x + (size_t) t * H, H, ly->sa_norm, cur + (size_t) t * H);
        orc_mul_mat(ly->sq.type, ly->sq.data, H, A, cur, n, q, m->act_mode);
        orc_mul_mat(ly->sk.type, ly->sk.data, H, kvH, cur, n, k, m->act_mode);
        orc_mul_mat(ly->sv.type, ly->sv.data, H, kvH, cur, n, v, m->act_mode);
        for (int b = 0; b < n; b++) {
            float *kc = s->k + ((size_t) l * 2 + b) * G * kvH, *vc = s->v + ((size_t) l * 2 + b) * G * kvH;
            neox_rope(q + (size_t) b * A, NH, hd, pos, NULL, 10000.0f);
            neox_rope(k + (size_t) b * kvH, NKV, hd, pos, NULL, 10000.0f);
            memcpy(kc + (size_t) pos * kvH, k + (size_t) b * kvH, (size_t) kvH * 4);
            memcpy(vc + (size_t) pos * kvH, v + (size_t) b * kvH, (size_t) kvH * 4);
            dia_attend(q + (size_t) b * A, kc, vc, (int) pos + 1, kvH, NH, rep, hd, NULL, att + (size_t) b * A, sc);
        }
        orc_mul_mat(ly->so.type, ly->so.data, A, H, att, n, tmp, m->act_mode);
        for (size_t i = 0; i < (size_t) n * H; i++) x[i] = tmp[i] + x[i];
        for (int t = 0; t < n; t++) llama_rms_norm(x + (size_t) t * H, H, ly->ca_norm, cur + (size_t) t * H);
        orc_mul_mat(ly->cq.type, ly->cq.data, H, A, cur, n, q, m->act_mode);
        for (int b = 0; b < n; b++) {
            const float *ck = s->ck + ((size_t) l * 2 + b) * S * A, *cv = s->cv + ((size_t) l * 2 + b) * S * A;
            if (!m->no_cross_rope) neox_rope(q + (size_t) b * A, NH, hd, pos, NULL, 10000.0f);
            dia_attend(q + (size_t) b * A, ck, cv, S, A, NH, 1, hd, NULL, att + (size_t) b * A, sc);
        }
        orc_mul_mat(ly->co.type, ly->co.data, A, H, att, n, tmp, m->act_mode);
        for (size_t i = 0; i < (size_t) n * H; i++) x[i] = tmp[i] + x[i];
        for (int t = 0; t < n; t++) llama_rms_norm(x + (size_t) t * H, H, ly->mlp_norm, cur + (size_t) t * H);
        orc_mul_mat(ly->gate.type, ly->gate.data, H, F, cur, n, gate, m->act_mode);
        orc_mul_mat(ly->up.type, ly->up.data, H, F, cur, n, up, m->act_mode);
        silu_mul(gate, up, (size_t) n * F);
        orc_mul_mat(ly->out.type, ly->out.data, F, H, gate, n, tmp, m->act_mode);
        for (size_t i = 0; i < (size_t) n * H; i++) x[i] = tmp[i] + x[i];
    }
    for (int t = 0; t < n; t++) llama_rms_norm(x + (size_t) t * H, H, m->dec_norm, cur + (size_t) t * H);
    for (int i = 0; i < NO; i++) {
        orc_mul_mat(m->heads[i].type, m->heads[i].data, H, V, cur, n, hl, m->act_mode);   /* build_dia_head_outputs :366-380 */
        for (int b = 0; b < n; b++) memcpy(raw + ((size_t) b * NO + i) * V, hl + (size_t) b * V, (size_t) V * 4);
    }
    for (size_t i = 0; i < (size_t) NO * V; i++) {   /* cfg_scale, util.cpp:194-196 */
        const float cr = raw[i], ur = raw[(size_t) NO * V + i];
        logits_out[i] = cr + m->cfg_scale * (cr - ur);
    }
    if (raw_out) memcpy(raw_out, raw, (size_t) n * NO * V * 4);
    free(x); free(cur); free(tmp); free(row); free(q); free(k); free(v); free(att); free(gate); free(up); free(sc); free(raw); free(hl);
}

int orc_dia_check_stopping(uint32_t *ids, int n_out, const uint32_t *delay_pattern, uint32_t max_delay, uint32_t eos, uint32_t pad,
                           uint32_t current_position, uint32_t max_generation_size, int *delay_steps) {
    if (*delay_steps == -1 && (ids[0] == eos || current_position >= max_generation_size - max_delay)) *delay_steps = (int) max_delay;
    if (*delay_steps > 0) {
        const int step_after_eos = (int) max_delay - *delay_steps;
        for (int i = 0; i < n_out; i++) {
            if (step_after_eos == (int) delay_pattern[i]) ids[i] = eos;
            else if (step_after_eos > (int) delay_pattern[i]) ids[i] = pad;
        }
        *delay_steps -= 1;
    }
    return *delay_steps == 0;
}

size_t orc_dia_adjust_output_tokens(const uint32_t *tokens, size_t size, int n_out, const uint32_t *delay_pattern, uint32_t max_delay,
                                    uint32_t audio_vocab, uint32_t *filtered) {
    size_t n = 0;
    for (int i = 0; i < (int) (size / (size_t) n_out) - (int) max_delay; i++) {
        int skip = 0;
        for (int ii = 0; ii < n_out; ii++) {
            const size_t next = (size_t) i * n_out + (size_t) delay_pattern[ii] * n_out + ii;
            if (next > size || tokens[next] >= audio_vocab) { skip = 1; break; }
        }
        if (skip) continue;
        for (int ii = 0; ii < n_out; ii++) filtered[n++] = tokens[(size_t) i * n_out + (size_t) delay_pattern[ii] * n_out + ii];
    }
    return n;
}

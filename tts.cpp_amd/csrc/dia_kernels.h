// dia_kernels.h — the pieces of the Dia step that are not shared with the Orpheus decoder (llama_kernels.h: rms norm,
// NEOX rope + cache append, grouped-query attention with key ranges, silu*up) or the GEMMs (parler_kernels.h).
//   dia_embed_kernel   build_dia_decoder_inp_embd: sum of the n_out codebook embeddings, the same row for both streams
//                      (/root/reference/src/models/dia/model.cpp:337-350, set_inputs :724-726)
//   dia_cfg_kernel     cfg_scale map at the end of the graph: cond + scale * (cond - uncond) (src/util.cpp:175-200; the
//                      "r > max_output -> -inf" statement there is overwritten by the next one, so nothing is masked)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct DiaEmbedArgs {
    const float *table[16];   // [V][H] fp32 each
    const uint32_t *ids;      // [n_utt][n_out]
    int n_out, H;
    float *x;                 // [2 * n_utt][H]: rows 2u (text stream) and 2u+1 (all-zero twin) of utterance u get the same embedding
};

static __global__ __launch_bounds__(256) void dia_embed_kernel(DiaEmbedArgs a) {
    const int e = blockIdx.x * 256 + threadIdx.x, u = blockIdx.y;
    if (e >= a.H) return;
    float acc = 0.0f;
    for (int i = 0; i < a.n_out; i++) {   // embds[0] first, then embds[i] + running (:343-347)
        const float v = a.table[i][(int64_t) a.ids[u * a.n_out + i] * a.H + e];
        acc = i == 0 ? v : v + acc;
    }
    a.x[(int64_t) (2 * u) * a.H + e] = acc;
    a.x[(int64_t) (2 * u + 1) * a.H + e] = acc;
}

// raw [2 * n_utt][ld] (ld >= n: the fused heads are padded to a multiple of 16 rows) -> guided [n_utt][n]; blockIdx.y = utterance
static __global__ __launch_bounds__(256) void dia_cfg_kernel(const float *raw, int ld, int n, float scale, float *guided) {
    const int i = blockIdx.x * 256 + threadIdx.x, u = blockIdx.y;
    if (i >= n) return;
    const float cr = raw[(int64_t) (2 * u) * ld + i], ur = raw[(int64_t) (2 * u + 1) * ld + i];
    guided[(int64_t) u * n + i] = cr + scale * (cr - ur);
}

// [R][ld] fp32 -> [R][K] fp16 (round to nearest even): the rounding ggml_mul_mat applies to the activations of an F16-weight product
// (vec_dot_type conversion of src1), done once here so that the encoder's 2 x 1024 rows can go through gemm_tile_kernel
static __global__ __launch_bounds__(256) void rows_to_f16_kernel(const float *x, int ld, int K, int64_t n8, _Float16 *y) {
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;   // one thread = 8 consecutive values
    if (i >= n8) return;
    const int64_t r = i / (K >> 3), c8 = i - r * (K >> 3);
    const float4 a = *(const float4 *) (x + r * ld + c8 * 8), b = *(const float4 *) (x + r * ld + c8 * 8 + 4);
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    h8 h;
    h[0] = (_Float16) a.x; h[1] = (_Float16) a.y; h[2] = (_Float16) a.z; h[3] = (_Float16) a.w;
    h[4] = (_Float16) b.x; h[5] = (_Float16) b.y; h[6] = (_Float16) b.z; h[7] = (_Float16) b.w;
    *(h8 *) (y + r * K + c8 * 8) = h;
}

// ------------------------------------------------------------------------------------------------
// device-resident generation loop (dia_runner::generate_from_batch, dia/model.cpp:806-870 with check_stopping :767-785 and the
// delay-pattern feedback :795-803): one thread per utterance slot keeps what the host loop keeps — the n_out input ids, the position,
// the countdown, the sampled history — so that a step (pre-step, forward, guidance, sampler, post-step) replays as one hipGraph and
// neither logits nor ids cross PCIe inside the loop.
//   dia_prestep_kernel   check_stopping before each decode: start the countdown when head 0 produced EOS or the position reaches
//                        max_gen - max_delay; during it force EOS / PAD into the heads whose delay has passed; mark the utterance done
//                        when the countdown reaches 0.  Also publishes the 1-based sampler call index of this step.
//   dia_poststep_kernel  record the sampled ids, advance the position (both guidance rows), feed head i its id once pos > i, BOS before.
// A finished utterance keeps its rows in the step (lock-step shapes stay fixed): its position stays, its ids stay, nothing is recorded.
// ------------------------------------------------------------------------------------------------
struct DiaLoopArgs {
    int n_utt, n_out;
    uint32_t bos, eos, pad, max_delay, max_gen;
    uint32_t delay_pattern[16];
    uint32_t *ids;        // [n_utt][n_out] the step's input ids (audio_tokens of the host loop)
    uint32_t *pos;        // [2 * n_utt] position of rows 2u, 2u+1
    int32_t *delay;       // [n_utt] countdown, -1 = not started
    uint32_t *done;       // [n_utt]
    uint32_t *call;       // [n_utt] 1-based index of the sampler call this step makes (sample_kernel's row_step)
    const uint32_t *tok;  // [n_utt][n_out] ids the sampler produced this step
    uint32_t *hist;       // [n_utt][max_gen][n_out]
};

static __global__ void dia_prestep_kernel(DiaLoopArgs a) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= a.n_utt || a.done[u]) return;
    uint32_t *aud = a.ids + u * a.n_out;
    const uint32_t p = a.pos[2 * u];
    int d = a.delay[u];
    if (d == -1 && (aud[0] == a.eos || p >= a.max_gen - a.max_delay)) d = (int) a.max_delay;
    if (d > 0) {
        const int after = (int) a.max_delay - d;
        for (int i = 0; i < a.n_out; i++) {
            if (after == (int) a.delay_pattern[i]) aud[i] = a.eos;
            else if (after > (int) a.delay_pattern[i]) aud[i] = a.pad;
        }
        d -= 1;
    }
    a.delay[u] = d;
    if (d == 0) a.done[u] = 1;
    a.call[u] = p + 1;
}

static __global__ void dia_poststep_kernel(DiaLoopArgs a) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= a.n_utt || a.done[u]) return;
    const uint32_t p = a.pos[2 * u];
    for (int i = 0; i < a.n_out; i++) a.hist[((int64_t) u * a.max_gen + p) * a.n_out + i] = a.tok[u * a.n_out + i];
    const uint32_t np = p + 1;
    a.pos[2 * u] = np;
    a.pos[2 * u + 1] = np;
    for (int i = 0; i < a.n_out; i++) a.ids[u * a.n_out + i] = np > (uint32_t) i ? a.tok[u * a.n_out + i] : a.bos;
}

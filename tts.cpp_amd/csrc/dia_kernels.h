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

__global__ __launch_bounds__(256) void dia_embed_kernel(DiaEmbedArgs a) {
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
__global__ __launch_bounds__(256) void dia_cfg_kernel(const float *raw, int ld, int n, float scale, float *guided) {
    const int i = blockIdx.x * 256 + threadIdx.x, u = blockIdx.y;
    if (i >= n) return;
    const float cr = raw[(int64_t) (2 * u) * ld + i], ur = raw[(int64_t) (2 * u + 1) * ld + i];
    guided[(int64_t) u * n + i] = cr + scale * (cr - ur);
}

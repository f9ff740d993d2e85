// shim_decoder.h — launchers of shim_decoder.hip that shim_llama.hip reuses (their argument records come from parler_kernels.h).
#pragma once
#include "shim_internal.h"
#include "parler_kernels.h"

enum { DIA_STREAM_SLABS = 8 };   // slab budget of the Dia step buffers (di_qkv, di_q, di_gu, di_parts)
int stream_slices(const tts_hip_ctx *c, const W &w, int R, int max_slabs);
bool stream_fold_ok(const tts_hip_ctx *c, const W &w, int R, int max_slabs);
int run_qgemm(tts_hip_ctx *c, int kclass, const W &w, GemmArgs a, int pro, int epi);
int qstream_slab_rows(int R);
int qstream_slices(const tts_hip_ctx *c, const W &w, int R, int max_slabs);
int launch_qstream(tts_hip_ctx *c, int kclass, const W &w, int R, float *out, int ldo, int64_t slab_stride, int ks);
int run_gemm(tts_hip_ctx *c, int kclass, const W &w, GemmArgs a, int pro, int epi);

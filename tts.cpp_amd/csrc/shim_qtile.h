// shim_qtile.h — launchers of shim_qtile.hip (the tiled integer GEMM) for shim_decoder.hip.
#pragma once
#include "shim_internal.h"
#include "qgemm_tile_kernels.h"

struct QTileShape { int BM, BN, threads; };
enum { N_QTILE_SHAPES = 5 };
extern const QTileShape QTILE_SHAPES[N_QTILE_SHAPES];
int launch_qtile(tts_hip_ctx *c, const QTileArgs &qa, int epi, int shape, int ks);
int qtile_quant_rows(tts_hip_ctx *c, const float *x, int lda, int K, int R);   // fp32 rows -> Q8_0 blocks in c->aq / c->adT

// gemm_tile_kernels.h — LDS-tiled MFMA GEMM for forwards that carry many rows (lock-step utterances).
//
//   y[r][n] = sum_k W[n][k] * act[r][k]        W fp16 [N][K] (ggml ne=[K,N]), act fp16 [R][lda]
//
// Same operation as gemm16_kernel (ggml_mul_mat with an F16 weight: activations rounded to fp16, fp32
// accumulate — /root/reference/src/models/parler/model.cpp:544-546,571,583,594,601-603), different shape:
// gemm16_kernel gives 16 features to a workgroup and is right for 1..32 rows (pure weight streaming); with
// R >= 64 rows every one of its N/16 workgroups re-reads all R activation rows from L2 and the matrix pipe
// sees 16-feature slivers.  Here a workgroup owns a BM x BN tile of the output:
//   * both operands are K-contiguous, so a k-tile of 64 is 128 B per row; it is staged into LDS with
//     global_load_lds_dwordx4 (8 lanes fetch one whole 128-B line, a wave instruction 8 rows), two LDS
//     buffers, the loads of tile t+1 in flight under the MFMAs of tile t, one barrier per k-tile;
//   * the LDS image is XOR-swizzled (16-B chunk c of row r lives at chunk c ^ (r & 7)) by permuting the
//     per-lane SOURCE address — the LDS destination of global_load_lds is lane-linear — so that the
//     ds_read_b128 fragment reads (16 rows at one chunk column) are bank-conflict free;
//   * workgroup id -> tile is XCD-aware: ids are dealt to the 8 XCDs round-robin by the hardware, so the
//     kernel remaps id -> (xcd-major) virtual id and keeps all row tiles of one feature tile on one XCD:
//     a weight tile leaves HBM once and is re-read by the other row tiles from that XCD's L2;
//   * split-K (residual GEMMs, N = hidden size) writes fp32 slabs that the following ln_rows_kernel folds
//     into the residual stream in slab order (deterministic, no atomics) — as gemm16_kernel does.
// Fragment orientation as in gemm16_kernel: MFMA A operand = 16 features, B operand = 16 rows, so a lane ends
// up with 4 consecutive features of one row (a float4 store / one KV-cache append).
#pragma once
#include "parler_kernels.h"

struct TileMap {
    int m_tiles, n_tiles, k_slices;  // grid = ceil8(m_tiles * n_tiles * k_slices) workgroups
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// BK: fp16 elements per k-tile (64 = one 128-B line per row, 128 = two); S: LDS buffers (S-1 k-tiles in flight).
template <int BM, int BN, int WM, int WN, int BK, int S, int EPI>
__global__ __launch_bounds__(WM * WN * 64) void gemm_tile_kernel(GemmArgs a, TileMap tm) {
    constexpr int NW = WM * WN;
    constexpr int MI = BM / WM / 16, NI = BN / WN / 16;  // 16x16 fragments per wave: rows x features
    constexpr int ROWB = BK * 2;                         // bytes per row of a k-tile
    constexpr int CH = BK / 8;                           // 16-B chunks per row
    constexpr int RPG = 64 / CH;                         // rows one wave instruction stages (1 KiB)
    constexpr int STAGE = (BM + BN) * ROWB;              // bytes per LDS buffer: W tile then activation tile
    constexpr int GROUPS = (BM + BN) / RPG, GPW = (GROUPS + NW - 1) / NW, GPW_MIN = GROUPS / NW;
    static_assert(MI >= 1 && NI >= 1 && BM % 16 == 0 && BN % 16 == 0 && (BK == 64 || BK == 128) && S >= 2 && GPW_MIN >= 1, "tile / wave shape");
    static_assert((S - 2) * GPW_MIN < 64, "vmcnt range");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // ---- workgroup -> (row tile, feature tile, k slice), XCD-aware ---------------------------------
    const int total = tm.m_tiles * tm.n_tiles * tm.k_slices;
    const int per_xcd = (int) (gridDim.x >> 3);
    const int v = (int) (blockIdx.x & 7) * per_xcd + (int) (blockIdx.x >> 3);
    if (v >= total) return;
    const int mt = v % tm.m_tiles;
    const int nt = (v / tm.m_tiles) % tm.n_tiles;
    const int kz = v / (tm.m_tiles * tm.n_tiles);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r0 = mt * BM, n0 = nt * BN;
    const int kc = a.kchunk ? a.kchunk : a.K;
    const int k0 = kz * kc;
    const int n_kt = kc / BK;

    // ---- staging: wave w copies row groups w, w+NW, ... of [W tile | activation tile] ---------------
    // a wave instruction writes 1 KiB of LDS lane-linearly = RPG rows; lane -> (row srow, chunk lane % CH) of the
    // LDS image, which must hold source chunk (lane % CH) ^ (row & (CH-1)): the swizzle goes on the source address
    const int srow = lane / CH;
    const _Float16 *src[GPW];
#pragma unroll
    for (int i = 0; i < GPW; i++) {
        const int g = min(wave + i * NW, GROUPS - 1);
        const int rowl = g * RPG + srow;              // row of the [BN + BM]-row LDS image (BN is a multiple of 16)
        const int schunk = (lane % CH) ^ (rowl & (CH - 1));
        if (g < BN / RPG) {
            const int n = min(n0 + rowl, a.N - 1);
            src[i] = (const _Float16 *) a.W + (int64_t) n * a.K + k0 + schunk * 8;
        } else {
            const int r = min(r0 + rowl - BN, a.R - 1);
            src[i] = (const _Float16 *) a.A + (int64_t) r * a.lda + k0 + schunk * 8;
        }
    }
    auto stage = [&](int buf, int kt) {
#pragma unroll
        for (int i = 0; i < GPW; i++) {
            const int g = wave + i * NW;
            if (GROUPS % NW != 0 && g >= GROUPS) break;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (src[i] + kt * BK),
                                             (__attribute__((address_space(3))) void *) (smem + buf * STAGE + g * 1024), 16, 0, 0);
        }
    };

    // ---- fragment read offsets (bytes inside a buffer), chunk XOR applied per k-step -----------------
    const int fl = lane & 15, fq = lane >> 4;
    int woff[NI], aoff[MI];
#pragma unroll
    for (int ni = 0; ni < NI; ni++) woff[ni] = (wn * (BN / WN) + ni * 16 + fl) * ROWB;
#pragma unroll
    for (int mi = 0; mi < MI; mi++) aoff[mi] = (BN + wm * (BM / WM) + mi * 16 + fl) * ROWB;
    const int sw = fl & (CH - 1);  // every fragment row index is fl mod 16 (tile / wave offsets are multiples of 16)

    float4v acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ni++)
#pragma unroll
        for (int mi = 0; mi < MI; mi++) acc[ni][mi] = (float4v){0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) {
        const char *base = smem + buf * STAGE;
#pragma unroll
        for (int ks = 0; ks < BK / 32; ks++) {
            const int coff = ((ks * 4 + fq) ^ sw) * 16;
            half8 wf[NI], af[MI];
#pragma unroll
            for (int ni = 0; ni < NI; ni++) wf[ni] = *(const half8 *) (base + woff[ni] + coff);
#pragma unroll
            for (int mi = 0; mi < MI; mi++) af[mi] = *(const half8 *) (base + aoff[mi] + coff);
#pragma unroll
            for (int ni = 0; ni < NI; ni++)
#pragma unroll
                for (int mi = 0; mi < MI; mi++) acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
        }
    };

    // ---- S-buffer pipeline: tiles kt+1 .. kt+S-2 stay in flight across the barrier of iteration kt ---
#pragma unroll
    for (int t = 0; t < S - 1; t++)
        if (t < n_kt) stage(t, t);
    int cur = 0, nxt = S - 1;  // buffer of tile kt, buffer freed by tile kt-1 (= (kt + S - 1) % S)
    for (int kt = 0; kt < n_kt; kt++) {
        if (kt + S - 2 < n_kt) wait_vmcnt<(S - 2) * GPW_MIN>();   // tile kt (this wave's pieces) has landed
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();   // every wave's pieces landed; every wave is done reading buffer nxt
        if (kt + S - 1 < n_kt) stage(nxt, kt + S - 1);
        compute(cur);
        cur = cur + 1 == S ? 0 : cur + 1;
        nxt = nxt + 1 == S ? 0 : nxt + 1;
    }

    // ---- epilogue ------------------------------------------------------------------------------------
    // Loads first, then stores.  The per-fragment form (gemm_epilogue4 under `if (r < R && n < N)`) compiled to load, s_waitcnt vmcnt(0),
    // store for each of the NI * MI fragments: the residual read-modify-write and the KV-cache append (row_seq / row_pos) were eight
    // dependent round trips at the end of a 10-20 us kernel (profiles/tools/isa_serial_loads.py).
    if constexpr (EPI == EPI_RESID) {
        float4v xo[NI][MI];
#pragma unroll
        for (int ni = 0; ni < NI; ni++)
#pragma unroll
            for (int mi = 0; mi < MI; mi++) {
                const int n = min(n0 + wn * (BN / WN) + ni * 16 + fq * 4, a.N - 4);
                const int r = min(r0 + wm * (BM / WM) + mi * 16 + fl, a.R - 1);
                xo[ni][mi] = *(const float4v *) (a.out + (int64_t) r * a.ldo + n);
            }
#pragma unroll
        for (int ni = 0; ni < NI; ni++) {
            const int n = n0 + wn * (BN / WN) + ni * 16 + fq * 4;
#pragma unroll
            for (int mi = 0; mi < MI; mi++) {
                const int r = r0 + wm * (BM / WM) + mi * 16 + fl;
                if (r < a.R && n < a.N) {
                    float4v o = xo[ni][mi];
                    o += acc[ni][mi];  // ggml_add(cur, residual)
                    *(float4v *) (a.out + (int64_t) r * a.ldo + n) = o;
                }
            }
        }
    } else if constexpr (EPI == EPI_CROSS) {
        // The 64 x 64 tile is ONE head's query for 64 rows (BN = 64 = the head width, n0 = 64 head).  Cross-attention over the voice prompt
        // (cross_E <= 32 positions; soft_max_ext + mul_mat, parler/model.cpp:586-593) right here: the tile goes to LDS (it is spread over the
        // 2 x 4 waves' accumulators), the head's K_c / V_c slices beside it, then 16 lanes per row — 4 channels each, 32 rows per pass — compute the
        // scores (row16_sum), the softmax (redundantly per lane) and the attended channels.  attn_short_kernel's arithmetic up to the order of the
        // 64-term score sum (a 16-lane butterfly of 4-term partial sums here, a 64-lane butterfly there); q never goes to memory and the launch
        // between the GEMM and the out projection (10.8 us at 1024 rows, 0.08 of the HBM peak) is gone.
        static_assert(BM == 64 && BN == 64 && WM * WN == 8, "EPI_CROSS is written for the 64 x 64 tile of 8 waves");
        __syncthreads();                                   // every wave is done with the k-tile buffers: the LDS is reused
        float *qs = (float *) smem;                        // [64][68] (the pad spreads the rows over the banks)
        float *ks = qs + 64 * 68, *vs = ks + 32 * 64;      // [E][64] each
        const int E = a.cross_E;
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
            *(float4v *) (qs + (wm * (BM / WM) + mi * 16 + fl) * 68 + wn * (BN / WN) + fq * 4) = acc[0][mi];
        for (int i = tid; i < E * 16; i += NW * 64) {
            const int e = i >> 4, c4 = (i & 15) * 4;
            *(float4v *) (ks + e * 64 + c4) = *(const float4v *) (a.cross_k + (int64_t) e * a.H + n0 + c4);
            *(float4v *) (vs + e * 64 + c4) = *(const float4v *) (a.cross_v + (int64_t) e * a.H + n0 + c4);
        }
        __syncthreads();
        const int cl = tid & 15;
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
            const int rl = pass * 32 + (tid >> 4), r = r0 + rl;
            const float4v q4 = *(const float4v *) (qs + rl * 68 + cl * 4);
            float sc[32];
            float m = -INFINITY;
#pragma unroll
            for (int e = 0; e < 32; e++) {
                if (e < E) {
                    const float4v k4 = *(const float4v *) (ks + e * 64 + cl * 4);
                    float d = q4[0] * k4[0] + q4[1] * k4[1] + q4[2] * k4[2] + q4[3] * k4[3];
                    sc[e] = row16_sum(d) * a.cross_scale;
                    m = fmaxf(m, sc[e]);
                }
            }
            float l = 0.0f;
            float4v o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 32; e++) {
                if (e < E) {
                    const float p = expf(sc[e] - m);
                    const float4v v4 = *(const float4v *) (vs + e * 64 + cl * 4);
                    l += p;
#pragma unroll
                    for (int j = 0; j < 4; j++) o[j] += p * v4[j];
                }
            }
            if (r < a.R) {
                if (a.cross_out16) {
                    half4 h;
#pragma unroll
                    for (int j = 0; j < 4; j++) h[j] = (_Float16) (o[j] / l);
                    *(half4 *) (a.cross_out16 + (int64_t) r * a.H + n0 + cl * 4) = h;
                } else {
                    float4v res;
#pragma unroll
                    for (int j = 0; j < 4; j++) res[j] = o[j] / l;
                    *(float4v *) (a.cross_out + (int64_t) r * a.H + n0 + cl * 4) = res;
                }
            }
        }
    } else if constexpr (EPI == EPI_QKV) {
        int64_t rowoff[MI];   // this lane's rows in the cache: sequence and position
#pragma unroll
        for (int mi = 0; mi < MI; mi++) {
            const int r = min(r0 + wm * (BM / WM) + mi * 16 + fl, a.R - 1);
            rowoff[mi] = (int64_t) a.row_seq[r] * a.seq_stride + (int64_t) a.row_pos[r] * a.H;
        }
#pragma unroll
        for (int ni = 0; ni < NI; ni++) {
            const int n = n0 + wn * (BN / WN) + ni * 16 + fq * 4;
            const int which = n / a.H, c = n - which * a.H;   // wave-uniform per ni up to fq: a feature quad never straddles q / k / v (H % 4 == 0)
#pragma unroll
            for (int mi = 0; mi < MI; mi++) {
                const int r = r0 + wm * (BM / WM) + mi * 16 + fl;
                if (r < a.R && n < a.N) {
                    const float4v v = acc[ni][mi];
                    if (which == 0) {
                        *(float4v *) (a.q + (int64_t) r * a.H + c) = v;
                    } else {
                        void *base = which == 1 ? a.kc : a.vc;
                        if (a.kv_f16) {
                            half4 h;
#pragma unroll
                            for (int e = 0; e < 4; e++) h[e] = (_Float16) v[e];
                            *(half4 *) ((_Float16 *) base + rowoff[mi] + c) = h;
                        } else {
                            *(float4v *) ((float *) base + rowoff[mi] + c) = v;
                        }
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int ni = 0; ni < NI; ni++) {
            const int n = n0 + wn * (BN / WN) + ni * 16 + fq * 4;
#pragma unroll
            for (int mi = 0; mi < MI; mi++) {
                const int r = r0 + wm * (BM / WM) + mi * 16 + fl;
                if (r < a.R && n < a.N) gemm_epilogue4(a, EPI, r, n, acc[ni][mi], kz);
            }
        }
    }
}

// engine_bench.hip — round 5, VERDICT r4 item 2: does ONE persistent launch that keeps the next phase's weights in flight across the phase boundary
// beat a chain of launches for a batch-1 decode layer of Orpheus-3B size (Q4_0: 66 MB of weights in four matrix phases)?
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/engine_bench profiles/engine_bench.hip && /tmp/engine_bench [layers]
//
// The four weight phases of a Llama decode layer at the Orpheus-3B shapes (orpheus/model.cpp:230-283), Q4_0 codes + fp16 block scales in the
// product's repacked layout (codes [N][K/32][16], scales [N][K/32]), activations as int8 with a fixed scale, integer block dots by v_dot4 exactly
// like gemv_q4_rows_lds_kernel (gemv_kernels.h):
//     qkv   [5120 x 3072]  ->  o [3072 x 3072]  ->  gate|up [16384 x 3072] with silu(gate) * up  ->  down [3072 x 8192]
// (no rms norm, rope or attention: the experiment is about the weight stream and the phase boundaries; the attention of the product step adds two
// more boundaries of the same kind).  Every layer has its own weights (layers x 66 MB > the 256 MB infinity cache for layers >= 4).
//
//   A  launch chain:  one kernel per phase, 256 workgroups x 4 waves, hipGraph-captured, 4 launches per layer; each workgroup requests its
//                     activations, then ALL of its weights, then computes (the order the product's kernels have since round 5).
//   B  persistent:    ONE launch for all layers, the same 256 x 4 waves, the same per-wave work and arithmetic (bit-identical results).  The
//                     weights of the NEXT phase are requested (into registers, 16-64 KB per CU in flight) BEFORE the workgroup computes the
//                     current phase and goes to the phase barrier, so the stream never stops at a boundary.  Boundary = XCD-hierarchical counter
//                     barrier (group = blockIdx % 8; MI355X_MICROARCH.md "barrier-xcd"); activations cross it as write-through (sc1) stores and
//                     sc1 loads; every spin is bounded (a timeout sets an error word and every workgroup leaves).
//   C  barrier only:  B without any work: the price of a boundary in the persistent form.
// Prints us per layer for A and B, us per barrier for C, and checks B's final activations against A's bit for bit.
//   D  the chain at the product's grid shapes, with and without its rms-norm staging prologue.
//   E  the chain with every kernel also requesting the NEXT launch's weight share into the L2 of its own XCD (ENGINE_BENCH_ONLY_E=1: only A and E):
//      0.98 x A (profiles/r05/engine_bench_call14_l2_prefetch.txt) — a launch-chain kernel is not waiting for its first weights.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#include <stdint.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#include "../tts.cpp_amd/csrc/wave_ops.h"   // the product's DPP / permlane reductions (wave_sum, lanes32_max)
typedef int int4v __attribute__((ext_vector_type(4)));
#define AGENT __HIP_MEMORY_SCOPE_AGENT

constexpr int H = 3072, QKV = 5120, F = 8192;
constexpr int NWG = 256, NTH = 256, NWAVES = NWG * 4;

struct LayerW {   // one layer's four matrices
    const uint8_t *w4[4];
    const _Float16 *wd[4];
};
struct Bufs {
    float *y_qkv, *y_o, *y_gu, *y_down;   // fp32 outputs of the four phases (y_down doubles as the next layer's input)
};
struct Sync {   // one counter per 128-byte line
    unsigned grp[8][32], top[32], gen[8][32], err[32];
};


// ---- one item: NF features x NP passes of 64 blocks, requested in one batch, consumed later ---------------------------------------------
template <int NF, int NP>
struct Frag {
    int4v wn[NF][NP];
    _Float16 dw[NF][NP];
};
template <int NF, int NP>
__device__ __forceinline__ void frag_load(Frag<NF, NP> &fr, const uint8_t *w4, const _Float16 *wd, const int (&nf)[NF], int K, int lane) {
    const int nb = K >> 5;
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const int b = min(lane + 64 * p, nb - 1);
#pragma unroll
        for (int f = 0; f < NF; f++) {
            fr.wn[f][p] = __builtin_nontemporal_load((const int4v *) (w4 + (int64_t) nf[f] * (K >> 1) + b * 16));
            fr.dw[f][p] = wd[(int64_t) nf[f] * nb + b];
        }
    }
}
template <int NF, int NP>
__device__ __forceinline__ void frag_dot(const Frag<NF, NP> &fr, const int8_t *sx, int K, int lane, float (&out)[NF]) {
    const int nb = K >> 5;
    float acc[NF];
#pragma unroll
    for (int f = 0; f < NF; f++) acc[f] = 0.0f;
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const int b = lane + 64 * p;
        if (b < nb) {
            const int4v x0 = *(const int4v *) (sx + b * 32), x1 = *(const int4v *) (sx + b * 32 + 16);
            int sxs = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                sxs = __builtin_amdgcn_sdot4(0x01010101, x0[e], sxs, false);
                sxs = __builtin_amdgcn_sdot4(0x01010101, x1[e], sxs, false);
            }
#pragma unroll
            for (int f = 0; f < NF; f++) {
                int s = 0;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int lo = fr.wn[f][p][e] & 0x0F0F0F0F, hi = (fr.wn[f][p][e] >> 4) & 0x0F0F0F0F;
                    s = __builtin_amdgcn_sdot4(lo, x0[e], s, false);
                    s = __builtin_amdgcn_sdot4(hi, x1[e], s, false);
                }
                acc[f] += (float) (s - 8 * sxs) * (float) fr.dw[f][p];
            }
        }
    }
#pragma unroll
    for (int f = 0; f < NF; f++) out[f] = wave_sum(acc[f]);
}

// activations: K floats -> int8 with a fixed scale, into LDS.  COHERENT: the floats were written by other workgroups of THIS launch (sc1 loads)
template <bool COHERENT, int K>
__device__ __forceinline__ void gather_x(const float *y, float scale, int8_t *sx, int tid) {
    float v[K / NTH];   // every load requested before the first is used (a loop of load-use pairs is K / 256 dependent L2 round trips)
#pragma unroll
    for (int j = 0; j < K / NTH; j++) v[j] = COHERENT ? __hip_atomic_load(y + tid + j * NTH, __ATOMIC_RELAXED, AGENT) : y[tid + j * NTH];
#pragma unroll
    for (int j = 0; j < K / NTH; j++) sx[tid + j * NTH] = (int8_t) fminf(fmaxf(rintf(v[j] * scale), -127.0f), 127.0f);
}
template <bool COHERENT>
__device__ __forceinline__ void put(float *p, float v) {
    if (COHERENT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, AGENT);   // write-through: visible to the other XCDs once vmcnt drains
    else *p = v;
}

// feature lists of this wave's items
constexpr int FQ = QKV / NWAVES, FO = H / NWAVES, FG = F / NWAVES / 2, FD = H / NWAVES;   // 5, 3, 4 (x 2 items, gate + up rows each), 3
static_assert(FQ * NWAVES == QKV && FO * NWAVES == H && FG * 2 * NWAVES == F && FD * NWAVES == H, "shapes");
constexpr float S_IN = 1.0f, S_O = 1.0f, S_GU = 1.0f, S_DOWN = 0.02f;

__device__ __forceinline__ void rows_qkv(int gw, int (&nf)[FQ]) {
#pragma unroll
    for (int f = 0; f < FQ; f++) nf[f] = gw * FQ + f;
}
__device__ __forceinline__ void rows_o(int gw, int (&nf)[FO]) {
#pragma unroll
    for (int f = 0; f < FO; f++) nf[f] = gw * FO + f;
}
__device__ __forceinline__ void rows_gu(int gw, int half, int (&nf)[2 * FG]) {   // FG gate rows and the matching FG up rows
#pragma unroll
    for (int f = 0; f < FG; f++) { nf[f] = (gw * 2 + half) * FG + f; nf[FG + f] = F + nf[f]; }
}

// ---- the four phases as functions of (weights already in registers, activations in LDS): results in lane-uniform registers --------------
template <int NF, int NP>
__device__ __forceinline__ void calc_plain(const Frag<NF, NP> &fr, const int8_t *sx, int K, int lane, float (&r)[NF]) {
    float o[NF];
    frag_dot<NF, NP>(fr, sx, K, lane, o);
#pragma unroll
    for (int f = 0; f < NF; f++) r[f] = o[f] * (1.0f / 256.0f);
}
__device__ __forceinline__ void calc_gu(const Frag<2 * FG, 2> &fr, const int8_t *sx, int lane, float (&r)[FG]) {
    float o[2 * FG];
    frag_dot<2 * FG, 2>(fr, sx, H, lane, o);
#pragma unroll
    for (int f = 0; f < FG; f++) {
        const float g = o[f] * (1.0f / 256.0f), u = o[FG + f] * (1.0f / 256.0f);
        r[f] = (g / (1.0f + expf(-g))) * u;
    }
}
template <int NF>
__device__ __forceinline__ void store_rows(float *dst, const float (&r)[NF], int lane) {   // dst: global (plain) or LDS
    if (lane == 0) {
#pragma unroll
        for (int f = 0; f < NF; f++) dst[f] = r[f];
    }
}

// ---- A: one kernel per phase ----------------------------------------------------------------------------------------------------------
template <int PH>
__global__ __launch_bounds__(NTH) void phase_kernel(LayerW w, Bufs b) {
    __shared__ __attribute__((aligned(16))) int8_t sx[F];
    const int tid = threadIdx.x, lane = tid & 63, gw = blockIdx.x * 4 + (tid >> 6);
    // activations first, weights right behind them (vmcnt retires in issue order), then the arithmetic
    const float *src = PH == 0 ? b.y_down : PH == 1 ? b.y_qkv : PH == 2 ? b.y_o : b.y_gu;
    const int K = PH == 3 ? F : H;
    const float sc = PH == 0 ? S_IN : PH == 1 ? S_O : PH == 2 ? S_GU : S_DOWN;
    constexpr int KJ = (PH == 3 ? F : H) / NTH;
    float v[KJ];
#pragma unroll
    for (int j = 0; j < KJ; j++) v[j] = src[tid + j * NTH];
    __builtin_amdgcn_sched_barrier(0);
    Frag<FQ, 2> fq; Frag<FO, 2> fo; Frag<2 * FG, 2> fa, fb; Frag<FD, 4> fd;
    if (PH == 0) { int nf[FQ]; rows_qkv(gw, nf); frag_load<FQ, 2>(fq, w.w4[0], w.wd[0], nf, H, lane); }
    if (PH == 1) { int nf[FO]; rows_o(gw, nf); frag_load<FO, 2>(fo, w.w4[1], w.wd[1], nf, H, lane); }
    if (PH == 2) {
        int nf[2 * FG];
        rows_gu(gw, 0, nf); frag_load<2 * FG, 2>(fa, w.w4[2], w.wd[2], nf, H, lane);
        rows_gu(gw, 1, nf); frag_load<2 * FG, 2>(fb, w.w4[2], w.wd[2], nf, H, lane);
    }
    if (PH == 3) { int nf[FD]; rows_o(gw, nf); frag_load<FD, 4>(fd, w.w4[3], w.wd[3], nf, F, lane); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < KJ; j++) sx[tid + j * NTH] = (int8_t) fminf(fmaxf(rintf(v[j] * sc), -127.0f), 127.0f);
    __syncthreads();
    if (PH == 0) { float r[FQ]; calc_plain<FQ, 2>(fq, sx, H, lane, r); store_rows<FQ>(b.y_qkv + gw * FQ, r, lane); }
    if (PH == 1) { float r[FO]; calc_plain<FO, 2>(fo, sx, H, lane, r); store_rows<FO>(b.y_o + gw * FO, r, lane); }
    if (PH == 2) {
        float r[FG];
        calc_gu(fa, sx, lane, r); store_rows<FG>(b.y_gu + (gw * 2 + 0) * FG, r, lane);
        calc_gu(fb, sx, lane, r); store_rows<FG>(b.y_gu + (gw * 2 + 1) * FG, r, lane);
    }
    if (PH == 3) { float r[FD]; calc_plain<FD, 4>(fd, sx, F, lane, r); store_rows<FD>(b.y_down + gw * FD, r, lane); }
}

__global__ void fill_x(float *p, int n);
// ---- E: A, and every kernel also asks for the NEXT launch's weights: into the L2 of its own XCD ------------------------------------------------
// A launch-chain kernel starts with nothing of its own in any cache: ~1 us of HBM latency + the transfer of its share before the first dot product.
// The previous kernel knows the addresses.  MODE 1: wave (blockIdx, w) of launch k touches every 64-byte piece of what wave (blockIdx, w) of launch
// k + 1 will stream (plain cached loads of one dword per lane, lane stride 64 B: 4 KB per instruction and ONE register, consumed by a never-true
// test at the end).  Workgroups go to the XCDs round-robin by blockIdx (checked by xcc_kernel below), so the lines wait in the L2 the consumer will
// ask.  MODE 2: the same requests shifted by one workgroup (the lines land in ANOTHER XCD's L2: what the memory-side cache alone gives).
// CAP: bytes of each region's head that are requested (gate|up is 3.1 MB per XCD in full, beside a 4 MB L2).
template <int BYTES, int NV>
__device__ __forceinline__ void pf_region(const void *p, int lane, unsigned (&v)[NV], int &n) {
#pragma unroll
    for (int i = 0; i < (BYTES + 4095) / 4096; i++) {
        const int off = min(lane * 64 + i * 4096, BYTES - 4);
        v[n++] = *(const unsigned *) ((const uint8_t *) p + off);
    }
}
constexpr int cmin(int a, int b) { return a < b ? a : b; }
template <int PH, int MODE, int CAP>
__global__ __launch_bounds__(NTH) void phase_kernel_e(LayerW w, LayerW wn, Bufs b) {
    __shared__ __attribute__((aligned(16))) int8_t sx[F];
    const int tid = threadIdx.x, lane = tid & 63, gw = blockIdx.x * 4 + (tid >> 6);
    const float *src = PH == 0 ? b.y_down : PH == 1 ? b.y_qkv : PH == 2 ? b.y_o : b.y_gu;
    const float sc = PH == 0 ? S_IN : PH == 1 ? S_O : PH == 2 ? S_GU : S_DOWN;
    constexpr int KJ = (PH == 3 ? F : H) / NTH;
    float v[KJ];
#pragma unroll
    for (int j = 0; j < KJ; j++) v[j] = src[tid + j * NTH];
    __builtin_amdgcn_sched_barrier(0);
    Frag<FQ, 2> fq; Frag<FO, 2> fo; Frag<2 * FG, 2> fa, fb; Frag<FD, 4> fd;
    if (PH == 0) { int nf[FQ]; rows_qkv(gw, nf); frag_load<FQ, 2>(fq, w.w4[0], w.wd[0], nf, H, lane); }
    if (PH == 1) { int nf[FO]; rows_o(gw, nf); frag_load<FO, 2>(fo, w.w4[1], w.wd[1], nf, H, lane); }
    if (PH == 2) {
        int nf[2 * FG];
        rows_gu(gw, 0, nf); frag_load<2 * FG, 2>(fa, w.w4[2], w.wd[2], nf, H, lane);
        rows_gu(gw, 1, nf); frag_load<2 * FG, 2>(fb, w.w4[2], w.wd[2], nf, H, lane);
    }
    if (PH == 3) { int nf[FD]; rows_o(gw, nf); frag_load<FD, 4>(fd, w.w4[3], w.wd[3], nf, F, lane); }
    __builtin_amdgcn_sched_barrier(0);
    // the next launch's share, behind this launch's own weights in the queue
    unsigned pfv[16];
    int npf = 0;
    if (MODE) {
        const int pg = MODE == 1 ? gw : (gw + 4) % NWAVES;   // MODE 2: the neighbour workgroup's share = another XCD's L2
        if (PH == 3) {   // next: qkv of the next layer
            pf_region<cmin(FQ * (H / 2), CAP)>(wn.w4[0] + (int64_t) pg * FQ * (H / 2), lane, pfv, npf);
            pf_region<FQ * (H / 32) * 2>(wn.wd[0] + (int64_t) pg * FQ * (H / 32), lane, pfv, npf);
        }
        if (PH == 0) {
            pf_region<cmin(FO * (H / 2), CAP)>(w.w4[1] + (int64_t) pg * FO * (H / 2), lane, pfv, npf);
            pf_region<FO * (H / 32) * 2>(w.wd[1] + (int64_t) pg * FO * (H / 32), lane, pfv, npf);
        }
        if (PH == 1) {   // gate rows of both items are neighbours, so are the up rows
            pf_region<cmin(2 * FG * (H / 2), CAP)>(w.w4[2] + (int64_t) pg * 2 * FG * (H / 2), lane, pfv, npf);
            pf_region<cmin(2 * FG * (H / 2), CAP)>(w.w4[2] + ((int64_t) F + pg * 2 * FG) * (H / 2), lane, pfv, npf);
            pf_region<2 * FG * (H / 32) * 2>(w.wd[2] + (int64_t) pg * 2 * FG * (H / 32), lane, pfv, npf);
            pf_region<2 * FG * (H / 32) * 2>(w.wd[2] + ((int64_t) F + pg * 2 * FG) * (H / 32), lane, pfv, npf);
        }
        if (PH == 2) {
            pf_region<cmin(FD * (F / 2), CAP)>(w.w4[3] + (int64_t) pg * FD * (F / 2), lane, pfv, npf);
            pf_region<FD * (F / 32) * 2>(w.wd[3] + (int64_t) pg * FD * (F / 32), lane, pfv, npf);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < KJ; j++) sx[tid + j * NTH] = (int8_t) fminf(fmaxf(rintf(v[j] * sc), -127.0f), 127.0f);
    __syncthreads();
    if (PH == 0) { float r[FQ]; calc_plain<FQ, 2>(fq, sx, H, lane, r); store_rows<FQ>(b.y_qkv + gw * FQ, r, lane); }
    if (PH == 1) { float r[FO]; calc_plain<FO, 2>(fo, sx, H, lane, r); store_rows<FO>(b.y_o + gw * FO, r, lane); }
    if (PH == 2) {
        float r[FG];
        calc_gu(fa, sx, lane, r); store_rows<FG>(b.y_gu + (gw * 2 + 0) * FG, r, lane);
        calc_gu(fb, sx, lane, r); store_rows<FG>(b.y_gu + (gw * 2 + 1) * FG, r, lane);
    }
    if (PH == 3) { float r[FD]; calc_plain<FD, 4>(fd, sx, F, lane, r); store_rows<FD>(b.y_down + gw * FD, r, lane); }
    if (MODE) {   // the requested dwords are "used": never true, and the compiler cannot know
        unsigned any = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) if (i < npf) any |= pfv[i];
        if (any == 0x5EEDF00Du && blockIdx.x == 0x7FFFFFFFu) b.y_qkv[0] = 1.0f;
    }
}
template <int MODE, int CAP>
static double time_chain_e(const std::vector<LayerW> &hl, Bufs b, hipStream_t st, int reps, float *result) {
    hipGraph_t g; hipGraphExec_t ge;
    const int L = (int) hl.size();
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int l = 0; l < L; l++) {
        const LayerW &wn = hl[(l + 1) % L];
        hipLaunchKernelGGL((phase_kernel_e<0, MODE, CAP>), dim3(NWG), dim3(NTH), 0, st, hl[l], wn, b);
        hipLaunchKernelGGL((phase_kernel_e<1, MODE, CAP>), dim3(NWG), dim3(NTH), 0, st, hl[l], wn, b);
        hipLaunchKernelGGL((phase_kernel_e<2, MODE, CAP>), dim3(NWG), dim3(NTH), 0, st, hl[l], wn, b);
        hipLaunchKernelGGL((phase_kernel_e<3, MODE, CAP>), dim3(NWG), dim3(NTH), 0, st, hl[l], wn, b);
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    fill_x<<<(H + 255) / 256, 256, 0, st>>>(b.y_down, H);
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    CK(hipMemcpy(result, b.y_down, H * 4, hipMemcpyDeviceToHost));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; i++) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return ms * 1e3 / reps / L;
}
__global__ void xcc_kernel(unsigned *out) {   // which XCD a workgroup runs on
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        out[blockIdx.x] = x & 0xF;
    }
}

// ---- D: the launch chain again, with the two things the product's kernels have and A does not ----------------------------------------------
// NF = features (gate|up: output pairs) per wave and item: fewer per wave = more, smaller workgroups, each repeating the staging prologue (the product
// launches 640 / 384 / 512 / 384 workgroups for the four phases).  RMS = the staging prologue of the product's qkv and gate|up kernels: every
// workgroup reads the row AND a norm weight, reduces the sum of squares over the workgroup (two barriers) and quantises with a per-block maximum
// (stage_rms_q8 of gemv_kernels.h; the block scales are computed and stored but the dot keeps the fixed scale: timing only, results differ from A).
template <int PH, int NF, bool RMS>
__global__ __launch_bounds__(NTH) void phase_kernel_d(LayerW w, Bufs b, const float *normw) {
    __shared__ __attribute__((aligned(16))) int8_t sx[F];
    __shared__ float sd[F / 32], red[4];
    const int tid = threadIdx.x, lane = tid & 63, gw = blockIdx.x * 4 + (tid >> 6);
    const float *src = PH == 0 ? b.y_down : PH == 1 ? b.y_qkv : PH == 2 ? b.y_o : b.y_gu;
    constexpr int K = PH == 3 ? F : H, KJ = K / NTH, NP = PH == 3 ? 4 : 2;
    const float sc = PH == 0 ? S_IN : PH == 1 ? S_O : PH == 2 ? S_GU : S_DOWN;
    float v[KJ], nw[KJ];
#pragma unroll
    for (int j = 0; j < KJ; j++) { v[j] = src[tid + j * NTH]; if (RMS) nw[j] = normw[tid + j * NTH]; }
    __builtin_amdgcn_sched_barrier(0);
    constexpr int ROWS = PH == 2 ? 2 * NF : NF;
    Frag<ROWS, NP> fa, fb;
    int nf[ROWS];
    if constexpr (PH == 2) {
#pragma unroll
        for (int f = 0; f < NF; f++) { nf[f] = (gw * 2 + 0) * NF + f; nf[NF + f] = F + nf[f]; }
        frag_load<ROWS, NP>(fa, w.w4[2], w.wd[2], nf, K, lane);
#pragma unroll
        for (int f = 0; f < NF; f++) { nf[f] = (gw * 2 + 1) * NF + f; nf[NF + f] = F + nf[f]; }
        frag_load<ROWS, NP>(fb, w.w4[2], w.wd[2], nf, K, lane);
    } else {
#pragma unroll
        for (int f = 0; f < NF; f++) nf[f] = gw * NF + f;
        frag_load<ROWS, NP>(fa, w.w4[PH], w.wd[PH], nf, K, lane);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (RMS) {
        float ss = 0.0f;
#pragma unroll
        for (int j = 0; j < KJ; j++) ss += v[j] * v[j];
        ss = wave_sum(ss);
        if (lane == 0) red[tid >> 6] = ss;
        __syncthreads();
        ss = (red[0] + red[1]) + (red[2] + red[3]);
        const float scale = 1.0f / sqrtf(ss / (float) K + 1e-5f);
#pragma unroll
        for (int j = 0; j < KJ; j++) {
            const float o = v[j] * scale * nw[j];
            const float amax = lanes32_max(fabsf(o));
            const float dd = amax / 127.0f, id = dd ? 1.0f / dd : 0.0f;
            sx[tid + j * NTH] = (int8_t) rintf(o * id);
            if ((tid & 31) == 0) sd[(tid + j * NTH) >> 5] = (float) (_Float16) dd;
        }
    } else {
#pragma unroll
        for (int j = 0; j < KJ; j++) sx[tid + j * NTH] = (int8_t) fminf(fmaxf(rintf(v[j] * sc), -127.0f), 127.0f);
    }
    __syncthreads();
    float *dst = PH == 0 ? b.y_qkv : PH == 1 ? b.y_o : PH == 2 ? b.y_gu : b.y_down;
    if constexpr (PH == 2) {
        float o[ROWS], r[NF];
        frag_dot<ROWS, NP>(fa, sx, K, lane, o);
#pragma unroll
        for (int f = 0; f < NF; f++) { const float g = o[f] * (1.0f / 256.0f), u = o[NF + f] * (1.0f / 256.0f); r[f] = (g / (1.0f + expf(-g))) * u * (RMS ? sd[0] * 0.0f + 1.0f : 1.0f); }
        store_rows<NF>(dst + (gw * 2 + 0) * NF, r, lane);
        frag_dot<ROWS, NP>(fb, sx, K, lane, o);
#pragma unroll
        for (int f = 0; f < NF; f++) { const float g = o[f] * (1.0f / 256.0f), u = o[NF + f] * (1.0f / 256.0f); r[f] = (g / (1.0f + expf(-g))) * u; }
        store_rows<NF>(dst + (gw * 2 + 1) * NF, r, lane);
    } else {
        float r[NF];
        calc_plain<NF, NP>(fa, sx, K, lane, r);
        if (RMS) r[0] += sd[0] * 0.0f;
        store_rows<NF>(dst + gw * NF, r, lane);
    }
}
template <int NQ, int NO, int NG, int ND, bool RMS>
static double time_chain_d(const std::vector<LayerW> &hl, Bufs b, const float *normw, hipStream_t st, int reps) {
    hipGraph_t g; hipGraphExec_t ge;
    const int L = (int) hl.size();
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int l = 0; l < L; l++) {
        hipLaunchKernelGGL((phase_kernel_d<0, NQ, RMS>), dim3(QKV / (4 * NQ)), dim3(NTH), 0, st, hl[l], b, normw);
        hipLaunchKernelGGL((phase_kernel_d<1, NO, false>), dim3(H / (4 * NO)), dim3(NTH), 0, st, hl[l], b, normw);
        hipLaunchKernelGGL((phase_kernel_d<2, NG, RMS>), dim3(F / (4 * 2 * NG)), dim3(NTH), 0, st, hl[l], b, normw);
        hipLaunchKernelGGL((phase_kernel_d<3, ND, false>), dim3(H / (4 * ND)), dim3(NTH), 0, st, hl[l], b, normw);
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; i++) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return ms * 1e3 / reps / L;
}

// ---- the boundary of the persistent form ---------------------------------------------------------------------------------------------------
// vmcnt counts loads and stores together and retires in issue order: a wave that publishes (stores + drain) cannot have prefetch loads in
// flight, or the drain waits for them too.  So ONE wave per workgroup (wave 0) publishes the workgroup's results (handed to it through LDS),
// drains, arrives at the barrier, only then requests its own share of the next phase's weights, and polls; waves 1-3 request theirs before
// they meet wave 0 at the workgroup barrier and keep them in flight across the whole boundary.
#define SPIN_LIMIT 400000u
__device__ __forceinline__ void bar_arrive(Sync *s, unsigned epoch) {   // one lane
    const int g = blockIdx.x & 7;
    const unsigned per = gridDim.x >> 3;
    const unsigned old = __hip_atomic_fetch_add(&s->grp[g][0], 1u, __ATOMIC_RELAXED, AGENT);
    if (old == per * epoch - 1) {                      // last of its group: one arrival at the top counter
        const unsigned old2 = __hip_atomic_fetch_add(&s->top[0], 1u, __ATOMIC_RELAXED, AGENT);
        if (old2 == 8 * epoch - 1) {                   // last group: open the generation of every group
#pragma unroll
            for (int i = 0; i < 8; i++) __hip_atomic_store(&s->gen[i][0], epoch, __ATOMIC_RELAXED, AGENT);
        }
    }
}
__device__ __forceinline__ unsigned bar_wait(Sync *s, unsigned epoch) {   // one lane; 0 = timed out
    const int g = blockIdx.x & 7;
    unsigned spins = 0;
    while (__hip_atomic_load(&s->gen[g][0], __ATOMIC_RELAXED, AGENT) < epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SPIN_LIMIT || __hip_atomic_load(&s->err[0], __ATOMIC_RELAXED, AGENT)) {
            __hip_atomic_store(&s->err[0], 1u, __ATOMIC_RELAXED, AGENT);
            return 0;
        }
    }
    return 1;
}

// ---- B: every layer, every phase in one launch --------------------------------------------------------------------------------------------
// One boundary: results of the phase are in LDS (outs[0 .. n_out)); `PREFETCH` is the request of the next phase's weights.
#define BOUNDARY(dst_global, n_out, PREFETCH)                                                                                   \
    do {                                                                                                                        \
        ++epoch;                                                                                                                \
        if (prefetch && wave != 0) { PREFETCH; }                                                                                \
        __syncthreads();                                                                                                        \
        if (wave == 0) {                                                                                                        \
            if (lane < (n_out)) __hip_atomic_store((dst_global) + lane, outs[lane], __ATOMIC_RELAXED, AGENT);                   \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                    \
            if (lane == 0) bar_arrive(s, epoch);                                                                              \
            if (prefetch) { PREFETCH; }                                                                                         \
            if (lane == 0) flag = bar_wait(s, epoch);                                                                           \
        }                                                                                                                       \
        __syncthreads();                                                                                                        \
        if (!flag) return;                                                                                                      \
    } while (0)

__global__ __launch_bounds__(NTH) void engine_kernel(const LayerW *layers, int n_layers, Bufs b, Sync *s, int prefetch) {
    __shared__ __attribute__((aligned(16))) int8_t sx[F];
    __shared__ float outs[64];
    __shared__ unsigned flag;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), gw = blockIdx.x * 4 + wave;
    unsigned epoch = 0;
    int nq[FQ], no[FO], na[2 * FG], nb2[2 * FG];
    rows_qkv(gw, nq); rows_o(gw, no); rows_gu(gw, 0, na); rows_gu(gw, 1, nb2);
    Frag<FQ, 2> fq; Frag<FO, 2> fo; Frag<2 * FG, 2> fa, fb; Frag<FD, 4> fd;
    LayerW w = layers[0];
    gather_x<false, H>(b.y_down, S_IN, sx, tid);   // layer 0's input: as the host left it (the launch boundary made it visible)
    if (prefetch) frag_load<FQ, 2>(fq, w.w4[0], w.wd[0], nq, H, lane);
    __syncthreads();
    for (int l = 0; l < n_layers; l++) {
        const LayerW wn = layers[min(l + 1, n_layers - 1)];
        // ---- qkv
        if (!prefetch) frag_load<FQ, 2>(fq, w.w4[0], w.wd[0], nq, H, lane);
        { float r[FQ]; calc_plain<FQ, 2>(fq, sx, H, lane, r); store_rows<FQ>(outs + wave * FQ, r, lane); }
        BOUNDARY(b.y_qkv + blockIdx.x * 4 * FQ, 4 * FQ, (frag_load<FO, 2>(fo, w.w4[1], w.wd[1], no, H, lane)));
        gather_x<true, H>(b.y_qkv, S_O, sx, tid);
        __syncthreads();
        // ---- o
        if (!prefetch) frag_load<FO, 2>(fo, w.w4[1], w.wd[1], no, H, lane);
        { float r[FO]; calc_plain<FO, 2>(fo, sx, H, lane, r); store_rows<FO>(outs + wave * FO, r, lane); }
        BOUNDARY(b.y_o + blockIdx.x * 4 * FO, 4 * FO, (frag_load<2 * FG, 2>(fa, w.w4[2], w.wd[2], na, H, lane)));
        gather_x<true, H>(b.y_o, S_GU, sx, tid);
        __syncthreads();
        // ---- gate | up (two items; the second one's weights are requested under the first one's dots)
        if (!prefetch) frag_load<2 * FG, 2>(fa, w.w4[2], w.wd[2], na, H, lane);
        frag_load<2 * FG, 2>(fb, w.w4[2], w.wd[2], nb2, H, lane);
        { float r[FG]; calc_gu(fa, sx, lane, r); store_rows<FG>(outs + (wave * 2 + 0) * FG, r, lane); }
        { float r[FG]; calc_gu(fb, sx, lane, r); store_rows<FG>(outs + (wave * 2 + 1) * FG, r, lane); }
        BOUNDARY(b.y_gu + blockIdx.x * 8 * FG, 8 * FG, (frag_load<FD, 4>(fd, w.w4[3], w.wd[3], no, F, lane)));
        gather_x<true, F>(b.y_gu, S_DOWN, sx, tid);
        __syncthreads();
        // ---- down
        if (!prefetch) frag_load<FD, 4>(fd, w.w4[3], w.wd[3], no, F, lane);
        { float r[FD]; calc_plain<FD, 4>(fd, sx, F, lane, r); store_rows<FD>(outs + wave * FD, r, lane); }
        BOUNDARY(b.y_down + blockIdx.x * 4 * FD, 4 * FD, (frag_load<FQ, 2>(fq, wn.w4[0], wn.wd[0], nq, H, lane)));
        gather_x<true, H>(b.y_down, S_IN, sx, tid);
        __syncthreads();
        w = wn;
    }
}

// ---- C: boundaries only ----------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTH) void barrier_kernel(Sync *s, int n) {
    __shared__ unsigned flag;
    for (int i = 1; i <= n; i++) {
        __syncthreads();
        if (threadIdx.x == 0) { bar_arrive(s, (unsigned) i); flag = bar_wait(s, (unsigned) i); }
        __syncthreads();
        if (!flag) return;
    }
}

__global__ void fill_bytes(uint8_t *p, size_t n, unsigned seed) {
    size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned) i * 2654435761u + seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    p[i] = (uint8_t) x;
}
__global__ void fill_scales(_Float16 *p, size_t n) {
    size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    if (i < n) p[i] = (_Float16) (1.0f + (float) (i % 7) * 0.0625f);
}
__global__ void fill_x(float *p, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (float) ((i * 37) % 201 - 100);
}

int main(int argc, char **argv) {
    const int L = argc > 1 ? atoi(argv[1]) : 8;
    const int reps = 20;
    const int N[4] = {QKV, H, 2 * F, H}, K[4] = {H, H, H, F};
    size_t layer_bytes = 0;
    for (int p = 0; p < 4; p++) layer_bytes += (size_t) N[p] * K[p] / 2 + (size_t) N[p] * K[p] / 32 * 2;
    printf("layers %d, %.1f MB of Q4_0 weights per layer (codes + fp16 block scales), %d workgroups x %d threads\n", L, layer_bytes / 1e6, NWG, NTH);
    std::vector<LayerW> hl((size_t) L);
    for (int l = 0; l < L; l++) {
        for (int p = 0; p < 4; p++) {
            const size_t nc = (size_t) N[p] * K[p] / 2, ns = (size_t) N[p] * K[p] / 32;
            uint8_t *c; _Float16 *d;
            CK(hipMalloc(&c, nc)); CK(hipMalloc(&d, ns * 2));
            fill_bytes<<<(unsigned) ((nc + 255) / 256), 256>>>(c, nc, 1000u * l + p);
            fill_scales<<<(unsigned) ((ns + 255) / 256), 256>>>(d, ns);
            hl[l].w4[p] = c; hl[l].wd[p] = d;
        }
    }
    LayerW *dl; CK(hipMalloc(&dl, sizeof(LayerW) * L)); CK(hipMemcpy(dl, hl.data(), sizeof(LayerW) * L, hipMemcpyHostToDevice));
    Bufs b; CK(hipMalloc(&b.y_qkv, QKV * 4)); CK(hipMalloc(&b.y_o, H * 4)); CK(hipMalloc(&b.y_gu, F * 4)); CK(hipMalloc(&b.y_down, H * 4));
    Sync *s; CK(hipMalloc(&s, sizeof(Sync)));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ra(H), rb(H);
    auto reset_x = [&]() { fill_x<<<(H + 255) / 256, 256, 0, st>>>(b.y_down, H); };

    // ---- A: graph of 4 L launches ----
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int l = 0; l < L; l++) {
        hipLaunchKernelGGL(phase_kernel<0>, dim3(NWG), dim3(NTH), 0, st, hl[l], b);
        hipLaunchKernelGGL(phase_kernel<1>, dim3(NWG), dim3(NTH), 0, st, hl[l], b);
        hipLaunchKernelGGL(phase_kernel<2>, dim3(NWG), dim3(NTH), 0, st, hl[l], b);
        hipLaunchKernelGGL(phase_kernel<3>, dim3(NWG), dim3(NTH), 0, st, hl[l], b);
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    reset_x(); CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    CK(hipMemcpy(ra.data(), b.y_down, H * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < 3; i++) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; i++) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us_a = ms * 1e3 / reps / L;
    printf("A  launch chain (4 launches per layer, graph replay):      %7.2f us per layer  (%.0f GB/s)\n", us_a, layer_bytes / us_a * 1e-3);

    // ---- E: the chain with every kernel warming its XCD's L2 for the next one ----
    {
        unsigned *xo; CK(hipMalloc(&xo, NWG * 4 * 2));
        xcc_kernel<<<NWG, NTH, 0, st>>>(xo); xcc_kernel<<<NWG, NTH, 0, st>>>(xo + NWG);
        std::vector<unsigned> hx(NWG * 2); CK(hipStreamSynchronize(st)); CK(hipMemcpy(hx.data(), xo, NWG * 8, hipMemcpyDeviceToHost));
        int rr = 0, same = 0;
        for (int i = 0; i < NWG; i++) { rr += hx[i] == (unsigned) (i % 8); same += hx[i] == hx[NWG + i]; }
        printf("E  XCD of a workgroup: %d of %d are blockIdx %% 8, %d of %d the same in two launches (first 16: ", rr, NWG, same, NWG);
        for (int i = 0; i < 16; i++) printf("%u", hx[i]);
        printf(")\n");
        std::vector<float> re(H);
        auto line = [&](const char *name, double us) {
            size_t diff = 0;
            for (int i = 0; i < H; i++) diff += memcmp(&ra[i], &re[i], 4) != 0;
            printf("E  %-58s %7.2f us per layer  (%.0f GB/s)  = %.2f x A, %zu results differ\n", name, us, layer_bytes / us * 1e-3, us / us_a, diff);
        };
        line("no requests for the next launch (= A again)", time_chain_e<0, 1 << 20>(hl, b, st, reps, re.data()));
        line("next launch's share into this XCD's L2, whole", time_chain_e<1, 1 << 20>(hl, b, st, reps, re.data()));
        line("... first 8 KB of each region", time_chain_e<1, 8192>(hl, b, st, reps, re.data()));
        line("... first 4 KB of each region", time_chain_e<1, 4096>(hl, b, st, reps, re.data()));
        line("the neighbour workgroup's share (another XCD's L2), whole", time_chain_e<2, 1 << 20>(hl, b, st, reps, re.data()));
    }
    if (getenv("ENGINE_BENCH_ONLY_E")) return 0;
    // ---- B: persistent, with and without the cross-boundary prefetch ----
    double us_b[2] = {0, 0};
    for (int pf = 1; pf >= 0; pf--) {
        unsigned err = 0;
        reset_x(); CK(hipMemsetAsync(s, 0, sizeof(Sync), st));
        hipLaunchKernelGGL(engine_kernel, dim3(NWG), dim3(NTH), 0, st, (const LayerW *) dl, L, b, s, pf);
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(rb.data(), b.y_down, H * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&err, &s->err[0], 4, hipMemcpyDeviceToHost));
        size_t diff = 0; double mx = 0;
        for (int i = 0; i < H; i++) { diff += memcmp(&ra[i], &rb[i], 4) != 0; mx = fmax(mx, fabs(ra[i])); }
        printf("B  persistent, prefetch %d: timeout flag %u, %zu of %d final activations differ from A (max |a| %.3g)\n", pf, err, diff, H, mx);
        if (err) continue;
        for (int i = 0; i < 3; i++) { CK(hipMemsetAsync(s, 0, sizeof(Sync), st)); hipLaunchKernelGGL(engine_kernel, dim3(NWG), dim3(NTH), 0, st, (const LayerW *) dl, L, b, s, pf); }
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; i++) { CK(hipMemsetAsync(s, 0, sizeof(Sync), st)); hipLaunchKernelGGL(engine_kernel, dim3(NWG), dim3(NTH), 0, st, (const LayerW *) dl, L, b, s, pf); }
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        us_b[pf] = ms * 1e3 / reps / L;
        printf("B  persistent (one launch, 4 boundaries per layer), prefetch %d: %7.2f us per layer  (%.0f GB/s)  = %.2f x A\n", pf, us_b[pf], layer_bytes / us_b[pf] * 1e-3, us_b[pf] / us_a);
    }
    // ---- D: what the product's shapes and its staging prologue cost on top of A ----
    {
        float *normw; CK(hipMalloc(&normw, F * 4));
        fill_x<<<(F + 255) / 256, 256, 0, st>>>(normw, F);
        reset_x();
        printf("D  launch chain, features per wave (qkv, o, gate|up pairs per item, down) and workgroups per phase; RMS = rms norm + block-max Q8_0 staging in qkv and gate|up\n");
        printf("   5,3,4,3 = 256/256/256/256 workgroups          : %7.2f us per layer   with RMS %7.2f\n", time_chain_d<5, 3, 4, 3, false>(hl, b, normw, st, reps), time_chain_d<5, 3, 4, 3, true>(hl, b, normw, st, reps));
        printf("   2,2,2,2 = 640/384/512/384 workgroups (product): %7.2f us per layer   with RMS %7.2f\n", time_chain_d<2, 2, 2, 2, false>(hl, b, normw, st, reps), time_chain_d<2, 2, 2, 2, true>(hl, b, normw, st, reps));
        printf("   1,1,1,1 = 1280/768/1024/768 workgroups        : %7.2f us per layer   with RMS %7.2f\n", time_chain_d<1, 1, 1, 1, false>(hl, b, normw, st, reps), time_chain_d<1, 1, 1, 1, true>(hl, b, normw, st, reps));
        printf("   10,6,8,6 = 128/128/128/128 workgroups         : %7.2f us per layer   with RMS %7.2f\n", time_chain_d<10, 6, 8, 6, false>(hl, b, normw, st, reps), time_chain_d<10, 6, 8, 6, true>(hl, b, normw, st, reps));
    }
    // ---- C: the boundary alone ----
    {
        const int nb = 200;
        unsigned err = 0;
        for (int i = 0; i < 3; i++) { CK(hipMemsetAsync(s, 0, sizeof(Sync), st)); hipLaunchKernelGGL(barrier_kernel, dim3(NWG), dim3(NTH), 0, st, s, nb); }
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; i++) { CK(hipMemsetAsync(s, 0, sizeof(Sync), st)); hipLaunchKernelGGL(barrier_kernel, dim3(NWG), dim3(NTH), 0, st, s, nb); }
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(&err, &s->err[0], 4, hipMemcpyDeviceToHost));
        printf("C  barrier alone (%d workgroups, group = blockIdx %% 8): %6.2f us per barrier (timeout flag %u)\n", NWG, ms * 1e3 / reps / nb, err);
    }
    return 0;
}

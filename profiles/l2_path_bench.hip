// What can a CU pull from a warm L2, by load path?  The tiled decoder GEMM stages 512 KB per workgroup through global_load_lds_dwordx4 and runs at
// ~13 B/clk/CU whether 1 or 3 k-tiles are in flight and whether the weights are cold or hot (profiles/r04/gemm_tile_sweep_r1024*.log): a
// throughput limit of some path.  Every workgroup (512 threads, one per CU, 256 of them) sweeps a 2 MB region shared by the workgroups of its XCD
// (blockIdx % 8: the dispatcher deals workgroups round-robin over the XCDs) ITER times, 32 KB per step, three ways:
//   dma   global_load_lds_dwordx4 straight into LDS (what gemm_tile_kernel does), 4 buffers, counted vmcnt
//   reg   global_load_dwordx4 into registers, ds_write_b128 into LDS one step later (two register sets in flight)
//   regx  global_load_dwordx4 into registers, consumed there (xor-folded): the load path alone
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int int4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int NT = 512, STEP = 32 * 1024, REGION = 2 * 1024 * 1024;

template <int MODE>
__global__ __launch_bounds__(512) void k(const char *base, int *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const char *reg = base + (size_t) (blockIdx.x & 7) * REGION;
    const int tid = threadIdx.x;
    int4v acc = {0, 0, 0, 0};
    const int n_steps = iters * (REGION / STEP);
    if (MODE == 3 || MODE == 4) {
        // the GEMM's pattern: a k-tile = 256 rows x 128 B at a row stride of 2 KB (K = 1024 fp16; MODE 4: 8 KB, K = 4096); a wave instruction = 8 rows
        constexpr int RS = MODE == 3 ? 2048 : 8192;
        const int lane = tid & 63, wave = tid >> 6;
        for (int s = 0; s < n_steps; s++) {
            const int kt = s % (RS / 128);
            const char *src = reg + (size_t) kt * 128 + (size_t) ((s / (RS / 128)) % (REGION / (256 * RS) > 0 ? REGION / (256 * RS) : 1)) * 256 * RS;
            char *dst = smem + (s & 3) * STEP;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int g = j * 8 + wave;                 // row group of 8 rows
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (src + (size_t) (g * 8 + (lane >> 3)) * RS + (lane & 7) * 16),
                                                 (__attribute__((address_space(3))) void *) (dst + g * 1024), 16, 0, 0);
            }
            if (s >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            if ((s & 3) == 3) { __syncthreads(); acc[0] ^= *(const int *) (smem + tid * 4); }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (MODE == 0) {
        for (int s = 0; s < n_steps; s++) {
            const char *src = reg + (size_t) (s % (REGION / STEP)) * STEP;
            char *dst = smem + (s & 3) * STEP;
#pragma unroll
            for (int j = 0; j < STEP / (NT * 16); j++) {   // 4 wave-instructions of 1 KiB per wave
                const int wave = tid >> 6;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (src + (size_t) (j * 8 + wave) * 1024 + (tid & 63) * 16),
                                                 (__attribute__((address_space(3))) void *) (dst + (j * 8 + wave) * 1024), 16, 0, 0);
            }
            if (s >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            if ((s & 3) == 3) { __syncthreads(); acc[0] ^= *(const int *) (smem + tid * 4); }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        int4v r[2][4];
        auto load = [&](int s, int set) {
            const char *src = reg + (size_t) (s % (REGION / STEP)) * STEP;
#pragma unroll
            for (int j = 0; j < 4; j++) r[set][j] = *(const int4v *) (src + (size_t) (j * NT + tid) * 16);
        };
        load(0, 0);
        for (int s = 0; s < n_steps; s += 2) {
            load(s + 1, 1);
            if (MODE == 1) {
#pragma unroll
                for (int j = 0; j < 4; j++) *(int4v *) (smem + (s & 3) * STEP + (j * NT + tid) * 16) = r[0][j];
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) acc ^= r[0][j];
            }
            load(s + 2, 0);
            if (MODE == 1) {
#pragma unroll
                for (int j = 0; j < 4; j++) *(int4v *) (smem + ((s + 1) & 3) * STEP + (j * NT + tid) * 16) = r[1][j];
                if ((s & 3) == 2) { __syncthreads(); acc[0] ^= *(const int *) (smem + tid * 4); }
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) acc ^= r[1][j];
            }
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678) out[tid] = 1;
}

template <int MODE>
static void run(const char *name, const char *buf, int *out, int grid) {
    CK(hipFuncSetAttribute((const void *) k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int iters = 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(NT), 4 * STEP, 0, buf, out, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(NT), 4 * STEP, 0, buf, out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double) grid * iters * REGION;
    printf("%-6s grid %4d: %7.3f ms  %6.2f TB/s aggregate  %5.1f GB/s per workgroup = %5.1f B/clk at 2.4 GHz\n", name, grid, ms, bytes / ms / 1e9, bytes / grid / ms / 1e6, bytes / grid / ms / 1e6 / 2.4);
}
int main() {
    char *buf; int *out;
    CK(hipMalloc(&buf, (size_t) 8 * REGION + STEP)); CK(hipMemset(buf, 1, (size_t) 8 * REGION + STEP)); CK(hipMalloc(&out, 4096));
    for (int grid : {256, 512}) {
        run<0>("dma", buf, out, grid);
        run<1>("reg", buf, out, grid);
        run<2>("regx", buf, out, grid);
        run<3>("dma2k", buf, out, grid);
        run<4>("dma8k", buf, out, grid);
    }
    return 0;
}

// cu_mask_probe.hip — does hipExtStreamCreateWithCUMask restrict a stream's kernels on this box?  A compute-bound kernel on a full stream and on
// streams masked to 1/2 and 1/4 of the CUs: the time must scale with the inverse share if the mask is honoured.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void spin(float *out, int n) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < n; i++) a = a * b + 0.5f;
    if (a == 123.0f) out[0] = a;
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    float *out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int m : {1, 2, 4}) {
        hipStream_t st;
        if (m == 1) hipStreamCreate(&st);
        else {
            std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
            for (int i = 0; i < ncu; i++) if (i % m == 0) mask[i / 32] |= 1u << (i % 32);
            hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t) mask.size(), mask.data());
            if (e != hipSuccess) { printf("share 1/%d: hipExtStreamCreateWithCUMask: %s\n", m, hipGetErrorString(e)); continue; }
        }
        spin<<<ncu * 8, 256, 0, st>>>(out, 1000);
        hipStreamSynchronize(st);
        hipEventRecord(e0, st);
        spin<<<ncu * 8, 256, 0, st>>>(out, 200000);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("CUs %d, share 1/%d of them: %.3f ms\n", ncu, m, ms);
    }
    return 0;
}

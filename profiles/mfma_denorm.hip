// mfma_denorm.hip — does v_mfma_f32_32x32x16_f16 honour fp16 subnormal INPUTS on gfx950?  (The fp16 hi/lo split of the codec needs the low parts of small
// values, which are fp16 subnormals, to count.)   hipcc --offload-arch=gfx950 -O2 -o mfma_denorm profiles/mfma_denorm.hip && ./mfma_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
__global__ void k(float *out, float av, float bv) {
    half8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16) av; b[i] = (_Float16) bv; }
    float16v c;
    for (int i = 0; i < 16; i++) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    float *d; hipMalloc(&d, 4);
    const float vals[][2] = {{1.0f, 1.0f}, {5.9604645e-8f, 1.0f}, {1.0f, 5.9604645e-8f}, {3.0e-6f, 2.0f}, {6.1035156e-5f, 1.0f}, {3.0517578e-5f, 1024.0f}};
    for (auto &v : vals) {
        k<<<1, 64>>>(d, v[0], v[1]);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("a = %.9g (fp16 %.9g)  b = %.9g : mfma sum over K = 16 -> %.9g, expected %.9g\n", v[0], (float) (_Float16) v[0], v[1], h, 16.0f * (float) (_Float16) v[0] * (float) (_Float16) v[1]);
    }
    return 0;
}

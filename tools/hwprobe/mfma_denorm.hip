// Hardware probe: does v_mfma_f32_32x32x16_f16 keep f16 subnormal inputs on gfx950?  (expects 16 * 2^-20)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float aval) {
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)aval; b[j] = (_Float16)1.0f; }
    f32x16 c;
    for (int e = 0; e < 16; ++e) c[e] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    float* d; hipMalloc(&d, 4);
    const float vals[3] = {9.5367431640625e-07f /* 2^-20 */, 5.9604644775390625e-08f /* 2^-24 */, 0.5f};
    for (float v : vals) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, v);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("a=%g  mfma sum=%g  expected=%g  %s\n", v, h, 16 * v, h == 16 * v ? "kept" : "FLUSHED/other");
    }
    return 0;
}

// Hardware probe: do MFMA (one wave) and VALU (another wave on the same SIMD) overlap on gfx950?
// Two workgroups of 4 waves per CU (80 KiB LDS each forces exactly two), so every SIMD holds one wave of each.
// mode 0: both run MFMA loops; 1: both VALU loops; 2: even workgroups MFMA, odd VALU; 3 / 4: one kind alone.
// PACKED = 1: the VALU loop is what hipcc makes of plain C (v_pk_fma_f32); 0: forced scalar v_fma_f32.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int PACKED>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, int mode) {
    extern __shared__ float smem[];
    if (mode >= 3 && (blockIdx.x & 1)) return;          // modes 3 / 4: the odd workgroups leave at once (one wave per SIMD)
    const bool mfma = mode == 0 || mode == 3 || (mode == 2 && (blockIdx.x & 1) == 0);
    float res = 0.f;
    if (mfma) {
        half8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f); b[j] = (_Float16)1.0f; }
        f32x16 c[4];
        for (int t = 0; t < 4; ++t) for (int e = 0; e < 16; ++e) c[t][e] = 0.f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int t = 0; t < 4; ++t) c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[t], 0, 0, 0);
        }
        for (int t = 0; t < 4; ++t) res += c[t][0];
    } else {
        float x[16];
        for (int j = 0; j < 16; ++j) x[j] = threadIdx.x * 0.01f + j;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if (PACKED) x[j] = __builtin_fmaf(x[j], 1.0001f, 0.5f);
                    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(1.0001f), "v"(0.5f));
                }
        }
        for (int j = 0; j < 16; ++j) res += x[j];
    }
    if (res == 12345.f) out[0] = res + smem[0];
}

int main() {
    float* d; (void)hipMalloc(&d, 4);
    (void)hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    const char* names[5] = {"MFMA + MFMA", "VALU + VALU", "MFMA + VALU", "MFMA alone", "VALU alone"};
    for (int packed = 0; packed < 2; ++packed)
        for (int mode = 0; mode < 5; ++mode) {
            hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            auto launch = [&](int iters) {
                if (packed) hipLaunchKernelGGL(k<1>, dim3(512), dim3(256), 80 * 1024, 0, d, iters, mode);
                else hipLaunchKernelGGL(k<0>, dim3(512), dim3(256), 80 * 1024, 0, d, iters, mode);
            };
            launch(1000);
            (void)hipEventRecord(a, 0);
            launch(20000);
            (void)hipEventRecord(b, 0);
            (void)hipDeviceSynchronize();
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            printf("%s VALU, %s: %.3f ms  (20000 x {12 MFMA | 96 FMA} per wave)\n", packed ? "packed" : "scalar", names[mode], ms);
        }
    return 0;
}

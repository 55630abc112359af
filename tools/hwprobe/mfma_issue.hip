// Hardware probe: MFMA throughput of ONE wave per SIMD vs the number of independent accumulators it rotates over
// (v_mfma_f32_32x32x16_f16, 8 passes = 32 cycles in the pipe).  Also two waves per SIMD for reference.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, int waves2) {
    extern __shared__ float smem[];
    if (!waves2 && (blockIdx.x & 1)) return;
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f); b[j] = (_Float16)1.0f; }
    f32x16 c[NACC];
    for (int t = 0; t < NACC; ++t) for (int e = 0; e < 16; ++e) c[t][e] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 24 / NACC; ++r)
#pragma unroll
            for (int t = 0; t < NACC; ++t) c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[t], 0, 0, 0);
    }
    float res = 0.f;
    for (int t = 0; t < NACC; ++t) res += c[t][0];
    if (res == 12345.f) out[0] = res + smem[0];
}

template <int NACC>
void run(float* d) {
    (void)hipFuncSetAttribute((const void*)k<NACC>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    for (int w2 = 0; w2 < 2; ++w2) {
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        hipLaunchKernelGGL(k<NACC>, dim3(512), dim3(256), 80 * 1024, 0, d, 500, w2);
        (void)hipEventRecord(a, 0);
        hipLaunchKernelGGL(k<NACC>, dim3(512), dim3(256), 80 * 1024, 0, d, 10000, w2);
        (void)hipEventRecord(b, 0);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        printf("%d accumulators, %d wave(s)/SIMD: %.3f ms for 240000 MFMAs per wave -> %.1f ns per MFMA per SIMD\n", NACC, w2 + 1, ms,
               ms * 1e6 / (240000.0 * (w2 + 1)));
    }
}
int main() {
    float* d; (void)hipMalloc(&d, 4);
    run<1>(d); run<2>(d); run<4>(d); run<8>(d); run<12>(d);
    return 0;
}

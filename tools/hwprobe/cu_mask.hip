// Hardware probe: does hipExtStreamCreateWithCUMask restrict a stream's kernels to the selected CUs on this box?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void spin(float* out, int iters) {
    float x = threadIdx.x;
    for (int i = 0; i < iters; ++i) x = x * 1.000001f + 0.5f;
    if (x == 123.f) out[0] = x;
}
int main() {
    float* d; (void)hipMalloc(&d, 4);
    const uint32_t pats[3] = {0xFFFFFFFFu, 0x55555555u, 0x11111111u};
    for (uint32_t p : pats) {
        std::vector<uint32_t> mask(8, p);
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask.data());
        if (e != hipSuccess) { printf("create failed: %s\n", hipGetErrorString(e)); continue; }
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        hipLaunchKernelGGL(spin, dim3(256 * 8), dim3(256), 0, s, d, 200000);
        (void)hipEventRecord(a, s);
        hipLaunchKernelGGL(spin, dim3(256 * 8), dim3(256), 0, s, d, 200000);
        (void)hipEventRecord(b, s);
        (void)hipStreamSynchronize(s);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        printf("mask %08x: %.3f ms\n", p, ms);
    }
    return 0;
}

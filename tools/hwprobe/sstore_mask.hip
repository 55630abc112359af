// Hardware probe (round 5): ReLU sign bits of an MFMA accumulator tile straight to memory through the SCALAR path.
//   v_cmp_gt_f32 s[n:n+1], v_e, 0   : one VALU instruction per accumulator register -> a 64-lane mask in an SGPR pair
//   s_store_dwordx2 s[n:n+1], base  : scalar store (gfx9 family: still there on gfx950), s_dcache_wb before the kernel ends
// against the VALU way (per lane: pack its own 16 sign bits with v_cmp + v_cndmask / shifts, one 4-byte vector store).
// Checks the stored words against the host's and times both variants: 256 workgroups x 512 threads, ITERS rounds of 16 registers.
//   hipcc --offload-arch=gfx950 -O3 -o sstore_mask sstore_mask.hip && ./sstore_mask
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

constexpr int ITERS = 64;

__global__ __launch_bounds__(512) void k_scalar(const float* __restrict__ in, unsigned long long* __restrict__ out) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const float* src = in + ((size_t)(blockIdx.x * 8 + wave) * ITERS) * 16 * 64;
    unsigned long long* dst = out + ((size_t)(blockIdx.x * 8 + wave) * ITERS) * 16;
    for (int it = 0; it < ITERS; ++it) {
        float v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = src[(it * 16 + e) * 64 + lane];
        unsigned long long* d = dst + it * 16;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            unsigned long long m;
            asm volatile("v_cmp_gt_f32_e64 %0, %1, 0" : "=s"(m) : "v"(v[e]));
            asm volatile("s_store_dwordx2 %0, %1, %2" ::"s"(m), "s"(d), "n"(e * 8) : "memory");
        }
    }
    asm volatile("s_dcache_wb" ::: "memory");
}

// the VALU way: lane packs its own 16 bits, one store per lane (64 x 4 bytes per 16 registers)
__global__ __launch_bounds__(512) void k_vector(const float* __restrict__ in, unsigned* __restrict__ out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* src = in + ((size_t)(blockIdx.x * 8 + wave) * ITERS) * 16 * 64;
    unsigned* dst = out + ((size_t)(blockIdx.x * 8 + wave) * ITERS) * 64;
    for (int it = 0; it < ITERS; ++it) {
        unsigned bits = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e) bits |= (src[(it * 16 + e) * 64 + lane] > 0.f ? 1u : 0u) << e;
        dst[it * 64 + lane] = bits;
    }
}

int main() {
    const size_t waves = 256 * 8, n = waves * ITERS * 16 * 64;
    std::vector<float> h(n);
    srand(1);
    for (size_t i = 0; i < n; ++i) h[i] = (float)(rand() % 2001 - 1000) * 1e-3f;
    float* d_in;
    unsigned long long* d_s;
    unsigned* d_v;
    hipMalloc(&d_in, n * 4);
    hipMalloc(&d_s, waves * ITERS * 16 * 8);
    hipMalloc(&d_v, waves * ITERS * 64 * 4);
    hipMemcpy(d_in, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(d_s, 0, waves * ITERS * 16 * 8);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float ms_s = 0, ms_v = 0;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k_scalar, dim3(256), dim3(512), 0, 0, d_in, d_s);
        hipEventRecord(b);
        hipEventSynchronize(b);
        hipEventElapsedTime(&ms_s, a, b);
        hipEventRecord(a);
        hipLaunchKernelGGL(k_vector, dim3(256), dim3(512), 0, 0, d_in, d_v);
        hipEventRecord(b);
        hipEventSynchronize(b);
        hipEventElapsedTime(&ms_v, a, b);
    }
    std::vector<unsigned long long> hs(waves * ITERS * 16);
    std::vector<unsigned> hv(waves * ITERS * 64);
    hipMemcpy(hs.data(), d_s, hs.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hv.data(), d_v, hv.size() * 4, hipMemcpyDeviceToHost);
    size_t bad_s = 0, bad_v = 0;
    for (size_t w = 0; w < waves * ITERS; ++w)
        for (int e = 0; e < 16; ++e) {
            unsigned long long want = 0;
            for (int l = 0; l < 64; ++l) want |= (unsigned long long)(h[(w * 16 + e) * 64 + l] > 0.f) << l;
            bad_s += hs[w * 16 + e] != want;
        }
    for (size_t w = 0; w < waves * ITERS; ++w)
        for (int l = 0; l < 64; ++l) {
            unsigned want = 0;
            for (int e = 0; e < 16; ++e) want |= (unsigned)(h[(w * 16 + e) * 64 + l] > 0.f) << e;
            bad_v += hv[w * 64 + l] != want;
        }
    printf("scalar path (v_cmp -> SGPR pair -> s_store_dwordx2): %zu wrong words of %zu, %.3f ms\n", bad_s, hs.size(), ms_s);
    printf("vector path (per-lane bit packing, 4-byte store):      %zu wrong words of %zu, %.3f ms\n", bad_v, hv.size(), ms_v);
    printf("(both read %.1f MB; 16 registers x %d rounds x %zu waves)\n", n * 4 / 1e6, ITERS, waves);
    return bad_s || bad_v;
}

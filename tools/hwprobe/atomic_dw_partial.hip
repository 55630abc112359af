// Hardware probe (round 5, review item 7): could the dX kernel accumulate dW itself, without the dY round trip through HBM?
// A dX workgroup holds dY of ONE 128-point tile in LDS; the dW product of that tile and layer is a 256 x 256 f32 block (256 KiB)
// that has to be ADDED to the layer's gradient - 4 081 tiles x 9.5 layers of them per 522 k-point launch.  Registers cannot
// hold it across tiles (8 waves x 128 accumulator registers = exactly one layer's block, and the chain needs its own 185), so
// the only fused form is red-to-memory: global_atomic_add_f32 (no return) into a per-XCD partial buffer that stays in L2.
// This probe measures just that traffic, with nothing else on the chip: every workgroup (512 threads, one per CU x ROUNDS)
// adds a 256 x 256 block (each lane 128 values, coalesced 256-byte rows per wave instruction) into partial[blockIdx % NPART].
//   NPART = 8  : one partial per XCD (workgroup i runs on XCD i % 8: all atomics of a partial stay in ONE L2)
//   NPART = 256: one partial per workgroup slot (no contention at all: the atomic unit's raw rate)
// and, for scale, the same bytes as plain stores.  Prints float-atomics per second and the time 4 081 x 9.5 blocks would take.
//   hipcc --offload-arch=gfx950 -O3 -o atomic_dw_partial atomic_dw_partial.hip && ./atomic_dw_partial
#include <hip/hip_runtime.h>
#include <stdio.h>

constexpr int BLK = 256 * 256;

template <int MODE>   // 0: atomic add (no return), 1: plain store
__global__ __launch_bounds__(512) void k(float* __restrict__ part, int npart, int rounds) {
    float* p = part + (size_t)(blockIdx.x % npart) * BLK;
    const int tid = threadIdx.x;
    for (int r = 0; r < rounds; ++r) {
#pragma unroll 8
        for (int i = 0; i < BLK / 512; ++i) {
            const float v = (float)(tid + i + r) * 1e-6f;
            if (MODE == 0) __builtin_amdgcn_global_atomic_fadd_f32(p + i * 512 + tid, v);
            else __builtin_nontemporal_store(v, p + i * 512 + tid);
        }
    }
}

int main() {
    float* d;
    hipMalloc(&d, (size_t)256 * BLK * 4);
    hipMemset(d, 0, (size_t)256 * BLK * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int rounds = 16, grid = 256;
    const double blocks = (double)grid * rounds, need = 4081 * 9.5;
    for (int npart : {8, 64, 256}) {
        for (int mode = 0; mode < 2; ++mode) {
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, d, npart, rounds);
                else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, d, npart, rounds);
                hipEventRecord(b);
                hipEventSynchronize(b);
                hipEventElapsedTime(&ms, a, b);
            }
            printf("%-12s into %3d partial block(s): %.3f ms for %.0f blocks = %.2f G float-%s/s, %.2f TB/s of operand bytes -> %.2f ms per 522 k-point launch (4 081 tiles x 9.5 layers)\n",
                   mode == 0 ? "atomic add" : "plain store", npart, ms, blocks, blocks * BLK / ms / 1e6, mode == 0 ? "atomics" : "stores",
                   blocks * BLK * 4 / ms / 1e9, need / blocks * ms);
        }
    }
    return 0;
}

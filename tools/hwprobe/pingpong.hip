// Hardware probe: MFMA-pipe utilisation of the split-f16 MLP stage loop (mlp_split.h gemm_stage + a full-cost
// epilogue: combine, ReLU, hi/lo split, 2-byte LDS writes, 8-byte global stores) under three schedules
//   (a) 4-wave workgroups, 64 points, two INDEPENDENT workgroups per CU (random relative phase: the round-1 kernels)
//   (b) 8-wave workgroup, 2 x 64 points, one per CU, the two 4-wave groups in LOCKSTEP (K-loops together, epilogues together)
//   (c) 8-wave workgroup, 2 x 64 points, one per CU, PING-PONG: group B runs one phase behind group A, every phase ends in
//       one workgroup barrier, so a K-loop (MFMA) phase of one group always sits beside an epilogue (VALU/LDS) phase of
//       the other on every SIMD
// plus (d) the MFMA issue rate of ONE wave per SIMD (one 256-thread workgroup per CU, enforced by 160 KiB of LDS).
#include "../../benerf_amd/csrc/mlp_split.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
using namespace mlp;

__device__ __forceinline__ uint64_t epi(f32x16 (&acc1)[2][2], f32x16 (&acc2)[2][2], _Float16* __restrict__ Th, _Float16* __restrict__ Tl,
                                        int ct0, int lane, const float* __restrict__ bias, _Float16* __restrict__ st, int64_t m0) {
    const int lr = lane & 31, r4 = 4 * (lane >> 5);
    uint64_t bits = 0;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int n = (ct0 + c) * 32 + lr;
        const float bv = bias[n];
        const int ns = (n >> 3) ^ ((lane >> 5) << 1);
        int base[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) base[q] = r4 * LD + ((((ns ^ ((q & 1) | ((q >> 1) << 2)))) << 3) | (n & 7));
        _Float16* st_lane = st + (((m0 >> 3) * 256 + n) * 8 + r4);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
#pragma unroll
            for (int eq = 0; eq < 4; ++eq) {
                Quad16 qh;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = eq * 4 + j;
                    float v = (acc1[r][c][e] + acc2[r][c][e] * LO_INV) + bv;
                    v = fmaxf(v, 0.f);
                    bits |= (uint64_t)(v > 0.f) << ((c * 2 + r) * 16 + e);
                    const _Float16 hi = (_Float16)v;
                    const _Float16 lo = (_Float16)((v - (float)hi) * LO_SCALE);
                    const int idx = base[((e >> 1) & 1) | (((e >> 2) & 1) << 1)] + (r * 32 + (e & 3) + 8 * (e >> 2)) * LD;
                    Th[idx] = hi;
                    Tl[idx] = lo;
                    qh.v[j] = hi;
                }
                *reinterpret_cast<uint2*>(st_lane + (int64_t)(r * 4 + eq) * 256 * 8) = __builtin_bit_cast(uint2, qh);
            }
        }
    }
    return bits;
}

// MODE 0: 4 waves, independent workgroups; 1: 8 waves lockstep; 2: 8 waves ping-pong
template <int MODE, bool GEMM = true, bool EPI = true>
__global__ __launch_bounds__(MODE == 0 ? 256 : 512, 2) void k(const float* __restrict__ packed, const float* __restrict__ bias,
                                                               _Float16* __restrict__ st, uint64_t* __restrict__ masks, int tiles_per_group) {
    extern __shared__ __attribute__((aligned(16))) _Float16 Tsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int grp = MODE == 0 ? 0 : __builtin_amdgcn_readfirstlane(tid >> 8);
    const int wave = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);
    _Float16* Th = Tsm + grp * 2 * TM * LD;
    _Float16* Tl = Th + TM * LD;
    for (int i = tid & 255; i < TM * LD; i += 256) {
        Th[i] = (_Float16)(0.001f * ((i * 7) & 255));
        Tl[i] = (_Float16)(0.01f * ((i * 13) & 63) - 0.3f);
    }
    lds_barrier();
    if (MODE == 2 && grp == 1) lds_barrier();
    const int ngroups = MODE == 0 ? 1 : 2;
    for (int t = 0; t < tiles_per_group; ++t) {
        const int64_t tile = ((int64_t)t * gridDim.x + blockIdx.x) * ngroups + grp;
        const int64_t m0 = (tile & 4095) * 64;       // 4096 tiles' worth of store area (256 MB), re-used
#pragma unroll 1
        for (int stg = 0; stg < 11; ++stg) {
            f32x16 acc1[2][2], acc2[2][2];
            zero_acc(acc1);
            zero_acc(acc2);
            if (GEMM) gemm_stage<16, 2>(Th, Tl, 0, packed + pack_offset(PF_L1 + (stg % 4)), wave * 2, lane, acc1, acc2);
            else { for (int r = 0; r < 2; ++r) for (int c = 0; c < 2; ++c) for (int e = 0; e < 16; ++e) { acc1[r][c][e] = Th[(e * 64 + lane + r * 7 + c * 3) & 8191]; acc2[r][c][e] = 1.f; } }
            lds_barrier();
            if (EPI) {
                const uint64_t bits = epi(acc1, acc2, Th, Tl, wave * 2, lane, bias, st + (int64_t)(stg % 8) * 4096 * 64 * 256, m0);
                masks[((int64_t)stg * 4096 + (tile & 4095)) * 256 + (tid & 255)] = bits;
            } else {
                float sink = 0.f;
                for (int c = 0; c < 2; ++c) sink += acc1[0][c][0] + acc2[1][c][3] + acc1[1][c][5] + acc2[0][c][9];
                if (sink == 12345.f) Th[lane] = (_Float16)sink;
            }
            lds_barrier();
        }
    }
    if (MODE == 2 && grp == 0) lds_barrier();
}

template <int NACC>
__global__ __launch_bounds__(256, 1) void kissue(float* out, int iters) {
    extern __shared__ float smem[];
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
void run_issue(float* d) {
    const int smem = 160 * 1024;
    (void)hipFuncSetAttribute((const void*)kissue<NACC>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(kissue<NACC>, dim3(256), dim3(256), smem, 0, d, 500);
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL(kissue<NACC>, dim3(256), dim3(256), smem, 0, d, 10000);
    (void)hipEventRecord(b, 0);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("(d) one wave per SIMD, %2d accumulators: %.3f ms for 240000 MFMAs per wave -> %.1f ns per MFMA (32 cycles at 2.4 GHz = 13.3 ns)\n",
           NACC, ms, ms * 1e6 / 240000.0);
}

template <int MODE, bool GEMM = true, bool EPI = true>
void run(const float* packed, const float* bias, _Float16* st, uint64_t* masks, const char* label) {
    const int nthreads = MODE == 0 ? 256 : 512;
    const int smem = (int)TILE_SMEM * (MODE == 0 ? 1 : 2), grid = MODE == 0 ? 512 : 256;
    (void)hipFuncSetAttribute((const void*)k<MODE, GEMM, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE, GEMM, EPI>), dim3(grid), dim3(nthreads), smem, 0, packed, bias, st, masks, 2);
    (void)hipEventRecord(a, 0);
    const int tiles = 12;
    hipLaunchKernelGGL((k<MODE, GEMM, EPI>), dim3(grid), dim3(nthreads), smem, 0, packed, bias, st, masks, tiles);
    (void)hipEventRecord(b, 0);
    (void)hipDeviceSynchronize();
    hipError_t e = hipGetLastError();
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double mfma = 512.0 * tiles * 11 * 16 * 48;                   // 512 64-point tile slots in flight either way
    const double ideal_ms = mfma * 32 / (1024.0 * 2.4e9) * 1e3;
    printf("%s: %.3f ms for %d tiles per slot; MFMA-bound at 2.4 GHz %.3f ms -> %.0f %%  (%s)\n", label, ms, tiles, ideal_ms,
           100 * ideal_ms / ms, hipGetErrorString(e));
}

int main() {
    float *packed, *bias, *d;
    _Float16* st;
    uint64_t* masks;
    const size_t np = 2 * PACKED_FLOATS;
    (void)hipMalloc(&packed, np * 4);
    std::vector<_Float16> hp(np * 2);
    srand(1);
    for (size_t i = 0; i < hp.size(); ++i) hp[i] = (_Float16)(((rand() & 1023) - 512) * (0.06f / 512));
    (void)hipMemcpy(packed, hp.data(), np * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&bias, 256 * 4);
    (void)hipMemset(bias, 0, 256 * 4);
    (void)hipMalloc(&st, (size_t)8 * 4096 * 64 * 256 * 2);
    (void)hipMalloc(&masks, (size_t)11 * 4096 * 256 * 8);
    (void)hipMalloc(&d, 4);
    run_issue<4>(d);
    run_issue<12>(d);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>(packed + PACKED_FLOATS, bias, st, masks, "(a) 4-wave workgroups, 2 independent per CU ");
        run<1>(packed + PACKED_FLOATS, bias, st, masks, "(b) 8-wave workgroup, lockstep groups       ");
        run<2>(packed + PACKED_FLOATS, bias, st, masks, "(c) 8-wave workgroup, ping-pong groups      ");
        run<2, true, false>(packed + PACKED_FLOATS, bias, st, masks, "(c') ping-pong, K-loops only                ");
        run<2, false, true>(packed + PACKED_FLOATS, bias, st, masks, "(c'') ping-pong, epilogues only             ");
        run<0, true, false>(packed + PACKED_FLOATS, bias, st, masks, "(a') independent, K-loops only              ");
        run<0, false, true>(packed + PACKED_FLOATS, bias, st, masks, "(a'') independent, epilogues only           ");
    }
    return 0;
}

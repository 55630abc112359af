// K3 backward part 2, split-f16 variant of mlp_dw.hip: dW_l = dY_l^T X_l over all sample points as three f16 MFMAs
// per product block (mlp_split.h), f32 accumulation, + bias sums and the alpha / rgb heads on the VALU.
//
// Operands arrive in the ST layout the split forward / dX kernels write (mlp_split.h): blocks of 8 points,
// feature-major, 16-byte units {hi x4, lo x4} - one split away from the MFMA fragment order for a contraction over
// points.  A workgroup (8 waves, one per CU) copies 32-point chunks into a double-buffered LDS image
// [block][plane][feature][8 points] (each unit lands as two 8-byte quads; conflict-free fragment reads, no
// transposition pass) while it multiplies the previous chunk; one barrier per chunk.
// Two accumulator sets (hi*hi and the cross terms) fill the register file at a 256 x 128 output block, so the eight
// 256x256 instances run as pairs of column halves (workgroup ids 8 apart = same XCD); both halves stream the same
// dY chunk.  Measured (FETCH_SIZE, profiles/): the second read is NOT absorbed by L2 whatever the id mapping or lag
// between the halves - 13.3 GB cross the fabric per launch at M = 522k against 9.4 GB of distinct operands - so the
// kernel runs at the fabric/HBM streaming rate for 1.5 KB per point and half-instance.  dY carries the call's
// global power-of-two scale s_g (max|d_raw| from the dX launch); the reduce kernel multiplies by 1/s_g.
#include "mlp_split.h"

namespace {
using namespace mlp;

constexpr int DWT = 512;
constexpr int CHP = 32;             // points per chunk = 4 blocks of 8 = 2 MFMA k-steps
// 16-byte unit as a first-class vector (arrays of HIP's uint4 struct were left in scratch memory by the compiler)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct DwArgs {
    const float* d_raw;
    const float* acts;
    const float* dacts;
    float* ws;
    int64_t M;
    int C;
};

struct Src {
    const u32x4* y;    // ST array of width N (16-byte units)
    const u32x4* x;    // ST array of width XW
    bool bias;
};

__device__ __forceinline__ Src inst_src(const DwArgs& a, int inst) {
    const int64_t Mp = m_pad(a.M);
    auto U = [](const float* p) { return reinterpret_cast<const u32x4*>(p); };
    const float* A = a.acts;
    const float* D = a.dacts;
    switch (inst) {
        case DW_L1: case DW_L2: case DW_L3: case DW_L4: case DW_L5H: case DW_L6: case DW_L7:
            return {U(D + sdact_h(Mp, 1 + (inst - DW_L1))), U(A + sact_h(Mp, inst - DW_L1)), true};
        case DW_FEAT: return {U(D + sdact_feat(Mp)), U(A + sact_h(Mp, 7)), true};
        case DW_VIEWSF: return {U(D + sdact_hv(Mp)), U(A + sact_feat(Mp)), true};
        case DW_L0: return {U(D + sdact_h(Mp, 0)), U(A + sact_pe(Mp)), true};
        case DW_L5P: return {U(D + sdact_h(Mp, 5)), U(A + sact_pe(Mp)), false};
        default: return {U(D + sdact_hv(Mp)), U(A + sact_ped(Mp)), false};   // DW_VIEWSP
    }
}

__device__ __forceinline__ float sum8(u32x4 hi, u32x4 lo) {
    const half8 h = __builtin_bit_cast(half8, hi), l = __builtin_bit_cast(half8, lo);
    float s = 0.f, t = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        s += (float)h[j];
        t += (float)l[j];
    }
    return s + t * LO_INV;
}

// Output block N x K at columns [k0, k0+K) of an instance whose X array is XW wide and whose partial block is KW
// wide.  Waves form a WN x (8/WN) grid; each owns TR x TC MFMA tiles.
template <int N, int K, int XW, int KW, int WN, int TR, int TC, bool ALPHA>
__device__ __forceinline__ void dw_gemm(const DwArgs& a, const Src src, int k0, bool lead, int64_t chunk_begin,
                                        int64_t chunk_end, float* __restrict__ part, u32x4* __restrict__ smem) {
    static_assert(WN * TR * 32 == N, "row tiling");
    constexpr int YU = 8 * N, XU = 8 * K;                      // 16-byte units per chunk (4 blocks x 2 planes x width)
    constexpr int NY = YU / DWT, NX = (XU + DWT - 1) / DWT;
    static_assert(YU % DWT == 0, "staging shape");
    constexpr bool XFULL = XU % DWT == 0;
    constexpr int BUF = YU + XU + 8;                           // + 32 floats of d_sigma
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int wn = wave % WN, wk = wave / WN;
    const bool mma_wave = wk * TC * 32 < K;
    const int64_t M = a.M;

    f32x16 acc1[TR][TC], acc2[TR][TC];
#pragma unroll
    for (int r = 0; r < TR; ++r)
#pragma unroll
        for (int c = 0; c < TC; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc1[r][c][e] = acc2[r][c][e] = 0.f;
    float bsum = 0.f, asum = 0.f, absum = 0.f;

    // Two chunks are always in flight (register sets A and B): with one chunk the loop runs at one HBM latency per
    // 48 KB per CU.  The barrier is LDS-only (no vmcnt(0)), so the younger set's loads stay in flight across it.
    // (Macros, not lambdas taking the sets by reference: those left the sets in scratch memory.)
    u32x4 ryA[NY], rxA[NX], ryB[NY], rxB[NX];
    float rdaA = 0.f, rdaB = 0.f;
#define DW_PREFETCH(RY, RX, RDA, CHUNK)                                                                   \
    if ((CHUNK) < chunk_end) {                                                                            \
        const u32x4* py = src.y + (CHUNK) * YU + tid;                                                     \
        _Pragma("unroll") for (int j = 0; j < NY; ++j) RY[j] = py[j * DWT];                               \
        _Pragma("unroll") for (int j = 0; j < NX; ++j) {                                                  \
            const int u = tid + j * DWT;                                                                  \
            RX[j] = u32x4{0u, 0u, 0u, 0u};                                                               \
            if (XFULL || u < XU)                                                                          \
                RX[j] = src.x[((((CHUNK) * 4 + u / (2 * K)) * XW) + k0 + ((u % (2 * K)) >> 1)) * 2 + (u & 1)];       \
        }                                                                                                 \
        if (ALPHA && tid < CHP) {                                                                         \
            const int64_t row = (CHUNK) * CHP + tid;                                                      \
            RDA = row < M ? a.d_raw[row * (a.C + 1) + a.C] : 0.f;                                         \
        }                                                                                                 \
    }
#define DW_STAGE(RY, RX, RDA, B)                                                                          \
    {                                                                                                     \
        u32x4* Ys_ = smem + (B) * BUF;                                                                    \
        u32x4* Xs_ = Ys_ + YU;                                                                            \
        /* global unit u = ((block * W + w) * 2 + half) holds {hi x4, lo x4} of 4 points: the two quads go to the */ \
        /* hi / lo fragment planes of the LDS image [block][plane][w][8 points] (8-byte writes, conflict free)   */ \
        _Pragma("unroll") for (int j = 0; j < NY; ++j) {                                                  \
            const int u = tid + j * DWT, mb = u / (2 * N), rem = u % (2 * N);                             \
            u32x2* d = reinterpret_cast<u32x2*>(Ys_) + ((mb * 2) * N + (rem >> 1)) * 2 + (rem & 1);       \
            d[0] = u32x2{RY[j].x, RY[j].y};                                                               \
            d[2 * N] = u32x2{RY[j].z, RY[j].w};                                                           \
        }                                                                                                 \
        _Pragma("unroll") for (int j = 0; j < NX; ++j) {                                                  \
            const int u = tid + j * DWT, mb = u / (2 * K), rem = u % (2 * K);                             \
            if (XFULL || u < XU) {                                                                        \
                u32x2* d = reinterpret_cast<u32x2*>(Xs_) + ((mb * 2) * K + (rem >> 1)) * 2 + (rem & 1);   \
                d[0] = u32x2{RX[j].x, RX[j].y};                                                           \
                d[2 * K] = u32x2{RX[j].z, RX[j].w};                                                       \
            }                                                                                             \
        }                                                                                                 \
        if (ALPHA && tid < CHP) reinterpret_cast<float*>(Xs_ + XU)[tid] = RDA;                            \
        /* buffer B was last read two chunks ago, and every wave has passed the barrier in between */    \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");                                   \
        __builtin_amdgcn_s_barrier();                                                                     \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");                                   \
    }
    auto compute = [&](int b) {
        const u32x4* Yl = smem + b * BUF;
        const u32x4* Xl = Yl + YU;
        const float* da = reinterpret_cast<const float*>(Xl + XU);
        if (mma_wave) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int mb = ks * 2 + lh;
                half8 ah[TR], al[TR], bh[TC], bl[TC];
#pragma unroll
                for (int r = 0; r < TR; ++r) {
                    ah[r] = __builtin_bit_cast(half8, Yl[(mb * 2 + 0) * N + (wn * TR + r) * 32 + lr]);
                    al[r] = __builtin_bit_cast(half8, Yl[(mb * 2 + 1) * N + (wn * TR + r) * 32 + lr]);
                }
#pragma unroll
                for (int c = 0; c < TC; ++c) {
                    bh[c] = __builtin_bit_cast(half8, Xl[(mb * 2 + 0) * K + (wk * TC + c) * 32 + lr]);
                    bl[c] = __builtin_bit_cast(half8, Xl[(mb * 2 + 1) * K + (wk * TC + c) * 32 + lr]);
                }
#pragma unroll
                for (int r = 0; r < TR; ++r)
#pragma unroll
                    for (int c = 0; c < TC; ++c) acc1[r][c] = mfma16(ah[r], bh[c], acc1[r][c]);
#pragma unroll
                for (int r = 0; r < TR; ++r)
#pragma unroll
                    for (int c = 0; c < TC; ++c) acc2[r][c] = mfma16(ah[r], bl[c], acc2[r][c]);
#pragma unroll
                for (int r = 0; r < TR; ++r)
#pragma unroll
                    for (int c = 0; c < TC; ++c) acc2[r][c] = mfma16(al[r], bh[c], acc2[r][c]);
            }
        }
        if (lead && src.bias && tid < N) {
            float s = 0.f;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) s += sum8(Yl[(mb * 2) * N + tid], Yl[(mb * 2 + 1) * N + tid]);
            bsum += s;
        }
        if (ALPHA && tid < K) {
            float s = 0.f, sb = 0.f;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const half8 h = __builtin_bit_cast(half8, Xl[(mb * 2) * K + tid]);
                const half8 l = __builtin_bit_cast(half8, Xl[(mb * 2 + 1) * K + tid]);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = da[mb * 8 + j];
                    s += d * ((float)h[j] + (float)l[j] * LO_INV);
                    sb += d;
                }
            }
            asum += s;
            absum += sb;
        }
    };

    DW_PREFETCH(ryA, rxA, rdaA, chunk_begin);
    DW_PREFETCH(ryB, rxB, rdaB, chunk_begin + 1);
    for (int64_t chunk = chunk_begin; chunk < chunk_end; chunk += 2) {
        DW_STAGE(ryA, rxA, rdaA, 0);
        DW_PREFETCH(ryA, rxA, rdaA, chunk + 2);
        __builtin_amdgcn_sched_barrier(0);   // keep the loads HERE: the scheduler otherwise sinks them below the MFMAs
        compute(0);
        if (chunk + 1 < chunk_end) {
            DW_STAGE(ryB, rxB, rdaB, 1);
            DW_PREFETCH(ryB, rxB, rdaB, chunk + 3);
            __builtin_amdgcn_sched_barrier(0);
            compute(1);
        }
    }
#undef DW_PREFETCH
#undef DW_STAGE

    // partial block -> workspace: [N][KW] then bias [N] (then alpha row [256] + alpha bias)
    if (mma_wave) {
#pragma unroll
        for (int r = 0; r < TR; ++r)
#pragma unroll
            for (int c = 0; c < TC; ++c)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = (wn * TR + r) * 32 + acc_row(e, lane);
                    part[(int64_t)row * KW + k0 + (wk * TC + c) * 32 + lr] = acc1[r][c][e] + acc2[r][c][e] * LO_INV;
                }
    }
    if (lead && tid < N) part[(int64_t)N * KW + tid] = src.bias ? bsum : 0.f;
    if (ALPHA && tid < K) {
        part[(int64_t)N * KW + N + k0 + tid] = asum;
        if (lead && tid == 0) part[(int64_t)N * KW + N + 256] = absum;
    }
}

// rgb head: dW_rgb[c][j] = sum_pt d_rgb[pt][c] * hv[pt][j], db_rgb[c] = sum_pt d_rgb[pt][c]   (unscaled d_raw, f32).
// Batches of 512 points: d_raw staged in LDS, then every thread (column j, phase ph) streams 16 blocks of hv with
// all its loads independent.
__device__ __forceinline__ void dw_rgb(const DwArgs& a, int64_t blk_begin, int64_t blk_end, float* __restrict__ part,
                                       float* __restrict__ smem) {
    constexpr int BB = 64;                                           // blocks of 8 points per batch
    const int tid = threadIdx.x, j = tid & 127, ph = tid >> 7;     // ph: block phase 0..3
    const u32x4* hv = reinterpret_cast<const u32x4*>(a.acts + sact_hv(m_pad(a.M)));
    const int C = a.C;
    const int64_t M = a.M;
    float* dr = smem;                                                // [BB*8][4]
    float s[3] = {0.f, 0.f, 0.f}, sb[3] = {0.f, 0.f, 0.f};
    for (int64_t b0 = blk_begin; b0 < blk_end; b0 += BB) {
        __syncthreads();
        {   // 512 points x 4 slots, one point per thread
            const int64_t m = b0 * 8 + tid;
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < M && m < blk_end * 8) {
                const float* src = a.d_raw + m * (C + 1);
                g.x = src[0];
                if (C > 1) g.y = src[1];
                if (C > 2) g.z = src[2];
            }
            reinterpret_cast<float4*>(dr)[tid] = g;
        }
        __syncthreads();
        u32x4 hh[BB / 4], hl[BB / 4];
#pragma unroll
        for (int i = 0; i < BB / 4; ++i) {
            const int64_t mb = b0 + ph + 4 * i;
            hh[i] = hl[i] = u32x4{0u, 0u, 0u, 0u};
            if (mb < blk_end) {
                hh[i] = hv[(mb * ACT_HV_W + j) * 2];          // points 0-3: {hi x4, lo x4}
                hl[i] = hv[(mb * ACT_HV_W + j) * 2 + 1];      // points 4-7
            }
        }
#pragma unroll
        for (int i = 0; i < BB / 4; ++i) {
            const half8 p0 = __builtin_bit_cast(half8, hh[i]), p1 = __builtin_bit_cast(half8, hl[i]);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const half8& pq = q < 4 ? p0 : p1;
                const float x = (float)pq[q & 3] + (float)pq[4 + (q & 3)] * LO_INV;
                const float4 g = reinterpret_cast<const float4*>(dr)[(ph + 4 * i) * 8 + q];   // zero beyond the range
                s[0] += g.x * x;
                s[1] += g.y * x;
                s[2] += g.z * x;
                sb[0] += g.x;
                sb[1] += g.y;
                sb[2] += g.z;
            }
        }
    }
    __syncthreads();
    float* red = smem + BB * 8 * 4;
    if (ph > 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            red[((ph - 1) * 6 + c) * 128 + j] = s[c];
            red[((ph - 1) * 6 + 3 + c) * 128 + j] = sb[c];
        }
    }
    __syncthreads();
    if (ph == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = 0.f;
            if (c < 3 && c < C) v = ((s[c] + red[(0 * 6 + c) * 128 + j]) + red[(1 * 6 + c) * 128 + j]) + red[(2 * 6 + c) * 128 + j];
            part[c * 128 + j] = v;
        }
        if (j < 4) {
            float v = 0.f;
            if (j < 3 && j < C) {
                const float own = j == 0 ? sb[0] : j == 1 ? sb[1] : sb[2];
                v = ((own + red[(0 * 6 + 3 + j) * 128 + j]) + red[(1 * 6 + 3 + j) * 128 + j]) + red[(2 * 6 + 3 + j) * 128 + j];
            }
            part[4 * 128 + j] = v;
        }
    }
}

constexpr size_t DWH_SMEM = 2 * (size_t)(8 * 256 + 8 * 128 + 8) * 16;        // two chunk images of the 256 x 128 block: 98 560 B
constexpr size_t DWH_SMEM_SMALL = 2 * (size_t)(8 * 256 + 8 * 64 + 8) * 16;   // 256 x 64 block: 82 176 B

__device__ __forceinline__ void chunk_range(const DwArgs& a, int inst, int split, int64_t& cb, int64_t& ce) {
    const int64_t nchunks = m_pad(a.M) / CHP;
    const int64_t per = (nchunks + dwh_splits(inst) - 1) / dwh_splits(inst);
    cb = (int64_t)split * per;
    ce = cb + per;
    if (cb > nchunks) cb = nchunks;
    if (ce > nchunks) ce = nchunks;
}

// the eight 256x256 instances as column-half pairs + the 128x256 views block: one workgroup per CU
__global__ __launch_bounds__(DWT, 2) void mlp_dw_split_big_kernel(DwArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32x4 smem_u[];
    const int id = blockIdx.x;
    int inst, split, half = 0;
    if (id < DWH_PAIR_BLOCKS) {           // id = 16*q + 8*half + x  <->  pair q*8 + x
        const int pair = (id >> 4) * 8 + (id & 7);
        half = (id >> 3) & 1;
        inst = pair / 15;                 // DW_L1 .. DW_FEAT
        split = pair % 15;
    } else {
        inst = DW_VIEWSF;
        split = id - DWH_PAIR_BLOCKS;
    }
    int64_t cb, ce;
    chunk_range(a, inst, split, cb, ce);
    float* part = a.ws + dwh_inst_offset(inst) + (int64_t)split * dw_inst_floats(inst);
    const Src src = inst_src(a, inst);
    if (inst == DW_FEAT) dw_gemm<256, 128, 256, 256, 4, 2, 2, true>(a, src, half * 128, half == 0, cb, ce, part, smem_u);
    else if (inst <= DW_L7) dw_gemm<256, 128, 256, 256, 4, 2, 2, false>(a, src, half * 128, half == 0, cb, ce, part, smem_u);
    else dw_gemm<128, 256, 256, 256, 2, 2, 2, false>(a, src, 0, true, cb, ce, part, smem_u);
}

// the thin instances: L0 and L5P (256 x 64, X = PE), VIEWSP (128 x 32, X = PE(dir)), rgb head (VALU)
__global__ __launch_bounds__(DWT, 2) void mlp_dw_split_small_kernel(DwArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32x4 smem_u[];
    const int inst = DW_L0 + blockIdx.x / 64, split = blockIdx.x % 64;
    int64_t cb, ce;
    chunk_range(a, inst, split, cb, ce);
    float* part = a.ws + dwh_inst_offset(inst) + (int64_t)split * dw_inst_floats(inst);
    if (inst == DW_RGB) {
        dw_rgb(a, cb * 4, ce * 4, part, reinterpret_cast<float*>(smem_u));
        return;
    }
    const Src src = inst_src(a, inst);
    if (inst == DW_VIEWSP) dw_gemm<128, 32, 32, 32, 4, 1, 1, false>(a, src, 0, true, cb, ce, part, smem_u);
    else dw_gemm<256, 64, 64, 64, 4, 2, 1, false>(a, src, 0, true, cb, ce, part, smem_u);   // DW_L0, DW_L5P
}

}  // namespace

int benerf_mlp_dw_reduce_launch(const float* ws, const BenerfMlpGrads* grads, int channels, int accumulate, int split_mode,
                                const float* absmax, hipStream_t stream);

int benerf_mlp_dw_split_launch(int channels, int64_t M, const float* d_raw, const float* acts, const float* dacts, float* dw_ws,
                               const BenerfMlpGrads* grads, int accumulate, hipStream_t stream) {
    DwArgs a;
    a.d_raw = d_raw;
    a.acts = acts;
    a.dacts = dacts;
    a.ws = dw_ws;
    a.M = M;
    a.C = channels;
    static_assert(DW_L0 + 1 == DW_L5P && DW_L5P + 1 == DW_VIEWSP && DW_VIEWSP + 1 == DW_RGB, "small-kernel instance order");
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)mlp_dw_split_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DWH_SMEM);
        (void)hipFuncSetAttribute((const void*)mlp_dw_split_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)DWH_SMEM_SMALL);
        attr_done = true;
    }
    hipLaunchKernelGGL(mlp_dw_split_small_kernel, dim3(mlp::DWH_SMALL_BLOCKS), dim3(DWT), DWH_SMEM_SMALL, stream, a);
    BENERF_LAUNCH_CHECK("mlp_bwd(dw small, split)");
    hipLaunchKernelGGL(mlp_dw_split_big_kernel, dim3(mlp::DWH_BIG_BLOCKS), dim3(DWT), DWH_SMEM, stream, a);
    BENERF_LAUNCH_CHECK("mlp_bwd(dw, split)");
    return benerf_mlp_dw_reduce_launch(dw_ws, grads, channels, accumulate, 1, dacts + mlp::sdact_scale(mlp::m_pad(M)), stream);
}

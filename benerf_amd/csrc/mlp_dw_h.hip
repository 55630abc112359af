// K3 backward part 2, f16 variant of mlp_dw.hip: dW_l = dY_l^T X_l over all sample points with f16 operands (the SH
// arrays the split forward / f16 dX kernels save, mlp_split.h), one v_mfma_f32_32x32x16_f16 per product block, f32
// accumulation, + bias sums and the alpha / rgb heads on the VALU.
//
// An SH array is blocks of 8 points, feature-major, one 16-byte unit per (block, feature) = exactly the MFMA fragment
// of a contraction over points, so a workgroup (8 waves, one per CU) copies 32-point chunks (two MFMA k-steps)
// verbatim into a triple-buffered LDS image [block][feature][8 points] with three chunks in flight in registers and
// one LDS-only barrier per chunk.  A workgroup holds a whole 256 x 256 output block (128 accumulator registers x 8
// waves), so every operand byte is read exactly once: HBM traffic = the algorithmic bytes, 2 bytes per element.
// The kernel is HBM-bound (9.9 KB per point against 16 MFMAs per wave and chunk).
// Gradient operands carry the per-call power-of-two scale s_s of the dX kernel (exact; divided out by the reduce
// kernel); activations are stored unscaled (f16 subnormals keep an absolute floor of 3e-8).
#include "mlp_split.h"

namespace {
using namespace mlp;

constexpr int DWT = 512;
constexpr int CHP = 32;             // points per chunk = 4 blocks of 8 = two MFMA k-steps
constexpr int CHB = CHP / 8;
// 16-byte unit as a first-class vector (arrays of HIP's uint4 struct were left in scratch memory by the compiler)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct DwArgs {
    const float* d_raw;
    const float* acts;
    const float* dacts;
    float* ws;
    int64_t M;
    int C;
};

struct Src {
    const u32x4* y;    // SH array of width N (16-byte units)
    const u32x4* x;    // SH array of width K ... or, for the thin instances (XROWS), f32 rows [Mp][K]
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
        case DW_L0: return {U(D + sdact_h(Mp, 0)), U(A + sact_pe32(Mp)), true};       // X = PE as f32 rows
        case DW_L5P: return {U(D + sdact_h(Mp, 5)), U(A + sact_pe32(Mp)), false};
        default: return {U(D + sdact_hv(Mp)), U(A + sact_ped32(Mp)), false};          // DW_VIEWSP: PE(dir) rows
    }
}

// Output block N x K (the whole instance: K = width of X).  Waves form a WN x (8/WN) grid; each owns TR x TC MFMA
// tiles: per 32-point chunk (= two MFMA k-steps) acc += Y^T X.  Three chunks are in flight in registers (sets A, B,
// C) and the LDS image is triple-buffered, so there is one LDS-only barrier per chunk.
template <int N, int K, int WN, int TR, int TC, bool ALPHA, bool XROWS = false>
__device__ __forceinline__ void dw_gemm(const DwArgs& a, const Src src, int64_t chunk_begin, int64_t chunk_end,
                                        float* __restrict__ part, u32x4* __restrict__ smem) {
    static_assert(WN * TR * 32 == N, "row tiling");
    constexpr int YU = CHB * N, XU = CHB * K;                  // 16-byte units per chunk
    static_assert(!XROWS || (CHP * K / 4 <= DWT), "one float4 of the f32 rows per thread");
    constexpr int NY = (YU + DWT - 1) / DWT, NX = (XU + DWT - 1) / DWT;
    constexpr bool YFULL = YU % DWT == 0, XFULL = XU % DWT == 0;
    // thin instances: X = PE / PE(dir) rows, whose values repeat along a ray (PE(dir)) or across the whole batch (the
    // raw direction components): their f16 rounding error would be COHERENT over the sum, so they are staged as hi + lo
    // (two MFMAs per block, 22-bit operand; the rows are f32 anyway)
    constexpr int XPL = XROWS ? 2 : 1;
    constexpr int BUF = YU + XPL * XU + CHP / 4;               // + CHP floats of d_sigma
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int wn = wave % WN, wk = wave / WN;
    const bool mma_wave = wk * TC * 32 < K;
    const int64_t M = a.M;

    f32x16 acc[TR][TC];
#pragma unroll
    for (int r = 0; r < TR; ++r)
#pragma unroll
        for (int c = 0; c < TC; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][c][e] = 0.f;
    float bsum = 0.f, asum = 0.f, absum = 0.f;

    u32x4 ryA[NY], rxA[NX], ryB[NY], rxB[NX], ryC[NY], rxC[NX];
    float rdaA = 0.f, rdaB = 0.f, rdaC = 0.f;
    // Loads are UNCONDITIONAL (a chunk index past the range is clamped to the last chunk and its staged dY zeroed):
    // with the loads under branches the compiler's waitcnt bookkeeping fell back to vmcnt(0) at every stage, i.e. one
    // chunk in flight instead of three.
#define DW_PREFETCH(RY, RX, RDA, CHUNK)                                                                   \
    {                                                                                                     \
        const int64_t cc = (CHUNK) < chunk_end ? (CHUNK) : chunk_end - 1;                                 \
        const u32x4* py = src.y + cc * YU + tid;                                                          \
        const u32x4* px = src.x + cc * XU + tid;                                                          \
        _Pragma("unroll") for (int j = 0; j < NY; ++j)                                                    \
            if (YFULL || tid + j * DWT < YU) RY[j] = py[j * DWT];                                         \
        if (XROWS) {   /* f32 rows [point][K]: one float4 (4 features of a point) per thread */                \
            if (tid < CHP * K / 4)                                                                        \
                RX[0] = reinterpret_cast<const u32x4*>(reinterpret_cast<const float*>(src.x) + cc * (CHP * K))[tid]; \
        } else {                                                                                          \
            _Pragma("unroll") for (int j = 0; j < NX; ++j)                                                \
                if (XFULL || tid + j * DWT < XU) RX[j] = px[j * DWT];                                     \
        }                                                                                                 \
        if (ALPHA) {                                                                                      \
            const int64_t row = cc * CHP + (tid & (CHP - 1));                                             \
            RDA = a.d_raw[(row < M ? row : M - 1) * (a.C + 1) + a.C];                                     \
            if (row >= M) RDA = 0.f;                                                                      \
        }                                                                                                 \
    }
    // global unit u = block * W + w of the chunk is unit u of the LDS image [block][w] (16-byte writes, conflict free)
#define DW_STAGE(RY, RX, RDA, B, VALID)                                                                   \
    {                                                                                                     \
        u32x4* Ys_ = smem + (B) * BUF;                                                                    \
        u32x4* Xs_ = Ys_ + YU;                                                                            \
        _Pragma("unroll") for (int j = 0; j < NY; ++j) {                                                  \
            const int u = tid + j * DWT;                                                                  \
            if (YFULL || u < YU) Ys_[u] = (VALID) ? RY[j] : u32x4{0u, 0u, 0u, 0u};   /* past the range: contributes nothing */ \
        }                                                                                                 \
        if (XROWS) {   /* round to f16 and scatter the 4 features of this thread's point into their fragments */ \
            if (tid < CHP * K / 4) {                                                                      \
                const int p = tid / (K / 4), w0 = (tid % (K / 4)) * 4;                                    \
                _Float16* img = reinterpret_cast<_Float16*>(Xs_);                                         \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                           \
                    const float v = __uint_as_float(RX[0][i]);                                            \
                    const _Float16 hi = (_Float16)v;                                                      \
                    img[((p >> 3) * K + w0 + i) * 8 + (p & 7)] = hi;                                      \
                    img[XU * 8 + ((p >> 3) * K + w0 + i) * 8 + (p & 7)] = (_Float16)(v - (float)hi);      \
                }                                                                                         \
            }                                                                                             \
        } else                                                                                            \
        _Pragma("unroll") for (int j = 0; j < NX; ++j) {                                                  \
            const int u = tid + j * DWT;                                                                  \
            if (XFULL || u < XU) Xs_[u] = RX[j];                                                          \
        }                                                                                                 \
        if (ALPHA && tid < CHP) reinterpret_cast<float*>(Xs_ + XPL * XU)[tid] = (VALID) ? RDA : 0.f;      \
        /* buffer B was last read three chunks ago, and every wave has passed two barriers in between */ \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");                                   \
        __builtin_amdgcn_s_barrier();                                                                     \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");                                   \
    }
    auto compute = [&](int b) {
        const u32x4* Yl = smem + b * BUF;
        const u32x4* Xl = Yl + YU;
        const float* da = reinterpret_cast<const float*>(Xl + XPL * XU);
        if (mma_wave) {
#pragma unroll
            for (int ks = 0; ks < CHP / 16; ++ks) {
                half8 ay[TR], bx[TC];
#pragma unroll
                for (int r = 0; r < TR; ++r) ay[r] = __builtin_bit_cast(half8, Yl[(ks * 2 + lh) * N + (wn * TR + r) * 32 + lr]);
#pragma unroll
                for (int c = 0; c < TC; ++c) bx[c] = __builtin_bit_cast(half8, Xl[(ks * 2 + lh) * K + (wk * TC + c) * 32 + lr]);
#pragma unroll
                for (int r = 0; r < TR; ++r)
#pragma unroll
                    for (int c = 0; c < TC; ++c) acc[r][c] = mfma16(ay[r], bx[c], acc[r][c]);
                if (XROWS) {
#pragma unroll
                    for (int c = 0; c < TC; ++c) bx[c] = __builtin_bit_cast(half8, Xl[XU + (ks * 2 + lh) * K + (wk * TC + c) * 32 + lr]);
#pragma unroll
                    for (int r = 0; r < TR; ++r)
#pragma unroll
                        for (int c = 0; c < TC; ++c) acc[r][c] = mfma16(ay[r], bx[c], acc[r][c]);
                }
            }
        }
        if (src.bias && tid < N) {
            float s = 0.f;
#pragma unroll
            for (int mb = 0; mb < CHB; ++mb) {
                const half8 h = __builtin_bit_cast(half8, Yl[mb * N + tid]);
#pragma unroll
                for (int j = 0; j < 8; ++j) s += (float)h[j];
            }
            bsum += s;
        }
        // the alpha head rides on waves 4..7 (column tid - 256): waves 0..3 already carry the bias sums, and with both on the
        // same four waves the FEAT workgroups were VALU-bound and finished 30 % behind every other instance (trace_dw.py)
        if (ALPHA && tid >= DWT - K) {
            const int ka = tid - (DWT - K);
            float s = 0.f, sb = 0.f;
#pragma unroll
            for (int mb = 0; mb < CHB; ++mb) {
                const half8 h = __builtin_bit_cast(half8, Xl[mb * K + ka]);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = da[mb * 8 + j];
                    s += d * (float)h[j];
                    sb += d;
                }
            }
            asum += s;
            absum += sb;
        }
    };

    if (chunk_begin < chunk_end) {
        DW_PREFETCH(ryA, rxA, rdaA, chunk_begin);
        DW_PREFETCH(ryB, rxB, rdaB, chunk_begin + 1);
        DW_PREFETCH(ryC, rxC, rdaC, chunk_begin + 2);
        for (int64_t chunk = chunk_begin; chunk < chunk_end; chunk += 3) {
            DW_STAGE(ryA, rxA, rdaA, 0, true);
            DW_PREFETCH(ryA, rxA, rdaA, chunk + 3);
            __builtin_amdgcn_sched_barrier(0);   // keep the loads HERE: the scheduler otherwise sinks them below the MFMAs
            compute(0);
            const bool v1 = chunk + 1 < chunk_end, v2 = chunk + 2 < chunk_end;
            DW_STAGE(ryB, rxB, rdaB, 1, v1);
            DW_PREFETCH(ryB, rxB, rdaB, chunk + 4);
            __builtin_amdgcn_sched_barrier(0);
            compute(1);
            DW_STAGE(ryC, rxC, rdaC, 2, v2);
            DW_PREFETCH(ryC, rxC, rdaC, chunk + 5);
            __builtin_amdgcn_sched_barrier(0);
            compute(2);
        }
    }
#undef DW_PREFETCH
#undef DW_STAGE

    // partial block -> workspace: [N][K] then bias [N] (then alpha row [256] + alpha bias).  Block and bias stay at
    // the gradient scale s_s (the reduce kernel divides it out); the alpha row is unscaled (d_raw is)
    if (mma_wave) {
#pragma unroll
        for (int r = 0; r < TR; ++r)
#pragma unroll
            for (int c = 0; c < TC; ++c)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = (wn * TR + r) * 32 + acc_row(e, lane);
                    part[(int64_t)row * K + (wk * TC + c) * 32 + lr] = acc[r][c][e];
                }
    }
    if (tid < N) part[(int64_t)N * K + tid] = src.bias ? bsum : 0.f;
    if (ALPHA && tid >= DWT - K) {
        part[(int64_t)N * K + N + tid - (DWT - K)] = asum;
        if (tid == DWT - K) part[(int64_t)N * K + N + 256] = absum;
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
        u32x4 hh[BB / 4];
#pragma unroll
        for (int i = 0; i < BB / 4; ++i) {
            const int64_t mb = b0 + ph + 4 * i;
            hh[i] = u32x4{0u, 0u, 0u, 0u};
            if (mb < blk_end) hh[i] = hv[mb * ACT_HV_W + j];     // 8 points of column j
        }
#pragma unroll
        for (int i = 0; i < BB / 4; ++i) {
            const half8 pq = __builtin_bit_cast(half8, hh[i]);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float x = (float)pq[q];
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

constexpr size_t DWH_SMEM = 3 * (size_t)(CHB * 256 + CHB * 256 + CHP / 4) * 16;       // three chunk images of the 256 x 256 block: 98 688 B
constexpr size_t DWH_SMEM_SMALL = 3 * (size_t)(CHB * 256 + 2 * CHB * 64 + CHP / 4) * 16;  // 256 x 64 block, X as hi + lo: 74 112 B

__device__ __forceinline__ void chunk_range(const DwArgs& a, int inst, int split, int64_t& cb, int64_t& ce) {
    const int64_t nchunks = m_pad(a.M) / CHP;
    const int64_t per = (nchunks + dwh_splits(inst) - 1) / dwh_splits(inst);
    cb = (int64_t)split * per;
    ce = cb + per;
    if (cb > nchunks) cb = nchunks;
    if (ce > nchunks) ce = nchunks;
}

// -DBENERF_TRACE_DW: thread 0 of every workgroup stamps the 100 MHz wall clock at its start and end behind the partial sums in
// the workspace (u64 [kernel: 0 small, 1 big][512 workgroups][2]) - tools/experiments/trace_dw.py prints per-instance finish times.
#ifdef BENERF_TRACE_DW
#define DW_TRACE(kern, which) do { if (threadIdx.x == 0) reinterpret_cast<unsigned long long*>(a.ws + ((dwh_inst_offset(DW_COUNT) + 63) & ~63LL))[((kern) * 512 + blockIdx.x) * 2 + (which)] = wall_clock64(); } while (0)
#else
#define DW_TRACE(kern, which) do { } while (0)
#endif

// the eight 256x256 instances + the 128x256 views block: one workgroup per CU, every operand byte read once
__global__ __launch_bounds__(DWT, 2) void mlp_dw_f16_big_kernel(DwArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32x4 smem_u[];
    DW_TRACE(1, 0);
    const int inst = dwh_big_inst(blockIdx.x), split = dwh_big_split(blockIdx.x);
    int64_t cb, ce;
    chunk_range(a, inst, split, cb, ce);
    float* part = a.ws + dwh_inst_offset(inst) + (int64_t)split * dw_inst_floats(inst);
    const Src src = inst_src(a, inst);
    if (inst == DW_FEAT) dw_gemm<256, 256, 4, 2, 4, true>(a, src, cb, ce, part, smem_u);
    else if (inst <= DW_L7) dw_gemm<256, 256, 4, 2, 4, false>(a, src, cb, ce, part, smem_u);
    else dw_gemm<128, 256, 2, 2, 2, false>(a, src, cb, ce, part, smem_u);
    DW_TRACE(1, 1);
}

// the thin instances: L0 and L5P (256 x 64, X = PE), VIEWSP (128 x 32, X = PE(dir)), rgb head (VALU)
__global__ __launch_bounds__(DWT, 4) void mlp_dw_f16_small_kernel(DwArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32x4 smem_u[];
    DW_TRACE(0, 0);
    const int inst = dwh_thin_inst(blockIdx.x), split = dwh_thin_split(blockIdx.x);
    int64_t cb, ce;
    chunk_range(a, inst, split, cb, ce);
    float* part = a.ws + dwh_inst_offset(inst) + (int64_t)split * dw_inst_floats(inst);
    if (inst == DW_RGB) {
        dw_rgb(a, cb * CHB, ce * CHB, part, reinterpret_cast<float*>(smem_u));
        DW_TRACE(0, 1);
        return;
    }
    const Src src = inst_src(a, inst);
    if (inst == DW_VIEWSP) dw_gemm<128, 32, 4, 1, 1, false, true>(a, src, cb, ce, part, smem_u);
    else dw_gemm<256, 64, 4, 2, 1, false, true>(a, src, cb, ce, part, smem_u);   // DW_L0, DW_L5P
    DW_TRACE(0, 1);
}

}  // namespace

int benerf_mlp_dw_reduce_launch(const float* ws, const BenerfMlpGrads* grads, int channels, int accumulate, int split_mode,
                                const float* grad_info, const float* pe_weights, hipStream_t stream, const BenerfMlpParams* params);

int benerf_mlp_dw_split_launch(int channels, int64_t M, const float* d_raw, const float* acts, const float* dacts, float* dw_ws,
                               const BenerfMlpGrads* grads, int accumulate, const float* pe_weights, hipStream_t stream) {
    DwArgs a;
    a.d_raw = d_raw;
    a.acts = acts;
    a.dacts = dacts;
    a.ws = dw_ws;
    a.M = M;
    a.C = channels;
    static_assert(DW_L0 + 1 == DW_L5P && DW_L5P + 1 == DW_VIEWSP && DW_VIEWSP + 1 == DW_RGB, "small-kernel instance order");
    static BenerfLdsAttr attr_big, attr_small;      // once per device
    if (!benerf_lds_attr(attr_big, (const void*)mlp_dw_f16_big_kernel, (int)DWH_SMEM) ||
        !benerf_lds_attr(attr_small, (const void*)mlp_dw_f16_small_kernel, (int)DWH_SMEM_SMALL)) {
        benerf_set_error("mlp_bwd(dw, f16): cannot reserve LDS");
        return BENERF_EHIP;
    }
    hipLaunchKernelGGL(mlp_dw_f16_small_kernel, dim3(mlp::DWH_SMALL_BLOCKS), dim3(DWT), DWH_SMEM_SMALL, stream, a);
    BENERF_LAUNCH_CHECK("mlp_bwd(dw small, f16)");
    hipLaunchKernelGGL(mlp_dw_f16_big_kernel, dim3(mlp::DWH_BIG_BLOCKS), dim3(DWT), DWH_SMEM, stream, a);
    BENERF_LAUNCH_CHECK("mlp_bwd(dw, f16)");
    return benerf_mlp_dw_reduce_launch(dw_ws, grads, channels, accumulate, 1, dacts + mlp::sdact_info(mlp::m_pad(M)), pe_weights, stream, nullptr);
}

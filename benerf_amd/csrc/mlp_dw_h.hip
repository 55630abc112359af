// K3 backward part 2, split-f16 variant of mlp_dw.hip: dW_l = dY_l^T X_l over all sample points as three f16 MFMAs
// per product block, f32 accumulation, + bias sums and the alpha / rgb heads on the VALU.
//
// Operands arrive in the ST layout the split forward / dX kernels write (mlp_split.h): blocks of 8 points,
// feature-major, 16-byte units {hi x4, lo x4} - one split away from the MFMA fragment order for a contraction over
// points.  A workgroup (8 waves, one per CU) copies 16-point chunks (one MFMA k-step) into a triple-buffered LDS
// image [block][plane][feature][8 points] (each unit lands as two 8-byte quads; conflict-free fragment reads, no
// transposition pass) with three chunks in flight in registers and one LDS-only barrier per chunk.
// The three products of a block go into ONE accumulator set: while staging, every operand unit is rescaled by its
// launch-wide power of two so that its lo part can be used UNSCALED ('dW operand formats', mlp_split.h).
// With one set a workgroup holds a whole 256 x 256 output block (128 accumulator registers x 8 waves), so every
// operand byte is read exactly once: HBM traffic = the algorithmic bytes.  (With two accumulator sets the block was
// 256 x 128 and dY was streamed twice: 13.3 GB against 9.4 GB of operands per launch at M = 522k.)
#include "mlp_split.h"

namespace {
using namespace mlp;

constexpr int DWT = 512;
constexpr int CHP = 16;             // points per chunk = 2 blocks of 8 = one MFMA k-step
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
    const u32x4* x;    // ST array of width K ... or, for the thin instances (XROWS), f32 rows [Mp][K]
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
// tiles in ONE accumulator set: per 16-point chunk (= one MFMA k-step) acc += Yh Xh + Yh Xl + Yl Xh with unscaled
// lo parts (mlp_split.h, 'dW operand formats').  Three chunks are in flight in registers (sets A, B, C) and the LDS
// image is triple-buffered, so there is one LDS-only barrier per chunk and every operand byte is read once.
template <int N, int K, int WN, int TR, int TC, bool ALPHA, bool XROWS = false>
__device__ __forceinline__ void dw_gemm(const DwArgs& a, const Src src, int kY, int kX, int64_t chunk_begin, int64_t chunk_end,
                                        float* __restrict__ part, u32x4* __restrict__ smem) {
    static_assert(WN * TR * 32 == N, "row tiling");
    constexpr int YU = 4 * N, XU = 4 * K;                      // 16-byte units per chunk (2 blocks x width x 2 halves)
    static_assert(!XROWS || (CHP * K / 4 <= DWT), "one float4 of the f32 rows per thread");
    constexpr int NY = (YU + DWT - 1) / DWT, NX = (XU + DWT - 1) / DWT;
    constexpr bool YFULL = YU % DWT == 0, XFULL = XU % DWT == 0;
    constexpr int BUF = YU + XU + 4;                           // + 16 floats of d_sigma
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

    // stored (hi, lo * 2^11)  ->  (hi * 2^k, lo * 2^(k-11)): the array's largest element lands in [2^14, 2^15) and lo is
    // unscaled.  2^k as a product of two f16-representable powers of two (packed-f16 multiplies, all exact).
    // a unit is {hi x4, lo x4} = words {hi, hi, lo, lo}: per operand four packed-f16 multipliers (2^k = m1 * m2 for the
    // hi words, 2^(k-11) for the lo words), kept as wave-uniform 32-bit patterns so they live in SGPRs
    typedef _Float16 half2 __attribute__((ext_vector_type(2)));
    struct Mul4 { unsigned int h1, h2, l1, l2; };
    auto mul4 = [](int k) -> Mul4 {
        const int kl = k - 11;
        const int h1 = k < -14 ? -14 : (k > 15 ? 15 : k), l1 = kl < -14 ? -14 : (kl > 15 ? 15 : kl);
        auto pat = [](int e) -> unsigned int {                       // {2^e, 2^e} as packed f16, e in [-14, 15]
            const unsigned int b = (unsigned int)((e + 15) << 10);
            return (unsigned int)__builtin_amdgcn_readfirstlane((int)(b | (b << 16)));
        };
        return Mul4{pat(h1), pat(k - h1), pat(l1), pat(kl - l1)};
    };
    const Mul4 ym = mul4(kY), xm = mul4(kX);
    auto mulw = [](unsigned int w, unsigned int m1, unsigned int m2) -> unsigned int {
        const half2 t = (__builtin_bit_cast(half2, w) * __builtin_bit_cast(half2, m1)) * __builtin_bit_cast(half2, m2);
        return __builtin_bit_cast(unsigned int, t);
    };
    auto rescale = [&](u32x4 unit, const Mul4& m) -> u32x4 {
        return u32x4{mulw(unit.x, m.h1, m.h2), mulw(unit.y, m.h1, m.h2), mulw(unit.z, m.l1, m.l2), mulw(unit.w, m.l1, m.l2)};
    };

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
    // global unit u = ((block * W + w) * 2 + half) holds {hi x4, lo x4} of 4 points: the two quads go to the hi / lo
    // fragment planes of the LDS image [block][plane][w][8 points] (8-byte writes, conflict free)
#define DW_STAGE(RY, RX, RDA, B, VALID)                                                                   \
    {                                                                                                     \
        u32x4* Ys_ = smem + (B) * BUF;                                                                    \
        u32x4* Xs_ = Ys_ + YU;                                                                            \
        _Pragma("unroll") for (int j = 0; j < NY; ++j) {                                                  \
            const int u = tid + j * DWT, mb = u / (2 * N), rem = u % (2 * N);                             \
            if (YFULL || u < YU) {                                                                        \
                u32x2* d = reinterpret_cast<u32x2*>(Ys_) + ((mb * 2) * N + (rem >> 1)) * 2 + (rem & 1);   \
                u32x4 q = rescale(RY[j], ym);                                                       \
                if (!(VALID)) q = u32x4{0u, 0u, 0u, 0u};       /* past the range: contributes nothing */ \
                d[0] = u32x2{q.x, q.y};                                                                   \
                d[2 * N] = u32x2{q.z, q.w};                                                               \
            }                                                                                             \
        }                                                                                                 \
        if (XROWS) {   /* split x * 2^kX directly (unscaled lo) and scatter the 4 features into their fragments */ \
            if (tid < CHP * K / 4) {                                                                      \
                const int p = tid / (K / 4), w0 = (tid % (K / 4)) * 4;                                    \
                _Float16* img = reinterpret_cast<_Float16*>(Xs_);                                         \
                const float xs = exp2i(kX);                                                               \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                           \
                    const float v = __uint_as_float(RX[0][i]) * xs;                                       \
                    const _Float16 hi = (_Float16)v;                                                      \
                    img[(((p >> 3) * 2 + 0) * K + w0 + i) * 8 + (p & 7)] = hi;                            \
                    img[(((p >> 3) * 2 + 1) * K + w0 + i) * 8 + (p & 7)] = (_Float16)(v - (float)hi);     \
                }                                                                                         \
            }                                                                                             \
        } else                                                                                            \
        _Pragma("unroll") for (int j = 0; j < NX; ++j) {                                                  \
            const int u = tid + j * DWT, mb = u / (2 * K), rem = u % (2 * K);                             \
            if (XFULL || u < XU) {                                                                        \
                u32x2* d = reinterpret_cast<u32x2*>(Xs_) + ((mb * 2) * K + (rem >> 1)) * 2 + (rem & 1);   \
                const u32x4 q = rescale(RX[j], xm);                                                 \
                d[0] = u32x2{q.x, q.y};                                                                   \
                d[2 * K] = u32x2{q.z, q.w};                                                               \
            }                                                                                             \
        }                                                                                                 \
        if (ALPHA && tid < CHP) reinterpret_cast<float*>(Xs_ + XU)[tid] = (VALID) ? RDA : 0.f;            \
        /* buffer B was last read three chunks ago, and every wave has passed two barriers in between */ \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");                                   \
        __builtin_amdgcn_s_barrier();                                                                     \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");                                   \
    }
    auto compute = [&](int b) {
        const u32x4* Yl = smem + b * BUF;
        const u32x4* Xl = Yl + YU;
        const float* da = reinterpret_cast<const float*>(Xl + XU);
        if (mma_wave) {
            half8 ah[TR], al[TR];
#pragma unroll
            for (int r = 0; r < TR; ++r) {
                ah[r] = __builtin_bit_cast(half8, Yl[(lh * 2 + 0) * N + (wn * TR + r) * 32 + lr]);
                al[r] = __builtin_bit_cast(half8, Yl[(lh * 2 + 1) * N + (wn * TR + r) * 32 + lr]);
            }
            // column tiles in groups of two (fragment registers: 16 + 16 instead of 16 + 32); inside a group all tiles
            // per product kind, so an accumulator is touched again only after TR*2 other MFMAs
            constexpr int CG = TC >= 2 ? 2 : 1;
#pragma unroll
            for (int c0 = 0; c0 < TC; c0 += CG) {
                half8 bh[CG], bl[CG];
#pragma unroll
                for (int c = 0; c < CG; ++c) {
                    bh[c] = __builtin_bit_cast(half8, Xl[(lh * 2 + 0) * K + (wk * TC + c0 + c) * 32 + lr]);
                    bl[c] = __builtin_bit_cast(half8, Xl[(lh * 2 + 1) * K + (wk * TC + c0 + c) * 32 + lr]);
                }
#pragma unroll
                for (int r = 0; r < TR; ++r)
#pragma unroll
                    for (int c = 0; c < CG; ++c) acc[r][c0 + c] = mfma16(ah[r], bh[c], acc[r][c0 + c]);
#pragma unroll
                for (int r = 0; r < TR; ++r)
#pragma unroll
                    for (int c = 0; c < CG; ++c) acc[r][c0 + c] = mfma16(ah[r], bl[c], acc[r][c0 + c]);
#pragma unroll
                for (int r = 0; r < TR; ++r)
#pragma unroll
                    for (int c = 0; c < CG; ++c) acc[r][c0 + c] = mfma16(al[r], bh[c], acc[r][c0 + c]);
            }
        }
        if (src.bias && tid < N) {
            float s = 0.f;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const half8 h = __builtin_bit_cast(half8, Yl[(mb * 2) * N + tid]), l = __builtin_bit_cast(half8, Yl[(mb * 2 + 1) * N + tid]);
#pragma unroll
                for (int j = 0; j < 8; ++j) s += (float)h[j] + (float)l[j];
            }
            bsum += s;
        }
        if (ALPHA && tid < K) {
            float s = 0.f, sb = 0.f;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const half8 h = __builtin_bit_cast(half8, Xl[(mb * 2) * K + tid]);
                const half8 l = __builtin_bit_cast(half8, Xl[(mb * 2 + 1) * K + tid]);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = da[mb * 8 + j];
                    s += d * ((float)h[j] + (float)l[j]);
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
    // the operand scales (the reduce kernel divides them out); the alpha row only carries the activation scale
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
    if (ALPHA && tid < K) {
        part[(int64_t)N * K + N + tid] = asum * exp2i(-kX);
        if (tid == 0) part[(int64_t)N * K + N + 256] = absum;
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

constexpr size_t DWH_SMEM = 3 * (size_t)(4 * 256 + 4 * 256 + 4) * 16;       // three chunk images of the 256 x 256 block: 98 496 B
constexpr size_t DWH_SMEM_SMALL = 3 * (size_t)(4 * 256 + 4 * 64 + 4) * 16;  // 256 x 64 block: 61 632 B

// rescale exponents of the operands from the absmax slots the forward / dX launches published
__device__ __forceinline__ void operand_scales(const DwArgs& a, int inst, int& kY, int& kX) {
    (void)inst;
    kY = __builtin_amdgcn_readfirstlane(rescale_exp(a.dacts[sdact_scale(m_pad(a.M)) + AY_ALL]));
    kX = __builtin_amdgcn_readfirstlane(rescale_exp(a.acts[sact_absmax(a.M) + AX_ALL]));
}

__device__ __forceinline__ void chunk_range(const DwArgs& a, int inst, int split, int64_t& cb, int64_t& ce) {
    const int64_t nchunks = m_pad(a.M) / CHP;
    const int64_t per = (nchunks + dwh_splits(inst) - 1) / dwh_splits(inst);
    cb = (int64_t)split * per;
    ce = cb + per;
    if (cb > nchunks) cb = nchunks;
    if (ce > nchunks) ce = nchunks;
}

// the eight 256x256 instances + the 128x256 views block: one workgroup per CU, every operand byte read once
__global__ __launch_bounds__(DWT, 2) void mlp_dw_split_big_kernel(DwArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32x4 smem_u[];
    const int id = blockIdx.x;
    int inst, split;
    if (id < DWH_FULL_BLOCKS) {
        inst = id / dwh_splits(DW_L1);    // DW_L1 .. DW_FEAT
        split = id % dwh_splits(DW_L1);
    } else {
        inst = DW_VIEWSF;
        split = id - DWH_FULL_BLOCKS;
    }
    int64_t cb, ce;
    chunk_range(a, inst, split, cb, ce);
    float* part = a.ws + dwh_inst_offset(inst) + (int64_t)split * dw_inst_floats(inst);
    const Src src = inst_src(a, inst);
    int kY, kX;
    operand_scales(a, inst, kY, kX);
    if (inst == DW_FEAT) dw_gemm<256, 256, 4, 2, 4, true>(a, src, kY, kX, cb, ce, part, smem_u);
    else if (inst <= DW_L7) dw_gemm<256, 256, 4, 2, 4, false>(a, src, kY, kX, cb, ce, part, smem_u);
    else dw_gemm<128, 256, 2, 2, 2, false>(a, src, kY, kX, cb, ce, part, smem_u);
}

// the thin instances: L0 and L5P (256 x 64, X = PE), VIEWSP (128 x 32, X = PE(dir)), rgb head (VALU)
__global__ __launch_bounds__(DWT, 2) void mlp_dw_split_small_kernel(DwArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32x4 smem_u[];
    const int inst = DW_L0 + blockIdx.x / 64, split = blockIdx.x % 64;
    int64_t cb, ce;
    chunk_range(a, inst, split, cb, ce);
    float* part = a.ws + dwh_inst_offset(inst) + (int64_t)split * dw_inst_floats(inst);
    if (inst == DW_RGB) {
        dw_rgb(a, cb * 2, ce * 2, part, reinterpret_cast<float*>(smem_u));
        return;
    }
    const Src src = inst_src(a, inst);
    int kY, kX;
    operand_scales(a, inst, kY, kX);
    if (inst == DW_VIEWSP) dw_gemm<128, 32, 4, 1, 1, false, true>(a, src, kY, kX, cb, ce, part, smem_u);
    else dw_gemm<256, 64, 4, 2, 1, false, true>(a, src, kY, kX, cb, ce, part, smem_u);   // DW_L0, DW_L5P
}

}  // namespace

int benerf_mlp_dw_reduce_launch(const float* ws, const BenerfMlpGrads* grads, int channels, int accumulate, int split_mode,
                                const float* absmax_y, const float* absmax_x, hipStream_t stream);

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
    return benerf_mlp_dw_reduce_launch(dw_ws, grads, channels, accumulate, 1, dacts + mlp::sdact_scale(mlp::m_pad(M)),
                                       acts + mlp::sact_absmax(M), stream);
}

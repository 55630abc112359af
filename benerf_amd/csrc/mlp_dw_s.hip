// K3 backward part 2, BENERF_MLP_SPLIT (the 22-bit backward): dW_l = dY_l^T X_l over all sample points with BOTH operands as
// hi + lo f16 pairs (the SH arrays and their lo twins the split forward / mlp_bwd_s.hip save, mlp_split.h):
//     dY^T X  =  dY_hi^T X_hi + dY_hi^T X_lo + dY_lo^T X_hi      (three v_mfma_f32_32x32x16_f16 per block, ONE f32 accumulator)
// + bias sums and the alpha / rgb heads on the VALU from hi + lo.  Why both lo halves: tools/experiments/
// backward_format_study.py - with an f16 X or an f16 dY the weight gradients sit 2-3e-4 of the largest entry from float64
// (same ReLU masks) where the exact-f32 kernel sits at 1e-6; with hi + lo on both they carry float32's own error.
//
// Structure = mlp_dw_h.hip's: an SH array is blocks of 8 points, feature-major, one 16-byte unit per (block, feature) = the
// MFMA fragment of a contraction over points, copied verbatim into a triple-buffered LDS image with three chunks in flight in
// registers and one LDS-only barrier per chunk; a workgroup (8 waves, one per CU) holds a whole 256 x 256 output block (128
// accumulator registers per wave), so every operand byte is read exactly once.  A chunk is 16 points (ONE MFMA k-step) of the
// four arrays = 32 KiB, the same bytes per barrier and per register set as the f16 kernel's 32-point chunk of two arrays.
// HBM-bound: 19.5 KB per point against 24 MFMAs per wave and chunk.
#include "mlp_split.h"

namespace {
using namespace mlp;

constexpr int DWT = 512;
constexpr int CHP = 16;             // points per chunk = 2 blocks of 8 = one MFMA k-step
constexpr int CHB = CHP / 8;
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
    const u32x4* y;    // SH array of width N (16-byte units), hi halves; lo twin at + ylo units
    const u32x4* x;    // SH array of width K ... or, for the thin instances (XROWS), f32 rows [Mp][K]
    int64_t ylo, xlo;  // unit offsets of the lo twins
    bool bias;
};

__device__ __forceinline__ Src inst_src(const DwArgs& a, int inst) {
    const int64_t Mp = m_pad(a.M);
    auto U = [](const float* p) { return reinterpret_cast<const u32x4*>(p); };
    const float* A = a.acts;
    const float* D = a.dacts;
    const int64_t yl = sdact_lo_delta(Mp) / 4, xl = sact_lo_delta(Mp) / 4;        // floats -> 16-byte units
    switch (inst) {
        case DW_L1: case DW_L2: case DW_L3: case DW_L4: case DW_L5H: case DW_L6: case DW_L7:
            return {U(D + sdact_h(Mp, 1 + (inst - DW_L1))), U(A + sact_h(Mp, inst - DW_L1)), yl, xl, true};
        case DW_FEAT: return {U(D + sdact_feat(Mp)), U(A + sact_h(Mp, 7)), yl, xl, true};
        case DW_VIEWSF: return {U(D + sdact_hv(Mp)), U(A + sact_feat(Mp)), yl, xl, true};
        case DW_L0: return {U(D + sdact_h(Mp, 0)), U(A + sact_pe32(Mp)), yl, 0, true};       // X = PE as f32 rows
        case DW_L5P: return {U(D + sdact_h(Mp, 5)), U(A + sact_pe32(Mp)), yl, 0, false};
        default: return {U(D + sdact_hv(Mp)), U(A + sact_ped32(Mp)), yl, 0, false};          // DW_VIEWSP: PE(dir) rows
    }
}

// Output block N x K (the whole instance: K = width of X).  Waves form a WN x (8/WN) grid; each owns TR x TC MFMA tiles: per
// 16-point chunk acc += Yh^T Xh + Yh^T Xl + Yl^T Xh.  Three chunks are in flight in registers (sets A, B, C) and the LDS image
// is triple-buffered, so there is one LDS-only barrier per chunk.
template <int N, int K, int WN, int TR, int TC, bool ALPHA, bool XROWS = false>
__device__ __forceinline__ void dw_gemm(const DwArgs& a, const Src src, int64_t chunk_begin, int64_t chunk_end,
                                        float* __restrict__ part, u32x4* __restrict__ smem) {
    static_assert(WN * TR * 32 == N, "row tiling");
    constexpr int YU = CHB * N, XU = CHB * K;                  // 16-byte units per chunk and plane
    static_assert(!XROWS || (CHP * K / 4 <= DWT), "one float4 of the f32 rows per thread");
    constexpr int NY = (YU + DWT - 1) / DWT, NX = XROWS ? 1 : (XU + DWT - 1) / DWT;
    constexpr bool YFULL = YU % DWT == 0, XFULL = XU % DWT == 0;
    constexpr int BUF = 2 * YU + 2 * XU + CHP / 4;             // Yh | Yl | Xh | Xl | CHP floats of d_sigma
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

    u32x4 ryA[2][NY], rxA[2][NX], ryB[2][NY], rxB[2][NX], ryC[2][NY], rxC[2][NX];
    float rdaA = 0.f, rdaB = 0.f, rdaC = 0.f;
    // Loads are UNCONDITIONAL (a chunk index past the range is clamped to the last chunk and its staged dY zeroed): with the
    // loads under branches the compiler's waitcnt bookkeeping falls back to vmcnt(0) at every stage (mlp_dw_h.hip)
#define DW_PREFETCH(RY, RX, RDA, CHUNK)                                                                   \
    {                                                                                                     \
        const int64_t cc = (CHUNK) < chunk_end ? (CHUNK) : chunk_end - 1;                                 \
        const u32x4* py = src.y + cc * YU + tid;                                                          \
        const u32x4* px = src.x + cc * XU + tid;                                                          \
        _Pragma("unroll") for (int j = 0; j < NY; ++j)                                                    \
            if (YFULL || tid + j * DWT < YU) {                                                            \
                RY[0][j] = py[j * DWT];                                                                   \
                RY[1][j] = py[src.ylo + j * DWT];                                                         \
            }                                                                                             \
        if (XROWS) {   /* f32 rows [point][K]: one float4 (4 features of a point) per thread */                \
            if (tid < CHP * K / 4)                                                                        \
                RX[0][0] = reinterpret_cast<const u32x4*>(reinterpret_cast<const float*>(src.x) + cc * (CHP * K))[tid]; \
        } else {                                                                                          \
            _Pragma("unroll") for (int j = 0; j < NX; ++j)                                                \
                if (XFULL || tid + j * DWT < XU) {                                                        \
                    RX[0][j] = px[j * DWT];                                                               \
                    RX[1][j] = px[src.xlo + j * DWT];                                                     \
                }                                                                                         \
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
        u32x4* Xs_ = Ys_ + 2 * YU;                                                                        \
        _Pragma("unroll") for (int j = 0; j < NY; ++j) {                                                  \
            const int u = tid + j * DWT;                                                                  \
            if (YFULL || u < YU) {   /* past the range: contributes nothing */                            \
                Ys_[u] = (VALID) ? RY[0][j] : u32x4{0u, 0u, 0u, 0u};                                      \
                Ys_[YU + u] = (VALID) ? RY[1][j] : u32x4{0u, 0u, 0u, 0u};                                 \
            }                                                                                             \
        }                                                                                                 \
        if (XROWS) {   /* split into hi + lo and scatter the 4 features of this thread's point into their fragments */ \
            if (tid < CHP * K / 4) {                                                                      \
                const int p = tid / (K / 4), w0 = (tid % (K / 4)) * 4;                                    \
                _Float16* img = reinterpret_cast<_Float16*>(Xs_);                                         \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                           \
                    const float v = __uint_as_float(RX[0][0][i]);                                         \
                    const _Float16 hi = (_Float16)v;                                                      \
                    img[((p >> 3) * K + w0 + i) * 8 + (p & 7)] = hi;                                      \
                    img[XU * 8 + ((p >> 3) * K + w0 + i) * 8 + (p & 7)] = (_Float16)(v - (float)hi);      \
                }                                                                                         \
            }                                                                                             \
        } else                                                                                            \
        _Pragma("unroll") for (int j = 0; j < NX; ++j) {                                                  \
            const int u = tid + j * DWT;                                                                  \
            if (XFULL || u < XU) {                                                                        \
                Xs_[u] = RX[0][j];                                                                        \
                Xs_[XU + u] = RX[1][j];                                                                   \
            }                                                                                             \
        }                                                                                                 \
        if (ALPHA && tid < CHP) reinterpret_cast<float*>(Xs_ + 2 * XU)[tid] = (VALID) ? RDA : 0.f;        \
        /* buffer B was last read three chunks ago, and every wave has passed two barriers in between */ \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");                                   \
        __builtin_amdgcn_s_barrier();                                                                     \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");                                   \
    }
    auto compute = [&](int b) {
        const u32x4* Yl = smem + b * BUF;
        const u32x4* Xl = Yl + 2 * YU;
        const float* da = reinterpret_cast<const float*>(Xl + 2 * XU);
        if (mma_wave) {
            half8 ayh[TR], ayl[TR];
#pragma unroll
            for (int r = 0; r < TR; ++r) {
                ayh[r] = __builtin_bit_cast(half8, Yl[lh * N + (wn * TR + r) * 32 + lr]);
                ayl[r] = __builtin_bit_cast(half8, Yl[YU + lh * N + (wn * TR + r) * 32 + lr]);
            }
            // column tiles in groups of CG: the three MFMAs of one accumulator are TR * CG issue slots apart, and only CG
            // fragment pairs of X are live at a time (all TC at once spilled)
            constexpr int CG = TC >= 2 ? 2 : 1;
#pragma unroll
            for (int c0 = 0; c0 < TC; c0 += CG) {
                half8 bxh[CG], bxl[CG];
#pragma unroll
                for (int c = 0; c < CG; ++c) {
                    bxh[c] = __builtin_bit_cast(half8, Xl[lh * K + (wk * TC + c0 + c) * 32 + lr]);
                    bxl[c] = __builtin_bit_cast(half8, Xl[XU + lh * K + (wk * TC + c0 + c) * 32 + lr]);
                }
#pragma unroll
                for (int r = 0; r < TR; ++r)
#pragma unroll
                    for (int c = 0; c < CG; ++c) acc[r][c0 + c] = mfma16(ayh[r], bxh[c], acc[r][c0 + c]);
#pragma unroll
                for (int r = 0; r < TR; ++r)
#pragma unroll
                    for (int c = 0; c < CG; ++c) acc[r][c0 + c] = mfma16(ayh[r], bxl[c], acc[r][c0 + c]);
#pragma unroll
                for (int r = 0; r < TR; ++r)
#pragma unroll
                    for (int c = 0; c < CG; ++c) acc[r][c0 + c] = mfma16(ayl[r], bxh[c], acc[r][c0 + c]);
            }
        }
        if (src.bias && tid < N) {
            float s = 0.f;
#pragma unroll
            for (int mb = 0; mb < CHB; ++mb) {
                const half8 h = __builtin_bit_cast(half8, Yl[mb * N + tid]);
                const half8 l = __builtin_bit_cast(half8, Yl[YU + mb * N + tid]);
#pragma unroll
                for (int j = 0; j < 8; ++j) s += (float)h[j] + (float)l[j];
            }
            bsum += s;
        }
        // the alpha head rides on waves 4..7 (column tid - 256): waves 0..3 already carry the bias sums (mlp_dw_h.hip)
        if (ALPHA && tid >= DWT - K) {
            const int ka = tid - (DWT - K);
            float s = 0.f, sb = 0.f;
#pragma unroll
            for (int mb = 0; mb < CHB; ++mb) {
                const half8 h = __builtin_bit_cast(half8, Xl[mb * K + ka]);
                const half8 l = __builtin_bit_cast(half8, Xl[XU + mb * K + ka]);
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

    // partial block -> workspace: [N][K] then bias [N] (then alpha row [256] + alpha bias).  Block and bias stay at the
    // gradient scale s_s (the reduce kernel divides it out); the alpha row is unscaled (d_raw is)
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

// ---- LDS-DMA variant of dw_gemm for the big instances -----------------------------------------------------------------
// The chunk image IS the global layout (16-byte units, contiguous per array and chunk), so the copy needs no registers at all:
// `buffer_load_dwordx4 ... lds` moves 1 KiB per wave-instruction straight into the ring slot.  Ring of DMA_RING slots,
// DMA_RING - 1 chunks in flight (96 KiB per CU against ~50 KiB of bandwidth-delay product at 6 TB/s), one barrier per chunk:
//   acquire(j): wait for this wave's own pieces of chunk j (vector-memory operations retire in order: only the pieces of the
//   chunks requested behind j may still be outstanding), barrier -> every piece of j has landed AND every wave is done with
//   chunk j - 1, whose slot the next request (chunk j + DMA_RING - 1) overwrites.
// Every wave issues the same number PW of pieces per chunk, so the wait counts are compile-time constants.
#ifndef DMA_RING
#define DMA_RING 4
#endif
template <int N, bool ALPHA>
struct DmaGeom {
    static constexpr int K = 256;
    static constexpr int YU = CHB * N, XU = CHB * K;                 // 16-byte units per chunk and plane
    static constexpr int YP = YU / 64, XP = XU / 64;                 // 1-KiB pieces per plane
    static_assert(XP == 8 && (YP == 8 || YP == 4), "eight waves share the pieces evenly");
    static constexpr int PW = (YP == 8 ? 2 : 1) + 2 + (ALPHA ? 1 : 0);
    static constexpr int SIG = (2 * YU + 2 * XU) * 16;               // byte offset of the d_raw rows (one 256-byte copy per wave)
    static constexpr int SLOT = SIG + (ALPHA ? 8 * 256 : 0);
};
template <int CNT>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory"); }

template <int N, int WN, int TR, int TC, bool ALPHA>
__device__ __forceinline__ void dw_gemm_dma(const DwArgs& a, const Src src, int64_t chunk_begin, int64_t chunk_end,
                                            float* __restrict__ part, char* __restrict__ smem) {
    typedef DmaGeom<N, ALPHA> G;
    constexpr int K = G::K, YU = G::YU, XU = G::XU, PW = G::PW, AHEAD = DMA_RING - 1;
    static_assert(WN * TR * 32 == N && (8 / WN) * TC * 32 == K, "tiling");
    static_assert(AHEAD * PW < 64, "vmcnt field");
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int wn = wave % WN, wk = wave / WN;
    auto rsrc_of = [](const void* p, uint32_t bytes) {
        const uint64_t wa = reinterpret_cast<uint64_t>(p);
        const uint64_t wau = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(wa >> 32)) << 32) |
                             (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)wa);
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(wau), 0, bytes, 0x00020000);
    };
    // descriptors on this workgroup's first chunk: per-chunk offsets stay below 2^31 (a split covers a few MB per array)
    const __amdgpu_buffer_rsrc_t rs_yh = rsrc_of(src.y + chunk_begin * YU, 0x7fffffff);
    const __amdgpu_buffer_rsrc_t rs_yl = rsrc_of(src.y + src.ylo + chunk_begin * YU, 0x7fffffff);
    const __amdgpu_buffer_rsrc_t rs_xh = rsrc_of(src.x + chunk_begin * XU, 0x7fffffff);
    const __amdgpu_buffer_rsrc_t rs_xl = rsrc_of(src.x + src.xlo + chunk_begin * XU, 0x7fffffff);
    // d_raw rows of a chunk (ALPHA: the alpha head's d_sigma): [M][C + 1] floats; rows past M read as zero (range check on the
    // VECTOR offset, which therefore carries the chunk's position)
    const int row_dw = a.C + 1;
    const __amdgpu_buffer_rsrc_t rs_sig = rsrc_of(a.d_raw, (uint32_t)(a.M * row_dw * 4));
    const int lane16 = lane * 16;

    auto request = [&](int64_t j) {
        char* slot = smem + (int)(j % DMA_RING) * G::SLOT;
        const int rel = (int)(j - chunk_begin);
        if (G::YP == 8) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_yh, (lds_ptr)(slot + wave * 1024), 16, lane16, rel * (YU * 16) + wave * 1024, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_yl, (lds_ptr)(slot + YU * 16 + wave * 1024), 16, lane16, rel * (YU * 16) + wave * 1024, 0, 0);
        } else {    // 4 pieces per plane: waves 0..3 take the hi plane, 4..7 the lo plane
            if (wave < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_yh, (lds_ptr)(slot + wave * 1024), 16, lane16, rel * (YU * 16) + wave * 1024, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_yl, (lds_ptr)(slot + YU * 16 + (wave - 4) * 1024), 16, lane16, rel * (YU * 16) + (wave - 4) * 1024, 0, 0);
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_xh, (lds_ptr)(slot + 2 * YU * 16 + wave * 1024), 16, lane16, rel * (XU * 16) + wave * 1024, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_xl, (lds_ptr)(slot + (2 * YU + XU) * 16 + wave * 1024), 16, lane16, rel * (XU * 16) + wave * 1024, 0, 0);
        if (ALPHA) {    // 16 rows x (C + 1) floats <= 64 dwords: one dword per lane, one copy per wave (uniform piece count)
            const int voff = ((int)j * CHP * row_dw + (lane < CHP * row_dw ? lane : 0)) * 4;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_sig, (lds_ptr)(slot + G::SIG + wave * 256), 4, voff, 0, 0, 0);
        }
    };
    auto acquire = [&](int newer) {
        if (newer >= 2) wait_vm<2 * PW>();
        else if (newer == 1) wait_vm<PW>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
    };
    static_assert(AHEAD == 3, "acquire() enumerates newer = 0, 1, 2");

    f32x16 acc[TR][TC];
#pragma unroll
    for (int r = 0; r < TR; ++r)
#pragma unroll
        for (int c = 0; c < TC; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][c][e] = 0.f;
    float bsum = 0.f, asum = 0.f, absum = 0.f;

    for (int i = 0; i < AHEAD; ++i)
        if (chunk_begin + i < chunk_end) request(chunk_begin + i);
    for (int64_t j = chunk_begin; j < chunk_end; ++j) {
        const int64_t behind = chunk_end - 1 - j;
        acquire(behind >= AHEAD - 1 ? AHEAD - 1 : (int)behind);
        if (j + AHEAD < chunk_end) request(j + AHEAD);
        __builtin_amdgcn_sched_barrier(0);      // requests first, then the chunk's MFMAs
        const char* slot = smem + (int)(j % DMA_RING) * G::SLOT;
        const u32x4* Yl = reinterpret_cast<const u32x4*>(slot);
        const u32x4* Xl = Yl + 2 * YU;
        {
            half8 ayh[TR], ayl[TR];
#pragma unroll
            for (int r = 0; r < TR; ++r) {
                ayh[r] = __builtin_bit_cast(half8, Yl[lh * N + (wn * TR + r) * 32 + lr]);
                ayl[r] = __builtin_bit_cast(half8, Yl[YU + lh * N + (wn * TR + r) * 32 + lr]);
            }
            constexpr int CG = 2;
#pragma unroll
            for (int c0 = 0; c0 < TC; c0 += CG) {
                half8 bxh[CG], bxl[CG];
#pragma unroll
                for (int c = 0; c < CG; ++c) {
                    bxh[c] = __builtin_bit_cast(half8, Xl[lh * K + (wk * TC + c0 + c) * 32 + lr]);
                    bxl[c] = __builtin_bit_cast(half8, Xl[XU + lh * K + (wk * TC + c0 + c) * 32 + lr]);
                }
#pragma unroll
                for (int r = 0; r < TR; ++r)
#pragma unroll
                    for (int c = 0; c < CG; ++c) acc[r][c0 + c] = mfma16(ayh[r], bxh[c], acc[r][c0 + c]);
#pragma unroll
                for (int r = 0; r < TR; ++r)
#pragma unroll
                    for (int c = 0; c < CG; ++c) acc[r][c0 + c] = mfma16(ayh[r], bxl[c], acc[r][c0 + c]);
#pragma unroll
                for (int r = 0; r < TR; ++r)
#pragma unroll
                    for (int c = 0; c < CG; ++c) acc[r][c0 + c] = mfma16(ayl[r], bxh[c], acc[r][c0 + c]);
            }
        }
        if (src.bias && tid < N) {
            float s = 0.f;
#pragma unroll
            for (int mb = 0; mb < CHB; ++mb) {
                const half8 h = __builtin_bit_cast(half8, Yl[mb * N + tid]);
                const half8 l = __builtin_bit_cast(half8, Yl[YU + mb * N + tid]);
#pragma unroll
                for (int q = 0; q < 8; ++q) s += (float)h[q] + (float)l[q];
            }
            bsum += s;
        }
        if (ALPHA && tid >= DWT - K) {      // waves 4..7: column tid - 256 (mlp_dw_h.hip); each reads its own wave's d_raw copy
            const int ka = tid - (DWT - K);
            const float* da = reinterpret_cast<const float*>(slot + G::SIG + wave * 256);
            float s = 0.f, sb = 0.f;
#pragma unroll
            for (int mb = 0; mb < CHB; ++mb) {
                const half8 h = __builtin_bit_cast(half8, Xl[mb * K + ka]);
                const half8 l = __builtin_bit_cast(half8, Xl[XU + mb * K + ka]);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float d = da[(mb * 8 + q) * row_dw + a.C];
                    s += d * ((float)h[q] + (float)l[q]);
                    sb += d;
                }
            }
            asum += s;
            absum += sb;
        }
    }

    // partial block -> workspace (dw_gemm's layout)
#pragma unroll
    for (int r = 0; r < TR; ++r)
#pragma unroll
        for (int c = 0; c < TC; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (wn * TR + r) * 32 + acc_row(e, lane);
                part[(int64_t)row * K + (wk * TC + c) * 32 + lr] = acc[r][c][e];
            }
    if (tid < N) part[(int64_t)N * K + tid] = src.bias ? bsum : 0.f;
    if (ALPHA && tid >= DWT - K) {
        part[(int64_t)N * K + N + tid - (DWT - K)] = asum;
        if (tid == DWT - K) part[(int64_t)N * K + N + 256] = absum;
    }
}

// rgb head: dW_rgb[c][j] = sum_pt d_rgb[pt][c] * hv[pt][j], db_rgb[c] = sum_pt d_rgb[pt][c]   (unscaled d_raw, f32; hv = hi + lo).
// Batches of 256 points: d_raw staged in LDS, then every thread (column j, phase ph) streams 8 blocks of hv (hi and lo) with
// all its loads independent.
__device__ __forceinline__ void dw_rgb(const DwArgs& a, int64_t blk_begin, int64_t blk_end, float* __restrict__ part,
                                       float* __restrict__ smem) {
    constexpr int BB = 32;                                           // blocks of 8 points per batch
    const int tid = threadIdx.x, j = tid & 127, ph = tid >> 7;     // ph: block phase 0..3
    const int64_t Mp = m_pad(a.M);
    const u32x4* hv = reinterpret_cast<const u32x4*>(a.acts + sact_hv(Mp));
    const int64_t lo = sact_lo_delta(Mp) / 4;
    const int C = a.C;
    const int64_t M = a.M;
    float* dr = smem;                                                // [BB*8][4]
    float s[3] = {0.f, 0.f, 0.f}, sb[3] = {0.f, 0.f, 0.f};
    for (int64_t b0 = blk_begin; b0 < blk_end; b0 += BB) {
        __syncthreads();
        if (tid < BB * 8) {   // 256 points x 4 slots, one point per thread
            const int64_t m = b0 * 8 + tid;
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < M && m < blk_end * 8) {
                const float* srcp = a.d_raw + m * (C + 1);
                g.x = srcp[0];
                if (C > 1) g.y = srcp[1];
                if (C > 2) g.z = srcp[2];
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
                hh[i] = hv[mb * ACT_HV_W + j];     // 8 points of column j
                hl[i] = hv[lo + mb * ACT_HV_W + j];
            }
        }
#pragma unroll
        for (int i = 0; i < BB / 4; ++i) {
            const half8 pq = __builtin_bit_cast(half8, hh[i]), pl = __builtin_bit_cast(half8, hl[i]);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float x = (float)pq[q] + (float)pl[q];
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

#ifdef DWS_NO_DMA
constexpr size_t DWS_SMEM = 3 * (size_t)(2 * CHB * 256 + 2 * CHB * 256 + CHP / 4) * 16;       // three chunk images of the 256 x 256 block: 98 496 B
#else
constexpr size_t DWS_SMEM = (size_t)DMA_RING * DmaGeom<256, true>::SLOT;                        // four ring slots of 32 KiB + d_raw copies: 139 264 B
#endif
constexpr size_t DWS_SMEM_SMALL = 3 * (size_t)(2 * CHB * 256 + 2 * CHB * 64 + CHP / 4) * 16;  // 256 x 64 block: 61 632 B
static_assert(DWS_SMEM_SMALL >= (32 * 8 * 4 + 18 * 128) * sizeof(float), "rgb head scratch fits the thin image");

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
__global__ __launch_bounds__(DWT, 2) void mlp_dw_split_big_kernel(DwArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32x4 smem_u[];
    DW_TRACE(1, 0);
    const int inst = dwh_big_inst(blockIdx.x), split = dwh_big_split(blockIdx.x);
    int64_t cb, ce;
    chunk_range(a, inst, split, cb, ce);
    float* part = a.ws + dwh_inst_offset(inst) + (int64_t)split * dw_inst_floats(inst);
    const Src src = inst_src(a, inst);
#ifdef DWS_NO_DMA      // register-staged copies (the f16 kernel's scheme): kept for A/B runs
    if (inst == DW_FEAT) dw_gemm<256, 256, 4, 2, 4, true>(a, src, cb, ce, part, smem_u);
    else if (inst <= DW_L7) dw_gemm<256, 256, 4, 2, 4, false>(a, src, cb, ce, part, smem_u);
    else dw_gemm<128, 256, 2, 2, 2, false>(a, src, cb, ce, part, smem_u);
#else
    char* smem_c = reinterpret_cast<char*>(smem_u);
    if (inst == DW_FEAT) dw_gemm_dma<256, 4, 2, 4, true>(a, src, cb, ce, part, smem_c);
    else if (inst <= DW_L7) dw_gemm_dma<256, 4, 2, 4, false>(a, src, cb, ce, part, smem_c);
    else dw_gemm_dma<128, 2, 2, 2, false>(a, src, cb, ce, part, smem_c);
#endif
    DW_TRACE(1, 1);
}

// the thin instances: L0 and L5P (256 x 64, X = PE), VIEWSP (128 x 32, X = PE(dir)), rgb head (VALU)
__global__ __launch_bounds__(DWT, 4) void mlp_dw_split_small_kernel(DwArgs a) {
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
                                const float* grad_info, const float* pe_weights, hipStream_t stream);

int benerf_mlp_dw_split22_launch(int channels, int64_t M, const float* d_raw, const float* acts, const float* dacts, float* dw_ws,
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
    if (!benerf_lds_attr(attr_big, (const void*)mlp_dw_split_big_kernel, (int)DWS_SMEM) ||
        !benerf_lds_attr(attr_small, (const void*)mlp_dw_split_small_kernel, (int)DWS_SMEM_SMALL)) {
        benerf_set_error("mlp_bwd(dw, split): cannot reserve LDS");
        return BENERF_EHIP;
    }
    hipLaunchKernelGGL(mlp_dw_split_small_kernel, dim3(mlp::DWH_SMALL_BLOCKS), dim3(DWT), DWS_SMEM_SMALL, stream, a);
    BENERF_LAUNCH_CHECK("mlp_bwd(dw small, split)");
    hipLaunchKernelGGL(mlp_dw_split_big_kernel, dim3(mlp::DWH_BIG_BLOCKS), dim3(DWT), DWS_SMEM, stream, a);
    BENERF_LAUNCH_CHECK("mlp_bwd(dw, split)");
    return benerf_mlp_dw_reduce_launch(dw_ws, grads, channels, accumulate, 1, dacts + mlp::sdact_info(mlp::m_pad(M)), pe_weights, stream);
}

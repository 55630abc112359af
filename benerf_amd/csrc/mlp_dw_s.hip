// K3 backward part 2, BENERF_MLP_SPLIT (the fp32-equivalent backward): dW_l = dY_l^T X_l over all sample points with BOTH operands as
// f16 pairs hi + lo:
//     dY^T X  =  dY_hi^T X_hi + dY_hi^T X_lo + dY_lo^T X_hi      (three v_mfma_f32_32x32x16_f16 per block, ONE f32 accumulator)
// + bias sums and the alpha / rgb heads on the VALU from hi + lo.  Why both low halves: tools/experiments/
// backward_format_study.py - with an f16 X or an f16 dY the weight gradients sit 2-3e-4 of the largest entry from float64
// (same ReLU masks) where the exact-f32 kernel sits at 1e-6.
// In HBM the low halves are 8-BIT RESIDUAL CODES (lo8 twins of the SH arrays the split forward / mlp_bwd_s.hip save,
// mlp_split.h: 19-bit operands, 3 bytes per value): this kernel is bound by exactly those bytes.  A code unit (8 bytes = 8 points
// of one feature) is decoded to its f16 low halves on the way from the load registers to LDS (h8_decode_unit: 6 packed VALU
// operations per pair, next to an idle VALU), so the MFMA side sees plain f16 pairs.
//
// Structure = mlp_dw_h.hip's: an SH array is blocks of 8 points, feature-major, one 16-byte unit per (block, feature) = the
// MFMA fragment of a contraction over points, staged through registers into a triple-buffered LDS image with three chunks in
// flight and one LDS-only barrier per chunk; a workgroup (8 waves, one per CU) holds a whole 256 x 256 output block (128
// accumulator registers per wave), so every operand byte is read exactly once.  A chunk is 16 points (ONE MFMA k-step) of the
// four arrays = 24 KiB from HBM, 32 KiB in LDS.  HBM-bound: 14.6 KB per point against 24 MFMAs per wave and chunk.
// (An LDS-DMA variant of the copy - ring of four slots, no staging registers - measured 5 % faster on f16 pairs in HBM, commit
// caf2f65; it cannot decode on the way and went with the 4-byte format.)
// Round 5: the X operand of the big instances (h0..h7, feature) arrives in the forward's SP layout (mlp_split.h: 16-byte units =
// 8 FEATURES of one point - what the forward's epilogue registers hold; the forward no longer transposes; a 16-point chunk is one
// contiguous run [slot][point]).  The chunk image keeps those units ([slot][point], XOR-swizzled) and the MFMA B fragments - 8 POINTS of one feature - come out of it with two
// ds_read_b64_tr_b16 each: the transposition rides on LDS reads this kernel does anyway.  dY stays SH (the dX kernel's
// accumulators hold 4 points of one feature: its natural layout).
// The thin instances (X = positional encoding: L0, the PE part of L5, the PE(dir) part of the views layer) stream WITHOUT LDS:
// see thin_stream below.
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
    const u32x4* y;     // SH array of width N (16-byte units), hi halves
    const uint2* y8;    // its lo8 twin (8-byte units)
    const u32x4* x;     // SH array of width K
    const uint2* x8;
    bool bias;
};

__device__ __forceinline__ Src inst_src(const DwArgs& a, int inst) {
    const int64_t Mp = m_pad(a.M);
    const float* A = a.acts;
    const float* D = a.dacts;
    auto Y = [&](int64_t off, int64_t xoff) {
        return Src{reinterpret_cast<const u32x4*>(D + off), reinterpret_cast<const uint2*>(sdact_lo8(D, Mp, off)),
                   reinterpret_cast<const u32x4*>(A + xoff), reinterpret_cast<const uint2*>(sact_lo8(A, Mp, xoff)), true};
    };
    switch (inst) {     // the big kernel's instances
        // G = dhv^T h7 (mlp_common.h, round 5): the views block reads h7, not `feature` - neither feature nor its gradient is saved;
        // the feature layer's and the views layer's weight gradients are composed from G in the reduce stage
        case DW_VIEWSF: return Y(sdact_hv(Mp), sact_h(Mp, 7));
        default: return Y(sdact_h(Mp, 1 + (inst - DW_L1)), sact_h(Mp, inst - DW_L1));     // DW_L1 .. DW_L7 (DW_L5H: the h4 part of layer 5)
    }
}

// buffer descriptor on a wave-uniform base (raw, 2^31 - 1 bytes)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dw_rsrc(const void* p) {
    const uint64_t wa = reinterpret_cast<uint64_t>(p);
    const uint64_t wau = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(wa >> 32)) << 32) |
                         (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)wa);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(wau), 0, 0x7fffffff, 0x00020000);
}
// cache policy of the operand loads: every byte of the saved activations / gradients is read by exactly ONE workgroup, once (aux 2 = nt on
// gfx950; measured: -0.3 % of the step for these loads alone, profiles/r05_cache_policy_ab.log)
#ifndef DWS_NT
#define DWS_NT 2
#endif
__device__ __forceinline__ uint2 dw_load_b64(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, DWS_NT);
    return uint2{v[0], v[1]};
}

// sum of the 8 + 8 halfs of a unit in f32 (v_dot2_f32_f16 against (1, 1): 8 instructions where convert-and-add takes 32)
__device__ __forceinline__ float dw_sum_unit(const half8 h, const half8 l, float s) {
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    const half2v one = {(_Float16)1.f, (_Float16)1.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s = __builtin_amdgcn_fdot2(half2v{h[2 * j], h[2 * j + 1]}, one, s, false);
        s = __builtin_amdgcn_fdot2(half2v{l[2 * j], l[2 * j + 1]}, one, s, false);
    }
    return s;
}

// Output block N x K (the whole instance: K = width of X).  Waves form a WN x (8/WN) grid; each owns TR x TC MFMA tiles: per
// 16-point chunk acc += Yh^T Xh + Yh^T Xl + Yl^T Xh.  Three chunks are in flight in registers (sets A, B, C) and the LDS image
// is triple-buffered, so there is one LDS-only barrier per chunk.
// ALPHA: the alpha head (dW_alpha = d_sigma^T X, X = h7) rides on waves 4..7; its partial row [256] + bias go to `apart`
template <int N, int K, int WN, int TR, int TC, bool ALPHA>
__device__ __forceinline__ void dw_gemm(const DwArgs& a, const Src src, int64_t chunk_begin, int64_t chunk_end,
                                        float* __restrict__ part, u32x4* __restrict__ smem, float* __restrict__ apart = nullptr) {
    static_assert(WN * TR * 32 == N, "row tiling");
    static_assert(K == 256 && CHP == 16, "X in SP layout: 32 slots x 16 points = one unit per thread and chunk");
    constexpr int YU = CHB * N, XU = CHB * K;                  // 16-byte units per chunk and plane
    constexpr int NY = (YU + DWT - 1) / DWT, NX = (XU + DWT - 1) / DWT;
    constexpr bool YFULL = YU % DWT == 0;
    static_assert(XU == DWT && NX == 1, "X: one SP unit per thread and chunk");
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

    // per chunk and thread: hi unit (16 bytes) + code unit (8 bytes) of Y and of X
    struct Regs { u32x4 y[NY]; uint2 y8[NY]; u32x4 x[NX]; uint2 x8[NX]; float da; };
    Regs rA, rB, rC;
    rA.da = rB.da = rC.da = 0.f;
    // Loads are UNCONDITIONAL (a chunk index past the range is clamped to the last chunk and its staged dY zeroed): with the
    // loads under branches the compiler's waitcnt bookkeeping falls back to vmcnt(0) at every stage (mlp_dw_h.hip)
#ifdef DWS_NO_CODES      /* timing variant: what do the 8-byte code loads cost? */
#define DWS_CODE_LOAD(expr) (uint2{0x80808080u, 0x80808080u})
#else
#define DWS_CODE_LOAD(expr) (expr)
#endif
    // Loads through buffer descriptors: the chunk part of every address is wave-uniform (one SGPR offset), the thread part a
    // constant (two VGPRs for the whole kernel) - plain pointers cost two 64-bit VGPR addresses per load and, with the SP layout's
    // index arithmetic on top, nine spilled registers.  Offsets are 32-bit and RELATIVE TO THIS SPLIT'S FIRST CHUNK (round 6: the
    // descriptors' bases are advanced by 64-bit pointer arithmetic, once): a split's share of one array must stay below 2^31 bytes,
    // i.e. < 128 M points per launch (checked by the launcher; absolute offsets had capped a launch at 4 M points - C5 with 8 dense
    // event bins is 5.8 M).
    const int64_t cb0 = chunk_begin;
    const __amdgpu_buffer_rsrc_t rs_y = dw_rsrc(src.y + cb0 * YU), rs_y8 = dw_rsrc(src.y8 + cb0 * YU), rs_x = dw_rsrc(src.x + cb0 * XU),
                                 rs_x8 = dw_rsrc(src.x8 + cb0 * XU);
    // X, SP layout: a chunk is 512 consecutive units [slot tid >> 4][point tid & 15]
#define DW_PREFETCH(R, CHUNK)                                                                             \
    {                                                                                                     \
        const int64_t ca = (CHUNK) < chunk_end ? (CHUNK) : chunk_end - 1;                                 \
        const int cc = (int)(ca - cb0);                                                                   \
        _Pragma("unroll") for (int j = 0; j < NY; ++j)                                                    \
            if (YFULL || tid + j * DWT < YU) {                                                            \
                R.y[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, (tid + j * DWT) * 16, cc * (YU * 16), DWS_NT);                  \
                R.y8[j] = DWS_CODE_LOAD(dw_load_b64(rs_y8, (tid + j * DWT) * 8, cc * (YU * 8)));          \
            }                                                                                             \
        R.x[0] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, tid * 16, cc * (XU * 16), DWS_NT);           \
        R.x8[0] = DWS_CODE_LOAD(dw_load_b64(rs_x8, tid * 8, cc * (XU * 8)));                              \
        if (ALPHA) {                                                                                      \
            const int64_t row = ca * CHP + (tid & (CHP - 1));                                             \
            R.da = a.d_raw[(row < M ? row : M - 1) * (a.C + 1) + a.C];                                    \
            if (row >= M) R.da = 0.f;                                                                     \
        }                                                                                                 \
    }
    // global unit u = block * W + w of the chunk is unit u of the LDS image [block][w] (16-byte writes, conflict free); the
    // low halves are decoded from the codes here, between the load registers and LDS
    auto put_pair = [&](u32x4* img_hi, u32x4* img_lo, int u, const u32x4 hi, const uint2 code, bool valid) {
        const uint32_t h4[4] = {hi[0], hi[1], hi[2], hi[3]};
        uint32_t l4[4];
        h8_decode_unit(h4, code, l4);
        img_hi[u] = valid ? hi : u32x4{0u, 0u, 0u, 0u};
        img_lo[u] = valid ? u32x4{l4[0], l4[1], l4[2], l4[3]} : u32x4{0u, 0u, 0u, 0u};
    };
    auto stage_y = [&](const Regs& R, int b, bool valid) {     // past the range: contributes nothing
        u32x4* Ys_ = smem + b * BUF;
#pragma unroll
        for (int j = 0; j < NY; ++j) {
            const int u = tid + j * DWT;
            if (YFULL || u < YU) put_pair(Ys_, Ys_ + YU, u, R.y[j], R.y8[j], valid);
        }
    };
    // X image: unit (slot s, point p) at s * 16 + (p ^ 4 (s & 3)): the four slots of a column tile start a quarter of the banks
    // apart, so the transpose reads of a fragment (4 points x 4 slots per half wave) and these 16-byte writes are conflict free
    auto stage_x = [&](const Regs& R, int b, bool valid) {
        u32x4* Xs_ = smem + b * BUF + 2 * YU;
        const int s_ = tid >> 4, p_ = tid & 15;
        put_pair(Xs_, Xs_ + XU, s_ * 16 + (p_ ^ (4 * (s_ & 3))), R.x[0], R.x8[0], true);
        if (ALPHA && tid < CHP) reinterpret_cast<float*>(Xs_ + 2 * XU)[tid] = valid ? R.da : 0.f;
    };
    // B fragment (8 points 8 lh .. + 7 of feature 32 ctile + lr) = two transpose reads (4 points each); byte offsets of this lane
    // inside the hi image for column tile 0: within a 16-lane group lane t addresses point (t >> 2), features 4 (t & 3) .. + 3 of
    // the group's 16 and receives feature t (tools/hwprobe/tr_read.hip); + ctile * 1024 per column tile, + XU * 16 for the lo image
    int xfo[2];
    {
        const int t = lane & 15, g1 = (lane >> 4) & 1;
        const int sl = 2 * g1 + ((t & 3) >> 1);                      // slot inside the column tile = slot & 3
#pragma unroll
        for (int h = 0; h < 2; ++h) xfo[h] = ((sl * 16 + ((8 * lh + 4 * h + (t >> 2)) ^ (4 * sl))) * 16) + 8 * (t & 1);
    }
    typedef short short4v __attribute__((ext_vector_type(4)));
    auto xfrag = [&](const u32x4* img, int ctile) -> half8 {
        const char* base = reinterpret_cast<const char*>(img) + ctile * 1024;
        const short4v a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)(reinterpret_cast<const short4v*>(base + xfo[0])));
        const short4v a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)(reinterpret_cast<const short4v*>(base + xfo[1])));
        typedef short short8v __attribute__((ext_vector_type(8)));
        return __builtin_bit_cast(half8, short8v{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]});
    };
    // one LDS-only barrier per chunk.  The image written before it (chunk j + 1) was last read two barriers earlier (three images)
#define DW_BARRIER()                                                                                      \
    {                                                                                                     \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");                                   \
        __builtin_amdgcn_s_barrier();                                                                     \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");                                   \
    }
    // every wave of the grid multiplies (the big instances, L0 / L5P): the staging of the NEXT chunk goes INTO the MFMA block
    constexpr bool ALLMMA = (8 / WN - 1) * TC * 32 < K;
    // Chunk b's MFMAs with the decode + staging of the next chunk (register set Rn -> image bn) between them: ONE basic block,
    // so the VALU / LDS-write instructions of the staging issue in the shadow of the MFMAs (an MFMA holds the matrix pipe for
    // 32 cycles, the in-order wave would otherwise just wait for it) - first half of the column tiles, Y staging, second half,
    // X staging.  With the staging as its own phase in front of the barrier (the f16 kernel's order, where it is a plain copy)
    // the decode sat on the critical path of every chunk: 2.25 ms per 522 k-point launch against 1.99 ms for f16 pairs.
    auto compute = [&](int b, const Regs& Rn, int bn, bool vn) {
        const u32x4* Yl = smem + b * BUF;
        const u32x4* Xl = Yl + 2 * YU;
        const float* da = reinterpret_cast<const float*>(Xl + 2 * XU);
        if (ALLMMA || mma_wave) {
            half8 ayh[TR], ayl[TR];
#pragma unroll
            for (int r = 0; r < TR; ++r) {
                ayh[r] = __builtin_bit_cast(half8, Yl[lh * N + (wn * TR + r) * 32 + lr]);
                ayl[r] = __builtin_bit_cast(half8, Yl[YU + lh * N + (wn * TR + r) * 32 + lr]);
            }
            // column tiles in groups of CG: the three MFMAs of one accumulator are TR * CG issue slots apart, and only CG
            // fragment pairs of X are live at a time (all TC at once spilled)
#ifndef DWS_CG
#define DWS_CG 2
#endif
            constexpr int CG = TC >= DWS_CG ? DWS_CG : 1;
#pragma unroll
            for (int c0 = 0; c0 < TC; c0 += CG) {
                half8 bxh[CG], bxl[CG];
#pragma unroll
                for (int c = 0; c < CG; ++c) {
                    bxh[c] = xfrag(Xl, wk * TC + c0 + c);
                    bxl[c] = xfrag(Xl + XU, wk * TC + c0 + c);
                }
#ifdef DWS_ONE_MFMA      /* timing variant (wrong results): one MFMA per block instead of three - how far from its byte bound is the kernel? */
#pragma unroll
                for (int r = 0; r < TR; ++r)
#pragma unroll
                    for (int c = 0; c < CG; ++c) acc[r][c0 + c] = mfma16(ayh[r], bxh[c], acc[r][c0 + c]);
                asm volatile("" :: "v"(bxl[0]), "v"(ayl[0]));
#else
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
#endif
                if (ALLMMA && c0 == 0) stage_y(Rn, bn, vn);
                if (ALLMMA && c0 + CG >= TC) stage_x(Rn, bn, vn);
            }
        }
        if (!ALLMMA) {
            stage_y(Rn, bn, vn);
            stage_x(Rn, bn, vn);
        }
        if (src.bias && tid < N) {
#pragma unroll
            for (int mb = 0; mb < CHB; ++mb)
                bsum = dw_sum_unit(__builtin_bit_cast(half8, Yl[mb * N + tid]), __builtin_bit_cast(half8, Yl[YU + mb * N + tid]), bsum);
        }
        // the alpha head rides on waves 4..7 (column tid - 256): waves 0..3 already carry the bias sums (mlp_dw_h.hip)
        if (ALPHA && tid >= DWT - K) {
            // column ka = tid - 256 = 64 (wave - 4) + lane of X = h7: the 16 points of the chunk by four transpose reads per image
            // (lane t of a 16-lane group g: feature 64 w' + 16 g + t, the group's 4 x 16 block = points 4 pg .. + 3)
            const int t = lane & 15, g = lane >> 4, wq = wave - 4;
            const int slot = 8 * wq + 2 * g + ((t & 3) >> 1), sl = slot & 3;
            float s = 0.f, sb = 0.f;
#pragma unroll
            for (int pg = 0; pg < 4; ++pg) {
                const int off = ((slot * 16 + ((4 * pg + (t >> 2)) ^ (4 * sl))) * 16) + 8 * (t & 1);
                const char* bh = reinterpret_cast<const char*>(Xl) + off;
                const short4v vh = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)(reinterpret_cast<const short4v*>(bh)));
                const short4v vl = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)(reinterpret_cast<const short4v*>(bh + XU * 16)));
                typedef _Float16 half4v __attribute__((ext_vector_type(4)));
                const half4v h = __builtin_bit_cast(half4v, vh), l = __builtin_bit_cast(half4v, vl);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = da[pg * 4 + j];
                    s += d * ((float)h[j] + (float)l[j]);
                    sb += d;
                }
            }
            asum += s;
            absum += sb;
        }
    };

    // Three register sets in flight, three LDS images.  (Four sets / two images / one column tile at a time - 96 KiB in flight
    // per CU instead of 72 - measured the same or 3 % slower: with 3 bytes per value the kernel is no longer bound by the bytes in
    // flight but by its per-chunk schedule, MFMA pipe busy ~55 %, like the forward and dX kernels.)
    if (chunk_begin < chunk_end) {
        DW_PREFETCH(rA, chunk_begin);
        DW_PREFETCH(rB, chunk_begin + 1);
        DW_PREFETCH(rC, chunk_begin + 2);
        stage_y(rA, 0, true);
        stage_x(rA, 0, true);
        DW_BARRIER();
        for (int64_t chunk = chunk_begin; chunk < chunk_end; chunk += 3) {
            const bool v1 = chunk + 1 < chunk_end, v2 = chunk + 2 < chunk_end, v3 = chunk + 3 < chunk_end;
            DW_PREFETCH(rA, chunk + 3);
            __builtin_amdgcn_sched_barrier(0);   // keep the loads HERE: the scheduler otherwise sinks them below the MFMAs
            compute(0, rB, 1, v1);
            DW_BARRIER();
            DW_PREFETCH(rB, chunk + 4);
            __builtin_amdgcn_sched_barrier(0);
            compute(1, rC, 2, v2);
            DW_BARRIER();
            DW_PREFETCH(rC, chunk + 5);
            __builtin_amdgcn_sched_barrier(0);
            compute(2, rA, 0, v3);
            DW_BARRIER();
        }
    }
#undef DW_BARRIER
#undef DW_PREFETCH

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
        apart[tid - (DWT - K)] = asum;
        if (tid == DWT - K) apart[256] = absum;
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
    const uint2* hv8 = reinterpret_cast<const uint2*>(sact_lo8(a.acts, Mp, sact_hv(Mp)));
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
        u32x4 hh[BB / 4];
        uint2 h8c[BB / 4];
#pragma unroll
        for (int i = 0; i < BB / 4; ++i) {
            const int64_t mb = b0 + ph + 4 * i;
            hh[i] = u32x4{0u, 0u, 0u, 0u};
            h8c[i] = uint2{0x80808080u, 0x80808080u};
            if (mb < blk_end) {
#if DWS_NT
                hh[i] = __builtin_nontemporal_load(&hv[mb * ACT_HV_W + j]);     // 8 points of column j
                {
                    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                    const u32x2 c2 = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(&hv8[mb * ACT_HV_W + j]));
                    h8c[i] = uint2{c2[0], c2[1]};
                }
#else
                hh[i] = hv[mb * ACT_HV_W + j];     // 8 points of column j
                h8c[i] = hv8[mb * ACT_HV_W + j];
#endif
            }
        }
#pragma unroll
        for (int i = 0; i < BB / 4; ++i) {
            const uint32_t h4[4] = {hh[i][0], hh[i][1], hh[i][2], hh[i][3]};
            uint32_t l4[4];
            h8_decode_unit(h4, h8c[i], l4);
            const half8 pq = __builtin_bit_cast(half8, hh[i]), pl = __builtin_bit_cast(half8, u32x4{l4[0], l4[1], l4[2], l4[3]});
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

constexpr size_t DWS_SMEM = 3 * (size_t)(2 * CHB * 256 + 2 * CHB * 256 + CHP / 4) * 16;       // three chunk images of the 256 x 256 block: 98 496 B
__device__ __forceinline__ int64_t chunks_per_split(const DwArgs& a, int inst) {
    return (m_pad(a.M) / CHP + dws_splits(inst) - 1) / dws_splits(inst);
}
__device__ __forceinline__ void chunk_range(const DwArgs& a, int inst, int split, int64_t& cb, int64_t& ce) {
    const int64_t nchunks = m_pad(a.M) / CHP;
    const int64_t per = (nchunks + dws_splits(inst) - 1) / dws_splits(inst);
    cb = (int64_t)split * per;
    ce = cb + per;
    if (cb > nchunks) cb = nchunks;
    if (ce > nchunks) ce = nchunks;
}

// -DBENERF_TRACE_DW: thread 0 of every workgroup stamps the 100 MHz wall clock at its start and end behind the partial sums in
// the workspace (u64 [kernel: 0 small, 1 big][512 workgroups][2]) - tools/experiments/trace_dw.py prints per-instance finish times.
#ifdef BENERF_TRACE_DW
#define DW_TRACE(kern, which) do { if (threadIdx.x == 0) reinterpret_cast<unsigned long long*>(a.ws + ((dws_inst_offset(DW_COUNT) + 63) & ~63LL))[((kern) * 512 + blockIdx.x) * 2 + (which)] = wall_clock64(); } while (0)
#else
#define DW_TRACE(kern, which) do { } while (0)
#endif

// the seven 256x256 instances L1 .. L7 + the 128x256 block G = dhv^T h7 (with the alpha head): one workgroup per CU, every operand
// byte read once
__global__ __launch_bounds__(DWT, 2) void mlp_dw_split_big_kernel(DwArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32x4 smem_u[];
    DW_TRACE(1, 0);
    const int inst = dws_big_inst(blockIdx.x), split = dws_big_split(blockIdx.x);
    int64_t cb, ce;
    chunk_range(a, inst, split, cb, ce);
    float* part = a.ws + dws_inst_offset(inst) + (int64_t)split * dw_inst_floats(inst);
    const Src src = inst_src(a, inst);
    if (inst <= DW_L7) dw_gemm<256, 256, 4, 2, 4, false>(a, src, cb, ce, part, smem_u);
    else     // alpha partials: the tail of this split's block in the (otherwise unused) FEAT instance part of the workspace
        dw_gemm<128, 256, 2, 2, 2, true>(a, src, cb, ce, part, smem_u,
                                         a.ws + dws_inst_offset(DW_FEAT) + (int64_t)split * dw_inst_floats(DW_FEAT) + 256 * 256 + 256);
    DW_TRACE(1, 1);
}

// ---- the thin instances: L0 and the PE part of L5 (256 x 64 each, X = PE), the PE(dir) part of the views layer (128 x 32), rgb head -----
// X is 64 / 32 features wide, so ONE wave can own a 32-row tile of dY against ALL of X (2 / 1 accumulator tiles) and nothing of
// the heavy operand (dY: 768 of the 960 bytes per point and instance) is shared between waves.  SH units ARE the MFMA fragments of
// a contraction over points (mlp_split.h), so a wave loads its fragments straight from global memory - lane (lr, lh) takes unit
// (block 2 k + lh, feature 32 rt + lr): 16 bytes of hi halves + 8 bytes of codes, 512 / 256 contiguous bytes per half wave -
// decodes the low halves in registers and multiplies: no LDS, no barrier, no staging pass; every wave streams at its own pace
// with TS_DEPTH k-steps (16 points each) in flight.  The X fragments (3 KiB per k-step against 24 of dY) are requested by all
// eight waves of a workgroup and come out of the vector L1 / L2.  L0 and L5P share X: one wave carries row tile rt of dY0 AND of
// dY5 (four accumulator tiles), so the encoding is fetched once for both.
// Before (round 4's first version): the dw_gemm structure above with X split from f32 rows while staging - six MFMAs per wave and
// barrier, 2.7 TB/s; 0.96 ms per C2 step for 6 % of the dW FLOPs.
#ifndef TS_DEPTH
#define TS_DEPTH 4
#endif
template <int NYA, int TC>
struct ThinRegs {
    u32x4 y[NYA];
    uint2 yc[NYA];
    u32x4 x[TC];
    uint2 xc[TC];
};

__device__ __forceinline__ half8 h8_lo_of(const u32x4 hi, const uint2 code) {
    const uint32_t h4[4] = {hi[0], hi[1], hi[2], hi[3]};
    uint32_t l4[4];
    h8_decode_unit(h4, code, l4);
    return __builtin_bit_cast(half8, u32x4{l4[0], l4[1], l4[2], l4[3]});
}
// sum of the 8 + 8 halfs of a unit in f32 (v_dot2_f32_f16 against (1, 1))
__device__ __forceinline__ float h8_sum_unit(const half8 h, const half8 l) {
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    const half2v one = {(_Float16)1.f, (_Float16)1.f};
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s = __builtin_amdgcn_fdot2(half2v{h[2 * j], h[2 * j + 1]}, one, s, false);
        s = __builtin_amdgcn_fdot2(half2v{l[2 * j], l[2 * j + 1]}, one, s, false);
    }
    return s;
}

// NYA dY arrays of width YW (row tile rt of each) x the X array of width 32 TC, k-steps [kb, ke).  part[i]: this split's partial
// block of dY array i ([YW][32 TC] then bias [YW]); bias sums for array 0 only when BIAS0.
// klen: the loop's length in k-steps, the SAME for every wave of the workgroup (>= ke - kb; the k-steps past ke are loaded clamped and
// multiplied by zero): the waves meet at a workgroup barrier every TS_LOCKSTEP ring cycles.  Nothing is exchanged there - the barrier
// keeps the eight streams of a workgroup within a few k-steps of each other, so that together they read contiguous 8-KiB runs of an
// array at the same time; drifting apart they are eight 512-byte-strided streams to the DRAM pages (round 5, end: thin kernel
// 290 -> 272 us at 522 k points with the barrier in the L0 + L5P workgroups alone: profiles/r05_thin_lockstep_ab.log).
#ifndef TS_LOCKSTEP
#define TS_LOCKSTEP 1
#endif
template <int NYA, int YW, int TC, bool BIAS0>
__device__ __forceinline__ void thin_stream(const u32x4* const (&yh)[NYA], const uint2* const (&y8)[NYA], const u32x4* __restrict__ xh,
                                            const uint2* __restrict__ x8, int rt, int64_t kb, int64_t ke, int64_t klen, int lane,
                                            float* const (&part)[NYA]) {
    constexpr int XW = 32 * TC;
    const int lr = lane & 31, lh = lane >> 5;
    const int yo = lh * YW + rt * 32 + lr;            // unit of this lane inside k-step 0 (a k-step = 2 blocks = 2 YW units)
    const int xo = lh * XW + lr;
    f32x16 acc[NYA][TC];
#pragma unroll
    for (int i = 0; i < NYA; ++i)
#pragma unroll
        for (int c = 0; c < TC; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][c][e] = 0.f;
    float bsum = 0.f;
    typedef ThinRegs<NYA, TC> Regs;
    Regs r[TS_DEPTH];
    // unconditional loads (a k-step past the range is clamped to the last one and not multiplied): loads under branches make
    // the compiler's waitcnt bookkeeping fall back to vmcnt(0) (mlp_dw_h.hip)
    auto load = [&](Regs& R, int64_t k) {
        const int64_t kk = k < ke ? k : (ke > 0 ? ke - 1 : 0);
#pragma unroll
        for (int i = 0; i < NYA; ++i) {
#if DWS_NT
            R.y[i] = __builtin_nontemporal_load(&yh[i][kk * (2 * YW) + yo]);
            {
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 c2 = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(&y8[i][kk * (2 * YW) + yo]));
                R.yc[i] = uint2{c2[0], c2[1]};
            }
#else
            R.y[i] = yh[i][kk * (2 * YW) + yo];
            R.yc[i] = y8[i][kk * (2 * YW) + yo];
#endif
        }
#pragma unroll
        for (int c = 0; c < TC; ++c) {
            R.x[c] = xh[kk * (2 * XW) + xo + c * 32];
            R.xc[c] = x8[kk * (2 * XW) + xo + c * 32];
        }
    };
    // vm: all ones for a k-step inside the range, zero past it (wave-uniform): the dY fragments of a clamped k-step are zeroed -
    // a branch around the MFMAs instead would cost the ring its wait counts (the compiler merges them to vmcnt(0) at the join)
    auto compute = [&](const Regs& R, uint32_t vm) {
        half8 bh[TC], bl[TC];
#pragma unroll
        for (int c = 0; c < TC; ++c) {
            bh[c] = __builtin_bit_cast(half8, R.x[c]);
#ifdef TS_X_NODECODE     /* timing variant: what does decoding the (wave-redundant) X codes cost? */
            bl[c] = __builtin_bit_cast(half8, u32x4{R.xc[c].x, R.xc[c].y, R.xc[c].x, R.xc[c].y});
#else
            bl[c] = h8_lo_of(R.x[c], R.xc[c]);
#endif
        }
#pragma unroll
        for (int i = 0; i < NYA; ++i) {
#ifdef TS_NO_TAIL_MASK   /* timing variant: the past-the-range masks (wrong sums at ragged split ends) */
            const u32x4 yv = R.y[i];
            const half8 ah = __builtin_bit_cast(half8, yv);
            const half8 al = h8_lo_of(R.y[i], R.yc[i]);
            (void)vm;
#else
            const u32x4 yv = R.y[i] & vm;
            const half8 ah = __builtin_bit_cast(half8, yv);
            const half8 al = __builtin_bit_cast(half8, __builtin_bit_cast(u32x4, h8_lo_of(R.y[i], R.yc[i])) & vm);
#endif
#pragma unroll
            for (int c = 0; c < TC; ++c) acc[i][c] = mfma16(ah, bh[c], acc[i][c]);
#pragma unroll
            for (int c = 0; c < TC; ++c) acc[i][c] = mfma16(ah, bl[c], acc[i][c]);
#pragma unroll
            for (int c = 0; c < TC; ++c) acc[i][c] = mfma16(al, bh[c], acc[i][c]);
            if (BIAS0 && i == 0) bsum += h8_sum_unit(ah, al);
        }
    };
    {
#pragma unroll
        for (int d = 0; d < TS_DEPTH; ++d) {
            load(r[d], kb + d);
            // in ring order: left alone the scheduler requests stage 0 LAST, and the wait counts of the loop header (merged from
            // this block and the loop's own back edge) degrade to vmcnt(0) - no prefetch distance left
            __builtin_amdgcn_sched_barrier(0);
        }
        int cyc = 0;
        for (int64_t k = kb; k < kb + klen; k += TS_DEPTH, ++cyc) {
            if (TS_LOCKSTEP > 0 && cyc % (TS_LOCKSTEP > 0 ? TS_LOCKSTEP : 1) == 0) __builtin_amdgcn_s_barrier();      // no data exchanged: see above
#pragma unroll
            for (int d = 0; d < TS_DEPTH; ++d) {
#ifdef TS_LOCKSTEP_K     /* experiment: a barrier in front of every TS_LOCKSTEP_K-th k-step instead of every ring cycle */
                if (d % TS_LOCKSTEP_K == 0 && d > 0) __builtin_amdgcn_s_barrier();
#endif
                compute(r[d], k + d < ke ? 0xffffffffu : 0u);
                load(r[d], k + d + TS_DEPTH);
                __builtin_amdgcn_sched_barrier(0);      // keep the ring as written: stage d's loads stay behind its own MFMAs
            }
        }
    }
    // partial blocks at the gradient scale s_s (the reduce kernel divides it out)
#pragma unroll
    for (int i = 0; i < NYA; ++i) {
#pragma unroll
        for (int c = 0; c < TC; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) part[i][(int64_t)(rt * 32 + acc_row(e, lane)) * XW + c * 32 + lr] = acc[i][c][e];
        float b = (BIAS0 && i == 0) ? bsum : 0.f;
        b += __shfl_xor(b, 32);                         // the two blocks of a k-step sit in the two half waves
        if (lh == 0) part[i][(int64_t)YW * XW + rt * 32 + lr] = b;
    }
}

// Workgroups [0, DWH_T0): split b of L0 + L5P, wave w = row tile w; [DWH_T0, + DWH_T1 / 2): VIEWSP, two splits per workgroup (waves
// 0-3 / 4-7, row tile w & 3); then DWH_T2 workgroups of the rgb head.
constexpr int DWS_SMALL_BLOCKS = DWH_T0 + DWH_T1 / 2 + DWH_T2;
static_assert(DWH_T1 % 2 == 0 && dws_splits(DW_L0) == dws_splits(DW_L5P), "thin split tables");
__global__ __launch_bounds__(DWT, 2) void mlp_dw_split_small_kernel(DwArgs a) {
    __shared__ __attribute__((aligned(16))) float rgb_smem[32 * 8 * 4 + 18 * 128];
    DW_TRACE(0, 0);
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t Mp = m_pad(a.M);
    const float* A = a.acts;
    const float* D = a.dacts;
    auto dy = [&](int64_t off) { return reinterpret_cast<const u32x4*>(D + off); };
    auto dy8 = [&](int64_t off) { return reinterpret_cast<const uint2*>(sdact_lo8(D, Mp, off)); };
    int64_t cb, ce;
    if (b < DWH_T0) {
        chunk_range(a, DW_L0, b, cb, ce);
        const u32x4* const yh[2] = {dy(sdact_h(Mp, 0)), dy(sdact_h(Mp, 5))};
        const uint2* const y8[2] = {dy8(sdact_h(Mp, 0)), dy8(sdact_h(Mp, 5))};
        float* const part[2] = {a.ws + dws_inst_offset(DW_L0) + (int64_t)b * dw_inst_floats(DW_L0),
                                a.ws + dws_inst_offset(DW_L5P) + (int64_t)b * dw_inst_floats(DW_L5P)};
        thin_stream<2, 256, 2, true>(yh, y8, reinterpret_cast<const u32x4*>(A + sact22_pe_hi(Mp)),
                                     reinterpret_cast<const uint2*>(A + sact22_pe_lo8(Mp)), wave, cb, ce, chunks_per_split(a, DW_L0), lane, part);
    } else if (b < DWH_T0 + DWH_T1 / 2) {
        const int split = 2 * (b - DWH_T0) + (wave >> 2);
        chunk_range(a, DW_VIEWSP, split, cb, ce);
        const u32x4* const yh[1] = {dy(sdact_hv(Mp))};
        const uint2* const y8[1] = {dy8(sdact_hv(Mp))};
        float* const part[1] = {a.ws + dws_inst_offset(DW_VIEWSP) + (int64_t)split * dw_inst_floats(DW_VIEWSP)};
        thin_stream<1, 128, 1, false>(yh, y8, reinterpret_cast<const u32x4*>(A + sact22_ped_hi(Mp)),
                                      reinterpret_cast<const uint2*>(A + sact22_ped_lo8(Mp)), wave & 3, cb, ce, chunks_per_split(a, DW_VIEWSP), lane, part);
    } else {
        const int split = b - (DWH_T0 + DWH_T1 / 2);
        chunk_range(a, DW_RGB, split, cb, ce);
        dw_rgb(a, cb * CHB, ce * CHB, a.ws + dws_inst_offset(DW_RGB) + (int64_t)split * dw_inst_floats(DW_RGB), rgb_smem);
    }
    DW_TRACE(0, 1);
}

}  // namespace

int benerf_mlp_dw_reduce_launch(const float* ws, const BenerfMlpGrads* grads, int channels, int accumulate, int split_mode,
                                const float* grad_info, const float* pe_weights, hipStream_t stream, const BenerfMlpParams* params);

int benerf_mlp_dw_split22_launch(const BenerfMlpParams* params, int channels, int64_t M, const float* d_raw, const float* acts,
                                 const float* dacts, float* dw_ws, const BenerfMlpGrads* grads, int accumulate, const float* pe_weights,
                                 hipStream_t stream) {
    DwArgs a;
    a.d_raw = d_raw;
    a.acts = acts;
    a.dacts = dacts;
    a.ws = dw_ws;
    a.M = M;
    a.C = channels;
    BENERF_REQUIRE(mlp::m_pad(M) < (1ll << 27), "mlp_bwd(dw, split): at most 128 M points per launch (32-bit buffer offsets inside a split)");
    static BenerfLdsAttr attr_big;      // once per device
    if (!benerf_lds_attr(attr_big, (const void*)mlp_dw_split_big_kernel, (int)DWS_SMEM)) {
        benerf_set_error("mlp_bwd(dw, split): cannot reserve LDS");
        return BENERF_EHIP;
    }
    (void)pe_weights;       // the saved encodings carry the BARF column weights already (mlp_split.h: sact22_*)
    hipLaunchKernelGGL(mlp_dw_split_small_kernel, dim3(DWS_SMALL_BLOCKS), dim3(DWT), 0, stream, a);
    BENERF_LAUNCH_CHECK("mlp_bwd(dw small, split)");
    hipLaunchKernelGGL(mlp_dw_split_big_kernel, dim3(mlp::DWS_BIG_BLOCKS), dim3(DWT), DWS_SMEM, stream, a);
    BENERF_LAUNCH_CHECK("mlp_bwd(dw, split)");
    return benerf_mlp_dw_reduce_launch(dw_ws, grads, channels, accumulate, 2, dacts + mlp::sdact_info(mlp::m_pad(M)), nullptr, stream, params);
}

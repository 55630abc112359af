// K3 backward part 1, BENERF_MLP_SPLIT (the 22-bit backward): the activation-gradient chain of mlp_bwd.hip (phases P0..P6, same
// d_pts / d_viewdirs outputs) with every GEMM as THREE f16 MFMAs per product block on hi/lo-split operands - the gradient as
// hi + lo (two f16, unscaled lo), the transposed weight as hi + lo (mlp_pack.hip, backward blocks) -
//     dY W  =  dY_hi W_hi + dY_hi W_lo + dY_lo W_hi        (lo x lo <= 2^-22 |dY W| dropped)
// all three into ONE f32 accumulator.  Both operands keep 22 significant bits: measured against float64 with identical ReLU
// masks the gradients carry the exact-f32 kernel's error (tools/experiments/backward_format_study.py is the CPU study that
// fixed the formats: an f16 gradient OR an f16 activation in either backward GEMM costs 2-8e-4 of the largest entry, 15-bit
// operands 2-4e-5, hi + lo 1e-6 = float32's own).  This is what makes the split mode fp32-equivalent in the backward pass
// too (the reference: fp32 nn.Linear, model/nerf.py:93-112, loss.backward() through them, train.py:340).
//
// Range: gradients are far outside f16's range (d_raw ~ 1/n_rays), but the chain is LINEAR in d_raw: each 128-point tile
// multiplies its d_raw by a power of two s (pow2_scale6: tile maximum -> [2^6, 2^7)), runs the chain on the scaled values and
// multiplies every output by 1/s - both exact.  The dY arrays for the dW kernel (f16 hi halves in SH layout + 8-bit residual codes,
// "lo8" twins: 19-bit operands, mlp_split.h) carry ONE power-of-two scale per call (s_s from max|d_raw| of the launch), divided
// out by the dW reduce kernel.  Inside the chain (LDS planes) the gradient stays an f16 pair.
//
// Tiling (the split forward kernel's): one workgroup of 8 waves per 128 points, two f16 planes Th / Tl (FEATURE-major,
// [320][128]: see fidx) = the CU's whole 160 KiB; wave w owns the 32 input features of column tile w x all four point tiles (one accumulator set of 64
// registers): a weight-fragment pair (hi, lo: 2 KiB from L2) feeds 12 MFMAs.  Un-transposed product (activations as the MFMA A
// operand): an accumulator lane holds 4 consecutive points of one feature, so the lane pair (l, l + 32) forms whole 16-byte SH
// units in registers (v_permlane32_swap) and the ReLU sign-bit words the forward pass saved line up with the accumulators.
// The planes' PE feature rows [256,320) are never a GEMM operand here: f32 scratch (hi plane: dPE(dir) at floats [0,27) of a
// point, the tile's maximum at [26] of points 0 / 1 during P0; lo plane: the scaled d_raw, channel-major - draw_cm), later the layer-5 skip's dPE block as hi + lo.
#include "mlp_split.h"
// Cache policy of the dY stores (aux of raw_buffer_store: 0 default, 2 = nt): written once, read by the dW launch after this one - nt keeps
// them from evicting the weight fragments out of L2 (mlp_fwd_h.hip: FWD_ST_AUX; profiles/r05_cache_policy_ab.log).  The forward's sign-bit
// words are read once, too: nt loads.
#ifndef BWS_ST_AUX
#define BWS_ST_AUX 2
#endif

// -DBENERF_TRACE_DX: thread 0 of the first 2048 workgroups stamps the 100 MHz wall clock at the phase boundaries into the d_viewdirs
// output (which is then not written) - tools/experiments/trace_phases.py prints the per-phase durations behind DESIGN.md 4.
#ifdef BENERF_TRACE_DX
#define TR(i) do { if (tid == 0 && blockIdx.x < 2048) reinterpret_cast<unsigned long long*>(a.d_vdir)[blockIdx.x * 64 + (i)] = wall_clock64(); } while (0)
#else
#define TR(i) do { } while (0)
#endif
namespace {
using namespace mlp;

constexpr int TMB = 128;                                              // points per workgroup
constexpr int BNT = 512;                                              // 8 waves
constexpr size_t BWD_SMEM = (size_t)2 * TMB * LD * sizeof(_Float16);  // 163 840 B
static_assert(TMB == SM_PAD, "the padded point count is a whole number of dX tiles");

struct BwdArgs {
    const float* d_raw;
    const float* acts;
    float* dacts;
    const float* packed;    // split-f16 section at + PACKED_FLOATS (backward blocks: hi + unscaled lo)
    const float* w_alpha;   // [256]
    const float* w_rgb;     // [C][128]
    const float* pe_w;      // BARF c2f column weights (include/benerf_hip.h) or null
    float* d_pts;           // [M][3]
    float* d_vdir;          // [M][3]
    uint32_t* status;       // [1]: max |tile-scaled gradient| bits once >= 2^15, [2]: acts buffer written by another mode (may be null)
    const float* absmax;    // max |d_raw| of the call from the compositing backward, or null: then it is in the dacts info word
    int64_t M;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));

// ---- LDS planes, FEATURE-MAJOR: plane[feature 0..319][128 points] f16 (256 bytes per feature row) -----------------------------
// An accumulator lane of the un-transposed product holds 4 consecutive points of one feature = 8 contiguous bytes of this layout:
// the epilogue writes whole quads (ds_write_b64, 32 per wave and stage for both planes) where a point-major plane needs 128
// two-byte writes; the K-loop gets its A fragments (row = point, 8 consecutive features) with two ds_read_b64_tr_b16 each
// (tools/hwprobe/tr_read.hip: within a 16-lane group lane t supplies the address of row t >> 2, halfs 4 (t & 3) .. + 3 of a
// 4 x 16 block and receives column t: here rows = features, columns = points).  Swizzle: the 64-byte segment (32 points) of a
// row is XORed with feature & 3, the 16-byte slot (8 points) inside it with (feature >> 2) & 3 - the four feature rows of a
// transpose read land in four different bank quarters, eight consecutive features of a quad write in eight different slots.
constexpr int PROW = 128;                                               // halfs per feature row
// Round 5: the two 8-byte halves of a slot are swapped for features with bit BWS_HALF_BIT set (below); the transpose reads of a
// 16-lane group fetch whole 32-byte runs of four feature rows in four different 64-byte segments, so they do not care which half is which.
// Which feature bit swaps the two 8-byte halves of a slot.  Bit 4 (round 5's first version) separated lanes n and n + 16 of an epilogue
// quad write: 57 M -> 41-43 M conflict cycles per 522 k-point launch.  What the counter then still showed was attributed with
// early-return builds (-DBWS_STOP_AFTER): 31 of the 41 M sit in the seven layer stages = the epilogue's 256 ds_write_b64 per layer and tile,
// each two-way conflicted.  An 8-byte write of a wave goes out 16 lanes per cycle over 128 bytes of banks; a feature row is 256 bytes, so
// its 64-byte segments s and s ^ 2 share banks, and among lanes 0..15 the segment (rt ^ (n & 3)) takes both - unless the half follows
// bit 1 of the feature: then (segment & 1, half) is a bijection on n & 3 and the sixteen lanes cover sixteen different 8-byte bank pairs.
// Bit 1: 41.0 M -> 5.5 M conflict cycles per launch (profiles/r05_dx_lds_conflicts.log); time-neutral (LDS conflicts were never what
// bounds the epilogue), and the A-fragment offsets no longer depend on the k-step's parity.
#ifndef BWS_HALF_BIT
#define BWS_HALF_BIT 1
#endif
__device__ __forceinline__ int fidx(int f, int p) {
#ifdef BWS_OLD_SWIZZLE
    return f * PROW + ((((p >> 5) ^ (f & 3)) << 5) | ((((p >> 3) & 3) ^ ((f >> 2) & 3)) << 3) | (p & 7));
#else
    return f * PROW + ((((p >> 5) ^ (f & 3)) << 5) | ((((p >> 3) & 3) ^ ((f >> 2) & 3)) << 3) | ((p & 7) ^ (((f >> BWS_HALF_BIT) & 1) << 2)));
#endif
}
// f32 scratch float i (0..31) of point `row`: the feature rows [256,320) of the hi plane (never a GEMM operand here) as 4096
// floats, 32 per point, groups of four floats XOR-swizzled by the point
__device__ __forceinline__ float* fscr1(_Float16* T, int row, int i) {
    return reinterpret_cast<float*>(T + 256 * PROW) + row * 32 + (i ^ ((row & 7) << 2));
}
// The tile's scaled d_raw, CHANNEL-major: float [C + 1][128 points] at the head of the LO plane's feature rows [256,320) (free until the
// layer-5 skip's dPE block): a lane of P1 / P2 wants the 4 consecutive points of an accumulator quad - one 16-byte read per channel
// where the point-major scratch rows took four (round 5: P1 was 4.4 us of a 105 us tile on 2-byte plane writes and per-element reads).
__device__ __forceinline__ float* draw_cm(_Float16* Tl, int c) { return reinterpret_cast<float*>(Tl + 256 * PROW) + c * 128; }
// A fragment pieces of this lane: half offsets at k-step 0 for piece j (features 8 (lane >> 5) + 4 j + ((lane & 15) >> 2)) and row
// tile rt (points rt * 32 + 16 ((lane >> 4) & 1) + 4 (lane & 3)); a k-step further is 16 feature rows = 16 * PROW halfs
__device__ __forceinline__ int frag_off(int lane, int j, int rt) {
    const int t = lane & 15;
    return fidx(8 * (lane >> 5) + 4 * j + (t >> 2), rt * 32 + 16 * ((lane >> 4) & 1) + 4 * (t & 3));
}
typedef short short4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ half8 frag_read(const _Float16* __restrict__ T, int off0, int off1) {
    const short4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)(reinterpret_cast<const short4v*>(T + off0)));
    const short4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)(reinterpret_cast<const short4v*>(T + off1)));
    typedef short short8v __attribute__((ext_vector_type(8)));
    return __builtin_bit_cast(half8, short8v{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]});
}

// buffer descriptor on a wave-uniform base address (per-lane addresses become ONE 32-bit VGPR offset: mlp_bwd_h.hip)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* p) {
    const uint64_t wa = reinterpret_cast<uint64_t>(p);
    const uint64_t wau = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(wa >> 32)) << 32) |
                         (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)wa);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(wau), 0, 0x7fffffff, 0x00020000);
}

// weight fragments of a packed block through a buffer descriptor; layout: mlp_pack.hip / mlp_split.h
struct WFrag {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff;
    __device__ __forceinline__ WFrag(const float* wp, int lane) : rsrc(uniform_rsrc(wp)), voff(lane * 16) {}
    // fragment of column tile t (wave-uniform), k-step ks of a block with KS k-steps; plane 0 = hi, 1 = lo (unscaled)
    __device__ __forceinline__ u32x4 load(int t_uniform, int ks, int KS, int plane) const {
        const int soff = (((t_uniform >> 1) * KS + ks) * 2 + (t_uniform & 1)) * 2048 + plane * 1024;
        return __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
    }
};

// offset of the backward block of hidden layer l (7..1) in closed form (a run-time pack_offset() becomes a scalar loop of branches)
__device__ __forceinline__ int bwd_layer_offset(int l) {
    return (int)pack_offset(PB_L7) + (7 - l) * (int)pack_floats(PB_L7) + (l < 5 ? (int)(pack_floats(PB_L5) - pack_floats(PB_L7)) : 0);
}
static_assert(pack_offset(PB_L7) + 1 * pack_floats(PB_L7) == pack_offset(PB_L6) && pack_offset(PB_L7) + 2 * pack_floats(PB_L7) == pack_offset(PB_L5) &&
              pack_offset(PB_L7) + 3 * pack_floats(PB_L7) + (pack_floats(PB_L5) - pack_floats(PB_L7)) == pack_offset(PB_L4) &&
              pack_offset(PB_L7) + 6 * pack_floats(PB_L7) + (pack_floats(PB_L5) - pack_floats(PB_L7)) == pack_offset(PB_L1), "bwd_layer_offset");

// lane-derived addresses stay local to a stage (hoisted out of the layer loop they overflow the register file: mlp_bwd_h.hip)
__device__ __forceinline__ int stage_local(int lane) {
#ifndef BWS_HOIST     /* experiment: let the compiler hoist the lane-derived addresses out of the layer loop */
    asm volatile("" : "+v"(lane));
#endif
    return lane;
}

// The lane-derived LDS offsets the 256-wide stages share, computed ONCE per tile (16 registers carried across the layer loop):
//   ao0[j][rt]  piece j of the A fragment of row tile rt at k-step 0, hi plane (frag_off)
//   ta[rt], b[q] plane offset of the epilogue's quad (row tile rt, block q) = ta[rt] + b[q]  (feature ct * 32 + (lane & 31))
// Recomputed per stage (stage_local: nothing carried) they were ~190 of a layer's 704 VALU instructions - 130 of them in front of the
// K-loop's first MFMA; carried ALL (the compiler's own hoisting: 32 fragment offsets + 16 quad offsets) they spill 33 registers.  What a
// stage derives from these 16 per layer: the odd k-steps' offsets (^ 4), the lo plane's (+ plane distance), the 16 quad sums.
struct LaneAddr {
    int ao0[2][4];
    int ta[4], b[4];
};

#ifndef BWS_PF
#define BWS_PF 2        /* k-steps of weight-fragment prefetch: see BWS_DEFER_RT */
#endif
template <int PF>
struct WRing { u32x4 q[PF + 1][2]; };

// the first PF k-steps of weight fragments of column tile ct.  Called BEFORE the previous stage's epilogue: vector-memory
// operations retire in order, fragments requested behind the epilogue's 16 stores would wait for every one of them.
template <int KS, int PF>
__device__ __forceinline__ void gemm3_head(const float* __restrict__ wp, int ct, int lane, WRing<PF>& r) {
    const WFrag wf(wp, lane);
    const int ctu = __builtin_amdgcn_readfirstlane(ct);
#pragma unroll
    for (int p = 0; p < PF; ++p)
        if (p < KS) {
            r.q[p][0] = wf.load(ctu, p, KS, 0);
            r.q[p][1] = wf.load(ctu, p, KS, 1);
        }
}

// acc[rt] += (Th + Tl)[rt*32.., 0 .. KS*16) x (W_hi + W_lo)(tile ct) without the lo x lo term, rt = 0..3.  Per k-step: the four
// hi x hi MFMAs, the four hi x lo, the four lo x hi - the three MFMAs on one accumulator are four issue slots apart.  Weight
// fragments PF k-steps ahead, ONE set of activation fragments, each reloaded right behind its last MFMA of the k-step.
struct NoKstepHook { __device__ __forceinline__ void operator()(int) const {} };
// hook(ks): caller's work placed in k-step ks's scheduling region, behind its MFMAs' operands (the previous stage's deferred units)
template <int KS, int PF, class HK = NoKstepHook>
__device__ __forceinline__ void gemm3_body(const _Float16* __restrict__ Th, const _Float16* __restrict__ Tl, const float* __restrict__ wp,
                                           int ct, int lane, WRing<PF>& r, f32x16 (&acc)[4], const LaneAddr& la, HK hook = HK()) {
    lane = stage_local(lane);
    const WFrag wf(wp, lane);
    const int ctu = __builtin_amdgcn_readfirstlane(ct);
    // The lo plane sits 80 KiB behind the hi plane: past the 16-bit immediate offset of a DS instruction.  Addressed as Tl + offset
    // the compiler folds plane distance and k-step into ONE constant, finds it too large and emits a v_add per transpose read
    // (8 per k-step = 0.67 VALU per MFMA of this loop).  A second set of lane offsets with the plane distance folded in, opaque
    // to the constant folder, leaves the k-step (<= 60 KiB) as the instruction's immediate.
    // [parity of the k-step]: with the half-swap on feature bit 4 the two 8-byte halves of a slot are swapped in every odd k-step (NPAR = 2 sets
    // of offsets); on a lower bit (the shipped bit 1) the k-step does not enter
#ifdef BWS_OLD_SWIZZLE
    constexpr int NPAR = 1;
#else
    constexpr int NPAR = BWS_HALF_BIT == 4 ? 2 : 1;
#endif
    int ao[NPAR][2][4], aol[NPAR][2][4];
    const int plane_delta = (int)(Tl - Th);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            {   // a per-stage copy of the carried offset: what is derived from it below stays inside the stage (not hoisted, not spilled)
                int v = la.ao0[j][rt];
                asm volatile("" : "+v"(v));
                ao[0][j][rt] = v;
            }
            if (NPAR == 2) ao[NPAR - 1][j][rt] = ao[0][j][rt] ^ 4;
#pragma unroll
            for (int par = 0; par < NPAR; ++par) {
                int v = ao[par][j][rt] + plane_delta;
                asm volatile("" : "+v"(v));
                aol[par][j][rt] = v;
            }
        }
    half8 ah[4], al[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        ah[rt] = frag_read(Th, ao[0][0][rt], ao[0][1][rt]);
        al[rt] = frag_read(Th, aol[0][0][rt], aol[0][1][rt]);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if (ks + PF < KS) {
            r.q[(ks + PF) % (PF + 1)][0] = wf.load(ctu, ks + PF, KS, 0);
            r.q[(ks + PF) % (PF + 1)][1] = wf.load(ctu, ks + PF, KS, 1);
        }
        const half8 bh = __builtin_bit_cast(half8, r.q[ks % (PF + 1)][0]);
        const half8 bl = __builtin_bit_cast(half8, r.q[ks % (PF + 1)][1]);
        hook(ks);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[rt] = mfma16(ah[rt], bh, acc[rt]);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            acc[rt] = mfma16(ah[rt], bl, acc[rt]);
            if (ks + 1 < KS) ah[rt] = frag_read(Th + (ks + 1) * 16 * PROW, ao[(ks + 1) & (NPAR - 1)][0][rt], ao[(ks + 1) & (NPAR - 1)][1][rt]);
        }
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            acc[rt] = mfma16(al[rt], bh, acc[rt]);
            if (ks + 1 < KS) al[rt] = frag_read(Th + (ks + 1) * 16 * PROW, aol[(ks + 1) & (NPAR - 1)][0][rt], aol[(ks + 1) & (NPAR - 1)][1][rt]);
        }
        __builtin_amdgcn_sched_barrier(0);          // one k-step per scheduling region: keeps the prefetch distances as written
    }
}

template <int KS, int PF = BWS_PF>
__device__ __forceinline__ void gemm3(const _Float16* __restrict__ Th, const _Float16* __restrict__ Tl, const float* __restrict__ wp, int ct,
                                      int lane, f32x16 (&acc)[4], const LaneAddr& la) {
    WRing<PF> r;
    gemm3_head<KS, PF>(wp, ct, lane, r);
    gemm3_body<KS, PF>(Th, Tl, wp, ct, lane, r, acc, la);
}

// One row tile x one column tile: out += (Th + Tl)[rt*32.., 0 .. KS*16) x W(tile).  Three MFMAs per k-step on ONE accumulator
// (dependent: the pipe idles between them), so only for the small dPE blocks; fragments PF k-steps ahead.
template <int KS, int PF = 4>
__device__ __forceinline__ void gemm_row3(const _Float16* __restrict__ Th, const _Float16* __restrict__ Tl, const float* __restrict__ wp,
                                          int tile, int rt, int lane, f32x16& out) {
    lane = stage_local(lane);
    const int o0 = frag_off(lane, 0, rt), o1 = frag_off(lane, 1, rt);
    int ol0 = o0 + (int)(Tl - Th), ol1 = o1 + (int)(Tl - Th);      // lo plane through the hi plane's pointer: see gemm3_body
    asm volatile("" : "+v"(ol0), "+v"(ol1));
#ifdef BWS_OLD_SWIZZLE
    constexpr int ODD = 0;
#else
    constexpr int ODD = BWS_HALF_BIT == 4 ? 4 : 0;      // fidx: odd k-steps (features with bit 4 set) have the halves of a slot swapped
#endif
    const WFrag wf(wp, lane);
    const int tu = __builtin_amdgcn_readfirstlane(tile);
    u32x4 bq[PF + 1][2];
#pragma unroll
    for (int p = 0; p < PF; ++p)
        if (p < KS) {
            bq[p][0] = wf.load(tu, p, KS, 0);
            bq[p][1] = wf.load(tu, p, KS, 1);
        }
    half8 ahn = frag_read(Th, o0, o1);
    half8 aln = frag_read(Th, ol0, ol1);
    // two partial accumulators (even / odd k-steps) halve the dependent chain
    f32x16 o2;
#pragma unroll
    for (int e = 0; e < 16; ++e) o2[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const half8 ah = ahn, al = aln;
        if (ks + PF < KS) {
            bq[(ks + PF) % (PF + 1)][0] = wf.load(tu, ks + PF, KS, 0);
            bq[(ks + PF) % (PF + 1)][1] = wf.load(tu, ks + PF, KS, 1);
        }
        if (ks + 1 < KS) {
            const int x = ((ks + 1) & 1) ? ODD : 0;
            ahn = frag_read(Th + (ks + 1) * 16 * PROW, o0 ^ x, o1 ^ x);
            aln = frag_read(Th + (ks + 1) * 16 * PROW, ol0 ^ x, ol1 ^ x);
        }
        const half8 bh = __builtin_bit_cast(half8, bq[ks % (PF + 1)][0]);
        const half8 bl = __builtin_bit_cast(half8, bq[ks % (PF + 1)][1]);
        if (ks & 1) {
            o2 = mfma16(ah, bh, o2);
            o2 = mfma16(ah, bl, o2);
            o2 = mfma16(al, bh, o2);
        } else {
            out = mfma16(ah, bh, out);
            out = mfma16(ah, bl, out);
            out = mfma16(al, bh, out);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) out[e] += o2[e];
}

__device__ __forceinline__ void zero4(f32x16 (&acc)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;
}

// bit k (compile-time) of a 32-bit mask word as an all-ones / all-zeros mask (one v_bfe_i32)
__device__ __forceinline__ uint32_t bit_mask32(uint32_t w, int k) {
    uint32_t m;     // asm: the compiler would turn "x & sbfe(...)" back into v_and + v_cmp + v_cndmask
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(w), "n"(k));
    return m;
}

// v -> (hi, lo) as packed f16 pairs: hi = rn16(v), lo = rn16(v - hi).  The low halves come from v_fma_mixlo_f16 / _mixhi_f16:
// fma(-hi (read as f16), 1.0, v) evaluated in f32 (v - hi is exact there) and rounded once into the low / high half of the
// result - two instructions per pair where "convert back, subtract, convert" takes four (round 5: the epilogue is over the
// issue budget of its K-loop, every instruction counts).  Same values bit for bit.
__device__ __forceinline__ void split2(float v0, float v1, half2v& hi, half2v& lo) {
    hi = __builtin_convertvector(float2v{v0, v1}, half2v);                          // v_cvt_pk_f16_f32 (RNE)
    uint32_t l;
    asm("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, -%1, 1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(l) : "v"(__builtin_bit_cast(uint32_t, hi)), "v"(v0), "v"(v1));
    lo = __builtin_bit_cast(half2v, l);
}
// amax = max(amax, |v0|, |v1|) in ONE instruction (the compiler's IEEE-mode fmaxf chain: v_max_f32 |v0|, |v1| + v_max_f32)
__device__ __forceinline__ void amax3(float& amax, float v0, float v1) {
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(v0), "v"(v1));
}

// The STORE half of an epilogue - lane exchange, rescale, residual codes, two stores per unit: ~34 of a unit's instructions, 40 % of the
// epilogue's VALU work - needs nothing but the packed (hi, lo) quads.  For the row tiles an epilogue finishes LAST (2 and 3) those quads
// wait in registers (the dX kernel has them to spare, the forward does not) and the units are formed in the NEXT stage's K-loop, one
// per k-step in its last k-steps, in the shadow of its MFMAs (the VALU is idle there; BWS_DEFER, BWS_DEFER_RT).
// First deferred row tile: the units of row tiles BWS_DEFER_RT .. 3 wait (16 registers per row tile).  2 (the first version: 32 registers, four
// units in k-steps 12..15): -1.7 % of the kernel; 1: +-0 on top of it; 0 - ALL eight units, k-steps 8..15, the epilogue stores nothing itself -
// another -1.8 % alone and -4 % inside the step (C2 7.38 -> 7.31 ms; profiles/r05_dx_deferred_units_ab.log).  64 registers of quads next to the
// accumulators fit only with the weight-fragment ring two k-steps deep (BWS_PF = 2: 253 VGPRs, no scratch; three deep: two values spilled across
// the layer loops) - the depth itself measures +-0.
#ifndef BWS_DEFER_RT
#define BWS_DEFER_RT 0
#endif
constexpr int NDEFER = 2 * (4 - BWS_DEFER_RT);     // deferred units per stage, one per k-step in the next K-loop's last NDEFER k-steps
struct DeferUnits {
    uint2 qh[4 - BWS_DEFER_RT][2][2], ql[4 - BWS_DEFER_RT][2][2];     // [row tile - BWS_DEFER_RT][ep][h]
};
#ifndef BWS_DEFER
#define BWS_DEFER 1
#endif
__device__ __forceinline__ void store_unit(u32x4 hi, uint2 code, __amdgpu_buffer_rsrc_t rs_hi, __amdgpu_buffer_rsrc_t rs_lo, int st_lane, int rt, int ep) {
#ifndef BWS_SKIP_STORE      // timing variants only (tools/experiments/build_variant.sh)
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b128(hi, rs_hi, st_lane + (rt * 4 + ep * 2) * 256 * 8 * 2, 0, BWS_ST_AUX);
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{code.x, code.y}, rs_lo, st_lane / 2 + (rt * 4 + ep * 2) * 256 * 8, 0, BWS_ST_AUX);
#endif
}
// the VALU half of a deferred unit: quads -> rescaled hi halves + residual codes
__device__ __forceinline__ void form_unit(const uint2 (&qh)[2], const uint2 (&ql)[2], float gf, u32x4& hi, uint2& code) {
    const _Float16 gh = (_Float16)gf, gl = (_Float16)(gf * 4096.f);
    const half2v g2 = {gh, gh}, g2l = {gl, gl};
    const uint4 uh = sh_pair_unit(qh[0], qh[1]);
    const uint4 ul = sh_pair_unit(ql[0], ql[1]);
    const uint32_t uhw[4] = {uh.x, uh.y, uh.z, uh.w}, ulw[4] = {ul.x, ul.y, ul.z, ul.w};
    uint32_t oh[4], ol[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        oh[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2v, uhw[i]) * g2);
        ol[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2v, ulw[i]) * g2l);
    }
    code = h8_encode_unit<12>(oh, ol);
    hi = u32x4{oh[0], oh[1], oh[2], oh[3]};
}
// one unit (row tile rt, block pair ep) of a 256-wide SH gradient array from its two quads: see epilogue3
__device__ __forceinline__ void finish_unit(const uint2 (&qh)[2], const uint2 (&ql)[2], __amdgpu_buffer_rsrc_t rs_hi, __amdgpu_buffer_rsrc_t rs_lo,
                                            int st_lane, int rt, int ep, float gf) {
    const _Float16 gh = (_Float16)gf, gl = (_Float16)(gf * 4096.f);
    const half2v g2 = {gh, gh}, g2l = {gl, gl};
    const uint4 uh = sh_pair_unit(qh[0], qh[1]);
    const uint4 ul = sh_pair_unit(ql[0], ql[1]);
    const uint32_t uhw[4] = {uh.x, uh.y, uh.z, uh.w}, ulw[4] = {ul.x, ul.y, ul.z, ul.w};
    uint32_t oh[4], ol[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        oh[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2v, uhw[i]) * g2);
        ol[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2v, ulw[i]) * g2l);
    }
    const uint2 code = h8_encode_unit<12>(oh, ol);
    store_unit(u32x4{oh[0], oh[1], oh[2], oh[3]}, code, rs_hi, rs_lo, st_lane, rt, ep);
}
// byte offset of this lane's unit (block lane >> 5 of a pair, feature n) in the tile's part of a 256-wide SH array
__device__ __forceinline__ int unit_lane_offset(int ct, int lane) { return (((lane >> 5) * 256 + ct * 32 + (lane & 31)) * 8) * 2; }

// dY = acc masked by the forward pass' ReLU sign bits (bits[rt]: the 32 points of row tile rt for THIS lane's feature, shifted
// right by 4 (lane >> 5): bit 8 (e >> 2) + (e & 3) <-> accumulator element e; mlp_split.h: sp_mask_word) -> both planes (hi, lo;
// tile scale) and, rescaled by gf = s_call / s_tile (a power of two <= 1; exact), the SH gradient arrays `st_hi` / `st_lo` of
// width 256 (tile part).
// STORE = false: planes only (d feature: the dW kernels do not need it, mlp_common.h: DWS_*)
// NRT row tiles starting at row tile rt0 (wave-uniform; P1: the views layer's 128-wide stage runs as column tile w & 3 x point half
// w >> 2), W: width of the SH arrays `st_hi` / `st_lo` (their tile part: block 0 = the tile's first 8 points).
template <bool MASK, bool STORE = true, int NRT = 4, int W = 256, bool USE_LA = false, bool DEFER = false>
__device__ __forceinline__ void epilogue3(f32x16 (&acc)[NRT], const uint32_t (&bits)[NRT], _Float16* __restrict__ Th, _Float16* __restrict__ Tl,
                                          int ct, int lane, const _Float16* __restrict__ st_hi, const uint8_t* __restrict__ st_lo, float gf,
                                          float& amax, int rt0, const LaneAddr& la, DeferUnits* du = nullptr) {
    static_assert(!DEFER || (NRT == 4 && W == 256 && STORE), "deferred units: the 256-wide stages");
    lane = stage_local(lane);
    const int lr = lane & 31, r4 = 4 * (lane >> 5);
    const __amdgpu_buffer_rsrc_t rs_hi = uniform_rsrc(st_hi);         // this tile's 16 blocks of the SH arrays (64 KiB each)
    const __amdgpu_buffer_rsrc_t rs_lo = uniform_rsrc(st_lo);
    const _Float16 gh = (_Float16)gf;             // a power of two (or 0 below 2^-24: such a tile's gradients are below f16 anyway)
    const half2v g2 = {gh, gh};
    const _Float16 gl = (_Float16)(gf * 4096.f);  // the residual goes into the encoder as lo * gf * 2^12 (h8_encode_unit<12>)
    const half2v g2l = {gl, gl};
    const int n = ct * 32 + lr;
    // plane offset of this lane's quad (4 points r4 .. r4 + 3 of an 8-point block) in row tile 0, block 0; row tile rt and block q
    // enter through the swizzle: + (((rt ^ (n & 3)) << 5) | ((q ^ ((n >> 2) & 3)) << 3))
#ifdef BWS_OLD_SWIZZLE
    const int tq = n * PROW + r4;
#else
    const int tq = n * PROW + (r4 ^ (((n >> BWS_HALF_BIT) & 1) << 2));      // fidx: the slot's halves swapped for features with that bit set
#endif
    const int sw_seg = n & 3, sw_slot = (n >> 2) & 3;
    const int st_lane = ((((lane >> 5) + rt0 * 4) * W + n) * 8) * 2;   // byte offset of unit (block, n); lanes 32-63: the odd block of a pair
    // the 256-wide stages: quad offsets from the carried pieces (LaneAddr), through per-stage copies so that the 16 sums are not hoisted
    int qb[4] = {0, 0, 0, 0};
    if (USE_LA) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int v = la.b[q];
            asm volatile("" : "+v"(v));
            qb[q] = v;
        }
    }
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int ep = 0; ep < 2; ++ep) {
            uint2 qh[2], ql[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint32_t wh[2], wl[2];
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    float v[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int e = (ep * 2 + h) * 4 + jp * 2 + t;
                        v[t] = acc[rt][e];
                        if (MASK) v[t] = __uint_as_float(__float_as_uint(v[t]) & bit_mask32(bits[rt], 8 * (e >> 2) + (e & 3)));
                    }
                    amax3(amax, v[0], v[1]);
                    half2v hv, lv;
                    split2(v[0], v[1], hv, lv);
                    wh[jp] = __builtin_bit_cast(uint32_t, hv);
                    wl[jp] = __builtin_bit_cast(uint32_t, lv);
                }
                {   // this quad (block q = 2 ep + h of row tile rt) -> both planes, 8 bytes each
                    const int o = USE_LA ? la.ta[rt] + qb[ep * 2 + h]
                                         : tq + ((((NRT == 4 ? rt : rt0 + rt) ^ sw_seg) << 5) | ((((ep * 2 + h) ^ sw_slot)) << 3));
                    *reinterpret_cast<uint2*>(Th + o) = uint2{wh[0], wh[1]};
#ifndef BWS_SKIP_PLANE_LO
                    *reinterpret_cast<uint2*>(Tl + o) = uint2{wl[0], wl[1]};
#endif
                }
                qh[h] = uint2{wh[0], wh[1]};
                ql[h] = uint2{wl[0], wl[1]};
            }
            if (!STORE) continue;
            if (DEFER && rt >= BWS_DEFER_RT) {     // the quads wait for the next K-loop (DeferUnits)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    du->qh[rt - BWS_DEFER_RT][ep][h] = qh[h];
                    du->ql[rt - BWS_DEFER_RT][ep][h] = ql[h];
                }
                continue;
            }
            // lanes exchange halves, then the exact rescale (v_pk_mul_f16 by a power of two) and the residual codes
            const uint4 uh = sh_pair_unit(qh[0], qh[1]);
            const uint4 ul = sh_pair_unit(ql[0], ql[1]);
            const uint32_t uhw[4] = {uh.x, uh.y, uh.z, uh.w}, ulw[4] = {ul.x, ul.y, ul.z, ul.w};
            uint32_t oh[4], ol[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                oh[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2v, uhw[i]) * g2);
                ol[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2v, ulw[i]) * g2l);
            }
            const uint2 code = h8_encode_unit<12>(oh, ol);
            // vector offset + zero scalar offset (mlp_bwd_h.hip: the scalar-offset form of a 16-byte store reads its data late)
#ifndef BWS_SKIP_STORE      // timing variants only (tools/experiments/build_variant.sh)
            typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{oh[0], oh[1], oh[2], oh[3]}, rs_hi, st_lane + (rt * 4 + ep * 2) * W * 8 * 2, 0, BWS_ST_AUX);
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{code.x, code.y}, rs_lo, st_lane / 2 + (rt * 4 + ep * 2) * W * 8, 0, BWS_ST_AUX);
#endif
        }
}

template <int C>
__global__ __launch_bounds__(BNT, 2) void mlp_bwd_split_kernel(BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) _Float16 Tsm[];   // Th | Tl
    _Float16* Th = Tsm;
    _Float16* Tl = Tsm + TMB * LD;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7, wave-uniform: weight pointers stay scalar
    const int64_t m0 = (int64_t)blockIdx.x * TMB;
    const int64_t M = a.M;
    const float* acts = a.acts;
    float* dacts = a.dacts;
    const float* packed_h = a.packed + PACKED_FLOATS;
    const int ct = wave;                                         // this wave's column tile in the 256-wide stages
    const int64_t Mp = m_pad(M);
    uint8_t* st8 = reinterpret_cast<uint8_t*>(dacts + sdact_lo8_base(Mp));      // lo8 region: byte i <-> half i of the SH region
    // ReLU sign-bit words.  h0..h7 (round 5, mlp_split.h): uint32 [layer][tile 128][row tile 4][column tile 8][32], written by the
    // forward's scalar stores; this lane's feature ct * 32 + (lane & 31) is word sp_mask_word(lane & 31) of block (rt, ct), and the
    // 16 points of its accumulator elements are bits 8 (e >> 2) + 4 (lane >> 5) + (e & 3): shifted down by 4 (lane >> 5) once.
    // hv (mask layer 8) keeps the uint64 [64-point tile][4 x 64 threads] format of mlp_common.h (P1 below).
    const __amdgpu_buffer_rsrc_t mask_rsrc =
        uniform_rsrc(reinterpret_cast<const uint64_t*>(acts + sact_mask(Mp)) + (int64_t)blockIdx.x * 2 * NTHREADS);
    const int mask_stride_b = (int)((Mp / TM) * NTHREADS * 8);           // bytes between layers (< 2^31 up to 8M points)
    auto load_bits = [&](int layer, uint32_t (&b)[4]) {
        const int so = __builtin_amdgcn_readfirstlane(layer * mask_stride_b);
        const int vo = (wave * 32 + sp_mask_word(lane & 31)) * 4;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) b[rt] = __builtin_amdgcn_raw_buffer_load_b32(mask_rsrc, vo + rt * 8 * 32 * 4, so, BWS_ST_AUX);
    };
    auto shift_bits = [&](uint32_t (&b)[4]) {      // at the point of use: the loads are requested a stage ahead
        const int sh = 4 * (lane >> 5);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) b[rt] >>= sh;
    };
    _Float16* st_dyh = reinterpret_cast<_Float16*>(dacts + sdact_h(Mp, 0));      // layer l: + l * Mp * 256 halfs
    // timing variant (-DBWS_STORE_WINDOW=256): every tile stores into the first WINDOW points - same instructions, bytes stay in L2
#ifdef BWS_STORE_WINDOW
    const int64_t ms0 = m0 & (int64_t)(BWS_STORE_WINDOW - 1);
#else
    const int64_t ms0 = m0;
#endif
    auto st_tile = [&](int l) { return st_dyh + ((int64_t)l * Mp + ms0) * 256; };   // tile's part of layer l's SH array (hi)
    auto st8_tile = [&](int l) { return st8 + ((int64_t)l * Mp + ms0) * 256; };      // ... and of its lo8 twin
    float s_g, inv_s_g;
    const float mx_call = a.absmax ? *a.absmax : dacts[sdact_info(Mp) + SD_DRAW];
    if (a.absmax && blockIdx.x == 0 && tid == 0) dacts[sdact_info(Mp) + SD_DRAW] = mx_call;   // the dW reduce reads it there
    pow2_scale6(mx_call, s_g, inv_s_g);                                          // scale of the dY arrays of this call
    if (a.status && blockIdx.x == 0 && tid == 0 && reinterpret_cast<const uint32_t*>(acts + sact_info(Mp))[SI_TAG] != SACT_TAG_SPLIT22)
        a.status[2] = 1u;
    float amax = 0.f;          // max |tile-scaled gradient| of this thread before its f16 split (range guard)
    TR(0);
#ifdef BENERF_TRACE_DX
    if (tid == 0 && blockIdx.x < 2048) { unsigned hw, xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); reinterpret_cast<unsigned long long*>(a.d_vdir)[blockIdx.x * 64 + 63] = ((unsigned long long)xcc << 32) | hw; }
#endif

    // ---- P0: d_raw tile, its power-of-two scale, scaled values -> scratch floats [28, 28+C] of each row ------
    float dr0[C + 1];
    if (tid < TMB) {
        const int64_t m = m0 + tid;
        float mx = 0.f;
#pragma unroll
        for (int c = 0; c <= C; ++c) {
            dr0[c] = m < M ? a.d_raw[m * (C + 1) + c] : 0.f;
            mx = fmaxf(mx, fabsf(dr0[c]));
        }
        mx = wave_max_nonneg(mx);                                           // lane 63 of waves 0 and 1
        if (lane == 63) *fscr1(Th, wave, 26) = mx;
    }
    lds_barrier();
    float s, inv_s;
    pow2_scale6(fmaxf(*fscr1(Th, 0, 26), *fscr1(Th, 1, 26)), s, inv_s);     // 2^(6 - exponent(max)), exact inverse
    if (tid < TMB) {
#pragma unroll
        for (int c = 0; c <= C; ++c) draw_cm(Tl, c)[tid] = dr0[c] * s;
    }
    lds_barrier();
    const float gf = s_g * inv_s;   // tile scale -> scale of the stored dY (power of two <= 1)
    TR(1);

    // ---- P1: rgb layer backward + ReLU mask of the views layer -> dYv in planes[:, 0:128) ----
    // wave w: column tile w & 3, point half w >> 2 (row tiles 2 (w >> 2), + 1) - the split forward's VIEWS mapping, so the hv sign
    // bits it saved (bit b*4 + j = point 8b + 4 (lane >> 5) + j of the half = accumulator element (rt & 1) * 16 + e) line up
    {
        const int vct = wave & 3, vrh = wave >> 2;
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 hvw = __builtin_amdgcn_raw_buffer_load_b64(mask_rsrc, (vct * 64 + lane) * 8 + vrh * NTHREADS * 8,
                                                               __builtin_amdgcn_readfirstlane(8 * mask_stride_b), 0);
        const int col = vct * 32 + (lane & 31), r4 = 4 * (lane >> 5);
        float wr[C];
#pragma unroll
        for (int c = 0; c < C; ++c) wr[c] = a.w_rgb[c * 128 + col];
        // g = d_rgb . w_rgb[:, col] for this lane's 2 x 16 points, in accumulator order (element e of row tile rt = point
        // rt * 32 + 8 (e >> 2) + r4 + (e & 3): a quad = 4 consecutive points = one 16-byte read per channel)
        f32x16 g2[2];
#pragma unroll
        for (int rtl = 0; rtl < 2; ++rtl)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int p0 = (vrh * 2 + rtl) * 32 + 8 * q + r4;
                float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float4 d = *reinterpret_cast<const float4*>(draw_cm(Tl, c) + p0);
                    g.x += d.x * wr[c];
                    g.y += d.y * wr[c];
                    g.z += d.z * wr[c];
                    g.w += d.w * wr[c];
                }
                g2[rtl][q * 4 + 0] = g.x;
                g2[rtl][q * 4 + 1] = g.y;
                g2[rtl][q * 4 + 2] = g.z;
                g2[rtl][q * 4 + 3] = g.w;
            }
        // the forward's hv sign bits (bit 16 rtl + e of this thread's word) in epilogue3's order (bit 8 (e >> 2) + (e & 3))
        uint32_t hb[2];
#pragma unroll
        for (int rtl = 0; rtl < 2; ++rtl) {
            const uint32_t x = hvw[0] >> (16 * rtl);
            hb[rtl] = (x & 0xFu) | ((x & 0xF0u) << 4) | ((x & 0xF00u) << 8) | ((x & 0xF000u) << 12);
        }
        epilogue3<true, true, 2, ACT_HV_W>(g2, hb, Th, Tl, vct, lane, reinterpret_cast<_Float16*>(dacts + sdact_hv(Mp)) + ms0 * ACT_HV_W,
                                           st8 + 2 * sdact_hv(Mp) + ms0 * ACT_HV_W, gf, amax, vrh * 2, LaneAddr());
    }
    lds_barrier();
    TR(2);

    f32x16 acc[4];
    uint32_t bits[4];
    LaneAddr la;
    {
        const int ln = stage_local(lane);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) la.ao0[j][rt] = frag_off(ln, j, rt);
        const int n = ct * 32 + (ln & 31), r4 = 4 * (ln >> 5);
#ifdef BWS_OLD_SWIZZLE
        const int tq = n * PROW + r4;
#else
        const int tq = n * PROW + (r4 ^ (((n >> BWS_HALF_BIT) & 1) << 2));
#endif
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            la.ta[k] = tq + ((k ^ (n & 3)) << 5);
            la.b[k] = (k ^ ((n >> 2) & 3)) << 3;
        }
    }

    // ---- P2: VIEWS^T and FEAT^T in ONE stage (round 5): dh7 = dYv x W_c + d_sigma w_alpha, W_c = W_v[:, :256] W_f (mlp_common.h:
    // PB_VIEWSC - the feature layer feeds the views layer without a ReLU, so its gradient never has to exist); dPE(dir) = dYv x
    // W_v[:, 256:283]; mask h7 -> dY7 ----------------------------------------------------------------------------------------
    load_bits(7, bits);
    if (wave < 4) {   // dPE(dir): tile 8 of the block, row tile = wave -> scratch floats [0,27)
        f32x16 ap;
#pragma unroll
        for (int e = 0; e < 16; ++e) ap[e] = 0.f;
        gemm_row3<8>(Th, Tl, packed_h + pack_offset(PB_VIEWSC), 8, wave, lane, ap);
        const int ln = stage_local(lane);
        if ((ln & 31) < 27) {
#pragma unroll
            for (int e = 0; e < 16; ++e) *fscr1(Th, wave * 32 + acc_row(e, ln), ln & 31) = ap[e];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    zero4(acc);
    gemm3<8>(Th, Tl, packed_h + pack_offset(PB_VIEWSC), ct, lane, acc, la);
    {
        const float wa = a.w_alpha[ct * 32 + (lane & 31)];
        const int r4 = 4 * (lane >> 5);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 ds = *reinterpret_cast<const float4*>(draw_cm(Tl, C) + rt * 32 + 8 * q + r4);
                acc[rt][q * 4 + 0] += ds.x * wa;
                acc[rt][q * 4 + 1] += ds.y * wa;
                acc[rt][q * 4 + 2] += ds.z * wa;
                acc[rt][q * 4 + 3] += ds.w * wa;
            }
    }
    lds_barrier();   // dYv fully consumed; dPE(dir) visible
    TR(3);
    if (tid < TMB && m0 + tid < M) {   // d viewdirs (per point) through PE(dir): sin / cos recomputed from the saved direction
        const int64_t m = m0 + tid;
        const float4 vd4 = reinterpret_cast<const float4*>(acts + sact22_pts(Mp))[m * 2 + 1];
        const float vd[3] = {vd4.x, vd4.y, vd4.z};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float sv = *fscr1(Th, tid, d);
            if (a.pe_w) sv *= a.pe_w[64 + d];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const int es = 3 + f * 6 + d, ec = es + 3;
                float sn, cs;
                pe_sincos(vd[d] * (float)(1 << f), sn, cs);       // the forward's own evaluation (mlp_fwd_h.hip): identical values
                const float ws = a.pe_w ? a.pe_w[64 + es] : 1.f, wc = a.pe_w ? a.pe_w[64 + ec] : 1.f;
                sv += (float)(1 << f) * (cs * (ws * *fscr1(Th, tid, es)) - sn * (wc * *fscr1(Th, tid, ec)));
            }
#ifndef BENERF_TRACE_DX
            a.d_vdir[m * 3 + d] = sv * inv_s;
#endif
        }
    }
    // Loads the next stage needs are requested BEFORE this stage's epilogue stores (in-order retirement): the sign bits of the
    // stage after (from HBM: a whole epilogue + K-loop of cover) and the first weight fragments.
    uint32_t bits_n[4];
    WRing<BWS_PF> ring;
    load_bits(6, bits_n);
    gemm3_head<16, BWS_PF>(packed_h + pack_offset(PB_L7), ct, lane, ring);
    shift_bits(bits);
    DeferUnits du;
    epilogue3<true, true, 4, 256, true, BWS_DEFER != 0>(acc, bits, Th, Tl, ct, lane, st_tile(7), st8_tile(7), gf, amax, 0, la, &du);
    lds_barrier();
    TR(4);
#if defined(BWS_STOP_AFTER) && BWS_STOP_AFTER == 2     /* attribution of counters to phases (wrong results): stop behind P2 */
    return;
#endif

    // ---- P4: L7 .. L1: dY_l x W_l, mask h_{l-1} -> dY_{l-1} ----------------------------------------------
    auto layer = [&](int l) __attribute__((always_inline)) {
        zero4(acc);
        // dY_l's deferred units (row tiles BWS_DEFER_RT .. 3: all eight) are formed and stored in the last NDEFER k-steps of this K-loop, one
        // per k-step (the first of them in front of the K-loop's last fragment requests: measured, they do not hold the fragments up)
        const __amdgpu_buffer_rsrc_t drs_hi = uniform_rsrc(st_tile(l)), drs_lo = uniform_rsrc(st8_tile(l));
#if BWS_DEFER == 2      // VALU half early (k-steps 2, 5, 8, 11), stores in k-steps 12..15 (BWS_DEFER_RT == 2 only)
        static_assert(BWS_DEFER_RT == 2, "the early-VALU variant is written for four deferred units");
        u32x4 ph[4];
        uint2 pc[4];
        gemm3_body<16, BWS_PF>(Th, Tl, packed_h + bwd_layer_offset(l), ct, lane, ring, acc, la, [&](int ks) {
            if (ks < 12 && ks % 3 == 2) {
                const int u = ks / 3;
                form_unit(du.qh[u >> 1][u & 1], du.ql[u >> 1][u & 1], gf, ph[u], pc[u]);
            }
            if (ks >= 12) {
                const int u = ks - 12;
                store_unit(ph[u], pc[u], drs_hi, drs_lo, unit_lane_offset(ct, stage_local(lane)), 2 + (u >> 1), u & 1);
            }
        });
#else
        gemm3_body<16, BWS_PF>(Th, Tl, packed_h + bwd_layer_offset(l), ct, lane, ring, acc, la, [&](int ks) {
            if (BWS_DEFER && ks >= 16 - NDEFER) {
                const int u = ks - (16 - NDEFER);
                finish_unit(du.qh[u >> 1][u & 1], du.ql[u >> 1][u & 1], drs_hi, drs_lo, unit_lane_offset(ct, stage_local(lane)), BWS_DEFER_RT + (u >> 1), u & 1, gf);
            }
        });
#endif
        lds_barrier();
        TR(5 + 2 * (7 - l));
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) bits[rt] = bits_n[rt];
        shift_bits(bits);
        if (l >= 2) {
            load_bits(l - 2, bits_n);
            gemm3_head<16, BWS_PF>(packed_h + bwd_layer_offset(l - 1), ct, lane, ring);
        }
        epilogue3<true, true, 4, 256, true, BWS_DEFER != 0>(acc, bits, Th, Tl, ct, lane, st_tile(l - 1), st8_tile(l - 1), gf, amax, 0, la, &du);
        lds_barrier();
        TR(6 + 2 * (7 - l));
    };
#pragma unroll 1
    for (int l = 7; l >= 6; --l) layer(l);
    // The layer-5 skip's dPE block [row tile = wave & 3][column tile 8 + (wave >> 2)] = dY5 x W5[:, PE part] waits for layer 0's
    // part in the planes' PE columns [256,320) as hi + lo (tile scale; free from P3 on).
    const int prt = wave & 3, pct = wave >> 2;
    {
        f32x16 dpe;
#pragma unroll
        for (int e = 0; e < 16; ++e) dpe[e] = 0.f;
        gemm_row3<16>(Th, Tl, packed_h + pack_offset(PB_L5), 8 + pct, prt, lane, dpe);
        const int ln = stage_local(lane);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            amax = __builtin_fmaxf(amax, __builtin_fabsf(dpe[e]));
            const _Float16 vh = (_Float16)dpe[e];
            const int o = fidx(256 + pct * 32 + (ln & 31), prt * 32 + acc_row(e, ln));
            Th[o] = vh;
            Tl[o] = (_Float16)(dpe[e] - (float)vh);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll 1
    for (int l = 5; l >= 1; --l) layer(l);
#if defined(BWS_STOP_AFTER) && BWS_STOP_AFTER == 4     /* ... behind the layer loops */
    return;
#endif
    if (BWS_DEFER) {     // dY0's deferred units: no K-loop of this shape follows
        const __amdgpu_buffer_rsrc_t drs_hi = uniform_rsrc(st_tile(0)), drs_lo = uniform_rsrc(st8_tile(0));
#pragma unroll
        for (int u = 0; u < NDEFER; ++u)
            finish_unit(du.qh[u >> 1][u & 1], du.ql[u >> 1][u & 1], drs_hi, drs_lo, unit_lane_offset(ct, stage_local(lane)), BWS_DEFER_RT + (u >> 1), u & 1, gf);
    }

    if (a.status) {   // range guard of the f16 gradient halves: one atomic per wave, only near f16's maximum
        const float wmax = wave_max_nonneg(amax);
        if (lane == 63 && !(wmax < 32768.f)) atomicMax(a.status + 1, __float_as_uint(wmax == wmax ? wmax : __builtin_inff()));
    }

    // ---- P5: L0^T: dPE += dY0 x W0 -----------------------------------------------------------------------
    // The point itself (saved by the forward as f32, 16 bytes) is requested here, its HBM round trip hides under the L0 GEMM; P6
    // recomputes sin / cos from it (the forward's own pe_sincos: identical values) instead of reading 256 bytes of saved rows.
    const float4 x4 = reinterpret_cast<const float4*>(acts + sact22_pts(Mp))[(m0 + (tid & (TMB - 1))) * 2];
    f32x16 dpe;
    {
        const int ln = stage_local(lane);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int o = fidx(256 + pct * 32 + (ln & 31), prt * 32 + acc_row(e, ln));
            dpe[e] = (float)Th[o] + (float)Tl[o];
        }
    }
    TR(19);
    gemm_row3<16>(Th, Tl, packed_h + pack_offset(PB_L0), pct, prt, lane, dpe);
    lds_barrier();      // every wave is done reading dY0: the planes become f32 scratch, 77 floats per point:
    // [0,64) dPE, [64,73) the partial sums of the other three frequency groups.  The odd row stride keeps P6's per-point walks
    // (lane = point, same column) free of bank conflicts.
    float* F = reinterpret_cast<float*>(Tsm);
    constexpr int FLD = 77;
    static_assert((size_t)TMB * FLD * sizeof(float) <= BWD_SMEM, "P6 scratch fits the planes");
    {
        const int ln = stage_local(lane);
#pragma unroll
        for (int e = 0; e < 16; ++e) F[(prt * 32 + acc_row(e, ln)) * FLD + pct * 32 + (ln & 31)] = dpe[e];
    }
    lds_barrier();
    TR(20);

    // ---- P6: dPE -> d_pts; sin / cos recomputed from the point; four threads per point (frequencies g, g + 4, g + 8) ------------
    {
        const int pt = tid & (TMB - 1), g = tid >> 7;
        const int64_t m = m0 + pt;
        const float* dp = F + pt * FLD;
        const float x[3] = {x4.x, x4.y, x4.z};
        float sp[3] = {0.f, 0.f, 0.f};
        for (int f = g; f < 10; f += 4) {
            const float sc = (float)(1 << f);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int es = 3 + f * 6 + d, ec = es + 3;
                float sn, cs;
                pe_sincos(x[d] * sc, sn, cs);
                const float ws = a.pe_w ? a.pe_w[es] : 1.f, wc = a.pe_w ? a.pe_w[ec] : 1.f;
                sp[d] += sc * (cs * (ws * dp[es]) - sn * (wc * dp[ec]));
            }
        }
        // fixed summation order: ((identity + group 0) + group 1) + group 2) + group 3
        float* part = F + pt * FLD + 64;                          // floats [64,73) of the row: past the dPE block
        if (g > 0) {
#pragma unroll
            for (int d = 0; d < 3; ++d) part[(g - 1) * 3 + d] = sp[d];
        }
        lds_barrier();
        if (g == 0 && m < M) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float id = a.pe_w ? a.pe_w[d] * dp[d] : dp[d];
                a.d_pts[m * 3 + d] = ((((id + sp[d]) + part[d]) + part[3 + d]) + part[6 + d]) * inv_s;
            }
        }
    }
    TR(21);
}

// max |d_raw| -> dacts info word (zeroed by the launcher; non-negative floats order like their bit patterns)
__global__ void grad_absmax22_kernel(const float* __restrict__ d_raw, int64_t n, float* __restrict__ out) {
    float mx = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        mx = fmaxf(mx, fabsf(d_raw[i]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(mx));
}

}  // namespace

int benerf_mlp_dx_split22_launch(const BenerfMlpParams* params, const float* packed, int channels, int64_t M, const float* d_raw,
                                 const float* acts, float* dacts, float* d_pts, float* d_vdir_pts, uint32_t* status,
                                 const float* d_raw_absmax, hipStream_t stream) {
    BwdArgs a;
    a.d_raw = d_raw;
    a.acts = acts;
    a.dacts = dacts;
    a.packed = packed;
    a.w_alpha = params->w[BENERF_L_ALPHA];
    a.w_rgb = params->w[BENERF_L_RGB];
    a.pe_w = params->pe_weights;
    a.d_pts = d_pts;
    a.d_vdir = d_vdir_pts;
    a.status = status;
    a.absmax = d_raw_absmax;
    a.M = M;
    const int64_t tiles = mlp::m_pad(M) / TMB;
    BENERF_REQUIRE(tiles < (1ll << 31), "mlp_bwd: too many points");
    if (!d_raw_absmax) {    // nobody computed max |d_raw| for us: one pass over d_raw into the info word
        float* info = dacts + mlp::sdact_info(mlp::m_pad(M));
        if (hipMemsetAsync(info, 0, mlp::SD_COUNT * sizeof(float), stream) != hipSuccess) {
            benerf_set_error("mlp_bwd: memset failed");
            return BENERF_EHIP;
        }
        hipLaunchKernelGGL(grad_absmax22_kernel, dim3(256), dim3(256), 0, stream, d_raw, M * (channels + 1), info + mlp::SD_DRAW);
    }
    dim3 grid((unsigned)tiles), block(BNT);
    const int smem = (int)BWD_SMEM;
    static BenerfLdsAttr attr[2];       // once per device and variant
    if (!benerf_lds_attr(attr[channels == 1 ? 0 : 1],
                         channels == 1 ? (const void*)mlp_bwd_split_kernel<1> : (const void*)mlp_bwd_split_kernel<3>, smem)) {
        benerf_set_error("mlp_bwd(dx, split): cannot reserve %d bytes of LDS", smem);
        return BENERF_EHIP;
    }
    if (channels == 1) hipLaunchKernelGGL((mlp_bwd_split_kernel<1>), grid, block, smem, stream, a);
    else hipLaunchKernelGGL((mlp_bwd_split_kernel<3>), grid, block, smem, stream, a);
    BENERF_LAUNCH_CHECK("mlp_bwd(dx, split)");
    return BENERF_OK;
}

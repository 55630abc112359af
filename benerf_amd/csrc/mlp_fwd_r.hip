// K3 forward, split-f16 arithmetic, REGISTER-CHAIN schedule (experimental, inference launches; BENERF_FWD_R=1).
//
// Same arithmetic as mlp_fwd_h.hip (three f16 MFMAs per product block on hi/lo-split operands, the transposed product: weights
// as the MFMA A operand, a lane = one point), different data flow:
//   * the ACTIVATIONS never leave registers.  A wave owns 32 points for the whole network; lane (p = l & 31, g = l >> 5) holds, per
//     k-step, the 8 consecutive input features 16 ks + 8 g .. + 7 of its point as hi / lo f16 - exactly the MFMA B fragment.  The
//     accumulator of output tile t holds features 32 t + (r & 3) + 8 (r >> 2) + 4 g of that point; after the combine / ReLU / split
//     one v_permlane32_swap per register pair turns them into the B fragments of k-steps 2 t and 2 t + 1 of the next layer
//     (mlp_fwd_h.hip writes the same 16-byte slots to LDS planes instead).  No activation planes, no per-layer barriers.
//   * the WEIGHTS go through LDS, once per workgroup: the network is one stream of 32-KiB chunks (mlp_r.h: 4 output tiles x 4
//     k-steps x hi / lo) that the four waves copy with LDS-DMA into a ring of four slots, three chunks ahead; every wave reads every
//     fragment (ds_read_b128, lane-linear, conflict free).  One barrier per chunk orders the ring.
// One workgroup = 4 waves (one per SIMD, up to 512 registers each) = 128 points; LDS 128 KiB ring + 12 KiB of biases / head weights.
#include "mlp_r.h"
#include <stdlib.h>

namespace {
using namespace mlp;

struct FwdRArgs {
    const float* rays_o;
    const float* rays_d;
    const float* viewdirs;
    const float* z;
    const float* stream;     // weight stream (packed + 2 * PACKED_FLOATS)
    const float* bias[10];
    const float* w_alpha;
    const float* b_alpha;
    const float* w_rgb;
    const float* b_rgb;
    const float* pe_w;
    float* raw;
    uint32_t* status;
    int64_t M;
    int S;
};

typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int RT = 256;
constexpr int RING = 3, AHEAD = 2;
constexpr int TAB = RING * R_CHUNK_BYTES;                  // byte offset of the f32 table behind the ring
// table (floats): bias L0..L7 [8][256] | FEAT [256] | VIEWS [128] | w_alpha [256] | w_rgb [3][128]
constexpr int TB_FEAT = 2048, TB_VIEWS = 2304, TB_WALPHA = 2432, TB_WRGB = 2688, TB_FLOATS = 3072;
constexpr int PED = TAB + TB_FLOATS * 4;                    // PE(viewdir): f16 planes hi | lo [128 points][32]
constexpr int PEP = PED + 2 * 128 * 32 * 2;                 // PE(pts): f16 planes hi | lo [128 points][64]
constexpr size_t R_SMEM = (size_t)PEP + 2 * 128 * 64 * 2;  // 159 744 B

struct Ring {
    __amdgpu_buffer_rsrc_t rsrc;
    char* smem;
    int lane16, wave;
    int next;                // next chunk to request
    // the four waves split a chunk's 32 fragments: 8 LDS-DMA instructions each
    __device__ __forceinline__ void request() {
        char* slot = smem + (next % RING) * R_CHUNK_BYTES;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int f = wave * 8 + i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(slot + f * 1024), 16, lane16,
                                                     next * R_CHUNK_BYTES + f * 1024, 0, 0);
        }
        ++next;
    }
    // Chunk j has landed for every wave and the slot of chunk j - 1 is free: wait for this wave's own copies of chunk j (vector-
    // memory operations retire in order: at most the 8 * NEWER requests issued behind them may still be in flight), then the barrier.
    // NEWER = chunks requested behind j: 2 in steady state, fewer at the end of the stream.
    __device__ __forceinline__ void acquire(int NEWER) {      // a constant after unrolling at every call site
        if (NEWER >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (NEWER == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
};

__device__ __forceinline__ f32x16 mfma16r(half8 a, half8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// Output tiles 4 grp .. 4 grp + 3 of one layer: acc1 += Whi x Xhi, acc2 += Wlo x Xhi + Whi x Xlo over KS4 chunks of 4 k-steps whose B
// fragments come from xh / xl (k-steps [0, KSH)) and from the staged encoding in LDS (k-steps [KSH, ...)).
// TAIL: the last chunks of the stream (nothing left to request behind them).
template <int KSH, int NCH, bool TAIL = false>
__device__ __forceinline__ void group_gemm(Ring& ring, int& j, const half8 (&xh)[16], const half8 (&xl)[16], const _Float16* __restrict__ pe_h,
                                           int pe_plane, f32x16 (&acc1)[4], f32x16 (&acc2)[4]) {
    // pe_h: this lane's first PE fragment in LDS (hi plane; lo plane pe_plane halfs behind, k-step stride 16 halfs) for k-steps >= KSH
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (TAIL) {
            ring.acquire(NCH - 1 - c < AHEAD - 1 ? NCH - 1 - c : AHEAD - 1);
            if (c + AHEAD < NCH) ring.request();
        } else {
            ring.acquire(AHEAD - 1);
            ring.request();
        }
        const char* slot = ring.smem + (j % RING) * R_CHUNK_BYTES + ring.lane16;
        // weight fragments one whole k-step ahead (two register sets): with one wave per SIMD nobody else covers an LDS round trip
        half8 w[2][8];
        auto load_w = [&](int set, int k4) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                w[set][2 * t + 0] = *reinterpret_cast<const half8*>(slot + ((t * 4 + k4) * 2 + 0) * 1024);
                w[set][2 * t + 1] = *reinterpret_cast<const half8*>(slot + ((t * 4 + k4) * 2 + 1) * 1024);
            }
        };
        load_w(0, 0);
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const int ks = 4 * c + k4, cur = k4 & 1;
            if (k4 < 3) load_w(cur ^ 1, k4 + 1);
            half8 bh, bl;
            if (ks < KSH) {
                bh = xh[ks < KSH ? ks : 0];
                bl = xl[ks < KSH ? ks : 0];
            } else {
                bh = *reinterpret_cast<const half8*>(pe_h + (ks - KSH) * 16);
                bl = *reinterpret_cast<const half8*>(pe_h + pe_plane + (ks - KSH) * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc1[t] = mfma16r(w[cur][2 * t], bh, acc1[t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc2[t] = mfma16r(w[cur][2 * t + 1], bh, acc2[t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc2[t] = mfma16r(w[cur][2 * t], bl, acc2[t]);
            __builtin_amdgcn_sched_barrier(0);
        }
        ++j;
    }
}

// accumulators start at the bias (hi x hi set) / zero; element r of tile t <-> feature 32 t + (r & 3) + 8 (r >> 2) + 4 g
__device__ __forceinline__ void acc_init(f32x16 (&acc1)[4], f32x16 (&acc2)[4], const float* __restrict__ bias_lds, int tile0, int g) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 b = *reinterpret_cast<const float4*>(bias_lds + (tile0 + t) * 32 + 8 * q + 4 * g);
            acc1[t][q * 4 + 0] = b.x;
            acc1[t][q * 4 + 1] = b.y;
            acc1[t][q * 4 + 2] = b.z;
            acc1[t][q * 4 + 3] = b.w;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[t][e] = 0.f;
    }
}

// combine (+ ReLU), split into hi / scaled lo, and hand the lane pair's halves around: tile t -> B fragments of k-steps 2 t, 2 t + 1
template <bool RELU>
__device__ __forceinline__ void group_epilogue(const f32x16 (&acc1)[4], const f32x16 (&acc2)[4], int tile0, half8 (&oh)[16], half8 (&ol)[16],
                                               float& amax) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        uint2 qh[4], ql[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t wh[2], wl[2];
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                const int e = q * 4 + jp * 2;
                float2v v = {acc1[t][e] + acc2[t][e] * LO_INV, acc1[t][e + 1] + acc2[t][e + 1] * LO_INV};
                if (RELU) {
                    v[0] = fmaxf(v[0], 0.f);
                    v[1] = fmaxf(v[1], 0.f);
                    amax = fmaxf(amax, fmaxf(v[0], v[1]));
                } else {
                    amax = fmaxf(amax, fmaxf(fabsf(v[0]), fabsf(v[1])));
                }
                const half2v hi = __builtin_convertvector(v, half2v);
                const float2v res = (v - __builtin_convertvector(hi, float2v)) * LO_SCALE;
                wh[jp] = __builtin_bit_cast(uint32_t, hi);
                wl[jp] = __builtin_bit_cast(uint32_t, __builtin_convertvector(res, half2v));
            }
            qh[q] = uint2{wh[0], wh[1]};
            ql[q] = uint2{wl[0], wl[1]};
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {   // lanes 0-31 end with features 32 t + 16 i + 0..7, lanes 32-63 with + 8..15: k-step 2 t + i, group g
            const uint4 uh = sh_pair_unit(qh[2 * i], qh[2 * i + 1]);
            const uint4 ul = sh_pair_unit(ql[2 * i], ql[2 * i + 1]);
            oh[2 * (tile0 + t) + i] = __builtin_bit_cast(half8, u32x4{uh.x, uh.y, uh.z, uh.w});
            ol[2 * (tile0 + t) + i] = __builtin_bit_cast(half8, u32x4{ul.x, ul.y, ul.z, ul.w});
        }
    }
}

// one layer with 8 output tiles: x (+ p) -> o
template <int KSH, int NCH, bool RELU>
__device__ __forceinline__ void layer8(Ring& ring, int& j, const float* __restrict__ bias_lds, int g, const half8 (&xh)[16],
                                       const half8 (&xl)[16], const _Float16* __restrict__ pe_h, int pe_plane, half8 (&oh)[16], half8 (&ol)[16],
                                       float& amax) {
    f32x16 acc1[4], acc2[4];
#pragma unroll
    for (int grp = 0; grp < 2; ++grp) {
        acc_init(acc1, acc2, bias_lds, 4 * grp, g);
        group_gemm<KSH, NCH>(ring, j, xh, xl, pe_h, pe_plane, acc1, acc2);
        group_epilogue<RELU>(acc1, acc2, 4 * grp, oh, ol, amax);
    }
}

__device__ __forceinline__ void split8(const float (&v)[8], half8& hi, half8& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        hi[i] = (_Float16)v[i];
        lo[i] = (_Float16)((v[i] - (float)hi[i]) * LO_SCALE);
    }
}

template <int C>
__global__ __launch_bounds__(RT, 1) void mlp_fwd_r_kernel(FwdRArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5;
    const int64_t M = a.M;
    const int64_t m = (int64_t)blockIdx.x * 128 + wave * 32 + (lane & 31);
    float* tab = reinterpret_cast<float*>(smem + TAB);

    Ring ring;
    {
        const uint64_t wa = reinterpret_cast<uint64_t>(a.stream);
        const uint64_t wau = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(wa >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)wa);
        ring.rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(wau), 0, 0x7fffffff, 0x00020000);
    }
    ring.smem = smem;
    ring.lane16 = lane * 16;
    ring.wave = wave;
    ring.next = 0;

    // ---- table: biases and head weights -> LDS (plain loads, drained before the first LDS-DMA request) ------------------------
    for (int i = tid; i < TB_FLOATS; i += RT) {
        float v;
        if (i < 2048) v = a.bias[i >> 8][i & 255];
        else if (i < TB_VIEWS) v = a.bias[BENERF_L_FEAT][i - TB_FEAT];
        else if (i < TB_WALPHA) v = a.bias[BENERF_L_VIEWS][i - TB_VIEWS];
        else if (i < TB_WRGB) v = a.w_alpha[i - TB_WALPHA];
        else v = (i - TB_WRGB) < C * 128 ? a.w_rgb[i - TB_WRGB] : 0.f;
        tab[i] = v;
    }
    // ---- positional encodings, computed like mlp_fwd_h.hip (two threads per point share the frequencies, each sincos gives a sin
    // and a cos column) and staged as f16 planes behind the table: PE(viewdir) hi | lo [128][32], PE(pts) hi | lo [128][64].  The layers that
    // consume an encoding (L0, L5, VIEWS) read their B fragments from there (16 bytes per lane and k-step).
    float amax = 0.f;
    {
        const int pt = tid & 127, grp = tid >> 7;
        const int64_t mp = (int64_t)blockIdx.x * 128 + pt;
        const int64_t mpc = mp < M ? mp : M - 1;
        const int64_t ray = (int64_t)((uint32_t)mpc / (uint32_t)a.S);
        const float zz = a.z[mpc];
        float x[3], vd[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            x[c] = __fadd_rn(a.rays_o[ray * 3 + c], __fmul_rn(a.rays_d[ray * 3 + c], zz));      // separately rounded like torch
            vd[c] = a.viewdirs[ray * 3 + c];
        }
        _Float16* sh = reinterpret_cast<_Float16*>(smem + PEP) + pt * 64;
        _Float16* sl = sh + 128 * 64;
        _Float16* dh = reinterpret_cast<_Float16*>(smem + PED) + pt * 32;
        _Float16* dl = dh + 128 * 32;
        auto put = [&](_Float16* hi_p, _Float16* lo_p, int col, float v, const float* w) {
            if (w) v *= w[col];
            const _Float16 hi = (_Float16)v;
            hi_p[col] = hi;
            lo_p[col] = (_Float16)((v - (float)hi) * LO_SCALE);
        };
        if (grp == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                put(sh, sl, c, x[c], a.pe_w);
                put(dh, dl, c, vd[c], a.pe_w ? a.pe_w + 64 : nullptr);
                amax = fmaxf(amax, fabsf(x[c]));
            }
            put(sh, sl, 63, 0.f, nullptr);
        } else {
#pragma unroll
            for (int k = 27; k < 32; ++k) put(dh, dl, k, 0.f, nullptr);
        }
        for (int q = grp; q < 30; q += 2) {                    // model/embedder.py:13-28
            const int f = q / 3, d = q - 3 * f;
            float sn, cs;
            pe_sincos(x[d] * (float)(1 << f), sn, cs);
            put(sh, sl, 3 + f * 6 + d, sn, a.pe_w);
            put(sh, sl, 3 + f * 6 + 3 + d, cs, a.pe_w);
        }
        for (int q = grp; q < 12; q += 2) {
            const int f = q / 3, d = q - 3 * f;
            float sn, cs;
            pe_sincos(vd[d] * (float)(1 << f), sn, cs);
            put(dh, dl, 3 + f * 6 + d, sn, a.pe_w ? a.pe_w + 64 : nullptr);
            put(dh, dl, 3 + f * 6 + 3 + d, cs, a.pe_w ? a.pe_w + 64 : nullptr);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // no plain load in flight when the LDS-DMA requests start
    __builtin_amdgcn_s_barrier();                                  // table and staged encodings visible to every wave
#pragma unroll
    for (int i = 0; i < AHEAD; ++i) ring.request();
    const int P = wave * 32 + (lane & 31);                         // this lane's point inside the tile (both halves g of the pair)
    const _Float16* pe_pts = reinterpret_cast<const _Float16*>(smem + PEP) + P * 64 + 8 * g;     // lo plane: + 128 * 64
    const _Float16* pe_dir = reinterpret_cast<const _Float16*>(smem + PED) + P * 32 + 8 * g;     // lo plane: + 128 * 32

    half8 ah[16], al[16], bh[16], bl[16];
    int j = 0;
    // L0: PE -> a
    layer8<0, 1, true>(ring, j, tab + 0 * 256, g, ah, al, pe_pts, 128 * 64, ah, al, amax);     // x operands unused for KSH = 0
    // L1..L4, L6, L7 as three passes over ONE pair body (a -> b -> a); L5 ([h4 | PE] -> b, copied back to a) sits in front of the third
#pragma unroll 1
    for (int pi = 0; pi < 3; ++pi) {
        int l = 1 + 2 * pi;
        if (pi == 2) {
            layer8<16, 5, true>(ring, j, tab + 5 * 256, g, ah, al, pe_pts, 128 * 64, bh, bl, amax);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                ah[i] = bh[i];
                al[i] = bl[i];
            }
            l = 6;
        }
        layer8<16, 4, true>(ring, j, tab + l * 256, g, ah, al, pe_pts, 128 * 64, bh, bl, amax);
        layer8<16, 4, true>(ring, j, tab + (l + 1) * 256, g, bh, bl, pe_pts, 128 * 64, ah, al, amax);
    }

    // ---- alpha head on h7 (this lane: features 16 ks + 8 g + i), partner lane holds the other half --------------------------------
    float sigma;
    {
        float s = 0.f;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const float4 w0 = *reinterpret_cast<const float4*>(tab + TB_WALPHA + 16 * ks + 8 * g);
            const float4 w1 = *reinterpret_cast<const float4*>(tab + TB_WALPHA + 16 * ks + 8 * g + 4);
            const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) s += ((float)ah[ks][i] + (float)al[ks][i] * LO_INV) * w[i];
        }
        s += __shfl_xor(s, 32, 64);
        sigma = s + a.b_alpha[0];
    }
    // FEAT (linear): a -> b
    layer8<16, 4, false>(ring, j, tab + TB_FEAT, g, ah, al, pe_dir, 128 * 32, bh, bl, amax);
    // VIEWS: [feature | PE(dir)] -> 128, ReLU: b (+ p) -> a[0..7]
    {
        f32x16 acc1[4], acc2[4];
        acc_init(acc1, acc2, tab + TB_VIEWS, 0, g);
        group_gemm<16, 5, true>(ring, j, bh, bl, pe_dir, 128 * 32, acc1, acc2);
        group_epilogue<true>(acc1, acc2, 0, ah, al, amax);
    }
    // ---- rgb head: 128 -> C ----------------------------------------------------------------------------------------------------------
    float rgb[C];
#pragma unroll
    for (int c = 0; c < C; ++c) rgb[c] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        float h[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) h[i] = (float)ah[ks][i] + (float)al[ks][i] * LO_INV;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float4 w0 = *reinterpret_cast<const float4*>(tab + TB_WRGB + c * 128 + 16 * ks + 8 * g);
            const float4 w1 = *reinterpret_cast<const float4*>(tab + TB_WRGB + c * 128 + 16 * ks + 8 * g + 4);
            rgb[c] += h[0] * w0.x + h[1] * w0.y + h[2] * w0.z + h[3] * w0.w + h[4] * w1.x + h[5] * w1.y + h[6] * w1.z + h[7] * w1.w;
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) rgb[c] += __shfl_xor(rgb[c], 32, 64);
    if (g == 0 && m < M) {
#pragma unroll
        for (int c = 0; c < C; ++c) a.raw[m * (C + 1) + c] = rgb[c] + a.b_rgb[c];
        a.raw[m * (C + 1) + C] = sigma;
    }
    if (a.status) {
        const float wmax = wave_max_nonneg(amax);
        if (lane == 63 && !(wmax < 32768.f)) {
            const uint32_t bits = __float_as_uint(wmax == wmax ? wmax : __builtin_inff());
            atomicMax(a.status, bits);
            atomicMax(a.status + 3, bits);
        }
    }
}

}  // namespace

// inference launches only (acts == NULL); returns BENERF_OK, or 1 when the experimental schedule is not selected
int benerf_mlp_fwd_r_launch(const BenerfMlpParams* params, const float* packed, int channels, int n_rays, int n_samples,
                            const float* rays_o, const float* rays_d, const float* viewdirs, const float* z, float* raw, uint32_t* status,
                            hipStream_t stream) {
    static const bool on = [] { const char* e = getenv("BENERF_FWD_R"); return e && e[0] == '1'; }();
    if (!on) return 1;
    FwdRArgs a;
    a.rays_o = rays_o;
    a.rays_d = rays_d;
    a.viewdirs = viewdirs;
    a.z = z;
    a.stream = packed + 2 * mlp::PACKED_FLOATS;
    for (int l = 0; l < 8; ++l) a.bias[l] = params->b[l];
    a.bias[BENERF_L_VIEWS] = params->b[BENERF_L_VIEWS];
    a.bias[BENERF_L_FEAT] = params->b[BENERF_L_FEAT];
    a.w_alpha = params->w[BENERF_L_ALPHA];
    a.b_alpha = params->b[BENERF_L_ALPHA];
    a.w_rgb = params->w[BENERF_L_RGB];
    a.b_rgb = params->b[BENERF_L_RGB];
    a.pe_w = params->pe_weights;
    a.raw = raw;
    a.status = status;
    a.M = (int64_t)n_rays * n_samples;
    a.S = n_samples;
    const int64_t tiles = (a.M + 127) / 128;
    BENERF_REQUIRE(tiles < (1ll << 31) && a.M < (1ll << 31), "mlp_fwd(r): too many points");
    const int smem = (int)R_SMEM;
    static const bool lds_ok[2] = {
        hipFuncSetAttribute((const void*)mlp_fwd_r_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) == hipSuccess,
        hipFuncSetAttribute((const void*)mlp_fwd_r_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) == hipSuccess};
    if (!lds_ok[channels == 1 ? 0 : 1]) {
        benerf_set_error("mlp_fwd(r): cannot reserve %d bytes of LDS", smem);
        return BENERF_EHIP;
    }
    if (channels == 1) hipLaunchKernelGGL((mlp_fwd_r_kernel<1>), dim3((unsigned)tiles), dim3(RT), smem, stream, a);
    else hipLaunchKernelGGL((mlp_fwd_r_kernel<3>), dim3((unsigned)tiles), dim3(RT), smem, stream, a);
    BENERF_LAUNCH_CHECK("mlp_fwd(r)");
    return BENERF_OK;
}

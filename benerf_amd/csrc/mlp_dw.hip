// K3 backward, part 2: weight gradients dW_l = dY_l^T X_l summed over ALL sample points,
// exact-f32 MFMA, + bias gradients (column sums) and the two tiny heads (alpha, rgb) on the
// VALU.  The reduction dimension is the point index (hundreds of thousands), so every GEMM
// instance is split dw_splits(inst) ways along it (cost-proportional); each workgroup keeps its whole output block
// (up to 256x256 = 128 accumulator registers per lane across 8 waves, 2 waves per SIMD) in registers while it
// streams 32-point chunks of dY and X through LDS (register-staged prefetch of the next
// chunk during the MFMAs).  Partials are then summed in a fixed order by dw_reduce_kernel,
// which also scatters into nn.Linear layout ([out,in], model/nerf.py:53-64) -> deterministic.
#include "mlp_split.h"

namespace {
using namespace mlp;

constexpr int CH = 32;   // points per chunk (64 measured slower: 6.3 vs 5.8 ms at M = 522k)

struct DwArgs {
    const float* d_raw;
    const float* acts;
    const float* dacts;
    float* ws;
    int64_t M;
    int C;
};

struct InstSrc {
    const float* dy;   // [M][n]
    const float* x;    // [M][k]
    bool bias;
};

__device__ __forceinline__ InstSrc inst_src(const DwArgs& a, int inst) {
    const int64_t M = a.M;
    switch (inst) {
        case DW_L1: return {a.dacts + dact_h(M, 1), a.acts + act_h(M, 0), true};
        case DW_L2: return {a.dacts + dact_h(M, 2), a.acts + act_h(M, 1), true};
        case DW_L3: return {a.dacts + dact_h(M, 3), a.acts + act_h(M, 2), true};
        case DW_L4: return {a.dacts + dact_h(M, 4), a.acts + act_h(M, 3), true};
        case DW_L5H: return {a.dacts + dact_h(M, 5), a.acts + act_h(M, 4), true};
        case DW_L6: return {a.dacts + dact_h(M, 6), a.acts + act_h(M, 5), true};
        case DW_L7: return {a.dacts + dact_h(M, 7), a.acts + act_h(M, 6), true};
        case DW_FEAT: return {a.dacts + dact_feat(M), a.acts + act_h(M, 7), true};
        case DW_VIEWSF: return {a.dacts + dact_hv(M), a.acts + act_feat(M), true};
        case DW_L0: return {a.dacts + dact_h(M, 0), a.acts + act_pe(M), true};
        case DW_L5P: return {a.dacts + dact_h(M, 5), a.acts + act_pe(M), false};
        default: return {a.dacts + dact_hv(M), a.acts + act_ped(M), false};   // DW_VIEWSP
    }
}

constexpr int DWT = 512;   // 8 waves = 2 per SIMD: one wave's LDS staging / VALU sums overlap the other's MFMAs

// Output block N x K (N = WR*32).  The 8 waves form a WR x (8/WR) grid: wave w owns row tile
// (w % WR) and NCT = max(1, K / (32 * 8/WR)) column tiles starting at (w / WR) * NCT.
template <int N, int K, int WR, bool ALPHA>
__device__ __forceinline__ void dw_gemm(const DwArgs& a, const InstSrc src, int64_t chunk_begin, int64_t chunk_end,
                                        float* __restrict__ part, float* __restrict__ smem) {
    static_assert(N == WR * 32, "row tiles");
    constexpr int WC = 8 / WR;
    constexpr int NCT = (K / 32 >= WC) ? K / 32 / WC : 1;
    constexpr int YQ = CH * N / 4, XQ = CH * K / 4;                       // float4 slots per chunk
    constexpr int NY4 = (YQ + DWT - 1) / DWT, NX4 = (XQ + DWT - 1) / DWT;  // float4 loads per thread per chunk
    constexpr bool XFULL = XQ % DWT == 0;                                  // the 128x32 block fills only half the threads
    static_assert(YQ % DWT == 0, "staging shape");
    float* Ys = smem;                 // [CH][N]
    float* Xs = smem + CH * N;        // [CH][K]
    float* da = Xs + CH * K;          // [CH] d_sigma of the chunk (ALPHA only)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int wr = wave % WR, wc = wave / WR;
    const bool mma_wave = wc * NCT * 32 < K;      // waves beyond the block's columns only help staging
    const int64_t M = a.M;

    f32x16 acc[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
    float bsum = 0.f, asum = 0.f, absum = 0.f;

    float4 ry[NY4], rx[NX4];
    float rda = 0.f;
    auto prefetch = [&](int64_t chunk) {
        const int64_t row0 = chunk * CH;
        const float4* py = reinterpret_cast<const float4*>(src.dy + row0 * N) + tid;
        const float4* px = reinterpret_cast<const float4*>(src.x + row0 * K) + tid;
        if (row0 + CH <= M) {   // block-uniform fast path: plain back-to-back loads
#pragma unroll
            for (int j = 0; j < NY4; ++j) ry[j] = py[j * DWT];
#pragma unroll
            for (int j = 0; j < NX4; ++j) {
                rx[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (XFULL || tid + j * DWT < XQ) rx[j] = px[j * DWT];
            }
        } else {                // ragged last chunk: rows >= M contribute zero
#pragma unroll
            for (int j = 0; j < NY4; ++j) {
                const int64_t row = row0 + ((tid + j * DWT) * 4) / N;
                ry[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < M) ry[j] = py[j * DWT];
            }
#pragma unroll
            for (int j = 0; j < NX4; ++j) {
                const int64_t row = row0 + ((tid + j * DWT) * 4) / K;
                rx[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < M && (XFULL || tid + j * DWT < XQ)) rx[j] = px[j * DWT];
            }
        }
        if (ALPHA && tid < CH) rda = row0 + tid < M ? a.d_raw[(row0 + tid) * (a.C + 1) + a.C] : 0.f;
    };

    if (chunk_begin < chunk_end) prefetch(chunk_begin);
    for (int64_t chunk = chunk_begin; chunk < chunk_end; ++chunk) {
        __syncthreads();   // previous chunk fully consumed
#pragma unroll
        for (int j = 0; j < NY4; ++j) *reinterpret_cast<float4*>(Ys + (tid + j * DWT) * 4) = ry[j];
#pragma unroll
        for (int j = 0; j < NX4; ++j)
            if (XFULL || tid + j * DWT < XQ) *reinterpret_cast<float4*>(Xs + (tid + j * DWT) * 4) = rx[j];
        if (ALPHA && tid < CH) da[tid] = rda;
        __syncthreads();
        if (chunk + 1 < chunk_end) prefetch(chunk + 1);

        if (mma_wave) {
            const float* yp = Ys + lh * N + wr * 32 + lr;
            const float* xp = Xs + lh * K + wc * NCT * 32 + lr;
#pragma unroll 4
            for (int pp = 0; pp < CH; pp += 2) {
                const float av = yp[pp * N];
                float bv[NCT];
#pragma unroll
                for (int c = 0; c < NCT; ++c) bv[c] = xp[pp * K + c * 32];
#pragma unroll
                for (int c = 0; c < NCT; ++c) acc[c] = mfma32(av, bv[c], acc[c]);
            }
        }
        if (src.bias && tid < N) {
            float s = 0.f;
#pragma unroll 8
            for (int p = 0; p < CH; ++p) s += Ys[p * N + tid];
            bsum += s;
        }
        if (ALPHA && tid < K) {   // K == 256
            float s = 0.f, sb = 0.f;
#pragma unroll 8
            for (int p = 0; p < CH; ++p) {
                s += da[p] * Xs[p * K + tid];
                sb += da[p];
            }
            asum += s;
            absum += sb;
        }
    }

    // partial block -> workspace: [N][K] then bias [N] (then alpha row [256] + alpha bias)
    if (mma_wave) {
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = wr * 32 + acc_row(e, lane);
                part[(int64_t)row * K + (wc * NCT + c) * 32 + lr] = acc[c][e];
            }
    }
    if (tid < N) part[(int64_t)N * K + tid] = src.bias ? bsum : 0.f;
    if (ALPHA && tid < K) {
        part[(int64_t)N * K + N + tid] = asum;
        if (tid == 0) part[(int64_t)N * K + N + 256] = absum;
    }
}

// rgb head: dW_rgb[c][j] = sum_pt d_rgb[pt][c] * hv[pt][j], db_rgb[c] = sum_pt d_rgb[pt][c]
__device__ __forceinline__ void dw_rgb(const DwArgs& a, int64_t row_begin, int64_t row_end, float* __restrict__ part,
                                       float* __restrict__ smem) {
    const int tid = threadIdx.x, j = tid & 127, half = tid >> 7;   // half = row phase 0..3 (512 threads)
    const float* hv = a.acts + act_hv(a.M);
    const int C = a.C;
    // a workgroup's share is tens of thousands of rows per thread (7 point-splits): the running sums are kept in double and fed
    // with f32 partial sums of 8 rows, so the rounding of this head does not grow with the length of the share
    double s[3] = {0.0, 0.0, 0.0}, sb[3] = {0.0, 0.0, 0.0};
    // 8 rows in flight per thread (independent loads), fixed summation order
    constexpr int U = 8;
    int64_t mrow = row_begin + half;
    for (; mrow + 4 * (U - 1) < row_end; mrow += 4 * U) {
        float h[U], g[U][3];
#pragma unroll
        for (int q = 0; q < U; ++q) {
            h[q] = hv[(mrow + 4 * q) * ACT_HV_W + j];
#pragma unroll
            for (int c = 0; c < 3; ++c) g[q][c] = c < C ? a.d_raw[(mrow + 4 * q) * (C + 1) + c] : 0.f;
        }
        float ts[3] = {0.f, 0.f, 0.f}, tb[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < U; ++q)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                ts[c] += g[q][c] * h[q];
                tb[c] += g[q][c];
            }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            s[c] += (double)ts[c];
            sb[c] += (double)tb[c];
        }
    }
    for (; mrow < row_end; mrow += 4) {
        const float h = hv[mrow * ACT_HV_W + j];
#pragma unroll
        for (int c = 0; c < 3; ++c)
            if (c < C) {
                const float g = a.d_raw[mrow * (C + 1) + c];
                s[c] += g * h;
                sb[c] += g;
            }
    }
    if (half > 0) {   // phases 1..3 -> LDS [phase-1][{w,b}][3][128]
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            smem[((half - 1) * 6 + c) * 128 + j] = (float)s[c];
            smem[((half - 1) * 6 + 3 + c) * 128 + j] = (float)sb[c];
        }
    }
    __syncthreads();
    if (half == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = 0.f;
            if (c < 3 && c < C) v = (((float)s[c] + smem[(0 * 6 + c) * 128 + j]) + smem[(1 * 6 + c) * 128 + j]) + smem[(2 * 6 + c) * 128 + j];
            part[c * 128 + j] = v;
        }
        if (j < 4) {
            float v = 0.f;
            if (j < 3 && j < C) {
                const float own = (float)(j == 0 ? sb[0] : j == 1 ? sb[1] : sb[2]);
                v = ((own + smem[(0 * 6 + 3 + j) * 128 + j]) + smem[(1 * 6 + 3 + j) * 128 + j]) + smem[(2 * 6 + 3 + j) * 128 + j];
            }
            part[4 * 128 + j] = v;
        }
    }
}

// -DBENERF_TRACE_DW: thread 0 of every workgroup stamps the 100 MHz wall clock at its start and end behind the partial sums
// (u64 [512 workgroups][2] at ws + DW_WS_FLOATS; benerf_mlp_dw_workspace_floats grows by 4096 in such a build) -
// tools/experiments/trace_dw.py f32
#ifdef BENERF_TRACE_DW
#define DW32_TRACE(which) do { if (threadIdx.x == 0) reinterpret_cast<unsigned long long*>(a.ws + DW_WS_FLOATS)[blockIdx.x * 2 + (which)] = wall_clock64(); } while (0)
#else
#define DW32_TRACE(which) do { } while (0)
#endif

__global__ __launch_bounds__(DWT, 2) void mlp_dw_kernel(DwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    DW32_TRACE(0);
    // workgroup -> (instance, split); instances in cost order, split counts proportional to cost
    int split = blockIdx.x, inst = 0;
    while (split >= dw_splits(inst)) {
        split -= dw_splits(inst);
        ++inst;
    }
    const int64_t nchunks = (a.M + CH - 1) / CH;
    const int64_t per = (nchunks + dw_splits(inst) - 1) / dw_splits(inst);
    int64_t cb = (int64_t)split * per, ce = cb + per;
    if (cb > nchunks) cb = nchunks;
    if (ce > nchunks) ce = nchunks;
    float* part = a.ws + dw_inst_offset(inst) + (int64_t)split * dw_inst_floats(inst);
    if (inst == DW_RGB) {
        int64_t rb = cb * CH, re = ce * CH;
        if (re > a.M) re = a.M;
        if (rb > a.M) rb = a.M;
        dw_rgb(a, rb, re, part, smem);
        DW32_TRACE(1);
        return;
    }
    const InstSrc src = inst_src(a, inst);
    if (inst == DW_FEAT) dw_gemm<256, 256, 8, true>(a, src, cb, ce, part, smem);
    else if (inst <= DW_L7) dw_gemm<256, 256, 8, false>(a, src, cb, ce, part, smem);
    else if (inst == DW_VIEWSF) dw_gemm<128, 256, 4, false>(a, src, cb, ce, part, smem);
    else if (inst == DW_VIEWSP) dw_gemm<128, 32, 4, false>(a, src, cb, ce, part, smem);
    else dw_gemm<256, 64, 8, false>(a, src, cb, ce, part, smem);   // DW_L0, DW_L5P
    DW32_TRACE(1);
}

constexpr size_t DW_SMEM = (size_t)(CH * 256 + CH * 256 + CH) * sizeof(float);

// ---- fixed-order reduction over the splits + scatter into nn.Linear layout --------------------------
struct ReduceArgs {
    const float* ws;
    float* gw[BENERF_NLAYERS];
    float* gb[BENERF_NLAYERS];
    int C;
    int accumulate;
    int split_mode;          // 1: partials written by mlp_dw_h.hip / 2: by mlp_dw_s.hip (their split tables); dY-derived sums carry the factor s_s
    // split_mode 2 (BENERF_MLP_SPLIT): the feature / views weight gradients are composed from G = dhv^T h7 (mlp_common.h)
    const float* w_views;    // [128][283]
    const float* w_feat;     // [256][256]
    const float* b_feat;     // [256]
    float* gbuf;             // reduced G [128][256] + sum dhv [128] (unscaled), written by dw_reduce_kernel<2> (the feature layer's blocks)
    const float* grad_info;  // split mode: info words of the dY arrays ([SD_DRAW] = max |d_raw|, s_s derives from it)
    const float* pe_w;       // BARF c2f: the saved encodings are unweighted, so the PE columns of layers 0 / 5 / views are
                             // scaled here (dW[:, col] = w[col] * sum dY * PE[col]); null = no weighting
};

template <int SPLIT>     // 0: f32 kernels' split table, 1: mlp_dw_h.hip's, 2: mlp_dw_s.hip's
__device__ __forceinline__ float sum_splits_t(const float* ws, int inst, int64_t elem) {
    const float* p = ws + (SPLIT == 2 ? dws_inst_offset(inst) : SPLIT ? dwh_inst_offset(inst) : dw_inst_offset(inst)) + elem;
    const int64_t stride = dw_inst_floats(inst);
    const int n = SPLIT == 2 ? dws_splits(inst) : SPLIT ? dwh_splits(inst) : dw_splits(inst);
    // fixed summation order (run-to-run reproducible); eight loads in flight per step - the partial sums were just
    // written and sit in L2 / MALL, so this kernel is bound by the length of its dependent load chains
    float s = 0.f;
    int sp = 0;
    for (; sp + 8 <= n; sp += 8) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = p[(sp + i) * stride];
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    for (; sp < n; ++sp) s += p[sp * stride];
    return s;
}
// split mode: weight blocks and bias sums carry the gradient scale s_s of the call (mlp_split.h, pow2_scale6); the
// alpha / rgb heads are computed from the unscaled d_raw
template <int SPLIT>
__device__ __forceinline__ float unscale_of(const float* grad_info, int inst) {
    if (!SPLIT || inst == DW_RGB) return 1.f;
    float s_s, inv_s_s;
    pow2_scale6(grad_info[SD_DRAW], s_s, inv_s_s);
    return inv_s_s;
}
#define sum_splits(ws, inst, elem) (sum_splits_t<SPLIT>(ws, inst, elem) * unscale_of<SPLIT>(a.grad_info, inst))
#define sum_bias(ws, inst, elem) (sum_splits_t<SPLIT>(ws, inst, elem) * unscale_of<SPLIT>(a.grad_info, inst))
#define sum_raw(ws, inst, elem) sum_splits_t<SPLIT>(ws, inst, elem)

__device__ __forceinline__ int layer_inst(int l) {   // instance holding the bias / main block of layer l
    switch (l) {
        case 0: return DW_L0;
        case 5: return DW_L5H;
        case 6: return DW_L6;
        case 7: return DW_L7;
        case BENERF_L_VIEWS: return DW_VIEWSF;
        case BENERF_L_FEAT: return DW_FEAT;
        case BENERF_L_ALPHA: return DW_FEAT;
        case BENERF_L_RGB: return DW_RGB;
        default: return DW_L1 + (l - 1);
    }
}

// BENERF_MLP_SPLIT: the two small GEMMs that turn G = dhv^T h7 into weight gradients (mlp_common.h: DWS_*), 32 x 32 output tiles
// through LDS, fixed summation order (k ascending), f32 FMAs:
//   blocks [0, 64):   dW_f[n][j]      = sum_i W_v[i][n] G[i][j]                          (256 x 256, K = 128)
//   blocks [64, 96):  dW_v[n][j<256]  = sum_k G[n][k] W_f[j][k] + (sum dhv)[n] b_f[j]    (128 x 256, K = 256)
//   block 96:         db_f[n] = sum_i W_v[i][n] (sum dhv)[i];   db_v[n] = (sum dhv)[n]
__global__ __launch_bounds__(256) void dw_compose_kernel(ReduceArgs a) {
    // Round 5 (end): the whole contraction range of a tile goes into LDS in ONE load phase (every global load of the block in flight
    // at once) instead of 4 / 8 chunks of 32 with two barriers each - the kernel was a chain of global-load latencies (26 us for
    // 25 MFLOP).  Same products, same summation order (k ascending): bit-identical results.  Tiles [32][K] / [K][32] f32, K <= 256:
    // 64 KiB; the operand stored transposed (lanes along k) is XOR-swizzled by k & 31 inside its 32-wide rows.
    __shared__ float As[32 * 256], Bs[256 * 32];
    const float* G = a.gbuf;
    const float* bs = a.gbuf + 128 * 256;
    const int t = threadIdx.x, tx = t & 31, ty = t >> 5;
    const int b = blockIdx.x;
    if (b == 96) {
        float acc = 0.f;
        for (int i = 0; i < 128; ++i) acc = fmaf(a.w_views[i * 283 + t], bs[i], acc);      // t = n: coalesced rows of W_v
        float* d = a.gb[BENERF_L_FEAT] + t;
        *d = a.accumulate ? *d + acc : acc;
        if (t < 128) {
            float* dv = a.gb[BENERF_L_VIEWS] + t;
            *dv = a.accumulate ? *dv + bs[t] : bs[t];
        }
        return;
    }
    const bool feat = b < 64;
    const int bb = feat ? b : b - 64;
    const int m0 = (bb >> 3) * 32, n0 = (bb & 7) * 32, K = feat ? 128 : 256;
    // A tile as [m][k] (row stride K), B tile as [k][n] (row stride 32)
    for (int k0 = 0; k0 < K; k0 += 32) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = ty + 8 * r;
            if (feat) {     // A[m][k] = W_v[k][m] (lanes along m: transposed store, swizzled by m = tx), B[k][n] = G[k][n]
                As[tx * K + ((k0 + q) ^ tx)] = a.w_views[(k0 + q) * 283 + m0 + tx];
                Bs[(k0 + q) * 32 + tx] = G[(k0 + q) * 256 + n0 + tx];
            } else {        // A[m][k] = G[m][k], B[k][n] = W_f[n][k] (lanes along k: transposed store, swizzled by k & 31 = tx)
                As[q * K + k0 + tx] = G[(m0 + q) * 256 + k0 + tx];
                Bs[(k0 + tx) * 32 + (q ^ tx)] = a.w_feat[(n0 + q) * 256 + k0 + tx];
            }
        }
    }
    __syncthreads();
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (feat) {
#pragma unroll 8
        for (int k = 0; k < 128; ++k) {
            const float bv = Bs[k * 32 + tx];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = ty + 8 * r;
                acc[r] = fmaf(As[m * 128 + (k ^ m)], bv, acc[r]);
            }
        }
    } else {
#pragma unroll 8
        for (int k = 0; k < 256; ++k) {
            const float bv = Bs[k * 32 + (tx ^ (k & 31))];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = fmaf(As[(ty + 8 * r) * 256 + k], bv, acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + ty + 8 * r, n = n0 + tx;
        float v = acc[r];
        float* d;
        if (feat) d = a.gw[BENERF_L_FEAT] + m * 256 + n;
        else {
            v = fmaf(bs[m], a.b_feat[n], v);
            d = a.gw[BENERF_L_VIEWS] + m * 283 + n;
        }
        *d = a.accumulate ? *d + v : v;
    }
}

template <int SPLIT>
__global__ void dw_reduce_kernel(ReduceArgs a) {
    const int l = blockIdx.y;
    const int C = a.C;
    const int in = layer_in(l), out = layer_out(l, C);
    const int64_t nw = (int64_t)in * out;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nw + out; e += (int64_t)gridDim.x * blockDim.x) {
        float v;
        float* dst;
        if (SPLIT == 2 && (l == BENERF_L_FEAT || l == BENERF_L_VIEWS)) {
            // the feature layer and the first 256 columns + bias of the views layer are composed from G by dw_compose_kernel;
            // what is left here: the PE(dir) columns of the views layer - and, on the feature layer's otherwise idle blocks, the
            // reduction of G = dhv^T h7 + sum dhv over its splits -> gbuf (32 896 floats, fixed order, unscaled; round 5: this was
            // a launch of its own in front of this one)
            if (l == BENERF_L_FEAT) {
                if (e < 128 * 256 + 128) a.gbuf[e] = sum_splits(a.ws, DW_VIEWSF, e);
                continue;
            }
            if (e >= nw) continue;
            const int n = (int)(e / in), j = (int)(e % in);
            if (j < 256) continue;
            v = sum_splits(a.ws, DW_VIEWSP, (int64_t)n * 32 + (j - 256));
            if (a.pe_w) v *= a.pe_w[64 + (j - 256)];
            dst = a.gw[l] + e;
            *dst = a.accumulate ? *dst + v : v;
            continue;
        }
        if (e < nw) {
            const int n = (int)(e / in), j = (int)(e % in);
            dst = a.gw[l] + e;
            if (l == 0) v = sum_splits(a.ws, DW_L0, (int64_t)n * 64 + j);
            else if (l == 5) v = j < 63 ? sum_splits(a.ws, DW_L5P, (int64_t)n * 64 + j) : sum_splits(a.ws, DW_L5H, (int64_t)n * 256 + (j - 63));
            else if (l == BENERF_L_VIEWS) v = j < 256 ? sum_splits(a.ws, DW_VIEWSF, (int64_t)n * 256 + j) : sum_splits(a.ws, DW_VIEWSP, (int64_t)n * 32 + (j - 256));
            else if (l == BENERF_L_ALPHA) v = sum_raw(a.ws, DW_FEAT, (int64_t)256 * 256 + 256 + j);
            else if (l == BENERF_L_RGB) v = sum_splits(a.ws, DW_RGB, (int64_t)n * 128 + j);
            else v = sum_splits(a.ws, layer_inst(l), (int64_t)n * 256 + j);
            if (a.pe_w) {
                if ((l == 0 || l == 5) && j < 63) v *= a.pe_w[j];
                else if (l == BENERF_L_VIEWS && j >= 256) v *= a.pe_w[64 + (j - 256)];
            }
        } else {
            const int n = (int)(e - nw);
            dst = a.gb[l] + n;
            if (l == BENERF_L_ALPHA) v = sum_raw(a.ws, DW_FEAT, (int64_t)256 * 256 + 256 + 256);
            else if (l == BENERF_L_RGB) v = sum_raw(a.ws, DW_RGB, (int64_t)4 * 128 + n);
            else {
                const int inst = layer_inst(l);
                v = sum_bias(a.ws, inst, (int64_t)dw_shape(inst).n * dw_shape(inst).k + n);
            }
        }
        *dst = a.accumulate ? *dst + v : v;
    }
}
#undef sum_splits
#undef sum_bias
#undef sum_raw

}  // namespace

int benerf_mlp_dw_reduce_launch(const float* ws, const BenerfMlpGrads* grads, int channels, int accumulate, int split_mode,
                                const float* grad_info, const float* pe_weights, hipStream_t stream, const BenerfMlpParams* params) {
    ReduceArgs r;
    r.ws = ws;
    r.w_views = r.w_feat = r.b_feat = nullptr;
    r.gbuf = nullptr;
    if (split_mode == 2) {
        r.w_views = params->w[BENERF_L_VIEWS];
        r.w_feat = params->w[BENERF_L_FEAT];
        r.b_feat = params->b[BENERF_L_FEAT];
        // nothing of the FEAT instance's first split block but its alpha tail (floats [65 792, 66 049)) is written by the kernels
        r.gbuf = const_cast<float*>(ws) + dws_inst_offset(DW_FEAT);
    }
    for (int l = 0; l < BENERF_NLAYERS; ++l) {
        r.gw[l] = grads->w[l];
        r.gb[l] = grads->b[l];
    }
    r.C = channels;
    r.accumulate = accumulate;
    r.split_mode = split_mode;
    r.grad_info = grad_info;
    r.pe_w = pe_weights;
    if (split_mode == 2) {
        hipLaunchKernelGGL(dw_reduce_kernel<2>, dim3(128, BENERF_NLAYERS), dim3(256), 0, stream, r);      // incl. G -> gbuf
        hipLaunchKernelGGL(dw_compose_kernel, dim3(97), dim3(256), 0, stream, r);
    } else if (split_mode) hipLaunchKernelGGL(dw_reduce_kernel<1>, dim3(128, BENERF_NLAYERS), dim3(256), 0, stream, r);
    else hipLaunchKernelGGL(dw_reduce_kernel<0>, dim3(128, BENERF_NLAYERS), dim3(256), 0, stream, r);
    BENERF_LAUNCH_CHECK("mlp_bwd(reduce)");
    return BENERF_OK;
}

// split-f16 variant (mlp_dw_h.hip)
int benerf_mlp_dw_split_launch(int channels, int64_t M, const float* d_raw, const float* acts, const float* dacts, float* dw_ws,
                               const BenerfMlpGrads* grads, int accumulate, const float* pe_weights, hipStream_t stream);

// 22-bit variant (mlp_dw_s.hip)
int benerf_mlp_dw_split22_launch(const BenerfMlpParams* params, int channels, int64_t M, const float* d_raw, const float* acts,
                                 const float* dacts, float* dw_ws, const BenerfMlpGrads* grads, int accumulate, const float* pe_weights,
                                 hipStream_t stream);

int benerf_mlp_dw_launch(const BenerfMlpParams* params, int precision, int channels, int64_t M, const float* d_raw, const float* acts,
                         const float* dacts, float* dw_ws, const BenerfMlpGrads* grads, int accumulate,
                         const float* pe_weights, hipStream_t stream) {
    if (precision == BENERF_MLP_SPLIT)
        return benerf_mlp_dw_split22_launch(params, channels, M, d_raw, acts, dacts, dw_ws, grads, accumulate, pe_weights, stream);
    if (precision == BENERF_MLP_SPLIT_F16BWD)
        return benerf_mlp_dw_split_launch(channels, M, d_raw, acts, dacts, dw_ws, grads, accumulate, pe_weights, stream);
    DwArgs a;
    a.d_raw = d_raw;
    a.acts = acts;
    a.dacts = dacts;
    a.ws = dw_ws;
    a.M = M;
    a.C = channels;
    static BenerfLdsAttr attr;
    if (!benerf_lds_attr(attr, (const void*)mlp_dw_kernel, (int)DW_SMEM)) {
        benerf_set_error("mlp_bwd(dw): cannot reserve LDS");
        return BENERF_EHIP;
    }
    hipLaunchKernelGGL(mlp_dw_kernel, dim3(mlp::DW_TOTAL_BLOCKS), dim3(DWT), DW_SMEM, stream, a);
    BENERF_LAUNCH_CHECK("mlp_bwd(dw)");
    return benerf_mlp_dw_reduce_launch(dw_ws, grads, channels, accumulate, 0, nullptr, pe_weights, stream, nullptr);
}

// K7: event accumulation (polarity histogram), row gather; K8: fused Adam.
//
// K7 follows utils/event_utils.py:246-259 (sparse COO -> dense scatter-add: duplicates
// summed) and the window selection of model/nerf.py:162-178.  Polarities are +-1 so float
// atomic adds are exact and order independent.
// K8 follows torch.optim.Adam with default hyper-parameters (model/optimize.py:36-55,
// train.py:343-352).
#include "common.h"

namespace {

__global__ void event_accumulate_kernel(const int32_t* __restrict__ xs, const int32_t* __restrict__ ys,
                                        const float* __restrict__ ps, int64_t begin, int64_t end, int H, int W,
                                        float* __restrict__ out) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < end; e += stride) {
        int x = xs[e], y = ys[e];
        if (x >= 0 && x < W && y >= 0 && y < H) atomicAdd(out + (int64_t)y * W + x, ps[e]);
    }
}

// first index with ts[idx] >= v (lower) / first index with ts[idx] > v (upper)
__device__ int64_t bound(const double* ts, int64_t n, double v, bool upper) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        bool go = upper ? (ts[mid] <= v) : (ts[mid] < v);
        if (go) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

__global__ void event_window_accumulate_kernel(const int32_t* __restrict__ xs, const int32_t* __restrict__ ys,
                                               const float* __restrict__ ps, const double* __restrict__ ts, int64_t n,
                                               double low_t, double upper_t, int H, int W, float* __restrict__ out) {
    // low_t <= ts <= upper_t (both inclusive, model/nerf.py:170-172)
    int64_t begin = bound(ts, n, low_t, false);
    int64_t end = bound(ts, n, upper_t, true);
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < end; e += stride) {
        int x = xs[e], y = ys[e];
        if (x >= 0 && x < W && y >= 0 && y < H) atomicAdd(out + (int64_t)y * W + x, ps[e]);
    }
}

__global__ void gather_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx, int64_t n_idx,
                                   int width, float* __restrict__ out) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_idx * width) return;
    int64_t r = e / width;
    int c = (int)(e % width);
    out[e] = src[idx[r] * width + c];
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t n, float step_size, float omb1, float beta2, float omb2,
                            float eps, float bc2_sqrt, float grad_scale, const uint32_t* __restrict__ skip_if_range,
                            double lr, double beta1_d, double beta2_d, int step) {
    // status words of the split-f16 MLP mode (include/benerf_hip.h): a step whose activations or scaled gradients left
    // the f16 range carries inf / NaN gradients - leave parameters and moments untouched, the host reports it
    // ([4]: the verdict of benerf_step_gate for this step - identical on every data-parallel rank)
    if (skip_if_range && (skip_if_range[BENERF_ST_SKIP] != 0u || skip_if_range[BENERF_ST_ACT] >= 0x477fe000u ||
                          skip_if_range[BENERF_ST_GRAD] >= 0x477fe000u)) return;   // 65504.f
    // Bias correction counts APPLIED steps: steps the range guard skipped did not update the moments (torch's GradScaler does
    // not advance a skipped step either).  The host computed step_size / bc2_sqrt for t = step; with skipped steps on record
    // (rare) they are recomputed here for t = step - [SKIPPED_TOTAL].
    if (skip_if_range) {
        const uint32_t sk = skip_if_range[BENERF_ST_SKIPPED_TOTAL];
        if (sk != 0u) {
            const double t = (double)((int64_t)step - (int64_t)sk < 1 ? 1 : (int64_t)step - (int64_t)sk);
            step_size = (float)(lr / (1.0 - pow(beta1_d, t)));
            bc2_sqrt = (float)sqrt(1.0 - pow(beta2_d, t));
        }
    }
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float gi = g[i] * grad_scale;
        float mi = m[i] + (gi - m[i]) * omb1;                    // exp_avg.lerp_(grad, 1-beta1)
        float vi = v[i] * beta2 + (gi * gi) * omb2;               // mul_(beta2).addcmul_(g, g, 1-beta2)
        float denom = sqrtf(vi) / bc2_sqrt + eps;
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] - step_size * (mi / denom);                    // addcdiv_(m, denom, -step_size)
    }
}

// Range-guard bookkeeping of one training step (include/benerf_hip.h, benerf_step_gate).  One thread.
__global__ void step_gate_kernel(uint32_t* __restrict__ st, float* __restrict__ flag, int phase) {
    // the two per-step words hold max |d_raw| of the step's networks (benerf_composite_bwd; NaN arrives as +inf): a loss gradient that
    // is not finite - NaN / inf weights or inputs, in either arithmetic mode - must not reach the parameters either
    const bool nonfinite = st[BENERF_ST_STEP_SCRATCH] >= 0x7f800000u || st[BENERF_ST_STEP_SCRATCH + 1] >= 0x7f800000u;
    const bool local = st[BENERF_ST_ACT] >= 0x477fe000u || st[BENERF_ST_GRAD] >= 0x477fe000u || st[BENERF_ST_MODE] != 0u || nonfinite;
    if (phase == 0) {          // this rank's verdict as a float, to be SUMMED over the ranks with the gradients
        flag[0] = local ? 1.f : 0.f;
        return;
    }
    const bool skip = flag ? (flag[0] > 0.f || local) : local;
    if (local) {               // keep what tripped the guard for the host's message
        st[BENERF_ST_LAST_ACT] = st[BENERF_ST_ACT];
        st[BENERF_ST_LAST_GRAD] = nonfinite ? 0x7f800000u : st[BENERF_ST_GRAD];
    }
    st[BENERF_ST_SKIP] = skip ? 1u : 0u;
    st[BENERF_ST_SKIPPED] += skip ? 1u : 0u;
    st[BENERF_ST_SKIPPED_TOTAL] += skip ? 1u : 0u;
    st[BENERF_ST_CONSECUTIVE] = skip ? st[BENERF_ST_CONSECUTIVE] + 1u : 0u;
    if (st[BENERF_ST_CONSECUTIVE] > st[BENERF_ST_MAX_CONSECUTIVE]) st[BENERF_ST_MAX_CONSECUTIVE] = st[BENERF_ST_CONSECUTIVE];
    st[BENERF_ST_STEPS] += 1u;
    st[BENERF_ST_ACT] = 0u;    // the next step starts clean: one violation does not disable training for good
    st[BENERF_ST_GRAD] = 0u;
    st[BENERF_ST_MODE] = 0u;
    st[BENERF_ST_STEP_SCRATCH] = 0u;
    st[BENERF_ST_STEP_SCRATCH + 1] = 0u;
}

}  // namespace

extern "C" int benerf_step_gate(uint32_t* status, float* reduce_flag, int phase, benerf_stream_t stream) {
    BENERF_REQUIRE(status && (phase == 0 || phase == 1) && (phase == 1 || reduce_flag), "step_gate: bad args");
    hipLaunchKernelGGL(step_gate_kernel, dim3(1), dim3(1), 0, as_stream(stream), status, reduce_flag, phase);
    BENERF_LAUNCH_CHECK("step_gate");
    return BENERF_OK;
}

extern "C" int benerf_event_accumulate(const int32_t* xs, const int32_t* ys, const float* ps, int64_t n, int H, int W,
                                       float* out, benerf_stream_t stream) {
    BENERF_REQUIRE(out && H > 0 && W > 0 && n >= 0, "event_accumulate: bad args");
    if (n == 0) return BENERF_OK;
    BENERF_REQUIRE(xs && ys && ps, "event_accumulate: null event arrays");
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(event_accumulate_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), xs, ys, ps, (int64_t)0, n,
                       H, W, out);
    BENERF_LAUNCH_CHECK("event_accumulate");
    return BENERF_OK;
}

extern "C" int benerf_event_window_accumulate(const int32_t* xs, const int32_t* ys, const float* ps, const double* ts,
                                              int64_t n, double low_t, double upper_t, int H, int W, float* out,
                                              benerf_stream_t stream) {
    BENERF_REQUIRE(out && H > 0 && W > 0 && n >= 0, "event_window_accumulate: bad args");
    if (n == 0) return BENERF_OK;
    BENERF_REQUIRE(xs && ys && ps && ts, "event_window_accumulate: null event arrays");
    hipLaunchKernelGGL(event_window_accumulate_kernel, dim3(1024), dim3(256), 0, as_stream(stream), xs, ys, ps, ts, n,
                       low_t, upper_t, H, W, out);
    BENERF_LAUNCH_CHECK("event_window_accumulate");
    return BENERF_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Pixel selection without replacement (train.py:171-175, 296-299: np.random.choice(H*W, N_rand, replace=False)).
// A keyed pseudo-random BIJECTION of [0, n_total) - 4-round Feistel network on ceil(log2 n) bits (even), round
// function = Philox block, cycle-walking for values >= n_total - evaluated at 0..count-1: distinct indices in O(count)
// work, no sort, and a pure function of (seed, offset), so every data-parallel rank draws the same vector.
__device__ __forceinline__ uint32_t feistel_perm(uint32_t x, int half_bits, uint64_t seed, uint64_t offset) {
    const uint32_t mask = (1u << half_bits) - 1u;
    uint32_t l = x >> half_bits, r = x & mask;
#pragma unroll
    for (int round = 0; round < 4; ++round) {
        const uint4 h = philox4x32_10(make_uint4(r, (uint32_t)round, (uint32_t)offset, (uint32_t)(offset >> 32)),
                                      make_uint2((uint32_t)seed, (uint32_t)(seed >> 32) ^ 0x5bd1e995u));
        const uint32_t nl = r;
        r = (l ^ h.x) & mask;
        l = nl;
    }
    return (l << half_bits) | r;
}

__global__ void sample_pixels_kernel(int64_t n_total, int64_t count, int half_bits, uint64_t seed, uint64_t offset,
                                     int64_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t x = (uint32_t)i;
    do {
        x = feistel_perm(x, half_bits, seed, offset);     // the domain is < 4 n_total: a few walks at most
    } while ((int64_t)x >= n_total);
    out[i] = (int64_t)x;
}

extern "C" int benerf_sample_pixels(int64_t n_total, int64_t count, uint64_t seed, uint64_t offset, int64_t* out,
                                    benerf_stream_t stream) {
    BENERF_REQUIRE(out, "sample_pixels: null pointer");
    BENERF_REQUIRE(n_total > 0 && n_total < (1ll << 31) && count >= 0 && count <= n_total, "sample_pixels: bad sizes");
    if (count == 0) return BENERF_OK;
    int bits = 2;
    while ((1ll << bits) < n_total) bits += 2;
    hipLaunchKernelGGL(sample_pixels_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, as_stream(stream), n_total,
                       count, bits / 2, seed, offset, out);
    BENERF_LAUNCH_CHECK("sample_pixels");
    return BENERF_OK;
}

extern "C" int benerf_gather_rows(const float* src, const int64_t* idx, int64_t n_idx, int width, float* out,
                                  benerf_stream_t stream) {
    BENERF_REQUIRE(src && idx && out && width > 0 && n_idx >= 0, "gather_rows: bad args");
    if (n_idx == 0) return BENERF_OK;
    int64_t total = n_idx * width;
    int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), src, idx, n_idx, width, out);
    BENERF_LAUNCH_CHECK("gather_rows");
    return BENERF_OK;
}

extern "C" int benerf_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                                double beta1, double beta2, double eps, int step, double grad_scale,
                                const uint32_t* skip_if_range, benerf_stream_t stream) {
    BENERF_REQUIRE(param && grad && exp_avg && exp_avg_sq && n >= 0 && step >= 1, "adam_step: bad args");
    if (n == 0) return BENERF_OK;
    double bc1 = 1.0 - pow(beta1, (double)step);
    double bc2 = 1.0 - pow(beta2, (double)step);
    float step_size = (float)(lr / bc1);
    float bc2_sqrt = (float)sqrt(bc2);
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), param, grad, exp_avg, exp_avg_sq, n,
                       step_size, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, bc2_sqrt,
                       (float)grad_scale, skip_if_range, lr, beta1, beta2, step);
    BENERF_LAUNCH_CHECK("adam_step");
    return BENERF_OK;
}

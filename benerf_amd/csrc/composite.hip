// K4: alpha compositing, forward + backward.  One wavefront (64 lanes) per ray; each lane
// owns a contiguous run of IPL samples; transmittance is a wave-level exclusive product
// scan (lane-local sequential product + 6-step shuffle scan across lanes).
//
// Follows NeRF.raw2output, model/nerf.py:118-148:
//   dists = [z[1:]-z[:-1], 1e10] * ||rays_d||;  rgb = sigmoid(raw[:C])
//   sigma = relu(raw[C] + noise);  alpha = 1 - exp(-sigma*dists)
//   T = cumprod([1, 1-alpha+1e-10])[:-1];  w = alpha*T
//   rgb_map = sum w rgb; depth = sum w z; acc = sum w; disp = 1/max(1e-10, depth/acc)
// (NaN propagates through max as in torch.max, so acc == 0 gives disp = NaN like the
//  reference - SURVEY hard part 6.)
#include "common.h"

namespace {

constexpr int MAX_IPL = 8;
// channel loop with a compile-time bound so per-channel arrays stay in registers
#define FOR_CH(c) _Pragma("unroll") for (int c = 0; c < 3; ++c) if (c < C)

struct RayState {
    float alpha[MAX_IPL], T[MAX_IPL], w[MAX_IPL], dist[MAX_IPL], sig[MAX_IPL], pre[MAX_IPL], zz[MAX_IPL];
};

__device__ __forceinline__ float wave_sum(float v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int IPL>
__device__ __forceinline__ void ray_forward(const float* __restrict__ raw, const float* __restrict__ z,
                                            const float* __restrict__ noise, float noise_std, uint64_t seed,
                                            uint64_t offset, int C, int S, int64_t ray, int lane, float norm,
                                            RayState& st) {
    const int base = lane * IPL;
    const float* zr = z + ray * S;
    const float* rr = raw + ray * (int64_t)S * (C + 1);
    float lane_prod = 1.0f;
#pragma unroll
    for (int k = 0; k < IPL; ++k) {
        int i = base + k;
        bool ok = i < S;
        float zi = ok ? zr[i] : 0.f;
        float zn = (i + 1 < S) ? zr[i + 1] : 0.f;
        float dist = ok ? ((i + 1 < S) ? (zn - zi) : 1e10f) * norm : 0.f;
        float nz = 0.f;
        if (ok) {
            if (noise) nz = noise[ray * S + i];
            else if (noise_std > 0.f) nz = philox_normal(seed, offset, (uint64_t)(ray * S + i)) * noise_std;
        }
        float pre = ok ? rr[(int64_t)i * (C + 1) + C] + nz : 0.f;
        float sig = fmaxf(pre, 0.f);
        float alpha = ok ? 1.0f - expf(-sig * dist) : 0.f;
        st.zz[k] = zi;
        st.dist[k] = dist;
        st.pre[k] = pre;
        st.sig[k] = sig;
        st.alpha[k] = alpha;
        st.T[k] = lane_prod;                       // lane-local exclusive product
        float f = ok ? (1.0f - alpha) + 1e-10f : 1.0f;
        lane_prod *= f;
    }
    // exclusive product scan of lane_prod across lanes
    float incl = lane_prod;
    for (int off = 1; off < 64; off <<= 1) {
        float up = __shfl_up(incl, off, 64);
        if (lane >= off) incl *= up;
    }
    float excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 1.0f;
#pragma unroll
    for (int k = 0; k < IPL; ++k) {
        st.T[k] *= excl;
        st.w[k] = st.alpha[k] * st.T[k];
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int IPL>
__global__ void composite_fwd_kernel(const float* __restrict__ raw, const float* __restrict__ z,
                                     const float* __restrict__ rays_d, const float* __restrict__ noise,
                                     float noise_std, uint64_t seed, uint64_t offset, int C, int n_rays, int S,
                                     float* __restrict__ rgb_map, float* __restrict__ disp, float* __restrict__ acc,
                                     float* __restrict__ weights, float* __restrict__ depth,
                                     float* __restrict__ sigma) {
    int64_t ray = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x / 64);
    int lane = threadIdx.x & 63;
    if (ray >= n_rays) return;
    float d0 = rays_d[ray * 3], d1 = rays_d[ray * 3 + 1], d2 = rays_d[ray * 3 + 2];
    float norm = sqrtf((d0 * d0 + d1 * d1) + d2 * d2);
    RayState st;
    ray_forward<IPL>(raw, z, noise, noise_std, seed, offset, C, S, ray, lane, norm, st);
    const float* rr = raw + ray * (int64_t)S * (C + 1);
    float s_rgb[3] = {0.f, 0.f, 0.f}, s_depth = 0.f, s_acc = 0.f;
#pragma unroll
    for (int k = 0; k < IPL; ++k) {
        int i = lane * IPL + k;
        if (i < S) {
            float w = st.w[k];
            FOR_CH(c) s_rgb[c] += w * sigmoidf_(rr[(int64_t)i * (C + 1) + c]);
            s_depth += w * st.zz[k];
            s_acc += w;
            if (weights) weights[ray * S + i] = w;
            if (sigma) sigma[ray * S + i] = st.sig[k];
        }
    }
    FOR_CH(c) s_rgb[c] = wave_sum(s_rgb[c]);
    s_depth = wave_sum(s_depth);
    s_acc = wave_sum(s_acc);
    if (lane == 0) {
        if (rgb_map)
            FOR_CH(c) rgb_map[ray * C + c] = s_rgb[c];
        if (depth) depth[ray] = s_depth;
        if (acc) acc[ray] = s_acc;
        if (disp) {
            float r = s_depth / s_acc;
            float m = (r != r) ? r : fmaxf(1e-10f, r);
            disp[ray] = 1.0f / m;
        }
    }
}

// Backward.  With g_i = dL/dw_i:
//   dL/dalpha_i = g_i T_i - (sum_{k>i} g_k w_k) / (1 - alpha_i + 1e-10)
//   dL/dsigma_i = dL/dalpha_i * dist_i * exp(-sigma_i dist_i) * [pre_i > 0]
//   dL/dnorm    = sum_i dL/dalpha_i * sigma_i * exp(-sigma_i dist_i) * dist_i / norm
// (cumprod backward in its no-zero-input form; inputs are >= 1e-10 by construction.)
// one ray's backward on one wave; returns max |d_raw| the wave wrote (all lanes)
template <int IPL>
__device__ __forceinline__ float composite_bwd_ray(const float* __restrict__ raw, const float* __restrict__ z,
                                                   const float* __restrict__ rays_d, const float* __restrict__ noise,
                                                   float noise_std, uint64_t seed, uint64_t offset, int C, int S,
                                                   const float* __restrict__ g_rgb, const float* __restrict__ g_acc,
                                                   const float* __restrict__ g_depth, const float* __restrict__ g_disp,
                                                   float* __restrict__ d_raw, float* __restrict__ d_rays_d, int accumulate,
                                                   int64_t ray, int lane) {
    float d0 = rays_d[ray * 3], d1 = rays_d[ray * 3 + 1], d2 = rays_d[ray * 3 + 2];
    float norm = sqrtf((d0 * d0 + d1 * d1) + d2 * d2);
    RayState st;
    ray_forward<IPL>(raw, z, noise, noise_std, seed, offset, C, S, ray, lane, norm, st);
    const float* rr = raw + ray * (int64_t)S * (C + 1);
    float* dr = d_raw + ray * (int64_t)S * (C + 1);
    float grgb[3] = {0.f, 0.f, 0.f};
    FOR_CH(c) grgb[c] = g_rgb[ray * C + c];
    float ga = g_acc ? g_acc[ray] : 0.f;
    float gdp = g_depth ? g_depth[ray] : 0.f;
    if (g_disp) {   // disp = 1/max(1e-10, depth/acc): fold into depth / acc gradients
        float s_depth = 0.f, s_acc = 0.f;
#pragma unroll
        for (int k = 0; k < IPL; ++k) {
            s_depth += st.w[k] * st.zz[k];
            s_acc += st.w[k];
        }
        s_depth = wave_sum(s_depth);
        s_acc = wave_sum(s_acc);
        float r = s_depth / s_acc;
        if (r > 1e-10f) {
            float gr = -g_disp[ray] / (r * r);
            gdp += gr / s_acc;
            ga += -gr * r / s_acc;
        }
    }
    float gw[MAX_IPL], sg[MAX_IPL][3];
    float lane_tail = 0.f;   // sum of g_k w_k over this lane's samples
#pragma unroll
    for (int k = 0; k < IPL; ++k) {
        int i = lane * IPL + k;
        float g = 0.f;
        if (i < S) {
            FOR_CH(c) {
                float s = sigmoidf_(rr[(int64_t)i * (C + 1) + c]);
                sg[k][c] = s;
                g += grgb[c] * s;
            }
            g += ga + gdp * st.zz[k];
        }
        gw[k] = g;
        lane_tail += g * st.w[k];
    }
    // exclusive suffix sum across lanes
    float incl = lane_tail;
    for (int off = 1; off < 64; off <<= 1) {
        float dn = __shfl_down(incl, off, 64);
        if (lane + off < 64) incl += dn;
    }
    float after = __shfl_down(incl, 1, 64);
    if (lane == 63) after = 0.f;
    float g_norm = 0.f;
    float mx = 0.f;      // max |d_raw| this lane writes (the split-f16 dX chain scales its input by it: saves that a pass over d_raw)
    float run = after;   // sum over samples after the current one
#pragma unroll
    for (int k = IPL - 1; k >= 0; --k) {
        int i = lane * IPL + k;
        if (i < S) {
            float f = (1.0f - st.alpha[k]) + 1e-10f;
            float dalpha = gw[k] * st.T[k] - run / f;
            float e = expf(-st.sig[k] * st.dist[k]);
            float dsig = (st.pre[k] > 0.f) ? dalpha * st.dist[k] * e : 0.f;
            float ddist = dalpha * st.sig[k] * e;
            g_norm += ddist * (st.dist[k] / norm);
            FOR_CH(c) {
                const float dv = grgb[c] * st.w[k] * sg[k][c] * (1.0f - sg[k][c]);
                dr[(int64_t)i * (C + 1) + c] = dv;
                mx = fmaxf(mx, dv == dv ? fabsf(dv) : __builtin_inff());      // fmaxf drops NaN: record it as +inf
            }
            dr[(int64_t)i * (C + 1) + C] = dsig;
            mx = fmaxf(mx, dsig == dsig ? fabsf(dsig) : __builtin_inff());
            run += gw[k] * st.w[k];
        }
    }
    g_norm = wave_sum(g_norm);
    if (d_rays_d && lane < 3) {
        float dv = lane == 0 ? d0 : (lane == 1 ? d1 : d2);
        float v = g_norm * (dv / norm);
        if (accumulate) d_rays_d[ray * 3 + lane] += v;
        else d_rays_d[ray * 3 + lane] = v;
    }
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    return mx;
}

// One wave per ray, BWD_WAVES rays per workgroup.  max |d_raw|: an atomic per ray on ONE word serialises in L2 (4 081 rays:
// ~45 us on top of a ~8 us kernel, and every wave reaches it at the same moment, so reading the running maximum first does
// not thin them out) - the workgroup's 16 rays are reduced through LDS first: 256 atomics at C2.
constexpr int bwd_waves(int ipl) { return ipl > 4 ? 8 : 16; }      // 8 samples per lane: 128 registers do not hold a ray's state
template <int IPL>
__global__ __launch_bounds__(64 * bwd_waves(IPL)) void composite_bwd_kernel(
    const float* __restrict__ raw, const float* __restrict__ z, const float* __restrict__ rays_d, const float* __restrict__ noise,
    float noise_std, uint64_t seed, uint64_t offset, int C, int n_rays, int S, const float* __restrict__ g_rgb,
    const float* __restrict__ g_acc, const float* __restrict__ g_depth, const float* __restrict__ g_disp, float* __restrict__ d_raw,
    float* __restrict__ d_rays_d, int accumulate, float* __restrict__ d_raw_absmax) {
    constexpr int BWD_WAVES = bwd_waves(IPL);
    __shared__ float wave_mx[BWD_WAVES];
    const int wave = threadIdx.x / 64, lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * BWD_WAVES + wave;
    float mx = 0.f;
    if (ray < n_rays)
        mx = composite_bwd_ray<IPL>(raw, z, rays_d, noise, noise_std, seed, offset, C, S, g_rgb, g_acc, g_depth, g_disp, d_raw,
                                    d_rays_d, accumulate, ray, lane);
    if (!d_raw_absmax) return;
    if (lane == 0) wave_mx[wave] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {   // non-negative floats order like their bit patterns; NaN (sign clear) sorts above everything
        unsigned int m = 0u;
        for (int w = 0; w < BWD_WAVES; ++w) {
            const unsigned int b = __float_as_uint(wave_mx[w]);
            m = b > m ? b : m;
        }
        if (m) atomicMax(reinterpret_cast<unsigned int*>(d_raw_absmax), m);
    }
}

}  // namespace

#define DISPATCH_IPL(S, CALL)                      \
    do {                                           \
        int ipl__ = ((S) + 63) / 64;               \
        if (ipl__ <= 1) { CALL(1); }               \
        else if (ipl__ <= 2) { CALL(2); }          \
        else if (ipl__ <= 3) { CALL(3); }          \
        else if (ipl__ <= 4) { CALL(4); }          \
        else { CALL(8); }                          \
    } while (0)

extern "C" int benerf_composite_fwd(const float* raw, const float* z, const float* rays_d, const float* noise,
                                    float noise_std, uint64_t seed, uint64_t offset, int channels, int n_rays,
                                    int n_samples, float* rgb_map, float* disp, float* acc, float* weights,
                                    float* depth, float* sigma, benerf_stream_t stream) {
    BENERF_REQUIRE(raw && z && rays_d, "composite_fwd: null input");
    BENERF_REQUIRE(channels >= 1 && channels <= 3, "composite_fwd: channels must be 1..3");
    BENERF_REQUIRE(n_rays > 0 && n_samples > 0 && n_samples <= 64 * MAX_IPL, "composite_fwd: n_samples must be <= 512");
    const int waves = 4;
    dim3 grid((n_rays + waves - 1) / waves), block(64 * waves);
#define CALL(IPL)                                                                                                      \
    hipLaunchKernelGGL(composite_fwd_kernel<IPL>, grid, block, 0, as_stream(stream), raw, z, rays_d, noise, noise_std, \
                       seed, offset, channels, n_rays, n_samples, rgb_map, disp, acc, weights, depth, sigma)
    DISPATCH_IPL(n_samples, CALL);
#undef CALL
    BENERF_LAUNCH_CHECK("composite_fwd");
    return BENERF_OK;
}

extern "C" int benerf_composite_bwd(const float* raw, const float* z, const float* rays_d, const float* noise,
                                    float noise_std, uint64_t seed, uint64_t offset, int channels, int n_rays,
                                    int n_samples, const float* d_rgb_map, const float* d_acc, const float* d_depth,
                                    const float* d_disp, float* d_raw, float* d_rays_d, int accumulate,
                                    float* d_raw_absmax, benerf_stream_t stream) {
    BENERF_REQUIRE(raw && z && rays_d && d_rgb_map && d_raw, "composite_bwd: null pointer");
    BENERF_REQUIRE(channels >= 1 && channels <= 3, "composite_bwd: channels must be 1..3");
    BENERF_REQUIRE(n_rays > 0 && n_samples > 0 && n_samples <= 64 * MAX_IPL, "composite_bwd: n_samples must be <= 512");
#define CALL(IPL)                                                                                                      \
    hipLaunchKernelGGL(composite_bwd_kernel<IPL>, dim3((n_rays + bwd_waves(IPL) - 1) / bwd_waves(IPL)), dim3(64 * bwd_waves(IPL)), 0, as_stream(stream), raw, z, rays_d, noise, noise_std, \
                       seed, offset, channels, n_rays, n_samples, d_rgb_map, d_acc, d_depth, d_disp, d_raw, d_rays_d,  \
                       accumulate, d_raw_absmax)
    DISPATCH_IPL(n_samples, CALL);
#undef CALL
    BENERF_LAUNCH_CHECK("composite_bwd");
    return BENERF_OK;
}

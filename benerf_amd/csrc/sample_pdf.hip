// K5: hierarchical inverse-CDF sampling + sorted merge with the coarse depths.
//
// Follows run_nerf_helpers.py:74-115 (sample_pdf) and model/nerf.py:322-326:
//   bins = z_mid = .5*(z[1:]+z[:-1])  (B = S-1 entries);  w = weights[1:-1] + 1e-5
//   pdf = w / sum(w);  cdf = [0, cumsum(pdf)];  inds = searchsorted(cdf, u, right=True)
//   below = max(0, inds-1); above = min(B-1, inds); denom<1e-5 -> 1
//   sample = bins[below] + (u-cdf[below])/denom * (bins[above]-bins[below])
//   z_fine = sort(cat[z, samples])
//
// The reduction orders torch leaves unspecified are FIXED here and in the oracle's
// sample_pdf_exact (sequential float64 accumulation, rounded to float32 per element) so
// kernel and oracle agree bit for bit, indices and values.  Built with -ffp-contract=off.
//
// One wavefront per ray; per-wave LDS scratch: cdf[B], bins[B], merged values [S+Ni].
#include "common.h"

namespace {

constexpr int WAVES = 4;

__global__ void sample_pdf_merge_kernel(const float* __restrict__ z_coarse, const float* __restrict__ weights,
                                        const float* __restrict__ u_in, uint64_t seed, uint64_t offset, int n_rays,
                                        int S, int Ni, int bins_mode, float* __restrict__ z_fine,
                                        float* __restrict__ z_samples, int64_t* __restrict__ inds_out) {
    // bins_mode: z_coarse is `bins` [n_rays,S-1... given as B = S-1 columns], weights is [n_rays,B-1]
    // (already sliced, run_nerf_helpers.sample_pdf's own signature); no merge, samples only.
    extern __shared__ float smem[];
    const int B = S - 1, F = S + Ni;
    const int per_wave = 2 * B + F;
    const int wave = threadIdx.x / 64, lane = threadIdx.x & 63;
    int64_t ray = (int64_t)blockIdx.x * WAVES + wave;
    const bool active = ray < n_rays;        // inactive waves still reach the block barriers
    if (!active) ray = n_rays - 1;
    float* cdf = smem + wave * per_wave;
    float* bins = cdf + B;
    float* vals = bins + B;
    if (bins_mode) {
        const float* br = z_coarse + ray * B;
        const float* wr = weights + ray * (B - 1);
        for (int k = lane; k < B; k += 64) bins[k] = br[k];
        for (int k = lane; k < B - 1; k += 64) cdf[k + 1] = wr[k] + 1e-5f;
    } else {
        const float* zr = z_coarse + ray * S;
        const float* wr = weights + ray * S;
        for (int k = lane; k < B; k += 64) bins[k] = 0.5f * (zr[k + 1] + zr[k]);
        for (int k = lane; k < S; k += 64) vals[k] = zr[k];
        // stage w' = weights[1:-1] + 1e-5 in cdf[1..B-1] (B-1 = S-2 values)
        for (int k = lane; k < B - 1; k += 64) cdf[k + 1] = wr[k + 1] + 1e-5f;
    }
    __syncthreads();
    if (lane == 0) {
        double tot = 0.0;
        for (int k = 1; k < B; ++k) tot += (double)cdf[k];
        float total = (float)tot;
        double run = 0.0;
        cdf[0] = 0.f;
        for (int k = 1; k < B; ++k) {
            float pdf = cdf[k] / total;
            run += (double)pdf;
            cdf[k] = (float)run;
        }
    }
    __syncthreads();

    for (int q = lane; q < Ni; q += 64) {
        float u = u_in ? u_in[ray * Ni + q] : philox_uniform(seed, offset, (uint64_t)(ray * Ni + q));
        // first index with cdf[idx] > u  (searchsorted right=True)
        int lo = 0, hi = B;
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (cdf[mid] <= u) lo = mid + 1;
            else hi = mid;
        }
        int ind = lo;
        int below = ind - 1 > 0 ? ind - 1 : 0;
        int above = ind < B - 1 ? ind : B - 1;
        float c0 = cdf[below], c1 = cdf[above];
        float b0 = bins[below], b1 = bins[above];
        float denom = c1 - c0;
        if (denom < 1e-5f) denom = 1.0f;
        float t = (u - c0) / denom;
        float smp = b0 + t * (b1 - b0);
        vals[S + q] = smp;
        if (active && z_samples) z_samples[ray * Ni + q] = smp;
        if (active && inds_out) inds_out[ray * Ni + q] = ind;
    }
    __syncthreads();
    if (!active || bins_mode) return;

    // rank sort (values only): rank = #less + #equal-with-lower-index
    float* out = z_fine + ray * F;
    for (int e = lane; e < F; e += 64) {
        float v = vals[e];
        int rank = 0;
        for (int k = 0; k < F; ++k) {
            float o = vals[k];
            rank += (o < v || (o == v && k < e)) ? 1 : 0;
        }
        out[rank] = v;
    }
}

}  // namespace

extern "C" int benerf_sample_pdf_merge(const float* z_coarse, const float* weights, const float* u, uint64_t seed,
                                       uint64_t offset, int n_rays, int n_samples, int n_importance, float* z_fine,
                                       float* z_samples, int64_t* inds, benerf_stream_t stream) {
    BENERF_REQUIRE(z_coarse && weights && z_fine, "sample_pdf_merge: null pointer");
    BENERF_REQUIRE(n_rays > 0 && n_samples >= 3 && n_samples <= 1024 && n_importance > 0 && n_importance <= 1024,
                   "sample_pdf_merge: need 3 <= n_samples <= 1024, 0 < n_importance <= 1024");
    int B = n_samples - 1, F = n_samples + n_importance;
    size_t smem = (size_t)WAVES * (2 * B + F) * sizeof(float);
    BENERF_REQUIRE(smem <= 64 * 1024, "sample_pdf_merge: sample counts too large for LDS scratch");
    dim3 grid((n_rays + WAVES - 1) / WAVES), block(64 * WAVES);
    hipLaunchKernelGGL(sample_pdf_merge_kernel, grid, block, smem, as_stream(stream), z_coarse, weights, u, seed, offset,
                       n_rays, n_samples, n_importance, 0, z_fine, z_samples, inds);
    BENERF_LAUNCH_CHECK("sample_pdf_merge");
    return BENERF_OK;
}

extern "C" int benerf_sample_pdf(const float* bins, const float* weights, const float* u, uint64_t seed, uint64_t offset,
                                 int n_rays, int n_bins, int n_draws, float* samples, int64_t* inds,
                                 benerf_stream_t stream) {
    BENERF_REQUIRE(bins && weights && samples, "sample_pdf: null pointer");
    BENERF_REQUIRE(n_rays > 0 && n_bins >= 2 && n_bins <= 1024 && n_draws > 0 && n_draws <= 1024,
                   "sample_pdf: need 2 <= n_bins <= 1024, 0 < n_draws <= 1024");
    int S = n_bins + 1, F = S + n_draws;
    size_t smem = (size_t)WAVES * (2 * n_bins + F) * sizeof(float);
    BENERF_REQUIRE(smem <= 64 * 1024, "sample_pdf: sizes too large for LDS scratch");
    dim3 grid((n_rays + WAVES - 1) / WAVES), block(64 * WAVES);
    hipLaunchKernelGGL(sample_pdf_merge_kernel, grid, block, smem, as_stream(stream), bins, weights, u, seed, offset,
                       n_rays, S, n_draws, 1, (float*)nullptr, samples, inds);
    BENERF_LAUNCH_CHECK("sample_pdf");
    return BENERF_OK;
}

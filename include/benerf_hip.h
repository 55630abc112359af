/*
 * benerf_hip.h - C ABI of libbenerf_hip.so (MI355X / gfx950 kernels for the BeNeRF
 * training + rendering hot path).
 *
 * The reference (WU-CVGL/BeNeRF) is pure Python/PyTorch and has no FFI; each entry
 * point below replaces the chain of ATen dispatches issued by the cited reference
 * lines (paths relative to the reference root).  The Python host side
 * (benerf_amd/) binds these with ctypes; see INTEGRATION.md for the stub.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless the
 *     name ends in _host; arrays are dense row-major, float32 / int64 / int32 exactly
 *     as the torch tensors on the reference side;
 *   - the caller owns every buffer and the stream (pass
 *     torch.cuda.current_stream().cuda_stream); no allocation, no implicit
 *     synchronisation (one documented exception: benerf_mlp_status_check), nothing is thrown
 *     across the ABI;
 *   - return 0 on success, BENERF_EBADARG (-1) bad argument, BENERF_EWORKSPACE (-2)
 *     workspace too small, BENERF_EHIP (-3) HIP error, BENERF_ERANGE (-4) value outside the arithmetic
 *     mode's range (benerf_mlp_status_check only); text via benerf_last_error();
 *   - thread-compatible: no global mutable state except the thread-local last error.
 */
#ifndef BENERF_HIP_H
#define BENERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BENERF_OK 0
#define BENERF_EBADARG (-1)
#define BENERF_EWORKSPACE (-2)
#define BENERF_EHIP (-3)
#define BENERF_ERANGE (-4)

typedef void* benerf_stream_t; /* hipStream_t */

/* ABI revision of this header: bumped whenever the signature or the argument meaning of an entry point changes (round 5's
 * `params` argument in front of benerf_mlp_bwd_dw made it 101; round 6: 102, 103 with the loss-glue operators).  benerf_version() returns the revision the LIBRARY
 * was built from - a binding compares the two before its first call (benerf_amd/_lib.py does and refuses a mismatch): a caller
 * compiled against another revision would pass shifted arguments. */
#define BENERF_ABI_VERSION 103
int benerf_version(void);
const char* benerf_last_error(void);

/* ---------------------------------------------------------------- K1: trajectory --- */
/* Cubic B-spline / linear pose interpolation in SE(3).
 * Replaces spline.cubic_spline_pose_unit_time (spline.py:247-303),
 * spline.linear_pose_unit_time (spline.py:305-331) and the knot/transform/linspace
 * plumbing of Graph.get_pose_evt / get_pose_rgb (model/optimize.py:58-111).
 *   knots [4,6] se(3); transform [6] added to every knot in se(3) or NULL;
 *   explicit_ts == 0: ts [2] device floats = (t_start, t_end), pose p is evaluated at
 *   torch.linspace(t0,t1,n)[p] (model/optimize.py:71,102); explicit_ts != 0: ts [n_poses]
 *   sample times as passed to the reference's spline functions;
 *   traj 0 = cubic B-spline, 1 = linear (knots 0 and 3), 2 = cubic Bezier (the evident intent of the reference's
 *   non-executable bezier.py:22-74: Bernstein translation, cumulative-Bernstein rotation); poses out [n_poses,3,4]. */
int benerf_spline_poses_fwd(const float* knots, const float* transform, const float* ts,
                            int n_poses, int traj, int explicit_ts, float* poses,
                            benerf_stream_t stream);
/* d_poses [n_poses,3,4] -> d_knots [4,6] (overwritten), d_transform [6] (overwritten, may
 * be NULL).  Forward-mode duals over the same float code; deterministic reduction. */
int benerf_spline_poses_bwd(const float* knots, const float* transform, const float* ts,
                            int n_poses, int traj, int explicit_ts, const float* d_poses,
                            float* d_knots, float* d_transform, benerf_stream_t stream);
/* The two trajectory evaluations of one training step (a: event camera, on the knots themselves; b: RGB camera, knots +
 * transform_b; both linspace timestamps ts_* [2]) as one launch.  Same results as two benerf_spline_poses_fwd calls
 * (model/optimize.py:58-111). */
int benerf_spline_poses_fwd_pair(const float* knots, const float* transform_b, const float* ts_a, int n_a,
                                 const float* ts_b, int n_b, int traj, float* poses_a, float* poses_b,
                                 benerf_stream_t stream);
/* The two trajectory backward passes of one training step (a: event camera, no transform; b: RGB camera with
 * transform; both linspace timestamps) as one launch.  Same results as two benerf_spline_poses_bwd calls. */
int benerf_spline_poses_bwd_pair(const float* knots, const float* transform_b, const float* ts_a, int n_a,
                                 const float* ts_b, int n_b, int traj, const float* d_poses_a,
                                 const float* d_poses_b, float* d_knots_a, float* d_knots_b,
                                 float* d_transform_b, benerf_stream_t stream);

/* The reference's public single-step helpers (spline.py:16-192) as element-wise kernels over n items, same device
 * code as the trajectory kernels.  op: 0 se3_2_qt_parallel ([6] -> [q xyzw | t] = [7]), 1 exp_r2q_parallel ([3] -> [4]),
 * 2 log_q2r_parallel ([4] -> [3]), 3 q_to_R_parallel ([4] -> [3,3]), 4 taylor_B, 5 taylor_C ([1] -> [1]),
 * 6 skew_symmetric ([3] -> [3,3]), 7 q_to_Q_parallel ([4] -> [4,4]), 8 q_to_q_conj_parallel ([4] -> [4]).
 * _bwd: d_in [n, in] = J^T d_out.  Branch selection follows the VALUE like torch.where's forward; the derivative is
 * that of the selected branch (the reference's backward is NaN where its unselected branch divides 0 by 0). */
int benerf_spline_op_fwd(int op, const float* in, int64_t n, float* out, benerf_stream_t stream);
int benerf_spline_op_bwd(int op, const float* in, int64_t n, const float* d_out, float* d_in,
                         benerf_stream_t stream);

/* ---------------------------------------------------------------- K2: rays --------- */
/* Pinhole ray generation (pose-major, N = n_poses*n_pix), view directions and LLFF NDC.
 * Replaces run_nerf_helpers.get_specific_rays / get_rays (run_nerf_helpers.py:13-44),
 * ndc_rays (run_nerf_helpers.py:46-71) and Graph.render's ray plumbing
 * (model/nerf.py:241-286).  ray_idx [n_pix] int64 pixel indices (row-major, idx = j*W+i).
 * remap: NULL, or the TUM_VIE undistortion look-up table [H,W,2] of (x, y) float coordinates
 * gathered per pixel (model/nerf.py:247-250, run_nerf_helpers.py:17-23).
 * Outputs rays_o, rays_d (NDC'd when ndc != 0), viewdirs: [N,3] each. */
int benerf_rays_fwd(const float* poses, const int64_t* ray_idx, int n_poses, int n_pix,
                    int H, int W, float fx, float fy, float cx, float cy, int ndc, const float* remap,
                    float* rays_o, float* rays_d, float* viewdirs, benerf_stream_t stream);
/* (d_rays_o, d_rays_d, d_viewdirs) [N,3] -> d_poses [n_poses,3,4] (overwritten).  workspace: caller-owned scratch of
 * benerf_rays_bwd_workspace_floats(n_poses, n_pix) floats (0 - NULL allowed - up to 256 pixels per pose): per-chunk partial
 * sums, added in chunk order (deterministic). */
size_t benerf_rays_bwd_workspace_floats(int n_poses, int n_pix);
int benerf_rays_bwd(const float* poses, const int64_t* ray_idx, int n_poses, int n_pix,
                    int H, int W, float fx, float fy, float cx, float cy, int ndc, const float* remap,
                    const float* d_rays_o, const float* d_rays_d, const float* d_viewdirs,
                    float* d_poses, float* workspace, size_t workspace_floats, benerf_stream_t stream);
/* Stratified coarse depths: z = lower + (upper-lower)*t_rand (model/nerf.py:297-307).
 * t_rand [n_rays,n_samples] uniform draws, or NULL for in-kernel Philox(seed, offset). */
int benerf_stratified_z(int n_rays, int n_samples, float near, float far, const float* t_rand,
                        uint64_t seed, uint64_t offset, float* z, benerf_stream_t stream);
/* Sums per-sample point gradients into ray gradients: pts = o + d*z (model/nerf.py:308,327)
 *   d_o = sum_s d_pts, d_d = sum_s z*d_pts, d_viewdirs = sum_s d_vdir_pts.
 * accumulate: 0 overwrites the outputs, 1 adds into all three, 2 adds into d_rays_d only (which then already holds
 * benerf_composite_bwd's part) and overwrites d_rays_o / d_viewdirs. */
int benerf_ray_grad_reduce(int n_rays, int n_samples, const float* z, const float* d_pts,
                           const float* d_vdir_pts, int accumulate, float* d_rays_o,
                           float* d_rays_d, float* d_viewdirs, benerf_stream_t stream);

/* ---------------------------------------------------------------- K3: fused MLP ---- */
/* NeRF parameter block in nn.Linear layout ([out,in] row-major), reference state-dict
 * order (model/nerf.py:53-64): pts_linears.0..7, views_linears.0, feature_linear,
 * alpha_linear, rgb_linear. */
enum { BENERF_NLAYERS = 12, BENERF_L_VIEWS = 8, BENERF_L_FEAT = 9, BENERF_L_ALPHA = 10, BENERF_L_RGB = 11 };
typedef struct BenerfMlpParams {
    const float* w[BENERF_NLAYERS];
    const float* b[BENERF_NLAYERS];
    /* NULL, or 96 device floats scaling the positional-encoding columns before the first / skip / view layers
     * (BARF coarse-to-fine, model/nerf.py:16-26,78-89): [0,64) the 63 PE(pts) columns (+ pad), [64,96) the 27 PE(dir)
     * columns (+ pad).  The reference's weight of encoding element e is w[e % L] (its view(-1, L)), 1 for the raw input. */
    const float* pe_weights;
} BenerfMlpParams;
typedef struct BenerfMlpGrads {
    float* w[BENERF_NLAYERS];
    float* b[BENERF_NLAYERS];
} BenerfMlpGrads;

/* floats needed for the MFMA-packed copies of one network's weights (f32 blocks + split-f16 blocks) */
size_t benerf_mlp_packed_floats(void);
/* Re-pack one network (call after every optimiser step). packed [benerf_mlp_packed_floats()] */
int benerf_mlp_pack_weights(const BenerfMlpParams* params, int channels, float* packed,
                            benerf_stream_t stream);
/* Both networks of a training step (coarse, fine: same channel count) in one launch; same results as two
 * benerf_mlp_pack_weights calls. */
int benerf_mlp_pack_weights_pair(const BenerfMlpParams* params_a, float* packed_a, const BenerfMlpParams* params_b,
                                 float* packed_b, int channels, benerf_stream_t stream);
/* floats of the saved-activation buffer for n_points sample points (layer outputs, PE tiles and
 * the per-tile ReLU sign-bit words), and floats per point of the activation-gradient scratch */
size_t benerf_mlp_act_floats(int64_t n_points);
size_t benerf_mlp_dact_floats_per_point(void);
/* floats of the activation-gradient scratch for n_points (valid for either arithmetic mode; >= n_points *
 * benerf_mlp_dact_floats_per_point()) */
size_t benerf_mlp_dact_floats(int64_t n_points);
/* The same two buffers sized for ONE arithmetic mode (`precision`: BENERF_MLP_F32 / _SPLIT / _SPLIT_F16BWD below): the 8-bit
 * residual twins of BENERF_MLP_SPLIT add a third to the saved activations and a half to the gradients, which the other modes
 * never touch (at C5, 2.1 M points and two networks, several GB).  Returns 0 for an unknown mode. */
size_t benerf_mlp_act_floats_for(int64_t n_points, int precision);
size_t benerf_mlp_dact_floats_for(int64_t n_points, int precision);
/* floats of the weight-gradient partial-sum workspace for n_points */
size_t benerf_mlp_dw_workspace_floats(int64_t n_points);
/* One size query for every caller-owned buffer of a render + backward over n_points = n_rays * n_samples sample points
 * of n_poses x n_pix rays, in BYTES (SURVEY 8b2's benerf_workspace_bytes): `which` =
 *   BENERF_WS_MLP_ACTS    saved activations of one benerf_mlp_fwd call          (= 4 * benerf_mlp_act_floats)
 *   BENERF_WS_MLP_DACTS   activation gradients of one benerf_mlp_bwd_dx call    (= 4 * benerf_mlp_dact_floats)
 *   BENERF_WS_MLP_DW      partial sums of one benerf_mlp_bwd_dw call            (= 4 * benerf_mlp_dw_workspace_floats)
 *   BENERF_WS_MLP_PACKED  the MFMA-shaped weight copy of one network            (= 4 * benerf_mlp_packed_floats)
 *   BENERF_WS_RAYS_BWD    chunk partials of benerf_rays_bwd                     (= 4 * benerf_rays_bwd_workspace_floats)
 * Returns 0 for an unknown `which`.  Sizes cover either MLP precision mode. */
enum { BENERF_WS_MLP_ACTS = 0, BENERF_WS_MLP_DACTS = 1, BENERF_WS_MLP_DW = 2, BENERF_WS_MLP_PACKED = 3, BENERF_WS_RAYS_BWD = 4 };
size_t benerf_workspace_bytes(int which, int64_t n_points, int n_poses, int n_pix);

/* MFMA arithmetic of the fused MLP kernels - a PER-CALL argument (the library keeps no mode state):
 *   BENERF_MLP_F32    exact f32 MFMA (v_mfma_f32_32x32x2_f32) in forward and backward, bit-for-bit f32 products;
 *   BENERF_MLP_SPLIT  every GEMM operand (activations, weights, gradients) as two f16 numbers hi + lo, three f16 MFMAs
 *                     per product block (hi x hi, hi x lo, lo x hi), f32 accumulation - 22-bit operands, in the forward pass
 *                     AND in both backward GEMMs (dX chain, dW).  Measured against float64 the outputs and every gradient
 *                     carry the error of the BENERF_MLP_F32 path (tests/test_f64_truth_gpu.py holds both modes to the same
 *                     bounds): fp32-equivalent, the default.  Gradients are rescaled by exact powers of two into f16's
 *                     range.  Activations must stay below 65504 in magnitude (f16 range): see `status`;
 *   BENERF_MLP_SPLIT_F16BWD  the forward of BENERF_MLP_SPLIT with a REDUCED-PRECISION backward: dX takes the gradient as one
 *                     f16 (weights hi + lo), dW takes f16 dY and f16 X, one MFMA per block.  Gradients sit 2-8e-4 of their
 *                     largest entry from the fp32 result (inside SURVEY 8c's 1e-3 contract on the mean-squared losses, not
 *                     fp32-equivalent); ~25 % faster steps.  Opt-in;
 *   BENERF_MLP_AUTO   inference only (acts == NULL): BENERF_MLP_SPLIT, followed by a BENERF_MLP_F32 launch whose
 *                     workgroups exit at once unless the split launch reported an activation outside the f16 range
 *                     - the output is always valid, at the price of one (normally empty) extra launch.
 * The saved-activation / gradient buffers have a mode-specific layout: forward and backward of one step must
 * use the same mode (the buffers are tagged; a mismatch is reported through status[2]).
 *
 * status: caller-owned DEVICE uint32[BENERF_ST_WORDS], zeroed by the caller, may be NULL for BENERF_MLP_F32:
 *   [ACT]  max |activation| seen by split forward launches (f32 bit pattern; written only once it passes 2^15)
 *   [GRAD] max |gradient| of the tile-scaled split dX chain, taken before its rounding to f16 (same convention)
 *   [MODE] != 0: a backward launch got activation buffers of another mode
 *   [AUTO] scratch of BENERF_MLP_AUTO (maximum of the current call)
 *   [SKIP .. LAST_GRAD] bookkeeping of benerf_step_gate (below).
 * [ACT], [GRAD], [MODE] stay set until benerf_step_gate or the caller clears them.  Nothing here synchronises;
 * benerf_mlp_status_check does (copy + stream sync) and returns BENERF_ERANGE when [ACT] or [GRAD] reached 65504 or
 * training steps were skipped since the words were last zeroed, BENERF_EBADARG for [MODE].
 * benerf_adam_step takes the same pointer and leaves the parameters untouched when [SKIP] is set or [ACT] / [GRAD]
 * show a violation. */
enum { BENERF_MLP_F32 = 0, BENERF_MLP_SPLIT = 1, BENERF_MLP_AUTO = 2, BENERF_MLP_SPLIT_F16BWD = 3 };
enum { BENERF_ST_ACT = 0, BENERF_ST_GRAD = 1, BENERF_ST_MODE = 2, BENERF_ST_AUTO = 3, BENERF_ST_SKIP = 4, BENERF_ST_SKIPPED = 5,
       BENERF_ST_CONSECUTIVE = 6, BENERF_ST_STEPS = 7, BENERF_ST_LAST_ACT = 8, BENERF_ST_LAST_GRAD = 9,
       BENERF_ST_STEP_SCRATCH = 10,     /* [10], [11]: max |d_raw| of the step's (up to) two networks - benerf_composite_bwd's
                                         * d_raw_absmax, NaN recorded as +inf; zeroed by benerf_step_gate phase 1, which treats a
                                         * non-finite value as a violation (the loss gradient of the step is not finite) */
       BENERF_ST_SKIPPED_TOTAL = 12,    /* skipped steps since the words were created (never cleared by benerf_mlp_status_check's callers'
                                         * resets of [SKIPPED]): benerf_adam_step counts only APPLIED steps in its bias correction,
                                         * t = step - [SKIPPED_TOTAL] (torch's GradScaler does not advance a skipped step either) */
       BENERF_ST_MAX_CONSECUTIVE = 13,  /* longest run of skipped steps since the host last cleared it (a run that ends between two
                                         * host polls is still seen) */
       BENERF_ST_WORDS = 16 };
int benerf_mlp_status_check(const uint32_t* status, benerf_stream_t stream);
/* Per-step verdict of the range guard, on the device (no synchronisation), between the backward pass and the
 * benerf_adam_step launches of a training iteration (train.py:340-352):
 *   phase 0: reduce_flag[0] = 1.f if this rank's [ACT] / [GRAD] / [MODE] / [STEP_SCRATCH] show a violation, else 0.f - data-parallel
 *            callers SUM it over the ranks together with the gradients, so that every replica takes the same decision
 *            (a rank that overflowed contributes inf / NaN to everybody's gradient sum);
 *   phase 1: [SKIP] = violation (reduce_flag[0] > 0 when reduce_flag != NULL, this rank's words otherwise), counters
 *            [SKIPPED] (total), [SKIPPED_TOTAL], [CONSECUTIVE], [MAX_CONSECUTIVE], [STEPS] updated, the tripping maxima kept in
 *            [LAST_ACT] / [LAST_GRAD],
 *            [ACT] / [GRAD] / [MODE] and the two [STEP_SCRATCH] words cleared for the next step: one violation costs one step,
 *            not the rest of the run. */
int benerf_step_gate(uint32_t* status, float* reduce_flag, int phase, benerf_stream_t stream);

/* Fused positional encoding + 8x256 MLP + view branch, forward.
 * Replaces Embedder.embed (model/embedder.py:9-34) and NeRF.forward
 * (model/nerf.py:67-116) incl. pts = o + d*z (model/nerf.py:308,327).
 *   rays_o, rays_d, viewdirs [n_rays,3]; z [n_rays,n_samples]; raw out
 *   [n_rays,n_samples,channels+1] = [rgb..., sigma] pre-activation.
 *   acts: NULL (inference) or [benerf_mlp_act_floats(n_points)] saved for backward. */
int benerf_mlp_fwd(const BenerfMlpParams* params, const float* packed, int channels,
                   int n_rays, int n_samples, const float* rays_o, const float* rays_d,
                   const float* viewdirs, const float* z, float* raw, float* acts,
                   int precision, uint32_t* status, benerf_stream_t stream);
/* Backward of the above (autograd of model/nerf.py:67-116), two launches so that a caller can time / overlap them.
 * d_raw [n_points,channels+1].
 *   _dx: activation-gradient chain -> dacts scratch [benerf_mlp_dact_floats(n_points)], d_pts [n_points,3],
 *        d_vdir_pts [n_points,3] (per point; reduce with benerf_ray_grad_reduce);
 *   _dw: weight gradients from acts + dacts; dw_ws scratch [benerf_mlp_dw_workspace_floats(n_points)];
 *        params: the network's weights - read by BENERF_MLP_SPLIT only (the feature layer's and the views layer's weight
 *        gradients are composed from ONE product dhv^T h7 and those two weights: the linear feature layer is neither saved nor
 *        back-propagated through HBM), may be NULL in the other modes;
 *        grads: overwritten when accumulate == 0, added to otherwise; pe_weights: the forward call's
 *        BenerfMlpParams.pe_weights (the saved encodings are unweighted; the columns are scaled in the reduce).
 * precision: BENERF_MLP_F32, BENERF_MLP_SPLIT or BENERF_MLP_SPLIT_F16BWD, the mode of the forward launch that wrote acts.
 * d_raw_absmax (_dx): NULL, or the device float benerf_composite_bwd filled with max |d_raw| for exactly this d_raw. */
int benerf_mlp_bwd_dx(const BenerfMlpParams* params, const float* packed, int channels,
                      int n_rays, int n_samples, const float* d_raw, const float* acts,
                      float* dacts, float* d_pts, float* d_vdir_pts, int precision, uint32_t* status, const float* d_raw_absmax,
                      benerf_stream_t stream);
int benerf_mlp_bwd_dw(const BenerfMlpParams* params, int channels, int n_rays, int n_samples, const float* d_raw,
                      const float* acts, const float* dacts, float* dw_ws, size_t dw_ws_floats,
                      const BenerfMlpGrads* grads, int accumulate, int precision, const float* pe_weights,
                      benerf_stream_t stream);

/* Known-answer access to the format of the saved operands of BENERF_MLP_SPLIT (csrc/mlp_split.h): every activation / activation
 * gradient the forward and dX kernels save for the dW kernel is an f16 hi = rn16(x) plus an 8-bit residual code
 *     code = clamp(rn((x - hi) * 2^(18 - E)) + 128, 0, 255),  E = max(exponent(hi), -6)     x ~ hi + (code - 128) * 2^(E - 18).
 * Encodes and decodes n values (a multiple of 8) with the kernels' own device functions; residual_log2_scale = 11 (the forward
 * kernel's path: residual carried as (x - hi) * 2^11 in f16) or 12 (the dX kernel's).  Test infrastructure of the format, not a
 * step of the path: hi_bits [n] f16 bit patterns, codes [n], decoded [n]. */
int benerf_mlp_h8_roundtrip(const float* x, int64_t n, int residual_log2_scale, uint16_t* hi_bits, uint8_t* codes, float* decoded,
                            benerf_stream_t stream);

/* ---------------------------------------------------------------- K4: compositing -- */
/* Alpha compositing, one wavefront per ray.  Replaces NeRF.raw2output
 * (model/nerf.py:118-148).  noise [n_rays,n_samples] = randn*raw_noise_std, or NULL with
 * noise_std > 0 for in-kernel Philox normals, or NULL with noise_std == 0 for none.
 * Any output pointer may be NULL. n_samples <= 512. */
int benerf_composite_fwd(const float* raw, const float* z, const float* rays_d,
                         const float* noise, float noise_std, uint64_t seed, uint64_t offset,
                         int channels, int n_rays, int n_samples, float* rgb_map, float* disp,
                         float* acc, float* weights, float* depth, float* sigma,
                         benerf_stream_t stream);
/* d_rgb_map [n_rays,C] (required); d_acc, d_depth, d_disp [n_rays] optional (NULL = 0).
 * Out: d_raw [n_rays,n_samples,C+1]; d_rays_d [n_rays,3] through ||rays_d|| in dists
 * (overwritten, or added to when accumulate != 0; may be NULL).
 * d_raw_absmax: NULL, or one device float the caller zeroed: receives max |d_raw| (atomic maximum over the launch; +inf if any
 * entry is NaN) - the split-f16 benerf_mlp_bwd_dx takes it instead of running its own pass over d_raw. */
int benerf_composite_bwd(const float* raw, const float* z, const float* rays_d,
                         const float* noise, float noise_std, uint64_t seed, uint64_t offset,
                         int channels, int n_rays, int n_samples, const float* d_rgb_map,
                         const float* d_acc, const float* d_depth, const float* d_disp,
                         float* d_raw, float* d_rays_d, int accumulate, float* d_raw_absmax, benerf_stream_t stream);

/* ---------------------------------------------------------------- K5: sample_pdf --- */
/* Inverse-CDF importance sampling + sorted merge with the coarse depths.
 * Replaces run_nerf_helpers.sample_pdf (run_nerf_helpers.py:74-115) and
 * model/nerf.py:322-326 (z_mid, weights[1:-1], detach, cat, sort).
 *   z_coarse, weights [n_rays,n_samples]; u [n_rays,n_importance] uniforms or NULL for
 *   Philox; z_fine out [n_rays,n_samples+n_importance] ascending.
 *   Optional test outputs: z_samples [n_rays,n_importance], inds int64 same shape.
 * Arithmetic is fully specified (see oracle sample_pdf_exact): bit-exact vs the oracle. */
int benerf_sample_pdf_merge(const float* z_coarse, const float* weights, const float* u,
                            uint64_t seed, uint64_t offset, int n_rays, int n_samples,
                            int n_importance, float* z_fine, float* z_samples, int64_t* inds,
                            benerf_stream_t stream);

/* Stand-alone sample_pdf with the reference's own signature (run_nerf_helpers.py:74):
 * bins [n_rays,n_bins], weights [n_rays,n_bins-1], u [n_rays,n_draws] or NULL (Philox);
 * samples out [n_rays,n_draws]; inds optional. */
int benerf_sample_pdf(const float* bins, const float* weights, const float* u, uint64_t seed,
                      uint64_t offset, int n_rays, int n_bins, int n_draws, float* samples,
                      int64_t* inds, benerf_stream_t stream);

/* ---------------------------------------------------------------- stand-alone helpers */
/* Forward-only single operators backing the reference's public helper functions.
 * benerf_pixel_rays: get_specific_rays / get_rays (run_nerf_helpers.py:13-44); c2w is
 *   [n,3,4] (per_ray_pose != 0) or one [3,4] pose shared by all rays; i = column, j = row.
 * benerf_ndc_rays: ndc_rays (run_nerf_helpers.py:46-71).
 * benerf_posenc: Embedder.embed (model/embedder.py:9-34): [x | sin(2^k x), cos(2^k x)]_k.
 * benerf_mse_fwd/bwd: MSELoss (loss/imgloss.py:3-5), out/grad_out are 1-element arrays.
 * benerf_bright_log_fwd/bwd: rgb2brightlog (utils/math_utils.py:4-23) on n values: linlog == 0 log(x + 1e-9) (BeNeRF_*),
 *   linlog != 0 the lin-log curve of 8-bit brightness (E2NeRF_*: c = 255 x; c < 20 ? c log(20)/20 : log(c + 1e-9));
 *   bwd: d_x = grad * d curve / dx.  The fused step evaluates the same curves inside K6.
 * benerf_rgb2gray_fwd/bwd: RGB2Gray (utils/img_utils.py:7-16): [n,3] -> [n] luma (0.299 r + 0.587 g) + 0.114 b, summed in that
 *   order; bwd: d_rgb[n,3] = grad[n] * weights.
 * (One launch each way per call in a reference-style loop that assembles the event loss itself, train.py:207-292, instead of
 *  2-8 element-wise torch operators and as many autograd nodes.) */
int benerf_pixel_rays(const float* c2w, int per_ray_pose, const int64_t* i, const int64_t* j,
                      int64_t n, float fx, float fy, float cx, float cy, float* rays_o,
                      float* rays_d, benerf_stream_t stream);
int benerf_ndc_rays(int H, int W, float focal, float near, const float* rays_o,
                    const float* rays_d, int64_t n, float* out_o, float* out_d,
                    benerf_stream_t stream);
int benerf_posenc(const float* x, int64_t n, int dims, int n_freqs, int include_input,
                  float* out, benerf_stream_t stream);
int benerf_mse_fwd(const float* a, const float* b, int64_t n, float* out, benerf_stream_t stream);
int benerf_mse_bwd(const float* a, const float* b, int64_t n, const float* grad_out, float* d_a,
                   float* d_b, benerf_stream_t stream);
int benerf_bright_log_fwd(const float* x, int64_t n, int linlog, float* out, benerf_stream_t stream);
int benerf_bright_log_bwd(const float* x, const float* grad, int64_t n, int linlog, float* d_x, benerf_stream_t stream);
int benerf_rgb2gray_fwd(const float* rgb, int64_t n, float* out, benerf_stream_t stream);
int benerf_rgb2gray_bwd(const float* grad, int64_t n, float* d_rgb, benerf_stream_t stream);

/* ---------------------------------------------------------------- K6: losses ------- */
typedef struct BenerfLossCfg {
    int32_t channels;        /* 1 or 3 */
    int32_t linlog;          /* 0 safelog (BeNeRF_*), 1 linlog (E2NeRF_*): utils/math_utils.py:4-23 */
    int32_t n_evt_pix;       /* local event pixels R_e (rgb_evt has 2*R_e rows) */
    int32_t n_rgb_pix;       /* local blur pixels R_r */
    int32_t n_poses;         /* virtual poses n (rgb_rgb has n*R_r rows) */
    int32_t n_evt_pix_global;/* global batch sizes (== local on one GPU) */
    int32_t n_rgb_pix_global;
    float event_threshold;   /* > 0 synthetic branch (train.py:207-236), else L2-normalised (train.py:238-292) */
    float event_coeff;       /* event_coeff_syn or event_coeff_real */
    float rgb_coeff;
} BenerfLossCfg;
enum { BENERF_LOSS_NSTATS = 16 };
/* Pass 1: partial sums over the local batch -> stats[BENERF_LOSS_NSTATS] doubles (sum over
 * ranks with an all-reduce when data-parallel).
 *   rgb_evt/rgb0_evt [2*R_e,C]; target_acc [R_e] accumulated events at the sampled pixels;
 *   rgb_rgb/rgb0_rgb [n*R_r,C] pose-major; target_rgb [R_r,C]. */
int benerf_loss_stats(const BenerfLossCfg* cfg, const float* rgb_evt, const float* rgb0_evt,
                      const float* target_acc, const float* rgb_rgb, const float* rgb0_rgb,
                      const float* target_rgb, double* stats, benerf_stream_t stream);
/* Pass 2: loss values (losses[8] floats: total, event, event_fine, event_coarse, rgb,
 * rgb_fine, rgb_coarse, 0) and gradients w.r.t. the four rendered colour arrays.
 * Replaces train.py:163-337 + loss/imgloss.py:3-5 + utils/img_utils.py:7-16.
 * losses and any gradient pointer may be NULL.  The gradients of the mean-squared losses (event_threshold > 0, and the
 * blur loss) use the global COUNTS only: a gradient-only call (losses == NULL) may pass stats == NULL then - pass 1 and a
 * values-only call (all gradient pointers NULL) can follow off the critical path. */
int benerf_loss_grads(const BenerfLossCfg* cfg, const double* stats, const float* rgb_evt,
                      const float* rgb0_evt, const float* target_acc, const float* rgb_rgb,
                      const float* rgb0_rgb, const float* target_rgb, float* losses,
                      float* d_rgb_evt, float* d_rgb0_evt, float* d_rgb_rgb, float* d_rgb0_rgb,
                      benerf_stream_t stream);

/* ---------------------------------------------------------------- K7: events ------- */
/* out[y,x] += p for every event (duplicates summed; +-1 values => order independent,
 * exact).  Replaces utils/event_utils.accumulate_events_on_gpu (utils/event_utils.py:246-259).
 * out [H,W] float32 must be zeroed (or hold the running image) by the caller. */
int benerf_event_accumulate(const int32_t* xs, const int32_t* ys, const float* ps, int64_t n,
                            int H, int W, float* out, benerf_stream_t stream);
/* Time-window variant over events kept sorted by ts on the device (model/nerf.py:162-197):
 * accumulates every event with low_t <= ts <= upper_t; window bounds found by binary
 * search in-kernel.  ts [n] float64 ascending. */
int benerf_event_window_accumulate(const int32_t* xs, const int32_t* ys, const float* ps,
                                   const double* ts, int64_t n, double low_t, double upper_t,
                                   int H, int W, float* out, benerf_stream_t stream);
/* `count` distinct pseudo-random pixel indices in [0, n_total), a pure function of (seed, offset) - the
 * np.random.choice(H*W, N_rand, replace=False) of train.py:171-175 / 296-299 as a keyed bijection evaluated at
 * 0..count-1 (no sort; identical on every data-parallel rank).  out [count] int64. */
int benerf_sample_pixels(int64_t n_total, int64_t count, uint64_t seed, uint64_t offset, int64_t* out,
                         benerf_stream_t stream);
/* out[i] = src[idx[i]] (target gather: train.py:177,301-302). src float32 [n_src,width] */
int benerf_gather_rows(const float* src, const int64_t* idx, int64_t n_idx, int width,
                       float* out, benerf_stream_t stream);

/* ---------------------------------------------------------------- K8: optimiser ---- */
/* torch.optim.Adam (defaults beta 0.9/0.999, eps 1e-8) on a flat fp32 buffer; step counts
 * from 1.  Replaces train.py:343-352 (+ model/optimize.py:36-55); the caller applies the
 * exponential LR decay of train.py:355-394 to `lr`.  grad_scale multiplies g first
 * (1/world_size after a sum all-reduce).  skip_if_range: NULL, or the MLP status words (K3): the update is
 * skipped on the device when they show a range violation of the split-f16 mode. */
int benerf_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                     int64_t n, double lr, double beta1, double beta2, double eps, int step,
                     double grad_scale, const uint32_t* skip_if_range, benerf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* BENERF_HIP_H */

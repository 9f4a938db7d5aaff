/*
 * f2nerf_hip.h -- C ABI of libf2nerf_hip.so: the MI355X (gfx950) kernels behind the F2-NeRF
 * rendering hot path (hash-grid encode, ray sampling / early termination, SH encode, ragged per-ray
 * compositing; forward and backward).
 *
 * This is the drop-in boundary.  Every entry point replaces one CUDA kernel launch site (or one
 * run of ATen launches) of SakodaShintaro/f2-nerf; the reference interface each one stands in for
 * is cited as file:line (relative to the reference checkout).  INTEGRATION.md shows the LibTorch
 * wrappers that bind them.
 *
 * Conventions (inherited from the reference's launch sites):
 *   - all pointers are DEVICE pointers to contiguous, caller-owned, caller-allocated buffers;
 *   - no entry point allocates, frees, synchronises or keeps state between calls (calls arrive
 *     from the forward thread and from the autograd engine's device thread); the one exception is
 *     the explicit, process-wide kernel-route table of f2n_set_option below (atomics; defaults =
 *     the production routes; nothing is read from the environment);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *   - return value: F2N_OK (0) or a negative F2N_E_* code; nothing throws;
 *   - counts are elements, never bytes; `idx`/`bounds` are [n_rays, 2] int32 {start, end}.
 */
#ifndef F2NERF_HIP_H_
#define F2NERF_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: f2n_set_option / f2n_get_option (explicit process-wide route table; nothing is read from the
 *    environment any more), f2n_density_margin, f2n_hash_bwd_set_overflow_counter; f2n_hash_bwd_binned
 *    returns F2N_E_UNSUPPORTED for a workspace that cannot hold one tile and applies to n >= 65536. */
#define F2N_ABI_VERSION 2

#define F2N_OK 0
#define F2N_E_INVALID_ARG (-1) /* null pointer, negative count, unsupported L/F/degree ...     */
#define F2N_E_LAUNCH (-2)      /* hipGetLastError() != hipSuccess right after the launch        */
#define F2N_E_UNSUPPORTED (-3) /* valid request the build does not cover (e.g. F not in 1,2,4,8) */

#define F2N_MAX_LEVELS 32

int f2n_abi_version(void);
const char * f2n_status_string(int status);

/* Kernel-route switches for A/B measurements and for tests that must cover every route.  They
 * select between implementations of the SAME operation (results agree within the documented bars);
 * value 0 is always the production default.  Process-wide, lock-free (relaxed atomics): safe to
 * call while other threads launch, takes effect for launches issued afterwards.
 * f2n_set_option returns the previous value, or F2N_E_INVALID_ARG for an unknown key / value. */
#define F2N_OPT_SHADE_FWD 0     /* 0 matrix-core forward, 1 one-sample-per-lane vector kernel       */
#define F2N_OPT_SHADE_BWD 1     /* 0 matrix-core backward, 1 vector (VALU + LDS) backward           */
#define F2N_OPT_SHADE_VARIANT 2 /* matrix-core backward: 0 phases may overlap, 1 phase-fenced; the
                                  matrix-core FORWARD: 0 four waves per SIMD, 3 three, 2 two          */
#define F2N_OPT_RAYTILE 3       /* samples per ray tile of f2n_hash_fwd_raytile: 0 auto, 16, 32     */
#define F2N_OPT_HASH_BWD 4      /* f2n_hash_bwd route: 0 auto, 1 global atomics, 2 LDS-sliced       */
#define F2N_OPT_BWD_COMBINE 5   /* binned backward: 0 combine coarse levels per tile, 1 never       */
#define F2N_OPT_RAYTILE_WALK 6  /* lanes of f2n_hash_fwd_raytile: 0 chosen per tile, 1 across rays at one
                                   sample index, 2 along a ray, 3 across rays in depth order        */
#define F2N_OPT_MARCH 7         /* f2n_density_march: 0 eight rays per wavefront in strides of 8 samples,
                                  1 one ray per wavefront in strides of 64 (round 2), 2 four rays in
                                  strides of 16; same counts                                          */
#define F2N_OPT_BWD_PHASES 8    /* binned backward, overlapping level windows (level_stride < T*F): 0 the
                                  slices of levels that share elements add with float atomics, 1 the
                                  reduce pass runs in ceil(T*F / level_stride) launches of levels that
                                  do not overlap: bit-reproducible also when accumulating into an
                                  existing gradient, 4-25 % slower                                   */
#define F2N_OPT_COUNT 9
int f2n_set_option(int key, int value);
int f2n_get_option(int key);

/* ------------------------------------------------------------------ hash grid (rows A1, A2) --- */

/* feat_pool.to(torch::kFloat16) -- src/hash_3d_anchored.cu:169,198.  RNE cast of n elements. */
int f2n_table_to_f16(const float * table_f32, uint16_t * table_f16, int64_t n, void * stream);

/* Hash3DAnchoredForwardKernel<__half><<<(ceil(n/512), L), 512>>> + out.to(kFloat32)
 * -- src/hash_3d_anchored.cu:60-93,164-178.
 *   pts        [n,3] f32, already contracted
 *   table_f16  f16 pool; level l starts at element level_stride*l (reference: level_stride = T,
 *              quirk Q2), rows of F channels, row index = hash % T
 *   primes     [L,3] i32; bias [L,3] f32; mul [L] f32 (host table, see f2n_level_mul in the host lib)
 *   out        f32 holding the f16-rounded feature; element (p, c) at out[p*out_ld_point + c*out_ld_chan],
 *              c = l*F + k.  Reference layout: out_ld_point = L*F, out_ld_chan = 1.
 *   idx_out    optional [n, L, 8] u32 hash rows (parity tests); NULL in production. */
int f2n_hash_fwd(
  const float * pts, const uint16_t * table_f16, const int32_t * primes, const float * bias,
  const float * mul, float * out, int64_t out_ld_point, int64_t out_ld_chan, uint32_t * idx_out,
  int64_t n, int L, int F, uint32_t T, int64_t level_stride, void * stream);

/* f2n_hash_fwd for a DENSE sample grid: pts [n_rays, S, 3] ray-major as f2n_sample_rays writes it
 * (the input of the first field evaluation, src/renderer.cpp:61), out_cm channel-major
 * [L*F, n_rays*S] (element (p, c) at out_cm[c*n + p], 16-byte aligned).  Same arithmetic, same
 * results; the lanes of a wavefront walk neighbouring RAYS at one sample index instead of
 * consecutive samples of one ray, so image-ordered ray batches (src/renderer.cpp render of a whole
 * view, src/main_functions/train_manager.cpp:170-190) share gathered table lines.
 * F2N_E_UNSUPPORTED unless S is a multiple of 16: call f2n_hash_fwd instead. */
int f2n_hash_fwd_raytile(
  const float * pts, const uint16_t * table_f16, const int32_t * primes, const float * bias,
  const float * mul, float * out_cm, int n_rays, int S, int L, int F, uint32_t T,
  int64_t level_stride, void * stream);

/* Hash3DAnchoredBackwardKernel<__half> + the /grad_scale epilogue
 * -- src/hash_3d_anchored.cu:95-145,190-215.
 *   grad_out    element (p, c) at grad_out[p*g_ld_point + c*g_ld_chan], f32
 *   table_grad  f32, same element indexing as table_f16, ACCUMULATED INTO (caller zeroes it):
 *               += f16(f16(grad_scale*g) * w_d) / grad_scale per corner.  The reference accumulates
 *               with order-dependent f16 atomics; here the sum is kept in f32 (documented in
 *               DESIGN.md).  grad_scale must be a power of two (reference: 128).
 *   pts_grad    [n,3] f32 or NULL (NULL = points need no gradient: the training case).  Overwritten. */
int f2n_hash_bwd(
  const float * pts, const uint16_t * table_f16, const int32_t * primes, const float * bias,
  const float * mul, const float * grad_out, int64_t g_ld_point, int64_t g_ld_chan,
  float * table_grad, float * pts_grad, int64_t n, int L, int F, uint32_t T, int64_t level_stride,
  float grad_scale, void * stream);

/* Same operation as f2n_hash_bwd without the point gradient, for large batches: contributions are
 * binned by table slice into a caller-provided workspace and summed EXACTLY (64-bit fixed point) in
 * LDS, so the scattered f16/f32 atomics of Hash3DAnchoredBackwardKernel
 * (src/hash_3d_anchored.cu:129-137) disappear and the result does not depend on summation order --
 * with one condition: a record that finds an LDS queue, a workspace region or a run full, or that
 * holds a non-finite half, is applied directly with a global float atomic, and where two such
 * records meet on one element the last bits depend on their order (as the reference's own atomics
 * do everywhere).  f2n_hash_bwd_set_overflow_counter() makes the passes count those records.
 * Tables with more than 64 LDS-sized slices per level (T*F > 2^20, e.g. T = 2^22, F = 8) take a
 * second binning pass; coarse levels whose cells are shared by the points of a tile are combined
 * before they are binned (F2N_OPT_BWD_COMBINE).
 * f2n_hash_bwd_workspace_bytes returns the recommended workspace size (at most 64 GiB), or 0 when
 * the binned path does not apply to (n, L, F, T): n < 65536 or T*F > 2^26 -- use f2n_hash_bwd then.
 * workspace: device memory, 256-byte aligned, contents undefined on entry and exit.  A smaller
 * workspace makes the passes run in several rounds over the points (same results);
 * F2N_E_UNSUPPORTED when it cannot hold one 1024-point tile. */
int64_t f2n_hash_bwd_workspace_bytes(int64_t n, int L, int F, uint32_t T);
/* Overflow accounting of f2n_hash_bwd_binned (process-wide, like f2n_set_option): a caller-owned
 * device word (8-byte aligned) that every later call increments once per record it applied with a
 * global float atomic instead of the exact sum (a full queue / region / run, a combined sum beyond
 * the f16 range) -- 0 afterwards means the result is independent of summation order, bit for bit.
 * NULL (the default) switches the counting off.  The reference's kernel has no such distinction:
 * every one of its adds is an order-dependent atomic (src/hash_3d_anchored.cu:129-137). */
int f2n_hash_bwd_set_overflow_counter(uint64_t * device_counter);
int f2n_hash_bwd_binned(
  const float * pts, const int32_t * primes, const float * bias, const float * mul,
  const float * grad_out, int64_t g_ld_point, int64_t g_ld_chan, float * table_grad, int64_t n,
  int L, int F, uint32_t T, int64_t level_stride, float grad_scale, void * workspace,
  int64_t workspace_bytes, void * stream);

/* Scene contraction of Hash3DAnchored::query -- src/hash_3d_anchored.cpp:79-82 (8 ATen launches):
 *   x = p if |p| <= 1 else (2 - 1/|p|) * p/|p|, evaluated as the reference's mask expression
 *   (|p| == 0 gives NaN, quirk Q6).  The backward is the Jacobian-vector product autograd builds. */
int f2n_contract_fwd(const float * pts, float * x, int64_t n, void * stream);
int f2n_contract_bwd(const float * pts, const float * dx, float * dpts, int64_t n, void * stream);

/* ------------------------------------------------------------------ SH encode (row A6) -------- */

/* SHKernel<<<ceil(n/512), 512>>> -- src/sh_shader.cu:11-115.  dirs [n,3] -> out [n, degree^2],
 * degree 1..8 (F2N_E_UNSUPPORTED above); out 16-byte aligned. */
int f2n_sh_encode(const float * dirs, float * out, int64_t n, int degree, void * stream);

/* ------------------------------------------------------------------ ragged per-ray ops (A7) --- */

/* FlexSumForwardKernel / BackwardKernel -- src/CustomOps/FlexOps.cu:6-27,98-153 */
int f2n_seg_sum_fwd(const float * val, const int32_t * idx, float * sum, int n_rays, void * stream);
int f2n_seg_sum_bwd(const float * dsum, const int32_t * idx, float * dval, int n_rays, void * stream);
/* FlexSumVecForwardKernel / BackwardKernel -- src/CustomOps/FlexOps.cu:29-54 */
int f2n_seg_sum_vec_fwd(
  const float * val, const int32_t * idx, float * sum, int n_rays, int vec, void * stream);
int f2n_seg_sum_vec_bwd(
  const float * dsum, const int32_t * idx, float * dval, int n_rays, int vec, void * stream);
/* FlexAccumulateSumForwardKernel / BackwardKernel -- src/CustomOps/FlexOps.cu:56-94,155-199 */
int f2n_seg_scan_fwd(
  const float * val, const int32_t * idx, float * sum, int n_rays, int include_this, void * stream);
int f2n_seg_scan_bwd(
  const float * dsum, const int32_t * idx, float * dval, int n_rays, int include_this,
  void * stream);

/* WeightVarLossForwardKernel / BackwardKernel -- src/CustomOps/CustomOps.cu:13-67 (row A10) */
int f2n_weight_var_fwd(
  const float * weights, const int32_t * idx, float * out_vars, int n_rays, void * stream);
int f2n_weight_var_bwd(
  const float * weights, const int32_t * idx, const float * dvars, float * dw, int n_rays,
  void * stream);

/* ------------------------------------------------------------------ scatter (row A9) ---------- */

/* ScatterIdxKernal -- src/CustomOps/Scatter.cu:111-132 */
int f2n_scatter_idx(
  const int32_t * idx, const int32_t * emb_idx, int32_t * all_emb_idx, int n_rays, void * stream);
/* ScatterAddFuncForward (+ the clone at :63) -- src/CustomOps/Scatter.cu:11-19,45-70.
 * sum[p,c] = to_add[p,c] + emb[scatter_idx[p], c]; sum may alias to_add. */
int f2n_scatter_add_fwd(
  const float * emb, const int32_t * scatter_idx, const float * to_add, float * sum, int64_t n_all,
  int C, void * stream);
/* ScatterAddFuncBackwardBlock + torch::sum(dim 1) -- src/CustomOps/Scatter.cu:21-41,72-101.
 * demb [n_emb, C] is overwritten (zeroed, then accumulated) on `stream`. */
int f2n_scatter_add_bwd(
  const int32_t * scatter_idx, const float * dsum, float * demb, int64_t n_all, int n_emb, int C,
  void * stream);

/* ------------------------------------------------------------------ sampler (rows A4, A5) ----- */

/* get_rays_from_pose -- src/rays.cpp:7-28, for both callers: a view's pixel grid
 * (src/renderer.cpp:153-172, src/dataset.cpp:128-146) and the random training batch with one camera
 * per ray (src/dataset.cpp:150-171, without its index_select of poses and intrinsics).
 *   poses      [n_cams] blocks of pose_ld floats: a row-major [3,4] (pose_ld = 12) or [4,4] (16)
 *   intrinsics [n_cams, 3, 3]
 *   cam_idx    [n] i32 camera of each ray; NULL: camera 0 when n_cams == 1, camera r when n_cams == n
 *   ij         [n, 2] i32 (row, col); NULL: ray r is pixel first_pixel + r of a `width`-wide image
 *   rays_o, rays_d [n, 3]: origin = t, dir = R . ((col+.5-cx)/fx, -(row+.5-cy)/fy, -1), not normalised */
int f2n_gen_rays(
  const float * poses, int pose_ld, const float * intrinsics, int64_t n_cams,
  const int32_t * cam_idx, const int32_t * ij, int64_t first_pixel, int width, float * rays_o,
  float * rays_d, int64_t n, void * stream);

/* PtsSampler::get_samples (about 20 ATen launches) -- src/points_sampler.cpp:20-64.
 *   noise   [n_rays, S] f32 step multipliers (TRAIN: U[0.5,1.5)), or NULL for all-ones (VALIDATE)
 *   outputs pts [n_rays*S, 3], dirs [n_rays*S, 3], dt [n_rays*S], t [n_rays*S], bounds [n_rays,2]
 *   S = MAX_SAMPLE_PER_RAY (1024), step = SAMPLE_L (1/256) in the reference (src/points_sampler.hpp:15,39) */
int f2n_sample_rays(
  const float * rays_o, const float * rays_d, const float * noise, float * pts, float * dirs,
  float * dt, float * t, int32_t * bounds, int n_rays, int S, float step, void * stream);

/* Early-stop pass of Renderer::render fused into one march -- src/renderer.cpp:58-90 together with
 * src/points_sampler.cpp:20-64, src/hash_3d_anchored.cpp:79-86 (contraction, hash encode, row 0 of the
 * Linear) and src/CustomOps/CustomOps.cpp:10-14 (TruncExp fwd).  One wavefront walks one ray in 64-sample
 * strides and stops at the first stride whose transmittance exp(-sum sigma*dt) falls to
 * <= t_thresh; kept[r] = number of leading samples with T > t_thresh (the mask of :68 is a prefix).
 *   w0 [L*F] = mlp.weight[0, :], b0 = mlp.bias[0]; density = exp(w0.enc + b0 - density_shift) */
int f2n_density_march(
  const float * rays_o, const float * rays_d, const float * noise, const uint16_t * table_f16,
  const int32_t * primes, const float * bias, const float * mul, const float * w0, const float * b0,
  int32_t * kept, int n_rays, int S, float step, int L, int F, uint32_t T, int64_t level_stride,
  float t_thresh, float density_shift, void * stream);

/* The same keep-prefix as f2n_density_march (src/renderer.cpp:61-68 semantics), computed from an
 * encoding that already exists for ALL n_rays*S samples: enc_cm channel-major [C, n_rays*S] as
 * f2n_hash_fwd writes it, dt [n_rays*S] as f2n_sample_rays writes it.  Identical FMA chain, scan and
 * threshold test, hence identical counts.  The host picks this route when most samples survive, so
 * that the field is evaluated once (level-major, L2-friendly) and reused by the shading pass --
 * the reference evaluates it twice (src/renderer.cpp:61 and :92). */
int f2n_density_scan(
  const float * enc_cm, int C, const float * dt, const float * w0, const float * b0, int32_t * kept,
  int n_rays, int S, float t_thresh, float density_shift, void * stream);

/* A cheap sufficient test for "f2n_density_scan would keep every sample of every ray": logit
 * [n_rays*S] are density logits of the dense grid (any evaluation of the field head, e.g. the fused
 * per-sample network's), dt as above.  flag[0] (int32, zeroed by the caller) is set when some ray's
 * total optical depth sum_k exp(logit_k - density_shift) dt_k is not below depth_limit; choose
 * depth_limit = -ln(t_thresh) - margin with a margin far above the rounding differences between two
 * evaluations of the logit (the host uses 0.5).  Part of the same block src/renderer.cpp:61-68. */
int f2n_density_margin(
  const float * logit, const float * dt, int32_t * flag, int n_rays, int S, float density_shift,
  float depth_limit, void * stream);

/* Channel-major companion of the four index() gathers at src/renderer.cpp:71-74 for [C, n] tensors:
 * dst[c, bounds[r].start + k] = src[c, r*S + k] for k < bounds[r].end - bounds[r].start. */
int f2n_compact_rows_cm(
  const float * src, int64_t n_src, float * dst, int64_t n_dst, int C, const int32_t * bounds,
  int n_rays, int S, void * stream);

/* cumsum of the per-ray counts -> bounds (src/renderer.cpp:76-83).  total[0] = sum(kept).
 * Single-workgroup scan; n_rays <= 2^24. */
int f2n_bounds_from_counts(
  const int32_t * kept, int32_t * bounds, int32_t * total, int n_rays, void * stream);

/* where(mask) + the four index() gathers (src/renderer.cpp:69-74) without materialising the dense
 * sample arrays: re-derives the first (end-start) samples of each ray straight into the compacted
 * outputs pts/dirs [n_kept,3], dt/t [n_kept]. */
int f2n_sample_compact(
  const float * rays_o, const float * rays_d, const float * noise, const int32_t * bounds,
  float * pts, float * dirs, float * dt, float * t, int n_rays, int S, float step, void * stream);

/* ------------------------------------------------------------------ compositing (rows A7, A8) - */

/* src/renderer.cpp:93,107-118 as one pass per ray:
 *   sigma = exp(logit - density_shift); s = sigma*dt; alpha = 1-exp(-s); T = exp(-excl_scan(s));
 *   w = T*alpha; T_last = exp(-sum s); C = sum w*rgb + T_last*bg; D = sum w*(t+t_shift)/(1-T_last+1e-4)
 *   logit element i at logit[i*logit_ld] (column 0 of the [n,16] field output: logit_ld = 16)
 *   outputs colors [n_rays,3], depths [n_rays], weights [n], last_trans [n_rays] (saved for bwd) */
int f2n_composite_fwd(
  const float * logit, int64_t logit_ld, const float * rgb, const float * dt, const float * t,
  const int32_t * bounds, const float * bg, float * colors, float * depths, float * weights,
  float * last_trans, int n_rays, float density_shift, float t_shift, void * stream);

/* Backward of the above = FlexSum/FlexSumVec/FlexAccumulateSum backward kernels plus the ATen
 * element-wise backward and TruncExp::backward (src/CustomOps/CustomOps.cpp:16-20: exp(clamp(x,-100,5))).
 *   in : d_colors [n_rays,3], d_depths [n_rays], d_weights [n] (NULL = zeros)
 *   out: d_logit [n] (dense), d_rgb [n,3] */
int f2n_composite_bwd(
  const float * logit, int64_t logit_ld, const float * rgb, const float * dt, const float * t,
  const int32_t * bounds, const float * bg, const float * weights, const float * last_trans,
  const float * d_colors, const float * d_depths, const float * d_weights, float * d_logit,
  float * d_rgb, int n_rays, float density_shift, float t_shift, void * stream);

/* ------------------------------------------------------------------ fused per-sample network -- */

/* Everything between the hash encode and the compositing, one kernel per direction:
 *   h = mlp(enc)                       Linear(C->16) of Hash3DAnchored::query -- src/hash_3d_anchored.cpp:86
 *   logit = h[0]                       density logit                          -- src/renderer.cpp:93
 *   X = cat(1, h[1:16]) (+ app_emb[sample_img]) ++ SH16(dirs)                 -- src/renderer.cpp:95-104,
 *                                                                                src/sh_shader.cpp:24-25
 *   rgb = (1+2e)*sigmoid(mlp2(relu(mlp1(X)))) - e, e = 1e-3                   -- src/sh_shader.cpp:26-28
 * i.e. the three nn::Linear GEMMs, torch::cat x2, ScatterAdd (src/CustomOps/Scatter.cu:11-19),
 * SHKernel (src/sh_shader.cu:11-103) and the element-wise tail.  enc_cm / d_enc_cm are
 * channel-major [C, n] (C = L*F in {8,16,32,64}); w_h [16,C], w1 [64,32], w2 [3,64] row-major as
 * nn::Linear stores them; sample_img [n] image id per sample or NULL (no appearance embedding).
 * pre_cm: optional [64, n] output of the hidden layer's pre-activations (what autograd would have
 * saved for the ReLU); give it to f2n_shade_bwd and that kernel loads them instead of recomputing. */
int f2n_shade_fwd(
  const float * enc_cm, int C, const float * dirs, const int32_t * sample_img, const float * w_h,
  const float * b_h, const float * w1, const float * b1, const float * w2, const float * b2,
  const float * app_emb, float * logit, float * rgb, float * pre_cm, int64_t n, void * stream);

/* Backward of the above (recomputes the forward per sample).  d_enc_cm is overwritten; the seven
 * parameter gradients are ACCUMULATED INTO (caller zeroes them); g_app_emb may be NULL when
 * app_emb / sample_img are; pre_cm = the forward's optional output or NULL (recompute).  Replaces the
 * autograd chain of the ops listed above, including
 * ScatterAddFuncBackwardBlock -- src/CustomOps/Scatter.cu:21-41,72-101. */
int f2n_shade_bwd(
  const float * enc_cm, int C, const float * dirs, const int32_t * sample_img, const float * w_h,
  const float * b_h, const float * w1, const float * b1, const float * w2, const float * b2,
  const float * app_emb, const float * d_logit, const float * d_rgb, float * d_enc_cm,
  float * g_w_h, float * g_b_h, float * g_w1, float * g_b1, float * g_w2, float * g_b2,
  float * g_app_emb, const float * pre_cm, int64_t n, void * stream);

/* ------------------------------------------------------------------ optimiser (section 8f) ----- */

/* One fused pass of torch::optim::Adam::step() over one f32 parameter tensor -- the call at
 * src/main_functions/train_manager.cpp:106 with the options of src/hash_3d_anchored.cpp:90-114
 * (betas 0.9/0.99, eps 1e-15, weight decay 0 for the table, 1e-6 elsewhere) -- that also writes the
 * RNE f16 copy of the updated parameter when shadow_f16 != NULL (the feat_pool.to(kFloat16) of
 * src/hash_3d_anchored.cu:169,198, done once here instead of three times per iteration).
 * `step` is the 1-based step count (bias corrections 1 - beta^step are formed on the host). */
int f2n_adam_step(
  float * param, const float * grad, float * exp_avg, float * exp_avg_sq, uint16_t * shadow_f16,
  int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
  void * stream);

/* The loss of the training iteration -- src/main_functions/train_manager.cpp:78-96:
 *   color_loss = mean sqrt((colors - gt)^2 + 1e-4), var_loss = mean sqrt(var + 1e-2),
 *   loss = color_loss + var_weight * var_loss, sq_err_sum = sum (colors - gt)^2 (PSNR, :95-96).
 * colors, gt [n_rays, 3]; var [n_rays] (CustomOps::WeightVar).  Writes out4 = {loss, color_loss,
 * var_loss, sq_err_sum} and the gradients of loss w.r.t. colors / var (closed forms; multiply by the
 * upstream gradient).  partial: scratch of f2n_loss_workspace_floats(n_rays) floats.  Deterministic. */
int64_t f2n_loss_workspace_floats(int n_rays);
int f2n_loss_fwd(
  const float * colors, const float * gt, const float * var, int n_rays, float var_weight,
  float * d_colors, float * d_var, float * partial, float * out4, void * stream);

#ifdef __cplusplus
}
#endif

#endif /* F2NERF_HIP_H_ */

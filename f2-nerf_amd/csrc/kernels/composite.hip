// composite.hip -- volumetric alpha compositing over ragged per-ray sample runs, forward and
// backward (arithmetic per appendix A.7 / A.8).
//
// One launch replaces, per direction, the chain the reference builds out of TruncExp, six
// element-wise ATen ops, FlexOps::AccumulateSum and three FlexOps::Sum calls
// (reference src/renderer.cpp:93,107-118 forward; src/CustomOps/FlexOps.cu:18-27,43-54,76-94 and
// src/CustomOps/CustomOps.cpp:16-20 backward).
//
// One 64-lane wavefront per ray; 64-sample strides are coalesced; the exclusive optical-depth scan
// and the backward suffix scan are DPP wave scans with a scalar carry.
#include "common.hiph"

namespace
{

__device__ __forceinline__ int ray_of_wave()
{
  return (int)blockIdx.x * F2N_WAVES_PER_BLOCK + (int)(threadIdx.x >> 6);
}

__global__ __launch_bounds__(F2N_BLOCK) void composite_fwd_kernel(
  const float * __restrict__ logit, int64_t logit_ld, const float * __restrict__ rgb,
  const float * __restrict__ dt, const float * __restrict__ t, const int32_t * __restrict__ bounds,
  const float * __restrict__ bg, float * __restrict__ colors, float * __restrict__ depths,
  float * __restrict__ weights, float * __restrict__ last_trans, int n_rays, float density_shift,
  float t_shift)
{
  const int r = ray_of_wave();
  if (r >= n_rays) return;
  const int lane = lane_id();
  const int s = bounds[2 * r], e = bounds[2 * r + 1];
  float depth_carry = 0.f;
  float cr = 0.f, cg = 0.f, cb = 0.f, cd = 0.f;
  for (int c = s; c < e; c += F2N_WAVE) {
    const int i = c + lane;
    const bool valid = i < e;
    float sec = 0.f;
    if (valid) sec = expf(logit[(int64_t)i * logit_ld] - density_shift) * dt[i];
    const float incl = wave_incl_scan(sec);
    const float trans = expf(-(depth_carry + wave_shift_up1(incl, 0.f)));
    if (valid) {
      const float alpha = 1.f - expf(-sec);
      const float w = trans * alpha;
      weights[i] = w;
      cr = fmaf(w, rgb[3 * (int64_t)i], cr);
      cg = fmaf(w, rgb[3 * (int64_t)i + 1], cg);
      cb = fmaf(w, rgb[3 * (int64_t)i + 2], cb);
      cd = fmaf(w, t[i] + t_shift, cd);
    }
    depth_carry += wave_bcast_last(incl);
  }
  cr = wave_sum(cr);
  cg = wave_sum(cg);
  cb = wave_sum(cb);
  cd = wave_sum(cd);
  if (lane == 0) {
    const float tl = expf(-depth_carry);
    last_trans[r] = tl;
    colors[3 * r] = fmaf(tl, bg[3 * r], cr);
    colors[3 * r + 1] = fmaf(tl, bg[3 * r + 1], cg);
    colors[3 * r + 2] = fmaf(tl, bg[3 * r + 2], cb);
    depths[r] = cd / (1.f - tl + 1e-4f);
  }
}

// Backward (appendix A.8).  With dw_k = dC.rgb_k + dD*t'_k/den + dW_k:
//   d_rgb_k = w_k dC
//   d_sec_k = dw_k (T_k - w_k)  -  sum_{j>k} dw_j w_j  -  T_last (dC.bg + dD*Nd/den^2)
//   d_logit_k = d_sec_k * dt_k * exp(clamp(logit_k - shift, -100, 5))        (TruncExp::backward)
// T_k is re-derived by the same forward scan as composite_fwd_kernel (pass 1, parked in d_logit),
// then pass 2 walks the strides backwards with a suffix scan of dw_j w_j.  Both passes use the
// same lane <-> sample mapping, so each lane re-reads only what it wrote itself.
__global__ __launch_bounds__(F2N_BLOCK) void composite_bwd_kernel(
  const float * __restrict__ logit, int64_t logit_ld, const float * __restrict__ rgb,
  const float * __restrict__ dt, const float * __restrict__ t, const int32_t * __restrict__ bounds,
  const float * __restrict__ bg, const float * __restrict__ weights,
  const float * __restrict__ last_trans, const float * __restrict__ d_colors,
  const float * __restrict__ d_depths, const float * __restrict__ d_weights,
  float * __restrict__ d_logit, float * __restrict__ d_rgb, int n_rays, float density_shift,
  float t_shift)
{
  const int r = ray_of_wave();
  if (r >= n_rays) return;
  const int lane = lane_id();
  const int s = bounds[2 * r], e = bounds[2 * r + 1];
  if (s >= e) return;
  const float dcr = d_colors[3 * r], dcg = d_colors[3 * r + 1], dcb = d_colors[3 * r + 2];
  const float dd = d_depths[r];
  const float tl = last_trans[r];
  const float den = 1.f - tl + 1e-4f;

  // pass 1: transmittance per sample (parked in d_logit) and Nd = sum w*t'
  float depth_carry = 0.f, nd = 0.f;
  for (int c = s; c < e; c += F2N_WAVE) {
    const int i = c + lane;
    const bool valid = i < e;
    float sec = 0.f;
    if (valid) sec = expf(logit[(int64_t)i * logit_ld] - density_shift) * dt[i];
    const float incl = wave_incl_scan(sec);
    const float trans = expf(-(depth_carry + wave_shift_up1(incl, 0.f)));
    if (valid) {
      d_logit[i] = trans;
      nd = fmaf(weights[i], t[i] + t_shift, nd);
    }
    depth_carry += wave_bcast_last(incl);
  }
  nd = wave_sum(nd);
  const float d_tl = fmaf(dcb, bg[3 * r + 2], fmaf(dcg, bg[3 * r + 1], dcr * bg[3 * r])) +
                     dd * nd / (den * den);
  const float d_total = -tl * d_tl;  // through T_last = exp(-sum sec)
  const float dd_over_den = dd / den;

  // pass 2: strides from the ray's end; suffix scan = prefix scan on the lane-reversed stride
  const int n = e - s;
  const int last_c = s + ((n - 1) / F2N_WAVE) * F2N_WAVE;
  float suffix_carry = 0.f;  // sum of dw_j w_j over later strides
  for (int c = last_c; c >= s; c -= F2N_WAVE) {
    const int i = c + lane;
    const bool valid = i < e;
    float w = 0.f, dw = 0.f, trans = 0.f;
    if (valid) {
      w = weights[i];
      trans = d_logit[i];
      const float r0 = rgb[3 * (int64_t)i], r1 = rgb[3 * (int64_t)i + 1],
                  r2 = rgb[3 * (int64_t)i + 2];
      dw = fmaf(dcb, r2, fmaf(dcg, r1, dcr * r0)) + dd_over_den * (t[i] + t_shift);
      if (d_weights) dw += d_weights[i];
      d_rgb[3 * (int64_t)i] = w * dcr;
      d_rgb[3 * (int64_t)i + 1] = w * dcg;
      d_rgb[3 * (int64_t)i + 2] = w * dcb;
    }
    const float q = dw * w;
    const float incl_rev = wave_incl_scan(wave_reverse(q));       // lane j: sum of the last j+1
    const float excl_rev = wave_shift_up1(incl_rev, 0.f);          // lane j: sum of the last j
    const float later = suffix_carry + wave_reverse(excl_rev);     // sum_{j>k} within + beyond
    if (valid) {
      const float x = logit[(int64_t)i * logit_ld] - density_shift;
      const float d_sec = fmaf(dw, trans - w, -later) + d_total;
      d_logit[i] = d_sec * dt[i] * expf(fminf(fmaxf(x, -100.f), 5.f));
    }
    suffix_carry += wave_bcast_last(incl_rev);
  }
}

}  // namespace

extern "C" int f2n_composite_fwd(
  const float * logit, int64_t logit_ld, const float * rgb, const float * dt, const float * t,
  const int32_t * bounds, const float * bg, float * colors, float * depths, float * weights,
  float * last_trans, int n_rays, float density_shift, float t_shift, void * stream)
{
  if (n_rays < 0 || logit_ld < 1) return F2N_E_INVALID_ARG;
  if (n_rays == 0) return F2N_OK;
  if (!bounds || !bg || !colors || !depths || !last_trans) return F2N_E_INVALID_ARG;
  hipLaunchKernelGGL(
    composite_fwd_kernel, dim3(f2n_div_up(n_rays, F2N_WAVES_PER_BLOCK)), dim3(F2N_BLOCK), 0,
    (hipStream_t)stream, logit, logit_ld, rgb, dt, t, bounds, bg, colors, depths, weights,
    last_trans, n_rays, density_shift, t_shift);
  return f2n_launch_status();
}

extern "C" int f2n_composite_bwd(
  const float * logit, int64_t logit_ld, const float * rgb, const float * dt, const float * t,
  const int32_t * bounds, const float * bg, const float * weights, const float * last_trans,
  const float * d_colors, const float * d_depths, const float * d_weights, float * d_logit,
  float * d_rgb, int n_rays, float density_shift, float t_shift, void * stream)
{
  if (n_rays < 0 || logit_ld < 1) return F2N_E_INVALID_ARG;
  if (n_rays == 0) return F2N_OK;
  if (!bounds || !bg || !last_trans || !d_colors || !d_depths) return F2N_E_INVALID_ARG;
  hipLaunchKernelGGL(
    composite_bwd_kernel, dim3(f2n_div_up(n_rays, F2N_WAVES_PER_BLOCK)), dim3(F2N_BLOCK), 0,
    (hipStream_t)stream, logit, logit_ld, rgb, dt, t, bounds, bg, weights, last_trans, d_colors,
    d_depths, d_weights, d_logit, d_rgb, n_rays, density_shift, t_shift);
  return f2n_launch_status();
}

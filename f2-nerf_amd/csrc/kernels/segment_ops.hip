// segment_ops.hip -- ragged per-ray reductions / scans over idx_start_end[n_rays,2]
// (SURVEY.md rows A7, A10).
//
// Replaces the six FlexOps kernels (reference src/CustomOps/FlexOps.cu:6-94) and the two WeightVar
// kernels (src/CustomOps/CustomOps.cu:13-67).  The reference gives each ray ONE thread walking its
// samples serially (stride-1 per thread = stride-len across the wavefront: uncoalesced, and 512 rays
// fill one workgroup).  Here each ray gets one 64-lane wavefront: a 64-sample stride is one
// coalesced 256-byte access, sums/scans across the stride are DPP wave scans, and a scalar carry
// links strides.  Floating-point sums are therefore re-associated with respect to the reference's
// serial order (1e-7-level drift; parity tolerance in tests/).
#include "common.hiph"

namespace
{

__device__ __forceinline__ int ray_of_wave()
{
  return (int)blockIdx.x * F2N_WAVES_PER_BLOCK + (int)(threadIdx.x >> 6);
}

// ---- FlexSum ------------------------------------------------------------------------------------

__global__ __launch_bounds__(F2N_BLOCK) void seg_sum_fwd_kernel(
  const float * __restrict__ val, const int32_t * __restrict__ idx, float * __restrict__ sum,
  int n_rays)
{
  const int r = ray_of_wave();
  if (r >= n_rays) return;
  const int lane = lane_id();
  const int s = idx[2 * r], e = idx[2 * r + 1];
  float acc = 0.f;
  for (int i = s + lane; i < e; i += F2N_WAVE) acc += val[i];
  acc = wave_sum(acc);
  if (lane == 0) sum[r] = acc;
}

__global__ __launch_bounds__(F2N_BLOCK) void seg_sum_bwd_kernel(
  const float * __restrict__ dsum, const int32_t * __restrict__ idx, float * __restrict__ dval,
  int n_rays)
{
  const int r = ray_of_wave();
  if (r >= n_rays) return;
  const int lane = lane_id();
  const int s = idx[2 * r], e = idx[2 * r + 1];
  const float g = dsum[r];
  for (int i = s + lane; i < e; i += F2N_WAVE) dval[i] = g;
}

// vec variant: the segment is a contiguous run of (e-s)*vec floats; lane j of each stride reads
// element j of the run, whose channel is (j % vec).  Per-channel totals are formed by letting lane
// c (< vec) pick up the partial sums of all lanes congruent to c.
__global__ __launch_bounds__(F2N_BLOCK) void seg_sum_vec_fwd_kernel(
  const float * __restrict__ val, const int32_t * __restrict__ idx, float * __restrict__ sum,
  int n_rays, int vec)
{
  extern __shared__ float lds[];  // [waves][64]
  const int r = ray_of_wave();
  const int lane = lane_id();
  float * my = lds + (threadIdx.x >> 6) * F2N_WAVE;
  if (r < n_rays) {
    const int s = idx[2 * r], e = idx[2 * r + 1];
    const int64_t base = (int64_t)s * vec;
    const int64_t len = (int64_t)(e - s) * vec;
    // stride = largest multiple of vec that fits the wave, so a lane always sees one channel
    const int stride = (F2N_WAVE / vec) * vec;
    float acc = 0.f;
    if (lane < stride)
      for (int64_t j = lane; j < len; j += stride) acc += val[base + j];
    my[lane] = acc;
    __builtin_amdgcn_wave_barrier();
    if (lane < vec) {
      float tot = 0.f;
      for (int j = lane; j < stride; j += vec) tot += my[j];
      sum[(int64_t)r * vec + lane] = tot;
    }
  }
}

__global__ __launch_bounds__(F2N_BLOCK) void seg_sum_vec_bwd_kernel(
  const float * __restrict__ dsum, const int32_t * __restrict__ idx, float * __restrict__ dval,
  int n_rays, int vec)
{
  const int r = ray_of_wave();
  if (r >= n_rays) return;
  const int lane = lane_id();
  const int s = idx[2 * r], e = idx[2 * r + 1];
  const int64_t base = (int64_t)s * vec;
  const int64_t len = (int64_t)(e - s) * vec;
  const int stride = (F2N_WAVE / vec) * vec;
  if (lane < stride) {
    const float g = dsum[(int64_t)r * vec + (lane % vec)];
    for (int64_t j = lane; j < len; j += stride) dval[base + j] = g;
  }
}

// ---- FlexAccumulateSum --------------------------------------------------------------------------

template <bool INCLUDE_THIS>
__global__ __launch_bounds__(F2N_BLOCK) void seg_scan_fwd_kernel(
  const float * __restrict__ val, const int32_t * __restrict__ idx, float * __restrict__ sum,
  int n_rays)
{
  const int r = ray_of_wave();
  if (r >= n_rays) return;
  const int lane = lane_id();
  const int s = idx[2 * r], e = idx[2 * r + 1];
  float carry = 0.f;
  for (int c = s; c < e; c += F2N_WAVE) {
    const int i = c + lane;
    const float v = (i < e) ? val[i] : 0.f;
    const float incl = wave_incl_scan(v);
    const float res = carry + (INCLUDE_THIS ? incl : wave_shift_up1(incl, 0.f));
    if (i < e) sum[i] = res;
    carry += wave_bcast_last(incl);
  }
}

// dval[i] = sum_{j > i} dsum[j]  (exclusive)  or  sum_{j >= i}  (inclusive): a suffix scan, done
// as a prefix scan over the lane-reversed stride, walking strides from the ray's end.
template <bool INCLUDE_THIS>
__global__ __launch_bounds__(F2N_BLOCK) void seg_scan_bwd_kernel(
  const float * __restrict__ dsum, const int32_t * __restrict__ idx, float * __restrict__ dval,
  int n_rays)
{
  const int r = ray_of_wave();
  if (r >= n_rays) return;
  const int lane = lane_id();
  const int s = idx[2 * r], e = idx[2 * r + 1];
  float carry = 0.f;
  for (int hi = e; hi > s; hi -= F2N_WAVE) {
    const int i = hi - 1 - lane;  // lane 0 = last element of the stride
    const float v = (i >= s) ? dsum[i] : 0.f;
    const float incl = wave_incl_scan(v);
    const float res = carry + (INCLUDE_THIS ? incl : wave_shift_up1(incl, 0.f));
    if (i >= s) dval[i] = res;
    carry += wave_bcast_last(incl);
  }
}

// ---- WeightVar ----------------------------------------------------------------------------------

struct VarStats
{
  float mean, wsum;
};

__device__ __forceinline__ VarStats var_stats(const float * __restrict__ w, int s, int e, int lane)
{
  float m = 0.f, ws = 0.f;
  for (int i = lane; i + s < e; i += F2N_WAVE) {
    const float wi = w[i + s];
    m = fmaf(wi, (float)i / 16.f, m);
    ws += wi;
  }
  VarStats st;
  st.wsum = 1e-6f + wave_sum(ws);
  st.mean = wave_sum(m) / st.wsum;
  return st;
}

__global__ __launch_bounds__(F2N_BLOCK) void weight_var_fwd_kernel(
  const float * __restrict__ w, const int32_t * __restrict__ idx, float * __restrict__ out_vars,
  int n_rays)
{
  const int r = ray_of_wave();
  if (r >= n_rays) return;
  const int lane = lane_id();
  const int s = idx[2 * r], e = idx[2 * r + 1];
  if (s >= e) {
    if (lane == 0) out_vars[r] = 0.f;
    return;
  }
  const VarStats st = var_stats(w, s, e, lane);
  float var = 0.f;
  for (int i = lane; i + s < e; i += F2N_WAVE) {
    const float b = (float)i / 16.f - st.mean;
    var = fmaf(w[i + s] * b, b, var);
  }
  var = wave_sum(var);
  if (lane == 0) out_vars[r] = var;
}

__global__ __launch_bounds__(F2N_BLOCK) void weight_var_bwd_kernel(
  const float * __restrict__ w, const int32_t * __restrict__ idx, const float * __restrict__ dvars,
  float * __restrict__ dw, int n_rays)
{
  const int r = ray_of_wave();
  if (r >= n_rays) return;
  const int lane = lane_id();
  const int s = idx[2 * r], e = idx[2 * r + 1];
  if (s >= e) return;
  const VarStats st = var_stats(w, s, e, lane);
  float tmp = 0.f;
  for (int i = lane; i + s < e; i += F2N_WAVE) {
    const float b = (float)i / 16.f - st.mean;
    tmp = fmaf(w[i + s] * 2.f, b, tmp);
  }
  tmp = wave_sum(tmp);
  const float g = dvars[r];
  for (int i = lane; i + s < e; i += F2N_WAVE) {
    const float x = (float)i / 16.f;
    const float b = x - st.mean;
    // as coded in the reference (quirk Q9): bias^2 - tmp*x/wsum
    dw[i + s] = g * fmaf(b, b, tmp * -x / st.wsum);
  }
}

inline dim3 ray_grid(int n_rays) { return dim3(f2n_div_up(n_rays, F2N_WAVES_PER_BLOCK)); }

}  // namespace

#define F2N_RAY_LAUNCH(kernel, shmem, ...)                                                     \
  do {                                                                                         \
    if (n_rays == 0) return F2N_OK;                                                            \
    hipLaunchKernelGGL(                                                                        \
      kernel, ray_grid(n_rays), dim3(F2N_BLOCK), shmem, (hipStream_t)stream, __VA_ARGS__);     \
    return f2n_launch_status();                                                                \
  } while (0)

extern "C" int f2n_seg_sum_fwd(
  const float * val, const int32_t * idx, float * sum, int n_rays, void * stream)
{
  if (!idx || !sum || n_rays < 0) return F2N_E_INVALID_ARG;
  F2N_RAY_LAUNCH(seg_sum_fwd_kernel, 0, val, idx, sum, n_rays);
}

extern "C" int f2n_seg_sum_bwd(
  const float * dsum, const int32_t * idx, float * dval, int n_rays, void * stream)
{
  if (!idx || !dsum || n_rays < 0) return F2N_E_INVALID_ARG;
  F2N_RAY_LAUNCH(seg_sum_bwd_kernel, 0, dsum, idx, dval, n_rays);
}

extern "C" int f2n_seg_sum_vec_fwd(
  const float * val, const int32_t * idx, float * sum, int n_rays, int vec, void * stream)
{
  if (!idx || !sum || n_rays < 0 || vec < 1) return F2N_E_INVALID_ARG;
  if (vec > F2N_WAVE) return F2N_E_UNSUPPORTED;
  F2N_RAY_LAUNCH(
    seg_sum_vec_fwd_kernel, F2N_BLOCK * sizeof(float), val, idx, sum, n_rays, vec);
}

extern "C" int f2n_seg_sum_vec_bwd(
  const float * dsum, const int32_t * idx, float * dval, int n_rays, int vec, void * stream)
{
  if (!idx || !dsum || n_rays < 0 || vec < 1) return F2N_E_INVALID_ARG;
  if (vec > F2N_WAVE) return F2N_E_UNSUPPORTED;
  F2N_RAY_LAUNCH(seg_sum_vec_bwd_kernel, 0, dsum, idx, dval, n_rays, vec);
}

extern "C" int f2n_seg_scan_fwd(
  const float * val, const int32_t * idx, float * sum, int n_rays, int include_this, void * stream)
{
  if (!idx || n_rays < 0) return F2N_E_INVALID_ARG;
  if (include_this) F2N_RAY_LAUNCH(seg_scan_fwd_kernel<true>, 0, val, idx, sum, n_rays);
  F2N_RAY_LAUNCH(seg_scan_fwd_kernel<false>, 0, val, idx, sum, n_rays);
}

extern "C" int f2n_seg_scan_bwd(
  const float * dsum, const int32_t * idx, float * dval, int n_rays, int include_this,
  void * stream)
{
  if (!idx || n_rays < 0) return F2N_E_INVALID_ARG;
  if (include_this) F2N_RAY_LAUNCH(seg_scan_bwd_kernel<true>, 0, dsum, idx, dval, n_rays);
  F2N_RAY_LAUNCH(seg_scan_bwd_kernel<false>, 0, dsum, idx, dval, n_rays);
}

extern "C" int f2n_weight_var_fwd(
  const float * weights, const int32_t * idx, float * out_vars, int n_rays, void * stream)
{
  if (!idx || !out_vars || n_rays < 0) return F2N_E_INVALID_ARG;
  F2N_RAY_LAUNCH(weight_var_fwd_kernel, 0, weights, idx, out_vars, n_rays);
}

extern "C" int f2n_weight_var_bwd(
  const float * weights, const int32_t * idx, const float * dvars, float * dw, int n_rays,
  void * stream)
{
  if (!idx || !dvars || n_rays < 0) return F2N_E_INVALID_ARG;
  F2N_RAY_LAUNCH(weight_var_bwd_kernel, 0, weights, idx, dvars, dw, n_rays);
}

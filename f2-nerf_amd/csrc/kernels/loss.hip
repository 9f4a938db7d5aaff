// loss.hip -- the loss of the training harness in two launches instead of ~23:
//   color_loss = mean_{r,c} sqrt((C - gt)^2 + 1e-4)          (reference src/main_functions/
//   var_loss   = mean_r sqrt(var_r + 1e-2)                     train_manager.cpp:78-83)
//   loss       = color_loss + var_weight * var_loss           (:84-91)
//   sq_err_sum = sum (C - gt)^2                                (for the PSNR of :95-96)
// The gradients have closed forms, so the forward kernel writes them as well:
//   d loss / d C_rc  = err / sqrt(err^2 + 1e-4) / (3 R),   d loss / d var_r = var_weight / (2 sqrt(var_r + 1e-2) R)
// Sums are deterministic: one partial per workgroup, then a single-workgroup tree.
#include "common.hiph"

namespace
{

constexpr int kLossBlock = 256;

__device__ __forceinline__ float block_sum(float v, float * smem)
{
  v = wave_sum(v);
  const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
  __syncthreads();
  if (lane == 0) smem[wave] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < kLossBlock / 64; i++) t += smem[i];
  }
  return t;  // valid in thread 0
}

__global__ __launch_bounds__(kLossBlock) void loss_partial_kernel(
  const float * __restrict__ colors, const float * __restrict__ gt, const float * __restrict__ var,
  float var_weight, float * __restrict__ d_colors, float * __restrict__ d_var,
  float * __restrict__ partial, int n_rays)
{
  __shared__ float smem[kLossBlock / 64];
  const int r = blockIdx.x * kLossBlock + threadIdx.x;
  float s_col = 0.f, s_var = 0.f, s_sq = 0.f;
  if (r < n_rays) {
    const float inv3r = 1.f / (3.f * (float)n_rays), invr = 1.f / (float)n_rays;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float e = colors[3 * r + c] - gt[3 * r + c];
      const float e2 = e * e;
      const float rt = sqrtf(e2 + 1e-4f);
      s_col += rt;
      s_sq += e2;
      d_colors[3 * r + c] = e / rt * inv3r;
    }
    const float rv = sqrtf(var[r] + 1e-2f);
    s_var = rv;
    d_var[r] = var_weight * 0.5f / rv * invr;
  }
  const float a = block_sum(s_col, smem), b = block_sum(s_var, smem), c = block_sum(s_sq, smem);
  if (threadIdx.x == 0) {
    partial[3 * blockIdx.x + 0] = a;
    partial[3 * blockIdx.x + 1] = b;
    partial[3 * blockIdx.x + 2] = c;
  }
}

// out = {loss, color_loss, var_loss, sq_err_sum}
__global__ __launch_bounds__(kLossBlock) void loss_finish_kernel(
  const float * __restrict__ partial, int n_blocks, int n_rays, float var_weight,
  float * __restrict__ out)
{
  __shared__ float smem[kLossBlock / 64];
  float a = 0.f, b = 0.f, c = 0.f;
  for (int i = threadIdx.x; i < n_blocks; i += kLossBlock) {
    a += partial[3 * i];
    b += partial[3 * i + 1];
    c += partial[3 * i + 2];
  }
  a = block_sum(a, smem);
  b = block_sum(b, smem);
  c = block_sum(c, smem);
  if (threadIdx.x == 0) {
    const float color_loss = a / (3.f * (float)n_rays), var_loss = b / (float)n_rays;
    out[0] = color_loss + var_loss * var_weight;
    out[1] = color_loss;
    out[2] = var_loss;
    out[3] = c;
  }
}

}  // namespace

extern "C" int64_t f2n_loss_workspace_floats(int n_rays)
{
  return 3 * (int64_t)f2n_div_up(n_rays > 0 ? n_rays : 1, kLossBlock);
}

extern "C" int f2n_loss_fwd(
  const float * colors, const float * gt, const float * var, int n_rays, float var_weight,
  float * d_colors, float * d_var, float * partial, float * out4, void * stream)
{
  if (!colors || !gt || !var || !d_colors || !d_var || !partial || !out4 || n_rays <= 0)
    return F2N_E_INVALID_ARG;
  const int n_blocks = (int)f2n_div_up(n_rays, kLossBlock);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(
    loss_partial_kernel, dim3(n_blocks), dim3(kLossBlock), 0, s, colors, gt, var, var_weight,
    d_colors, d_var, partial, n_rays);
  hipLaunchKernelGGL(
    loss_finish_kernel, dim3(1), dim3(kLossBlock), 0, s, partial, n_blocks, n_rays, var_weight, out4);
  return f2n_launch_status();
}

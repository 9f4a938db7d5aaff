// adam.hip -- fused Adam step for the hash table (the step on the far
// side of the hot path).  One pass over the parameter does what torch::optim::Adam::step() does in
// ~8 element-wise launches (reference: optimizer_->step(), src/main_functions/train_manager.cpp:106,
// options from src/hash_3d_anchored.cpp:90-114) AND emits the f16 working copy of the table that the
// encode kernels read -- so the 3 full-pool f32->f16 casts per iteration of the reference
// (src/hash_3d_anchored.cu:169,198) disappear entirely.  Arithmetic order follows LibTorch's Adam:
//   g' = g + wd*p ; m = b1*m + (1-b1)*g' ; v = b2*v + (1-b2)*g'*g' ;
//   p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
#include "common.hiph"

namespace
{

template <bool SHADOW>
__global__ __launch_bounds__(F2N_BLOCK) void adam_step_kernel(
  float * __restrict__ param, const float * __restrict__ grad, float * __restrict__ exp_avg,
  float * __restrict__ exp_avg_sq, uint16_t * __restrict__ shadow, int64_t n, float beta1,
  float beta2, float eps, float weight_decay, float step_size, float inv_sqrt_bc2)
{
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n) {
      float4 p = *reinterpret_cast<const float4 *>(param + i);
      const float4 g4 = *reinterpret_cast<const float4 *>(grad + i);
      float4 m = *reinterpret_cast<const float4 *>(exp_avg + i);
      float4 v = *reinterpret_cast<const float4 *>(exp_avg_sq + i);
      float * pp = &p.x;
      float * mm = &m.x;
      float * vv = &v.x;
      const float * gg = &g4.x;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float g = fmaf(weight_decay, pp[j], gg[j]);
        mm[j] = fmaf(1.f - beta1, g, mm[j] * beta1);
        vv[j] = fmaf((1.f - beta2) * g, g, vv[j] * beta2);
        const float denom = sqrtf(vv[j]) * inv_sqrt_bc2 + eps;
        pp[j] = fmaf(-step_size, mm[j] / denom, pp[j]);
        // pin the f32 result: the shadow must be the RNE cast of the STORED f32 master, and hipcc
        // would otherwise fold fma + cvt into one v_fma_mixlo_f16 (single rounding)
        asm volatile("" : "+v"(pp[j]));
      }
      *reinterpret_cast<float4 *>(param + i) = p;
      *reinterpret_cast<float4 *>(exp_avg + i) = m;
      *reinterpret_cast<float4 *>(exp_avg_sq + i) = v;
      if (SHADOW) {
        uint2 h;
        h.x = (uint32_t)__half_as_ushort(__float2half_rn(p.x)) |
              ((uint32_t)__half_as_ushort(__float2half_rn(p.y)) << 16);
        h.y = (uint32_t)__half_as_ushort(__float2half_rn(p.z)) |
              ((uint32_t)__half_as_ushort(__float2half_rn(p.w)) << 16);
        *reinterpret_cast<uint2 *>(shadow + i) = h;
      }
    } else {
      for (int64_t k = i; k < n; k++) {
        const float g = fmaf(weight_decay, param[k], grad[k]);
        const float m = fmaf(1.f - beta1, g, exp_avg[k] * beta1);
        const float v = fmaf((1.f - beta2) * g, g, exp_avg_sq[k] * beta2);
        const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
        float p = fmaf(-step_size, m / denom, param[k]);
        asm volatile("" : "+v"(p));
        param[k] = p;
        exp_avg[k] = m;
        exp_avg_sq[k] = v;
        if (SHADOW) shadow[k] = __half_as_ushort(__float2half_rn(p));
      }
    }
  }
}

}  // namespace

extern "C" int f2n_adam_step(
  float * param, const float * grad, float * exp_avg, float * exp_avg_sq, uint16_t * shadow_f16,
  int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
  void * stream)
{
  if (!param || !grad || !exp_avg || !exp_avg_sq || n < 0 || step < 1) return F2N_E_INVALID_ARG;
  if (n == 0) return F2N_OK;
  const uintptr_t align = reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) |
                          reinterpret_cast<uintptr_t>(exp_avg) |
                          reinterpret_cast<uintptr_t>(exp_avg_sq);
  if ((align & 15u) || (reinterpret_cast<uintptr_t>(shadow_f16) & 7u)) return F2N_E_INVALID_ARG;
  const double bc1 = 1.0 - std::pow((double)beta1, (double)step);
  const double bc2 = 1.0 - std::pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / std::sqrt(bc2));
  const int64_t work = (n + 3) / 4;
  const unsigned grid = (unsigned)std::min<int64_t>((work + F2N_BLOCK - 1) / F2N_BLOCK, 256 * 16);
  hipStream_t s = (hipStream_t)stream;
  if (shadow_f16)
    hipLaunchKernelGGL(
      adam_step_kernel<true>, dim3(grid), dim3(F2N_BLOCK), 0, s, param, grad, exp_avg, exp_avg_sq,
      shadow_f16, n, beta1, beta2, eps, weight_decay, step_size, inv_sqrt_bc2);
  else
    hipLaunchKernelGGL(
      adam_step_kernel<false>, dim3(grid), dim3(F2N_BLOCK), 0, s, param, grad, exp_avg, exp_avg_sq,
      shadow_f16, n, beta1, beta2, eps, weight_decay, step_size, inv_sqrt_bc2);
  return f2n_launch_status();
}

// ragged_ops.hpp -- the reference's free-function operator surface over ragged per-ray runs:
//   FlexOps::Sum / FlexOps::AccumulateSum          (reference src/CustomOps/FlexOps.hpp:14-20)
//   CustomOps::WeightVar                           (src/CustomOps/CustomOps.hpp:23-28)
//   CustomOps::ScatterAdd / CustomOps::ScatterIdx  (src/CustomOps/Scatter.hpp:16-21)
//   torch::autograd::TruncExp                      (src/CustomOps/CustomOps.hpp:13-19)
// Same names, argument meaning and autograd behaviour; the kernels behind them are the wave-per-ray
// HIP kernels of libf2nerf_hip.so.  Also declares the fused compositing Function the Renderer uses.
#pragma once

#include "common.hpp"

namespace torch::autograd
{

class TruncExp : public Function<TruncExp>
{
public:
  static variable_list forward(AutogradContext * ctx, Tensor input);
  static variable_list backward(AutogradContext * ctx, variable_list grad_output);
};

}  // namespace torch::autograd

namespace FlexOps
{
torch::Tensor Sum(torch::Tensor val, torch::Tensor idx_start_end);
torch::Tensor AccumulateSum(torch::Tensor val, torch::Tensor idx_start_end, bool include_this);
}  // namespace FlexOps

namespace CustomOps
{
torch::Tensor WeightVar(torch::Tensor weights, torch::Tensor idx_start_end);
torch::Tensor ScatterAdd(torch::Tensor emb, torch::Tensor idx, torch::Tensor to_add);
torch::Tensor ScatterIdx(int n_all_pts, torch::Tensor idx_start_end, torch::Tensor emb_idx);
}  // namespace CustomOps

namespace f2n
{

struct CompositeOut
{
  Tensor colors;   // [n_rays, 3]
  Tensor depths;   // [n_rays]
  Tensor weights;  // [n]
};

// Fused statement of reference src/renderer.cpp:93,107-118 (density activation, alpha, exclusive
// optical-depth scan, weights, colour/depth sums, background blend), differentiable in `field_out`
// (column 0 = density logit; gradient returned dense, zero in the other columns) and `rgb`.
CompositeOut composite(
  const Tensor & field_out, const Tensor & rgb, const Tensor & dt, const Tensor & t,
  const Tensor & idx_start_end, const Tensor & bg_color);

}  // namespace f2n

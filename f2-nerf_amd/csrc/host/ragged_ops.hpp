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
// `bounds_tile_samples`: the caller guarantees that the rays' [start, end) ranges tile [0, n) (the
// Renderer's own bounds do): per-sample outputs are then written whole and need no zero fill.
CompositeOut composite(
  const Tensor & field_out, const Tensor & rgb, const Tensor & dt, const Tensor & t,
  const Tensor & idx_start_end, const Tensor & bg_color, bool bounds_tile_samples = false);


struct ShadeOut
{
  Tensor logit;  // [n]    density logit = row 0 of the field head
  Tensor rgb;    // [n, 3]
};

// The per-sample network between hash encode and compositing as one kernel per direction
// (f2n_shade_fwd / f2n_shade_bwd): field head Linear(C->16), shading-feature assembly with the
// per-image appearance embedding, SH(dirs), colour MLP 32-64-3 and the scaled sigmoid
// (reference src/hash_3d_anchored.cpp:86, src/renderer.cpp:93-106, src/sh_shader.cpp:22-29).
// `enc` is the [n, C] hash encoding (channel-major storage is consumed in place); `sample_img`
// [n] int32 or undefined (no embedding).  Differentiable in enc and all seven parameter tensors.
ShadeOut shade(
  const Tensor & enc, const Tensor & dirs, const Tensor & sample_img, const Tensor & w_h,
  const Tensor & b_h, const Tensor & w1, const Tensor & b1, const Tensor & w2, const Tensor & b2,
  const Tensor & app_emb);

// The loss of the training iteration (reference src/main_functions/train_manager.cpp:78-96) as one
// autograd node over f2n_loss_fwd: returns [4] = {loss, color_loss, var_loss, sum of squared colour
// error}; differentiable in colors [n_rays, 3] and var [n_rays] through element 0 (the others carry no
// gradient).  loss = mean sqrt((colors-gt)^2 + 1e-4) + var_loss_weight * mean sqrt(var + 1e-2).
Tensor train_loss(
  const Tensor & colors, const Tensor & gt_colors, const Tensor & var, float var_loss_weight);

// C = L*F values the fused kernel is built for
inline bool shade_supported(int64_t C) { return C == 8 || C == 16 || C == 32 || C == 64; }

}  // namespace f2n

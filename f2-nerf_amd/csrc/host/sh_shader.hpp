// sh_shader.hpp -- SHShader: spherical-harmonics view encoding + colour MLP.
// Public surface of reference src/sh_shader.hpp:10-27; module name "mlp" (Sequential 32-64-3) kept.
#pragma once

#include "common.hpp"

class SHShader : public torch::nn::Module
{
  using Tensor = torch::Tensor;

public:
  explicit SHShader(const torch::Device & device = f2n::default_device());

  Tensor query(const Tensor & feats, const Tensor & dirs);

  std::vector<torch::optim::OptimizerParamGroup> optim_param_groups(float lr);

  // SH basis of unit directions: [n,3] -> [n, DEGREE^2]; no gradient to dirs (as in the reference).
  Tensor encode(const Tensor & dirs);

  static constexpr int DEGREE = 4;

  // {w1 [64,32], b1 [64], w2 [3,64], b2 [3]} of mlp_, for the Renderer's fused per-sample kernel
  std::vector<Tensor> mlp_params() const;

private:
  torch::nn::Sequential mlp_ = nullptr;
};

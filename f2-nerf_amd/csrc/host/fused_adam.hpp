// fused_adam.hpp -- torch::optim::Optimizer with the reference's Adam settings whose step() is one
// fused kernel per parameter (f2n_adam_step) and which hands the hash table's f16 working copy to the
// field as a by-product of the update.  Drop-in for the
// torch::optim::Adam built at reference src/main_functions/train_manager.cpp:55:
//   optimizer_ = std::make_shared<FusedAdam>(renderer_->optim_param_groups(lr), renderer_->scene_field());
// Same options type (AdamOptions per group: lr, betas, eps, weight_decay), same update order.
#pragma once

#include <unordered_map>

#include "hash_3d_anchored.hpp"

class FusedAdam : public torch::optim::Optimizer
{
  using Tensor = torch::Tensor;

public:
  explicit FusedAdam(
    std::vector<torch::optim::OptimizerParamGroup> param_groups,
    std::shared_ptr<Hash3DAnchored> field = nullptr);

  Tensor step(LossClosure closure = nullptr) override;

private:
  struct State
  {
    Tensor exp_avg, exp_avg_sq;
    int64_t step = 0;
  };
  std::unordered_map<void *, State> state_;
  std::shared_ptr<Hash3DAnchored> field_;
};

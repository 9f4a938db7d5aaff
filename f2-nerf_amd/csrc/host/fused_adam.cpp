// fused_adam.cpp -- see fused_adam.hpp.  Update rule as LibTorch's Adam::step (no amsgrad).
#include "fused_adam.hpp"

using Tensor = torch::Tensor;

FusedAdam::FusedAdam(
  std::vector<torch::optim::OptimizerParamGroup> param_groups, std::shared_ptr<Hash3DAnchored> field)
: torch::optim::Optimizer(std::move(param_groups), std::make_unique<torch::optim::AdamOptions>()),
  field_(std::move(field))
{
}

Tensor FusedAdam::step(LossClosure closure)
{
  torch::NoGradGuard no_grad;
  Tensor loss;
  if (closure != nullptr) {
    at::AutoGradMode enable_grad(true);
    loss = closure();
  }
  for (auto & group : param_groups()) {
    auto & opt = static_cast<torch::optim::AdamOptions &>(group.options());
    TORCH_CHECK(!opt.amsgrad(), "FusedAdam: amsgrad is not supported");
    for (auto & p : group.params()) {
      if (!p.grad().defined()) continue;
      TORCH_CHECK(
        p.is_cuda() && p.scalar_type() == torch::kFloat32 && p.is_contiguous(),
        "FusedAdam: parameters must be contiguous float32 GPU tensors");
      Tensor grad = f2n::dev_f32(p.grad(), "FusedAdam grad");
      State & st = state_[p.unsafeGetTensorImpl()];
      if (!st.exp_avg.defined()) {
        st.exp_avg = torch::zeros_like(p);
        st.exp_avg_sq = torch::zeros_like(p);
      }
      st.step += 1;
      // the hash table also gets its f16 working copy rewritten in the same pass
      uint16_t * shadow = nullptr;
      const bool is_table = field_ && p.data_ptr() == field_->feat_pool_.data_ptr();
      if (is_table) shadow = reinterpret_cast<uint16_t *>(field_->shadow_storage().data_ptr());
      f2n::check(
        f2n_adam_step(
          p.data_ptr<float>(), grad.data_ptr<float>(), st.exp_avg.data_ptr<float>(),
          st.exp_avg_sq.data_ptr<float>(), shadow, p.numel(), (float)opt.lr(),
          (float)std::get<0>(opt.betas()), (float)std::get<1>(opt.betas()), (float)opt.eps(),
          (float)opt.weight_decay(), (int)st.step, f2n::current_stream(p)),
        "f2n_adam_step");
      p.unsafeGetTensorImpl()->bump_version();  // an in-place update, as far as autograd is concerned
      if (is_table) field_->mark_shadow_fresh();
    }
  }
  return loss;
}

// sh_shader.cpp -- see sh_shader.hpp; behaviour per reference src/sh_shader.cpp:11-39 and the host
// half of src/sh_shader.cu:105-115.
#include "sh_shader.hpp"

using Tensor = torch::Tensor;

SHShader::SHShader(const torch::Device & device)
{
  const int d_in = 16 + DEGREE * DEGREE, d_hidden = 64, d_out = 3;
  mlp_ = torch::nn::Sequential(
    torch::nn::Linear(d_in, d_hidden), torch::nn::ReLU(), torch::nn::Linear(d_hidden, d_out));
  mlp_->to(device);
  register_module("mlp", mlp_);
}

Tensor SHShader::encode(const Tensor & dirs_raw)
{
  Tensor dirs = f2n::dev_f32(dirs_raw.detach(), "SHShader dirs");
  const int64_t n = dirs.size(0);
  Tensor out = torch::empty({n, DEGREE * DEGREE}, dirs.options());
  f2n::check(
    f2n_sh_encode(dirs.data_ptr<float>(), out.data_ptr<float>(), n, DEGREE, f2n::current_stream(dirs)),
    "f2n_sh_encode");
  return out;
}

Tensor SHShader::query(const Tensor & feats, const Tensor & dirs)
{
  Tensor enc = encode(dirs);
  Tensor output = mlp_->forward(torch::cat({feats, enc}, -1));
  const float eps = 1e-3f;  // colours in (-eps, 1+eps)
  return (1.f + 2.f * eps) / (1.f + torch::exp(-output)) - eps;
}

std::vector<Tensor> SHShader::mlp_params() const
{
  auto l1 = mlp_->ptr<torch::nn::LinearImpl>(0);
  auto l2 = mlp_->ptr<torch::nn::LinearImpl>(2);
  return {l1->weight, l1->bias, l2->weight, l2->bias};
}

std::vector<torch::optim::OptimizerParamGroup> SHShader::optim_param_groups(float lr)
{
  auto opt = std::make_unique<torch::optim::AdamOptions>(lr);
  opt->betas(std::make_tuple(0.9, 0.99)).eps(1e-15).weight_decay(1e-6);
  return {torch::optim::OptimizerParamGroup(mlp_->parameters(), std::move(opt))};
}

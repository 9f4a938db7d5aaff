// points_sampler.cpp -- see points_sampler.hpp; behaviour per reference src/points_sampler.cpp:20-64.
#include "points_sampler.hpp"

using Tensor = torch::Tensor;

PtsSampler::PtsSampler(const PtsSamplerOptions & opt) : options_(opt)
{
  TORCH_CHECK(opt.max_samples >= 1 && opt.step > 0.f, "PtsSampler: bad options");
}

Tensor PtsSampler::draw_noise(int64_t n_rays, RunningMode mode, const torch::Device & device) const
{
  if (mode == RunningMode::VALIDATE) return Tensor();
  return ((torch::rand({n_rays, (int64_t)options_.max_samples}, f2n::float_on(device)) - .5f) + 1.f)
    .contiguous();
}

SampleResultFlex PtsSampler::get_samples(
  const Tensor & rays_o, const Tensor & rays_d, RunningMode mode)
{
  return get_samples(rays_o, rays_d, draw_noise(rays_o.size(0), mode, rays_o.device()));
}

SampleResultFlex PtsSampler::get_samples(
  const Tensor & rays_o_raw, const Tensor & rays_d_raw, const Tensor & noise_raw)
{
  Tensor rays_o = f2n::dev_f32(rays_o_raw, "rays_o");
  Tensor rays_d = f2n::dev_f32(rays_d_raw, "rays_d");
  const int n_rays = (int)rays_o.size(0);
  const int S = options_.max_samples;
  const int64_t n_all = (int64_t)n_rays * S;
  Tensor noise;
  if (noise_raw.defined()) {
    noise = f2n::dev_f32(noise_raw, "noise");
    TORCH_CHECK(noise.numel() == n_all, "noise must hold n_rays * max_samples values");
  }
  const auto fopt = rays_o.options();
  SampleResultFlex res;
  res.pts = torch::empty({n_all, 3}, fopt);
  res.dirs = torch::empty({n_all, 3}, fopt);
  res.dt = torch::empty({n_all}, fopt);
  res.t = torch::empty({n_all}, fopt);
  res.pts_idx_bounds = torch::empty({n_rays, 2}, f2n::int_on(rays_o.device()));
  f2n::check(
    f2n_sample_rays(
      rays_o.data_ptr<float>(), rays_d.data_ptr<float>(), f2n::fptr(noise),
      res.pts.data_ptr<float>(), res.dirs.data_ptr<float>(), res.dt.data_ptr<float>(),
      res.t.data_ptr<float>(), res.pts_idx_bounds.data_ptr<int32_t>(), n_rays, S, options_.step,
      f2n::current_stream(rays_o)),
    "f2n_sample_rays");
  return res;
}

SampleResultFlex PtsSampler::get_samples_aten(
  const Tensor & rays_o_raw, const Tensor & rays_d_raw, const Tensor & noise_raw)
{
  const int64_t S = options_.max_samples;
  Tensor rays_o = rays_o_raw.contiguous();
  Tensor rays_d = (rays_d_raw / torch::linalg_norm(rays_d_raw, 2, -1, true)).contiguous();
  const int64_t n_rays = rays_o.size(0);
  const int64_t n_all = n_rays * S;
  const auto fopt = f2n::float_on(rays_o.device());
  Tensor noise = noise_raw.defined() ? noise_raw.reshape({n_rays, S}) : torch::ones({n_rays, S}, fopt);
  Tensor cum = torch::cumsum(noise, 1) * options_.step;
  SampleResultFlex res;
  res.t = cum.reshape({n_all}).contiguous();
  Tensor pts = rays_o.view({n_rays, 1, 3}) + rays_d.view({n_rays, 1, 3}) * cum.unsqueeze(-1);
  Tensor dist = torch::diff(pts, 1, 1).norm(2, -1);
  res.dt = torch::cat({torch::zeros({n_rays, 1}, fopt), dist}, 1).reshape({n_all}).contiguous();
  res.pts = pts.reshape({n_all, 3});
  res.dirs = rays_d.view({n_rays, 1, 3}).expand({-1, S, -1}).reshape({n_all, 3}).contiguous();
  Tensor starts = torch::arange(0, n_all, S, f2n::int_on(rays_o.device()));
  res.pts_idx_bounds = torch::stack({starts, starts + (int)S}, 1).contiguous();
  return res;
}

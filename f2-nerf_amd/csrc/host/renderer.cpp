// renderer.cpp -- see renderer.hpp.  Arithmetic per reference src/renderer.cpp:18-197.
#include "renderer.hpp"

#include "kernel_timer.hpp"

#include "ragged_ops.hpp"
#include "rays.hpp"

#include <c10/hip/HIPGuard.h>
#include <hip/hip_runtime_api.h>

using Tensor = torch::Tensor;

f2n::HostCount::~HostCount()
{
  if (event_) hipEventDestroy((hipEvent_t)event_);
  if (pinned_) hipHostFree(pinned_);
}

void f2n::HostCount::request(const torch::Tensor & device_int32, void * stream)
{
  TORCH_CHECK(
    device_int32.is_cuda() && device_int32.scalar_type() == torch::kInt32 && device_int32.numel() == 1,
    "HostCount: one int32 on the device");
  const int device = (int)device_int32.device().index();
  c10::hip::HIPGuard on_device(device);
  if (pending_) {  // a read nobody waited for (an exception between request and wait): drain it
    hipEventSynchronize((hipEvent_t)event_);
    pending_ = false;
  }
  if (!pinned_)
    TORCH_CHECK(hipHostMalloc((void **)&pinned_, sizeof(int32_t), hipHostMallocDefault) == hipSuccess,
                "HostCount: pinned allocation failed");
  if (!event_ || device_ != device) {
    // an event belongs to the device it was created on: a Renderer whose inputs moved to another
    // GPU gets a new one (the pinned word is host memory and stays)
    if (event_) hipEventDestroy((hipEvent_t)event_);
    hipEvent_t ev;
    TORCH_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess,
                "HostCount: event creation failed");
    event_ = ev;
    device_ = device;
  }
  TORCH_CHECK(
    hipMemcpyAsync(pinned_, device_int32.data_ptr<int32_t>(), sizeof(int32_t), hipMemcpyDeviceToHost,
                   (hipStream_t)stream) == hipSuccess &&
      hipEventRecord((hipEvent_t)event_, (hipStream_t)stream) == hipSuccess,
    "HostCount: copy failed");
  pending_ = true;
}

int64_t f2n::HostCount::wait()
{
  TORCH_CHECK(pending_, "HostCount: wait() without request()");
  TORCH_CHECK(hipEventSynchronize((hipEvent_t)event_) == hipSuccess, "HostCount: wait failed");
  pending_ = false;
  return (int64_t)*pinned_;
}

Renderer::Renderer(int n_images, const RendererOptions & opt) : options_(opt)
{
  pts_sampler_ = std::make_shared<PtsSampler>(opt.sampler);

  scene_field_ = std::make_shared<Hash3DAnchored>(opt.field);
  register_module("scene_field", scene_field_);

  shader_ = std::make_shared<SHShader>(opt.field.device);
  register_module("shader", shader_);

  app_emb_ = torch::randn({n_images, 16}, f2n::float_on(opt.field.device)) * .1f;
  app_emb_.requires_grad_(true);
  register_parameter("app_emb", app_emb_);
}

RenderResult Renderer::render(
  const Tensor & rays_o, const Tensor & rays_d, const Tensor & emb_idx, RunningMode mode)
{
  return render(rays_o, rays_d, emb_idx, mode, Tensor(), Tensor());
}

RenderResult Renderer::render(
  const Tensor & rays_o, const Tensor & rays_d, const Tensor & emb_idx, RunningMode mode,
  const Tensor & noise_in, const Tensor & bg_in)
{
  const int64_t n_rays = rays_o.size(0);
  const auto fopt = f2n::float_on(rays_o.device());
  Tensor noise = noise_in.defined() ? noise_in
                                    : pts_sampler_->draw_noise(n_rays, mode, rays_o.device());
  Tensor bg_color = bg_in.defined() ? bg_in
                    : (mode == RunningMode::TRAIN) ? torch::rand({n_rays, 3}, fopt)
                                                   : torch::ones({n_rays, 3}, fopt) * .5f;
  if (n_rays <= 0) {
    last_n_samples_ = 0;
    return {bg_color, torch::zeros({n_rays}, fopt), torch::full({n_rays}, 512.f, fopt), Tensor()};
  }
  const bool rays_need_grad =
    torch::GradMode::is_enabled() && (rays_o.requires_grad() || rays_d.requires_grad());
  RenderResult res = (options_.fused && !rays_need_grad)
                       ? render_fused(rays_o, rays_d, emb_idx, mode, noise, bg_color)
                       : render_op_by_op(rays_o, rays_d, emb_idx, mode, noise, bg_color);
  if (options_.check_finite) CHECK(std::isfinite(res.colors.mean().item<float>()));
  return res;
}

// ---- fused path ----------------------------------------------------------------------------------

RenderResult Renderer::render_fused(
  const Tensor & rays_o_raw, const Tensor & rays_d_raw, const Tensor & emb_idx, RunningMode mode,
  const Tensor & noise_raw, const Tensor & bg_color)
{
  Tensor rays_o = f2n::dev_f32(rays_o_raw.detach(), "rays_o");
  Tensor rays_d = f2n::dev_f32(rays_d_raw.detach(), "rays_d");
  Tensor noise = noise_raw.defined() ? f2n::dev_f32(noise_raw, "noise") : Tensor();
  const int n_rays = (int)rays_o.size(0);
  const int S = pts_sampler_->options_.max_samples;
  const float step = pts_sampler_->options_.step;
  TORCH_CHECK(!noise.defined() || noise.numel() == (int64_t)n_rays * S, "noise shape");
  const auto fopt = rays_o.options();
  const auto iopt = f2n::int_on(rays_o.device());
  void * stream = f2n::current_stream(rays_o);
  Hash3DAnchored & field = *scene_field_;
  const int L = (int)field.options_.n_levels, F = (int)field.options_.n_channels;

  const int64_t C = (int64_t)L * F;
  const bool dense_ok = options_.fused_shade && f2n::shade_supported(C) &&
                        field.options_.mlp_out_dim == 16;
  const bool dense = dense_ok && (options_.dense_first_pass == 1 ||
                                  (options_.dense_first_pass < 0 && last_kept_fraction_ > 0.4f));
  if (dense) {
    // Dense first pass: every sample is encoded once (level-major kernel), the keep-prefix comes
    // from that encoding, and the shading pass reuses it -- same counts as the march, bit for bit.
    SampleResultFlex all = pts_sampler_->get_samples(rays_o, rays_d, noise);
    const int64_t n_all = all.pts.size(0);
    SampleResultFlex kept;
    Tensor enc_kept_cm, contracted_kept, contracted_all, enc_all_cm;
    Tensor total = torch::empty({1}, iopt);
    {
      torch::NoGradGuard no_grad;
      // all.pts is the dense [n_rays, S] grid of the sampler: ray-tile mapping of the encode
      enc_all_cm = field.encode(all.pts, S, &contracted_all).t();  // [C, n_all] contiguous storage
      TORCH_CHECK(enc_all_cm.is_contiguous(), "encode() must return channel-major storage");
    }
    // The number of survivors sizes everything downstream, so the host has to read it: one blocking
    // read per chunk.  When the previous chunk kept every sample the next one most likely does too
    // (no density yet, or validation of empty space), so that case is tried first and for free: the
    // shading pass is run over ALL samples (fresh buffers, nothing observable), the density logits
    // it produces anyway are summed per ray (f2n_density_margin: 67 MB instead of the 1 GB encoding
    // the exact scan reads) and a flag says whether some ray comes within a factor e^0.5 of the
    // early-stop threshold.  Flag clear = the exact scan would keep everything too (its logits
    // differ from these in the last bits only): the guess IS the result, and neither the scan
    // (0.22 ms per 8.4 M samples) nor an idle GPU across the read (the flag travels while
    // compositing runs) was paid.  Flag set = drop the guess, run the exact scan, compact.
    //   Small chunks (the 512-ray training batch) keep the exact scan instead -- there it costs
    // 20 us, less than the GPU would idle while the host waits for a flag that only exists after
    // the shading pass -- and hide ITS read-back behind the same guess (further down).
    if (options_.deferred_check && options_.fused_shade) {
      // no host read: exact scan -> device-side "kept fewer than shaded" flag, all samples shaded
      torch::NoGradGuard no_grad;
      auto head = field.density_head();
      Tensor counts = torch::empty({n_rays}, iopt);
      {
        f2n::ScopedKernelTimer timer("density_scan", stream, (double)n_rays);
        f2n::check(
          f2n_density_scan(
            enc_all_cm.data_ptr<float>(), (int)C, all.dt.data_ptr<float>(),
            head.first.data_ptr<float>(), head.second.data_ptr<float>(),
            counts.data_ptr<int32_t>(), n_rays, S, options_.early_stop_trans, 3.f, stream),
          "f2n_density_scan");
      }
      Tensor scratch_bounds = torch::empty({n_rays, 2}, iopt);
      f2n::check(
        f2n_bounds_from_counts(
          counts.data_ptr<int32_t>(), scratch_bounds.data_ptr<int32_t>(), total.data_ptr<int32_t>(),
          n_rays, stream),
        "f2n_bounds_from_counts");
      if (!deferred_bad_.defined()) deferred_bad_ = torch::zeros({1}, iopt);
      deferred_bad_.add_(total.ne(n_all).to(torch::kInt32));
      last_n_samples_ = n_all;
      last_kept_fraction_ = 1.f;
    }
    if (options_.deferred_check && options_.fused_shade)
      return shade_and_composite(all, emb_idx, mode, bg_color, enc_all_cm, contracted_all);
    const bool may_guess = options_.speculate_dense && last_kept_fraction_ >= 1.f && n_all > 0;
    const int64_t kMarginMinSamples = options_.margin_min_samples;
    if (may_guess && n_all >= kMarginMinSamples) {
      Tensor near_threshold = torch::zeros({1}, iopt);
      RenderResult guess = shade_and_composite(
        all, emb_idx, mode, bg_color, enc_all_cm, contracted_all, &near_threshold, S);
      if (survivors_.wait() == 0) {
        last_n_samples_ = n_all;
        last_kept_fraction_ = 1.f;
        return guess;
      }
    }
    {
      torch::NoGradGuard no_grad;
      auto head = field.density_head();
      Tensor counts = torch::empty({n_rays}, iopt);
      {
        f2n::ScopedKernelTimer timer("density_scan", stream, (double)n_rays);
        f2n::check(
          f2n_density_scan(
            enc_all_cm.data_ptr<float>(), (int)C, all.dt.data_ptr<float>(),
            head.first.data_ptr<float>(), head.second.data_ptr<float>(),
            counts.data_ptr<int32_t>(), n_rays, S, options_.early_stop_trans, 3.f, stream),
          "f2n_density_scan");
      }
      kept.pts_idx_bounds = torch::empty({n_rays, 2}, iopt);
      f2n::check(
        f2n_bounds_from_counts(
          counts.data_ptr<int32_t>(), kept.pts_idx_bounds.data_ptr<int32_t>(),
          total.data_ptr<int32_t>(), n_rays, stream),
        "f2n_bounds_from_counts");
    }
    survivors_.request(total, stream);
    RenderResult guess;
    bool guessed = false;
    if (may_guess && n_all < kMarginMinSamples) {
      // enqueued BEFORE the host waits for the count: the GPU does not idle across the read
      guess = shade_and_composite(all, emb_idx, mode, bg_color, enc_all_cm, contracted_all);
      guessed = true;
    }
    const int64_t n_kept = survivors_.wait();
    last_n_samples_ = n_kept;
    last_kept_fraction_ = n_all > 0 ? (float)n_kept / (float)n_all : 0.f;
    if (guessed && n_kept == n_all) return guess;
    guess = RenderResult();
    {
      torch::NoGradGuard no_grad;
      if (n_kept == n_all) {
        // nothing terminated: the uncompacted arrays ARE the compacted ones
        kept.pts = all.pts;
        kept.dirs = all.dirs;
        kept.dt = all.dt;
        kept.t = all.t;
        enc_kept_cm = enc_all_cm;
        contracted_kept = contracted_all;  // ... and so are their contracted positions
      } else {
        kept.pts = torch::empty({n_kept, 3}, fopt);
        kept.dirs = torch::empty({n_kept, 3}, fopt);
        kept.dt = torch::empty({n_kept}, fopt);
        kept.t = torch::empty({n_kept}, fopt);
        f2n::check(
          f2n_sample_compact(
            rays_o.data_ptr<float>(), rays_d.data_ptr<float>(), f2n::fptr(noise),
            kept.pts_idx_bounds.data_ptr<int32_t>(), kept.pts.data_ptr<float>(),
            kept.dirs.data_ptr<float>(), kept.dt.data_ptr<float>(), kept.t.data_ptr<float>(),
            n_rays, S, step, stream),
          "f2n_sample_compact");
        enc_kept_cm = torch::empty({C, n_kept}, fopt);
        f2n::check(
          f2n_compact_rows_cm(
            enc_all_cm.data_ptr<float>(), n_all, enc_kept_cm.data_ptr<float>(), n_kept, (int)C,
            kept.pts_idx_bounds.data_ptr<int32_t>(), n_rays, S, stream),
          "f2n_compact_rows_cm");
      }
    }
    return shade_and_composite(kept, emb_idx, mode, bg_color, enc_kept_cm, contracted_kept);
  }

  SampleResultFlex kept;
  {
    // First pass (renderer.cpp:58-90): density only, never differentiated by the loss.
    torch::NoGradGuard no_grad;
    Tensor table16 = field.table_f16();
    auto head = field.density_head();
    Tensor counts = torch::empty({n_rays}, iopt);
    {
      f2n::ScopedKernelTimer timer("density_march", stream, (double)n_rays);
      f2n::check(
      f2n_density_march(
        rays_o.data_ptr<float>(), rays_d.data_ptr<float>(), f2n::fptr(noise),
        reinterpret_cast<const uint16_t *>(table16.data_ptr()), field.prim_pool_.data_ptr<int32_t>(),
        field.bias_pool_.data_ptr<float>(), field.level_mul_.data_ptr<float>(),
        head.first.data_ptr<float>(), head.second.data_ptr<float>(), counts.data_ptr<int32_t>(),
        n_rays, S, step, L, F, (uint32_t)field.local_size_, field.level_stride_,
        options_.early_stop_trans, 3.f, stream),
      "f2n_density_march");
    }
    kept.pts_idx_bounds = torch::empty({n_rays, 2}, iopt);
    Tensor total = torch::empty({1}, iopt);
    f2n::check(
      f2n_bounds_from_counts(
        counts.data_ptr<int32_t>(), kept.pts_idx_bounds.data_ptr<int32_t>(),
        total.data_ptr<int32_t>(), n_rays, stream),
      "f2n_bounds_from_counts");
    const int64_t n_kept = total.item<int>();  // the one host sync of the fused path (sizes tensors)
    last_n_samples_ = n_kept;
    last_kept_fraction_ = (float)n_kept / (float)((int64_t)n_rays * S);
    kept.pts = torch::empty({n_kept, 3}, fopt);
    kept.dirs = torch::empty({n_kept, 3}, fopt);
    kept.dt = torch::empty({n_kept}, fopt);
    kept.t = torch::empty({n_kept}, fopt);
    f2n::check(
      f2n_sample_compact(
        rays_o.data_ptr<float>(), rays_d.data_ptr<float>(), f2n::fptr(noise),
        kept.pts_idx_bounds.data_ptr<int32_t>(), kept.pts.data_ptr<float>(),
        kept.dirs.data_ptr<float>(), kept.dt.data_ptr<float>(), kept.t.data_ptr<float>(), n_rays, S,
        step, stream),
      "f2n_sample_compact");
  }
  return shade_and_composite(kept, emb_idx, mode, bg_color);
}

bool Renderer::deferred_check_ok()
{
  if (!deferred_bad_.defined()) return true;
  const bool ok = deferred_bad_.item<int32_t>() == 0;  // the caller's sync point
  deferred_bad_.zero_();
  return ok;
}

// Second pass on the survivors (renderer.cpp:92-118).
RenderResult Renderer::shade_and_composite(
  const SampleResultFlex & kept, const Tensor & emb_idx, RunningMode mode, const Tensor & bg_color,
  const Tensor & enc_cm, const Tensor & contracted, Tensor * near_threshold, int64_t grid_samples)
{
  const int64_t n_kept = kept.pts.size(0);
  const int64_t C = scene_field_->options_.n_levels * scene_field_->options_.n_channels;
  if (options_.fused_shade && f2n::shade_supported(C) && scene_field_->options_.mlp_out_dim == 16) {
    // hash encode -> one kernel for field head + embedding + SH + colour MLP -> composite
    Tensor enc = enc_cm.defined() ? scene_field_->encode_cached(kept.pts, enc_cm, contracted)
                                  : scene_field_->encode(kept.pts);
    Tensor sample_img;
    if (mode == RunningMode::TRAIN)
      sample_img = CustomOps::ScatterIdx((int)n_kept, kept.pts_idx_bounds, emb_idx);
    auto mlp = shader_->mlp_params();
    f2n::ShadeOut sh = f2n::shade(
      enc, kept.dirs, sample_img, scene_field_->mlp_->weight, scene_field_->mlp_->bias, mlp[0],
      mlp[1], mlp[2], mlp[3], mode == RunningMode::TRAIN ? app_emb_ : Tensor());
    if (near_threshold) {
      // `kept` is the sampler's dense [n_rays, grid_samples] grid: see render_fused
      torch::NoGradGuard no_grad;
      void * stream = f2n::current_stream(sh.logit);
      const float limit = -std::log(options_.early_stop_trans) - 0.5f;
      f2n::check(
        f2n_density_margin(
          sh.logit.data_ptr<float>(), kept.dt.data_ptr<float>(), near_threshold->data_ptr<int32_t>(),
          (int)(n_kept / grid_samples), (int)grid_samples, 3.f, limit, stream),
        "f2n_density_margin");
      survivors_.request(*near_threshold, stream);
    }
    // (kept.pts_idx_bounds comes from f2n_bounds_from_counts / the sampler: the ranges tile [0, n))
    f2n::CompositeOut out = f2n::composite(
      sh.logit.unsqueeze(1), sh.rgb, kept.dt, kept.t, kept.pts_idx_bounds, bg_color, true);
    return {out.colors, out.depths, out.weights, kept.pts_idx_bounds};
  }
  Tensor scene_feat = scene_field_->query(kept.pts);  // [n, 16]: col 0 density logit, 1.. shading

  Tensor shading_feat = torch::cat(
    {torch::ones({n_kept, 1}, scene_feat.options()),
     scene_feat.index({Slc(), Slc(1, torch::indexing::None)})},
    1);
  if (mode == RunningMode::TRAIN) {
    Tensor all_emb_idx = CustomOps::ScatterIdx((int)n_kept, kept.pts_idx_bounds, emb_idx);
    shading_feat = CustomOps::ScatterAdd(app_emb_, all_emb_idx, shading_feat);
  }
  Tensor sampled_colors = shader_->query(shading_feat, kept.dirs);

  f2n::CompositeOut out =
    f2n::composite(scene_feat, sampled_colors, kept.dt, kept.t, kept.pts_idx_bounds, bg_color);
  return {out.colors, out.depths, out.weights, kept.pts_idx_bounds};
}

// ---- op-by-op path (the reference's own sequence on the drop-in operators) -----------------------

RenderResult Renderer::render_op_by_op(
  const Tensor & rays_o, const Tensor & rays_d, const Tensor & emb_idx, RunningMode mode,
  const Tensor & noise, const Tensor & bg_color)
{
  const int64_t n_rays = rays_o.size(0);
  const int64_t S = pts_sampler_->options_.max_samples;
  const bool rays_need_grad =
    torch::GradMode::is_enabled() && (rays_o.requires_grad() || rays_d.requires_grad());
  SampleResultFlex all = rays_need_grad ? pts_sampler_->get_samples_aten(rays_o, rays_d, noise)
                                        : pts_sampler_->get_samples(rays_o, rays_d, noise);

  auto density_act = [](const Tensor & x) {
    return torch::autograd::TruncExp::apply(x - 3.f)[0];
  };

  SampleResultFlex kept;
  {
    Tensor scene_feat = scene_field_->query(all.pts);
    Tensor density = density_act(scene_feat.index({Slc(), Slc(0, 1)}));
    Tensor sec = density.index({Slc(), 0}) * all.dt;
    Tensor acc = FlexOps::AccumulateSum(sec, all.pts_idx_bounds, false);
    Tensor mask = torch::exp(-acc) > options_.early_stop_trans;
    Tensor mask_idx = torch::where(mask)[0];
    kept.pts = all.pts.index({mask_idx}).contiguous();
    kept.dirs = all.dirs.index({mask_idx}).contiguous();
    kept.dt = all.dt.index({mask_idx}).contiguous();
    kept.t = all.t.index({mask_idx}).contiguous();
    Tensor num = mask.reshape({n_rays, S}).sum(1);
    Tensor cum = torch::cumsum(num, 0);
    kept.pts_idx_bounds = torch::stack({cum - num, cum}, 1).to(torch::kInt32).contiguous();
  }
  last_n_samples_ = kept.pts.size(0);

  Tensor scene_feat = scene_field_->query(kept.pts);
  Tensor density = density_act(scene_feat.index({Slc(), Slc(0, 1)}));
  Tensor shading_feat = torch::cat(
    {torch::ones_like(scene_feat.index({Slc(), Slc(0, 1)})),
     scene_feat.index({Slc(), Slc(1, torch::indexing::None)})},
    1);
  if (mode == RunningMode::TRAIN) {
    Tensor all_emb_idx =
      CustomOps::ScatterIdx((int)kept.pts.size(0), kept.pts_idx_bounds, emb_idx);
    shading_feat = CustomOps::ScatterAdd(app_emb_, all_emb_idx, shading_feat);
  }
  Tensor sampled_colors = shader_->query(shading_feat, kept.dirs);
  Tensor sampled_t = (kept.t + 1e-2f).contiguous();
  Tensor sec = density.index({Slc(), 0}) * kept.dt;
  Tensor alphas = 1.f - torch::exp(-sec);
  Tensor idx = kept.pts_idx_bounds;
  Tensor trans = torch::exp(-FlexOps::AccumulateSum(sec, idx, false));
  Tensor weights = trans * alphas;
  Tensor last_trans = torch::exp(-FlexOps::Sum(sec, idx));
  Tensor colors = FlexOps::Sum(weights.unsqueeze(-1) * sampled_colors, idx) +
                  last_trans.unsqueeze(-1) * bg_color;
  Tensor depths = FlexOps::Sum(weights * sampled_t, idx) / (1.f - last_trans + 1e-4f);
  return {colors, depths, weights, idx};
}

// ---- whole-image helpers -------------------------------------------------------------------------

std::tuple<Tensor, Tensor> Renderer::render_all_rays(
  const Tensor & rays_o, const Tensor & rays_d, const int batch_size)
{
  const int64_t n_rays = rays_d.size(0);
  std::vector<Tensor> colors, depths;
  for (int64_t lo = 0; lo < n_rays; lo += batch_size) {
    const int64_t hi = std::min<int64_t>(lo + batch_size, n_rays);
    RenderResult r = render(
      rays_o.index({Slc(lo, hi)}).contiguous(), rays_d.index({Slc(lo, hi)}).contiguous(), Tensor(),
      RunningMode::VALIDATE);
    colors.push_back(r.colors);
    depths.push_back(r.depths.reshape({-1, 1}));
  }
  return {torch::cat(colors, 0), torch::cat(depths, 0)};
}

std::tuple<Tensor, Tensor> Renderer::render_image(
  const Tensor & pose, const Tensor & intrinsic, const int h, const int w, const int batch_size)
{
  Rays rays = get_view_rays(pose, intrinsic, h, w);  // pixel grid generated in the kernel
  // The view is traversed in B x B pixel tiles rather than row by row: the ray-tile encode takes 64
  // consecutive rays per workgroup, and an 8 x 8 block of pixels is a bundle 8 pixels wide both
  // ways instead of a strip 64 pixels long -- a third fewer distinct table lines per gather at the
  // middle levels (src/renderer.cpp:153-172 walks the image in row-major batches; the image that
  // comes out is the same).
  const int B = options_.pixel_tiles;
  Tensor order;
  if (B > 1 && h % B == 0 && w % B == 0) {
    order = torch::arange((int64_t)h * w, torch::TensorOptions().dtype(torch::kInt64).device(pose.device()))
              .view({h / B, B, w / B, B})
              .permute({0, 2, 1, 3})
              .reshape({-1});
    rays.origins = rays.origins.index_select(0, order);
    rays.dirs = rays.dirs.index_select(0, order);
  }
  auto [colors, depths] = render_all_rays(rays.origins, rays.dirs, batch_size);
  if (order.defined()) {
    colors = torch::empty_like(colors).index_copy_(0, order, colors);
    depths = torch::empty_like(depths).index_copy_(0, order, depths);
  }
  colors = colors.reshape({h, w, 3}).clip(0.f, 1.f);
  depths = depths.reshape({h, w, 1}).repeat({1, 1, 3});
  return {colors, depths};
}

std::vector<torch::optim::OptimizerParamGroup> Renderer::optim_param_groups(float lr)
{
  std::vector<torch::optim::OptimizerParamGroup> groups;
  for (auto & g : scene_field_->optim_param_groups(lr)) groups.emplace_back(g);
  for (auto & g : shader_->optim_param_groups(lr)) groups.emplace_back(g);
  auto opt = std::make_unique<torch::optim::AdamOptions>(lr);
  opt->betas(std::make_tuple(0.9, 0.99)).eps(1e-15).weight_decay(1e-6);
  groups.emplace_back(std::vector<Tensor>{app_emb_}, std::move(opt));
  return groups;
}

// ---- training-step harness -----------------------------------------------------------------------

f2n::TrainStepResult f2n::train_step(
  Renderer & renderer, const Tensor & rays_o, const Tensor & rays_d, const Tensor & emb_idx,
  const Tensor & gt_colors, float var_loss_weight, const Tensor & noise, const Tensor & bg_color,
  bool run_backward)
{
  RenderResult res =
    renderer.render(rays_o, rays_d, emb_idx, RunningMode::TRAIN, noise, bg_color);
  // colour loss + variance loss + squared error in two launches (f2n_loss_fwd; the ATen spelling of
  // the reference, train_manager.cpp:78-96, is ~23 launches of a few microseconds each)
  Tensor var = CustomOps::WeightVar(res.weights, res.idx_start_end);
  // (a zero weight -- the reference's schedule starts there, train_manager.cpp:85-91 -- makes the
  // variance term's gradient exactly zero: its backward kernel is not run at all)
  if (var_loss_weight == 0.f) var = var.detach();
  Tensor stats = f2n::train_loss(res.colors, gt_colors, var, var_loss_weight);  // {loss, c, v, sq}
  Tensor loss = stats[0];
  TrainStepResult out;
  out.loss = loss.detach();
  out.sq_err_sum = stats[3].detach();
  out.n_values = res.colors.numel();
  out.n_samples = renderer.last_n_samples_;
  if (run_backward && loss.requires_grad()) loss.backward();
  return out;
}

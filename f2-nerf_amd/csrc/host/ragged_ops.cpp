// ragged_ops.cpp -- autograd wrappers binding the ragged per-ray kernels of libf2nerf_hip.so to the
// reference's FlexOps / CustomOps surface (reference src/CustomOps/{FlexOps,CustomOps,Scatter}.cu
// host halves: FlexOps.cu:96-216, CustomOps.cu:69-118, Scatter.cu:43-132, CustomOps.cpp:10-20).
#include "ragged_ops.hpp"

#include "kernel_timer.hpp"

using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

namespace
{

class FlexSumFn : public torch::autograd::Function<FlexSumFn>
{
public:
  static variable_list forward(AutogradContext * ctx, Tensor val, Tensor idx)
  {
    val = f2n::dev_f32(val, "FlexOps::Sum val");
    idx = f2n::dev_i32(idx, "FlexOps::Sum idx_start_end");
    TORCH_CHECK(val.dim() == 1 || val.dim() == 2, "FlexOps::Sum expects [n] or [n, c]");
    const int n_rays = (int)idx.size(0);
    void * s = f2n::current_stream(val);
    Tensor sum;
    if (val.dim() == 1) {
      sum = torch::empty({n_rays}, val.options());
      f2n::check(
        f2n_seg_sum_fwd(f2n::fptr(val), f2n::iptr(idx), sum.data_ptr<float>(), n_rays, s),
        "f2n_seg_sum_fwd");
    } else {
      const int vec = (int)val.size(1);
      sum = torch::empty({n_rays, vec}, val.options());
      f2n::check(
        f2n_seg_sum_vec_fwd(f2n::fptr(val), f2n::iptr(idx), sum.data_ptr<float>(), n_rays, vec, s),
        "f2n_seg_sum_vec_fwd");
    }
    ctx->save_for_backward({idx});
    ctx->saved_data["n_all"] = (int64_t)val.size(0);
    ctx->saved_data["vec"] = (int64_t)(val.dim() == 1 ? 0 : val.size(1));
    return {sum};
  }

  static variable_list backward(AutogradContext * ctx, variable_list grad_output)
  {
    Tensor dsum = f2n::dev_f32(grad_output[0], "FlexOps::Sum grad");
    Tensor idx = ctx->get_saved_variables()[0];
    const int64_t n_all = ctx->saved_data["n_all"].toInt();
    const int vec = (int)ctx->saved_data["vec"].toInt();
    const int n_rays = (int)idx.size(0);
    void * s = f2n::current_stream(dsum);
    Tensor dval;
    // zeros, not empty: samples outside every [start,end) get a defined (zero) gradient
    if (vec == 0) {
      dval = torch::zeros({n_all}, dsum.options());
      f2n::check(
        f2n_seg_sum_bwd(f2n::fptr(dsum), f2n::iptr(idx), dval.data_ptr<float>(), n_rays, s),
        "f2n_seg_sum_bwd");
    } else {
      dval = torch::zeros({n_all, vec}, dsum.options());
      f2n::check(
        f2n_seg_sum_vec_bwd(f2n::fptr(dsum), f2n::iptr(idx), dval.data_ptr<float>(), n_rays, vec, s),
        "f2n_seg_sum_vec_bwd");
    }
    return {dval, Tensor()};
  }
};

class FlexAccumulateSumFn : public torch::autograd::Function<FlexAccumulateSumFn>
{
public:
  static variable_list forward(AutogradContext * ctx, Tensor val, Tensor idx, bool include_this)
  {
    val = f2n::dev_f32(val, "FlexOps::AccumulateSum val");
    idx = f2n::dev_i32(idx, "FlexOps::AccumulateSum idx_start_end");
    TORCH_CHECK(val.dim() == 1, "FlexOps::AccumulateSum expects [n]");
    Tensor sum = torch::zeros_like(val);
    f2n::check(
      f2n_seg_scan_fwd(
        f2n::fptr(val), f2n::iptr(idx), sum.data_ptr<float>(), (int)idx.size(0), include_this,
        f2n::current_stream(val)),
      "f2n_seg_scan_fwd");
    ctx->save_for_backward({idx});
    ctx->saved_data["include_this"] = include_this;
    return {sum};
  }

  static variable_list backward(AutogradContext * ctx, variable_list grad_output)
  {
    Tensor dsum = f2n::dev_f32(grad_output[0], "FlexOps::AccumulateSum grad");
    Tensor idx = ctx->get_saved_variables()[0];
    const bool include_this = ctx->saved_data["include_this"].toBool();
    Tensor dval = torch::zeros_like(dsum);
    f2n::check(
      f2n_seg_scan_bwd(
        f2n::fptr(dsum), f2n::iptr(idx), dval.data_ptr<float>(), (int)idx.size(0), include_this,
        f2n::current_stream(dsum)),
      "f2n_seg_scan_bwd");
    return {dval, Tensor(), Tensor()};
  }
};

class WeightVarFn : public torch::autograd::Function<WeightVarFn>
{
public:
  static variable_list forward(AutogradContext * ctx, Tensor weights, Tensor idx)
  {
    weights = f2n::dev_f32(weights, "CustomOps::WeightVar weights");
    idx = f2n::dev_i32(idx, "CustomOps::WeightVar idx_start_end");
    const int n_rays = (int)idx.size(0);
    Tensor out = torch::empty({n_rays}, weights.options());
    f2n::check(
      f2n_weight_var_fwd(
        f2n::fptr(weights), f2n::iptr(idx), out.data_ptr<float>(), n_rays,
        f2n::current_stream(weights)),
      "f2n_weight_var_fwd");
    ctx->save_for_backward({weights, idx});
    return {out};
  }

  static variable_list backward(AutogradContext * ctx, variable_list grad_output)
  {
    Tensor dvar = f2n::dev_f32(grad_output[0], "CustomOps::WeightVar grad");
    auto saved = ctx->get_saved_variables();
    Tensor & weights = saved[0];
    Tensor & idx = saved[1];
    Tensor dw = torch::zeros_like(weights);
    f2n::check(
      f2n_weight_var_bwd(
        f2n::fptr(weights), f2n::iptr(idx), f2n::fptr(dvar), dw.data_ptr<float>(),
        (int)idx.size(0), f2n::current_stream(dvar)),
      "f2n_weight_var_bwd");
    return {dw, Tensor()};
  }
};

class ScatterAddFn : public torch::autograd::Function<ScatterAddFn>
{
public:
  static variable_list forward(AutogradContext * ctx, Tensor emb, Tensor idx, Tensor to_add)
  {
    emb = f2n::dev_f32(emb, "CustomOps::ScatterAdd emb");
    idx = f2n::dev_i32(idx, "CustomOps::ScatterAdd idx");
    to_add = f2n::dev_f32(to_add, "CustomOps::ScatterAdd to_add");
    const int64_t n_all = idx.size(0);
    const int C = (int)emb.size(1);
    TORCH_CHECK(to_add.size(0) == n_all && to_add.size(1) == C, "ScatterAdd shape mismatch");
    Tensor sum = torch::empty_like(to_add);
    f2n::check(
      f2n_scatter_add_fwd(
        f2n::fptr(emb), f2n::iptr(idx), f2n::fptr(to_add), sum.data_ptr<float>(), n_all, C,
        f2n::current_stream(to_add)),
      "f2n_scatter_add_fwd");
    ctx->save_for_backward({idx});
    ctx->saved_data["n_emb"] = (int64_t)emb.size(0);
    return {sum};
  }

  static variable_list backward(AutogradContext * ctx, variable_list grad_output)
  {
    Tensor dsum = f2n::dev_f32(grad_output[0], "CustomOps::ScatterAdd grad");
    Tensor idx = ctx->get_saved_variables()[0];
    const int n_emb = (int)ctx->saved_data["n_emb"].toInt();
    const int C = (int)dsum.size(1);
    Tensor demb = torch::empty({n_emb, C}, dsum.options());
    f2n::check(
      f2n_scatter_add_bwd(
        f2n::iptr(idx), f2n::fptr(dsum), demb.data_ptr<float>(), idx.size(0), n_emb, C,
        f2n::current_stream(dsum)),
      "f2n_scatter_add_bwd");
    return {demb, Tensor(), dsum};
  }
};

class CompositeFn : public torch::autograd::Function<CompositeFn>
{
public:
  static variable_list forward(
    AutogradContext * ctx, Tensor field_out, Tensor rgb, Tensor dt, Tensor t, Tensor idx, Tensor bg,
    bool tiled)
  {
    field_out = f2n::dev_f32(field_out, "composite field_out");
    rgb = f2n::dev_f32(rgb, "composite rgb");
    dt = f2n::dev_f32(dt, "composite dt");
    t = f2n::dev_f32(t, "composite t");
    idx = f2n::dev_i32(idx, "composite idx_start_end");
    bg = f2n::dev_f32(bg, "composite bg_color");
    TORCH_CHECK(field_out.dim() == 2 && rgb.dim() == 2 && rgb.size(1) == 3, "composite shapes");
    const int64_t n = field_out.size(0);
    const int n_rays = (int)idx.size(0);
    auto opt = field_out.options();
    Tensor colors = torch::empty({n_rays, 3}, opt), depths = torch::empty({n_rays}, opt);
    Tensor weights = tiled ? torch::empty({n}, opt) : torch::zeros({n}, opt);
    Tensor last_trans = torch::empty({n_rays}, opt);
    ctx->saved_data["tiled"] = tiled;
    f2n::check(
      f2n_composite_fwd(
        f2n::fptr(field_out), field_out.size(1), f2n::fptr(rgb), f2n::fptr(dt), f2n::fptr(t),
        f2n::iptr(idx), f2n::fptr(bg), colors.data_ptr<float>(), depths.data_ptr<float>(),
        weights.data_ptr<float>(), last_trans.data_ptr<float>(), n_rays, 3.f, 1e-2f,
        f2n::current_stream(field_out)),
      "f2n_composite_fwd");
    ctx->save_for_backward({field_out, rgb, dt, t, idx, bg, weights, last_trans});
    return {colors, depths, weights};
  }

  static variable_list backward(AutogradContext * ctx, variable_list grad_output)
  {
    auto sv = ctx->get_saved_variables();
    Tensor &field_out = sv[0], &rgb = sv[1], &dt = sv[2], &t = sv[3], &idx = sv[4], &bg = sv[5],
           &weights = sv[6], &last_trans = sv[7];
    const int n_rays = (int)idx.size(0);
    const int64_t n = field_out.size(0);
    auto opt = field_out.options();
    Tensor d_colors = grad_output[0].defined() ? f2n::dev_f32(grad_output[0], "d_colors")
                                               : torch::zeros({n_rays, 3}, opt);
    Tensor d_depths = grad_output[1].defined() ? f2n::dev_f32(grad_output[1], "d_depths")
                                               : torch::zeros({n_rays}, opt);
    Tensor d_weights =
      grad_output[2].defined() ? f2n::dev_f32(grad_output[2], "d_weights") : Tensor();
    const bool tiled = ctx->saved_data["tiled"].toBool();
    Tensor d_logit = tiled ? torch::empty({n}, opt) : torch::zeros({n}, opt);
    Tensor d_rgb = tiled ? torch::empty({n, 3}, opt) : torch::zeros({n, 3}, opt);
    f2n::check(
      f2n_composite_bwd(
        f2n::fptr(field_out), field_out.size(1), f2n::fptr(rgb), f2n::fptr(dt), f2n::fptr(t),
        f2n::iptr(idx), f2n::fptr(bg), f2n::fptr(weights), f2n::fptr(last_trans),
        f2n::fptr(d_colors), f2n::fptr(d_depths), f2n::fptr(d_weights), d_logit.data_ptr<float>(),
        d_rgb.data_ptr<float>(), n_rays, 3.f, 1e-2f, f2n::current_stream(field_out)),
      "f2n_composite_bwd");
    if (field_out.size(1) == 1)  // the fused path hands over the logit column alone
      return {d_logit.unsqueeze(1), d_rgb, Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    Tensor d_field = torch::zeros_like(field_out);
    d_field.select(1, 0).copy_(d_logit);
    return {d_field, d_rgb, Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

// channel-major [C, n] storage behind a logical [n, C] tensor (no copy when it already is)
Tensor as_channel_major(const Tensor & t)
{
  if (t.dim() == 2 && t.stride(0) == 1 && t.stride(1) == t.size(0)) return t.t();
  return t.t().contiguous();
}

class ShadeFn : public torch::autograd::Function<ShadeFn>
{
public:
  static variable_list forward(
    AutogradContext * ctx, Tensor enc, Tensor dirs, Tensor sample_img, Tensor w_h, Tensor b_h,
    Tensor w1, Tensor b1, Tensor w2, Tensor b2, Tensor app_emb)
  {
    TORCH_CHECK(enc.is_cuda() && enc.scalar_type() == torch::kFloat32 && enc.dim() == 2, "enc");
    const int64_t n = enc.size(0);
    const int C = (int)enc.size(1);
    Tensor enc_cm = as_channel_major(enc);  // [C, n] contiguous
    dirs = f2n::dev_f32(dirs.detach(), "shade dirs");
    // "no embedding" arrives as empty tensors (undefined tensors cannot pass through apply())
    const bool use_emb = sample_img.defined() && app_emb.defined() && sample_img.numel() > 0 &&
                         app_emb.numel() > 0;
    if (use_emb) sample_img = f2n::dev_i32(sample_img, "shade sample_img");
    w_h = f2n::dev_f32(w_h, "w_h");
    b_h = f2n::dev_f32(b_h, "b_h");
    w1 = f2n::dev_f32(w1, "w1");
    b1 = f2n::dev_f32(b1, "b1");
    w2 = f2n::dev_f32(w2, "w2");
    b2 = f2n::dev_f32(b2, "b2");
    Tensor emb = use_emb ? f2n::dev_f32(app_emb, "app_emb") : Tensor();
    TORCH_CHECK(
      w_h.size(0) == 16 && w_h.size(1) == C && w1.size(0) == 64 && w1.size(1) == 32 &&
        w2.size(0) == 3 && w2.size(1) == 64,
      "shade: layer shapes must be 16xC, 64x32, 3x64");
    Tensor logit = torch::empty({n}, enc.options()), rgb = torch::empty({n, 3}, enc.options());
    {
    f2n::ScopedKernelTimer timer("shade_fwd", f2n::current_stream(enc_cm), (double)n);
    f2n::check(
      f2n_shade_fwd(
        enc_cm.data_ptr<float>(), C, dirs.data_ptr<float>(), use_emb ? f2n::iptr(sample_img) : nullptr,
        f2n::fptr(w_h), f2n::fptr(b_h), f2n::fptr(w1), f2n::fptr(b1), f2n::fptr(w2), f2n::fptr(b2),
        f2n::fptr(emb), logit.data_ptr<float>(), rgb.data_ptr<float>(),
        /*pre_cm=*/nullptr,  // the matrix-core backward recomputes the hidden layer (cheaper than
                             // 256 B per sample through HBM twice)
        n, f2n::current_stream(enc_cm)),
      "f2n_shade_fwd");
    }
    ctx->save_for_backward(
      {enc_cm, dirs, use_emb ? sample_img : Tensor(), w_h, b_h, w1, b1, w2, b2, emb});
    return {logit, rgb};
  }

  static variable_list backward(AutogradContext * ctx, variable_list grad_output)
  {
    auto sv = ctx->get_saved_variables();
    Tensor &enc_cm = sv[0], &dirs = sv[1], &sample_img = sv[2], &w_h = sv[3], &b_h = sv[4],
           &w1 = sv[5], &b1 = sv[6], &w2 = sv[7], &b2 = sv[8], &emb = sv[9];
    const int C = (int)enc_cm.size(0);
    const int64_t n = enc_cm.size(1);
    auto opt = enc_cm.options();
    Tensor d_logit = grad_output[0].defined() ? f2n::dev_f32(grad_output[0], "d_logit")
                                              : torch::zeros({n}, opt);
    Tensor d_rgb = grad_output[1].defined() ? f2n::dev_f32(grad_output[1], "d_rgb")
                                            : torch::zeros({n, 3}, opt);
    Tensor d_enc_cm = torch::empty({(int64_t)C, n}, opt);
    // the seven small parameter gradients are views of ONE zero-filled buffer (one fill launch
    // instead of seven; each view starts 16-byte aligned)
    const bool use_emb = emb.defined() && sample_img.defined();
    auto pad4 = [](int64_t v) { return (v + 3) / 4 * 4; };
    const int64_t sizes[7] = {w_h.numel(), b_h.numel(), w1.numel(), b1.numel(),
                              w2.numel(), b2.numel(), use_emb ? emb.numel() : 0};
    int64_t total = 0;
    for (int64_t v : sizes) total += pad4(v);
    Tensor gbuf = torch::zeros({total}, opt);
    int64_t off = 0;
    auto take = [&](const Tensor & like, int64_t numel) {
      Tensor v = gbuf.narrow(0, off, numel).view(like.sizes());
      off += pad4(numel);
      return v;
    };
    Tensor g_w_h = take(w_h, sizes[0]), g_b_h = take(b_h, sizes[1]), g_w1 = take(w1, sizes[2]),
           g_b1 = take(b1, sizes[3]), g_w2 = take(w2, sizes[4]), g_b2 = take(b2, sizes[5]);
    Tensor g_emb = use_emb ? take(emb, sizes[6]) : Tensor();  // undefined = no gradient
    {
    f2n::ScopedKernelTimer timer("shade_bwd", f2n::current_stream(enc_cm), (double)n);
    f2n::check(
      f2n_shade_bwd(
        enc_cm.data_ptr<float>(), C, dirs.data_ptr<float>(), use_emb ? f2n::iptr(sample_img) : nullptr,
        f2n::fptr(w_h), f2n::fptr(b_h), f2n::fptr(w1), f2n::fptr(b1), f2n::fptr(w2), f2n::fptr(b2),
        f2n::fptr(emb), f2n::fptr(d_logit), f2n::fptr(d_rgb), d_enc_cm.data_ptr<float>(),
        g_w_h.data_ptr<float>(), g_b_h.data_ptr<float>(), g_w1.data_ptr<float>(),
        g_b1.data_ptr<float>(), g_w2.data_ptr<float>(), g_b2.data_ptr<float>(),
        use_emb ? g_emb.data_ptr<float>() : nullptr, /*pre_cm=*/nullptr, n,
        f2n::current_stream(enc_cm)),
      "f2n_shade_bwd");
    }
    // d_enc goes back as an [n, C] view of channel-major storage: f2n_hash_bwd reads it in place
    return {d_enc_cm.t(), Tensor(), Tensor(), g_w_h, g_b_h, g_w1, g_b1, g_w2, g_b2, g_emb};
  }
};


class TrainLossFn : public torch::autograd::Function<TrainLossFn>
{
public:
  static variable_list forward(
    AutogradContext * ctx, Tensor colors, Tensor gt, Tensor var, double var_weight)
  {
    colors = f2n::dev_f32(colors, "train_loss colors");
    gt = f2n::dev_f32(gt, "train_loss gt_colors");
    var = f2n::dev_f32(var, "train_loss var");
    const int64_t n_rays = colors.size(0);
    TORCH_CHECK(
      colors.dim() == 2 && colors.size(1) == 3 && gt.sizes() == colors.sizes() &&
        var.numel() == n_rays && n_rays > 0 && n_rays <= INT32_MAX,
      "train_loss: colors/gt [n_rays,3], var [n_rays]");
    Tensor d_colors = torch::empty_like(colors), d_var = torch::empty({n_rays}, colors.options());
    Tensor partial = torch::empty({f2n_loss_workspace_floats((int)n_rays)}, colors.options());
    Tensor out = torch::empty({4}, colors.options());
    f2n::check(
      f2n_loss_fwd(
        colors.data_ptr<float>(), gt.data_ptr<float>(), var.data_ptr<float>(), (int)n_rays,
        (float)var_weight, d_colors.data_ptr<float>(), d_var.data_ptr<float>(),
        partial.data_ptr<float>(), out.data_ptr<float>(), f2n::current_stream(colors)),
      "f2n_loss_fwd");
    ctx->save_for_backward({d_colors, d_var});
    ctx->saved_data["var_shape"] = var.sizes().vec();
    return {out};
  }

  static variable_list backward(AutogradContext * ctx, variable_list grad_output)
  {
    auto sv = ctx->get_saved_variables();
    Tensor g = grad_output[0].select(0, 0);  // only the loss element carries gradient
    Tensor d_var = (sv[1] * g).view(ctx->saved_data["var_shape"].toIntVector());
    return {sv[0] * g, Tensor(), d_var, Tensor()};
  }
};

}  // namespace

namespace torch::autograd
{

// exp with a clamped backward (reference src/CustomOps/CustomOps.cpp:10-20): pure ATen there and here
variable_list TruncExp::forward(AutogradContext * ctx, Tensor input)
{
  ctx->save_for_backward({input});
  return {torch::exp(input)};
}

variable_list TruncExp::backward(AutogradContext * ctx, variable_list grad_output)
{
  Tensor x = ctx->get_saved_variables()[0];
  return {grad_output[0] * torch::exp(x.clamp(-100.f, 5.f))};
}

}  // namespace torch::autograd

Tensor FlexOps::Sum(Tensor val, Tensor idx_start_end)
{
  return FlexSumFn::apply(val.contiguous(), idx_start_end.contiguous())[0];
}

Tensor FlexOps::AccumulateSum(Tensor val, Tensor idx_start_end, bool include_this)
{
  return FlexAccumulateSumFn::apply(val.contiguous(), idx_start_end.contiguous(), include_this)[0];
}

Tensor CustomOps::WeightVar(Tensor weights, Tensor idx_start_end)
{
  return WeightVarFn::apply(weights.contiguous(), idx_start_end.contiguous())[0];
}

Tensor CustomOps::ScatterAdd(Tensor emb, Tensor idx, Tensor to_add)
{
  return ScatterAddFn::apply(emb, idx, to_add)[0];
}

Tensor CustomOps::ScatterIdx(int n_all_pts, Tensor idx_start_end, Tensor emb_idx)
{
  idx_start_end = f2n::dev_i32(idx_start_end, "CustomOps::ScatterIdx idx_start_end");
  emb_idx = f2n::dev_i32(emb_idx, "CustomOps::ScatterIdx emb_idx");
  Tensor ret = torch::empty({n_all_pts}, idx_start_end.options());
  f2n::check(
    f2n_scatter_idx(
      f2n::iptr(idx_start_end), f2n::iptr(emb_idx), ret.data_ptr<int32_t>(),
      (int)idx_start_end.size(0), f2n::current_stream(idx_start_end)),
    "f2n_scatter_idx");
  return ret;
}

f2n::ShadeOut f2n::shade(
  const Tensor & enc, const Tensor & dirs, const Tensor & sample_img, const Tensor & w_h,
  const Tensor & b_h, const Tensor & w1, const Tensor & b1, const Tensor & w2, const Tensor & b2,
  const Tensor & app_emb)
{
  const Tensor no_img = torch::empty({0}, f2n::int_on(enc.device()));
  const Tensor no_emb = torch::empty({0, 16}, enc.options());
  const bool use_emb = sample_img.defined() && app_emb.defined();
  auto out = ShadeFn::apply(
    enc, dirs, use_emb ? sample_img : no_img, w_h, b_h, w1, b1, w2, b2, use_emb ? app_emb : no_emb);
  return {out[0], out[1]};
}

f2n::CompositeOut f2n::composite(
  const Tensor & field_out, const Tensor & rgb, const Tensor & dt, const Tensor & t,
  const Tensor & idx_start_end, const Tensor & bg_color, bool bounds_tile_samples)
{
  auto out = CompositeFn::apply(field_out, rgb, dt, t, idx_start_end, bg_color, bounds_tile_samples);
  return {out[0], out[1], out[2]};
}

Tensor f2n::train_loss(
  const Tensor & colors, const Tensor & gt_colors, const Tensor & var, float var_loss_weight)
{
  return TrainLossFn::apply(colors, gt_colors, var, (double)var_loss_weight)[0];
}

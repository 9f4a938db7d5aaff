// hash_3d_anchored.cpp -- host half of the hash-grid field (see hash_3d_anchored.hpp).
// Behaviour follows reference src/hash_3d_anchored.cpp:19-114 (construction, query, optimiser
// groups) and src/hash_3d_anchored.cu:150-218 (the autograd Function around the kernels).
#include "hash_3d_anchored.hpp"

#include "kernel_timer.hpp"

#include <cmath>

#include <c10/hip/HIPGuard.h>
#include <hip/hip_runtime_api.h>

using Tensor = torch::Tensor;

TORCH_LIBRARY(dec_hash3d_anchored, m)
{
  m.class_<Hash3DAnchoredInfo>("Hash3DAnchoredInfo").def(torch::init());
}

namespace
{

// x = p inside the unit ball, (2 - 1/|p|) p/|p| outside; |p| == 0 gives NaN like the reference's
// mask expression.  Backward = the Jacobian-vector product autograd would assemble from it.
class ContractFn : public torch::autograd::Function<ContractFn>
{
public:
  static torch::autograd::variable_list forward(torch::autograd::AutogradContext * ctx, Tensor points)
  {
    points = f2n::dev_f32(points, "Hash3DAnchored::query points");
    TORCH_CHECK(points.dim() == 2 && points.size(1) == 3, "points must be [n, 3]");
    Tensor x = torch::empty_like(points);
    f2n::check(
      f2n_contract_fwd(
        points.data_ptr<float>(), x.data_ptr<float>(), points.size(0), f2n::current_stream(points)),
      "f2n_contract_fwd");
    ctx->save_for_backward({points});
    return {x};
  }

  static torch::autograd::variable_list backward(
    torch::autograd::AutogradContext * ctx, torch::autograd::variable_list grad_output)
  {
    Tensor points = ctx->get_saved_variables()[0];
    Tensor dx = f2n::dev_f32(grad_output[0], "contraction grad");
    Tensor dp = torch::empty_like(points);
    f2n::check(
      f2n_contract_bwd(
        points.data_ptr<float>(), dx.data_ptr<float>(), dp.data_ptr<float>(), points.size(0),
        f2n::current_stream(points)),
      "f2n_contract_bwd");
    return {dp};
  }
};

bool is_prime(int64_t x)
{
  if (x < 2) return false;
  for (int64_t i = 2; i * i <= x; i++)
    if (x % i == 0) return false;
  return true;
}

}  // namespace

Hash3DAnchored::Hash3DAnchored(const Hash3DAnchoredOptions & opt) : options_(opt)
{
  const int64_t L = opt.n_levels, F = opt.n_channels;
  TORCH_CHECK(L >= 1 && L <= F2N_MAX_LEVELS, "n_levels out of range");
  TORCH_CHECK(F == 1 || F == 2 || F == 4 || F == 8, "n_channels must be 1, 2, 4 or 8");
  const auto fopt = f2n::float_on(opt.device);

  pool_size_ = (int)((int64_t(1) << opt.log2_table) * L);
  local_size_ = (int)(((pool_size_ / L) >> 4) << 4);  // rows per level, reference .cpp:57-58
  level_stride_ = opt.level_stride > 0 ? opt.level_stride : local_size_;
  TORCH_CHECK(level_stride_ % F == 0, "level_stride must be a multiple of n_channels");
  // rows needed so the last level's window [stride*(L-1), stride*(L-1) + T*F) stays in bounds
  const int64_t need_rows = (level_stride_ * (L - 1) + int64_t(local_size_) * F + F - 1) / F;
  const int64_t rows = std::max<int64_t>(pool_size_, need_rows);

  // (U[0,1) * 0.2 - 1) * 1e-4, reference .cpp:24
  feat_pool_ = (torch::rand({rows, F}, fopt) * .2f - 1.f) * 1e-4f;
  feat_pool_.requires_grad_(true);

  // 3L random primes in [2^28, 2^30), drawn on the CPU generator like the reference (.cpp:29-48)
  std::vector<int32_t> primes;
  const auto cpu_int = torch::TensorOptions().dtype(torch::kInt32).device(torch::kCPU);
  while ((int64_t)primes.size() < 3 * L) {
    const int v = torch::randint(1 << 28, 1 << 30, {1}, cpu_int).item<int>();
    if (is_prime(v)) primes.push_back(v);
  }
  prim_pool_ = torch::from_blob(primes.data(), {L, 3}, cpu_int).clone().to(opt.device).contiguous();
  bias_pool_ = (torch::rand({L, 3}, fopt) * 1000.f + 100.f).contiguous();

  // per-level scale, reference hash_3d_anchored.cu:72-73, evaluated in f32 with glibc exp2f
  std::vector<float> mul((size_t)L);
  for (int64_t l = 0; l < L; l++) {
    const float e = (L > 1) ? ((10.f - 3.f) * float(l) / float(L - 1) + 3.f) : 3.f;
    mul[(size_t)l] = exp2f(e);
  }
  level_mul_ =
    torch::from_blob(mul.data(), {L}, torch::TensorOptions().dtype(torch::kFloat32)).clone().to(
      opt.device);

  mlp_ = torch::nn::Linear(L * F, opt.mlp_out_dim);
  mlp_->to(opt.device);

  register_parameter("feat_pool", feat_pool_);
  register_parameter("prim_pool", prim_pool_, false);
  register_parameter("bias_pool", bias_pool_);
  register_module("mlp", mlp_);
}

Tensor Hash3DAnchored::table_f16()
{
  TORCH_CHECK(feat_pool_.is_cuda(), "Hash3DAnchored: feat_pool must live on the GPU");
  const void * src = feat_pool_.data_ptr();
  const uint32_t ver = feat_pool_._version();
  if (!feat_pool_f16_.defined() || src != shadow_src_ || ver != shadow_version_ ||
      feat_pool_f16_.numel() != feat_pool_.numel()) {
    Tensor master = feat_pool_.detach().contiguous();
    if (!feat_pool_f16_.defined() || feat_pool_f16_.numel() != master.numel() ||
        feat_pool_f16_.device() != master.device())
      feat_pool_f16_ = torch::empty(master.sizes(), master.options().dtype(torch::kFloat16));
    f2n::check(
      f2n_table_to_f16(
        master.data_ptr<float>(), reinterpret_cast<uint16_t *>(feat_pool_f16_.data_ptr()),
        master.numel(), f2n::current_stream(master)),
      "f2n_table_to_f16");
    shadow_src_ = src;
    shadow_version_ = ver;
  }
  return feat_pool_f16_;
}

Tensor Hash3DAnchored::shadow_storage()
{
  if (!feat_pool_f16_.defined() || feat_pool_f16_.numel() != feat_pool_.numel() ||
      feat_pool_f16_.device() != feat_pool_.device())
    feat_pool_f16_ = torch::empty(feat_pool_.sizes(), feat_pool_.options().dtype(torch::kFloat16));
  return feat_pool_f16_;
}

void Hash3DAnchored::mark_shadow_fresh()
{
  shadow_src_ = feat_pool_.data_ptr();
  shadow_version_ = feat_pool_._version();
}

Tensor Hash3DAnchored::table_for(const Tensor & feat_pool)
{
  if (feat_pool.data_ptr() == feat_pool_.data_ptr()) return table_f16();
  // a table other than the module's own parameter: cast it on the spot, as the reference does
  Tensor master = f2n::dev_f32(feat_pool.detach(), "feat_pool");
  Tensor t16 = torch::empty(master.sizes(), master.options().dtype(torch::kFloat16));
  f2n::check(
    f2n_table_to_f16(
      master.data_ptr<float>(), reinterpret_cast<uint16_t *>(t16.data_ptr()), master.numel(),
      f2n::current_stream(master)),
    "f2n_table_to_f16");
  return t16;
}

std::pair<Tensor, Tensor> Hash3DAnchored::density_head() const
{
  return {mlp_->weight.detach().select(0, 0).contiguous(),
          mlp_->bias.detach().slice(0, 0, 1).contiguous()};
}

Tensor Hash3DAnchored::encode(const Tensor & points, int64_t samples_per_ray, Tensor * contracted_out)
{
  auto info = torch::make_intrusive<Hash3DAnchoredInfo>();
  info->hash3d_ = this;
  info->samples_per_ray_ = samples_per_ray;

  // scene contraction (reference .cpp:79-82: eight ATen launches) as one kernel each way
  Tensor x = ContractFn::apply(points)[0];
  if (contracted_out) *contracted_out = x.detach();
  return torch::autograd::Hash3DAnchoredFunction::apply(x, feat_pool_, torch::IValue(info))[0];
}

Tensor Hash3DAnchored::encode_cached(
  const Tensor & points, const Tensor & enc_cm, const Tensor & contracted)
{
  TORCH_CHECK(
    enc_cm.dim() == 2 && enc_cm.is_contiguous() && enc_cm.size(1) == points.size(0) &&
      enc_cm.size(0) == options_.n_levels * options_.n_channels,
    "encode_cached: enc_cm must be contiguous [L*F, n]");
  auto info = torch::make_intrusive<Hash3DAnchoredInfo>();
  info->hash3d_ = this;
  info->precomputed_cm_ = enc_cm;
  const bool reuse = contracted.defined() && contracted.sizes() == points.sizes() &&
                     !(torch::GradMode::is_enabled() && points.requires_grad());
  Tensor x = reuse ? contracted : ContractFn::apply(points)[0];
  return torch::autograd::Hash3DAnchoredFunction::apply(x, feat_pool_, torch::IValue(info))[0];
}

Tensor Hash3DAnchored::query(const Tensor & points)
{
  return mlp_->forward(encode(points));
}

std::vector<torch::optim::OptimizerParamGroup> Hash3DAnchored::optim_param_groups(float lr)
{
  // Adam, betas (0.9, 0.99), eps 1e-15; weight decay 1e-6 on the MLP only (reference .cpp:90-114)
  std::vector<torch::optim::OptimizerParamGroup> groups;
  auto table_opt = std::make_unique<torch::optim::AdamOptions>(lr);
  table_opt->betas(std::make_tuple(0.9, 0.99)).eps(1e-15);
  groups.emplace_back(std::vector<Tensor>{feat_pool_}, std::move(table_opt));

  auto mlp_opt = std::make_unique<torch::optim::AdamOptions>(lr);
  mlp_opt->betas(std::make_tuple(0.9, 0.99)).eps(1e-15).weight_decay(1e-6);
  groups.emplace_back(mlp_->parameters(), std::move(mlp_opt));
  return groups;
}

namespace torch::autograd
{

variable_list Hash3DAnchoredFunction::forward(
  AutogradContext * ctx, Tensor points, Tensor feat_pool, IValue hash3d_info)
{
  auto info = hash3d_info.toCustomClass<Hash3DAnchoredInfo>();
  Hash3DAnchored * field = info->hash3d_;
  ctx->saved_data["hash3d_info"] = hash3d_info;
  points = f2n::dev_f32(points, "Hash3DAnchoredFunction points");
  TORCH_CHECK(points.dim() == 2 && points.size(1) == 3, "points must be [n, 3]");
  ctx->save_for_backward({points, feat_pool});

  const int64_t n = points.size(0);
  const int L = (int)field->options_.n_levels, F = (int)field->options_.n_channels;
  if (info->precomputed_cm_.defined()) return {info->precomputed_cm_.t()};
  Tensor table16 = field->table_for(feat_pool);
  // f32 tensor holding f16-rounded values (the reference's out_feat.to(kFloat32), fused).  Storage
  // is channel-major [L*F, n] -- every wavefront store is one coalesced 256-byte row segment instead
  // of 64 scattered 8-byte pieces (2.5x faster encode on MI355X) -- and the caller receives the
  // reference's logical shape [n, L*F] as a transposed view.
  Tensor out_cm = torch::empty({(int64_t)L * F, n}, points.options());
  {
    f2n::ScopedKernelTimer timer("hash_fwd", f2n::current_stream(points), (double)n);
    const int64_t spr = info->samples_per_ray_;
    if (spr > 0 && spr % 16 == 0 && n % spr == 0 && n / spr <= INT32_MAX) {
      f2n::check(
        f2n_hash_fwd_raytile(
          points.data_ptr<float>(), reinterpret_cast<const uint16_t *>(table16.data_ptr()),
          field->prim_pool_.data_ptr<int32_t>(), field->bias_pool_.data_ptr<float>(),
          field->level_mul_.data_ptr<float>(), out_cm.data_ptr<float>(), (int)(n / spr), (int)spr, L,
          F, (uint32_t)field->local_size_, field->level_stride_, f2n::current_stream(points)),
        "f2n_hash_fwd_raytile");
    } else {
      f2n::check(
        f2n_hash_fwd(
          points.data_ptr<float>(), reinterpret_cast<const uint16_t *>(table16.data_ptr()),
          field->prim_pool_.data_ptr<int32_t>(), field->bias_pool_.data_ptr<float>(),
          field->level_mul_.data_ptr<float>(), out_cm.data_ptr<float>(), 1, n, nullptr, n, L, F,
          (uint32_t)field->local_size_, field->level_stride_, f2n::current_stream(points)),
        "f2n_hash_fwd");
    }
  }
  Tensor out = out_cm.t();
  return {out};
}

variable_list Hash3DAnchoredFunction::backward(AutogradContext * ctx, variable_list grad_output)
{
  auto info = ctx->saved_data["hash3d_info"].toCustomClass<Hash3DAnchoredInfo>();
  Hash3DAnchored * field = info->hash3d_;
  auto saved = ctx->get_saved_variables();
  Tensor & points = saved[0];
  Tensor & feat_pool = saved[1];

  const int64_t n = points.size(0);
  const int L = (int)field->options_.n_levels, F = (int)field->options_.n_channels;
  const float grad_scale = 128.f;  // reference hash_3d_anchored.cu:190

  // the incoming gradient may be row-major [n, C] or a transposed (channel-major) view; both are
  // consumed in place through the kernel's (point, channel) strides
  Tensor grad_in = grad_output[0];
  TORCH_CHECK(grad_in.is_cuda() && grad_in.scalar_type() == torch::kFloat32, "hash grad dtype");
  const int64_t C = (int64_t)L * F;
  int64_t ld_point, ld_chan;
  if (grad_in.stride(1) == 1 && grad_in.stride(0) >= C) {
    ld_point = grad_in.stride(0);
    ld_chan = 1;
  } else if (grad_in.stride(0) == 1 && grad_in.stride(1) >= n) {
    ld_point = 1;
    ld_chan = grad_in.stride(1);
  } else {
    grad_in = grad_in.contiguous();
    ld_point = C;
    ld_chan = 1;
  }
  Tensor table16 = field->table_for(feat_pool);
  // points need a gradient only for pose optimisation; training rays are data
  const bool want_points = ctx->needs_input_grad(0);
  Tensor points_grad = want_points ? torch::empty({n, 3}, points.options()) : Tensor();
  // add into the gradient feat_pool already has (options_.accumulate_in_place), or start a new one
  Tensor embeds_grad;
  bool in_place = false;
  if (field->options_.accumulate_in_place && !torch::GradMode::is_enabled() && feat_pool.is_leaf()) {
    const Tensor & have = feat_pool.grad();
    if (have.defined() && have.is_contiguous() && have.scalar_type() == torch::kFloat32 &&
        have.device() == feat_pool.device() && have.sizes() == feat_pool.sizes()) {
      embeds_grad = have;
      in_place = true;
    }
  }
  if (!in_place) embeds_grad = torch::zeros_like(feat_pool);
  void * stream = f2n::current_stream(points);
  // Large training batches: bin the contributions by table slice into a scratch workspace and
  // reduce them in LDS (no scattered atomics; exact, order-independent sums).  The recommended
  // workspace is about 3 bytes per algorithmic byte (16 GiB for 8.4 M samples at L = 16, F = 2,
  // capped at 64 GiB: bigger batches run in rounds); it is further capped at half of the device's
  // free memory, and when even that cannot be allocated the atomic kernel takes over.
  int64_t ws_bytes =
    want_points ? 0 : f2n_hash_bwd_workspace_bytes(n, L, F, (uint32_t)field->local_size_);
  Tensor ws;
  if (ws_bytes > 0 && field->options_.binned_backward) {
    size_t free_b = 0, total_b = 0;
    c10::hip::HIPGuard on_device(points.device().index());  // the query is about THIS tensor's device
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    hipStreamIsCapturing((hipStream_t)stream, &capturing);
    // (not while a hipGraph is being captured: the query is not a stream operation, and the capture's
    // private pool serves the allocation below either way)
    if (capturing == hipStreamCaptureStatusNone && hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
      // memory held by the caching allocator is reusable too: only cap when the device is tight
      const int64_t cap = (int64_t)(free_b / 2);
      if (cap < ws_bytes && cap >= ws_bytes / 16) ws_bytes = cap / 256 * 256;
    }
    try {
      ws = torch::empty({ws_bytes}, points.options().dtype(torch::kUInt8));
    } catch (const c10::Error &) {
      ws = Tensor();  // out of memory: fall through to the kernel that needs no workspace
    }
  }
  bool done = false;
  if (ws.defined()) {
    f2n::ScopedKernelTimer timer("hash_bwd", stream, (double)n);
    const int st = f2n_hash_bwd_binned(
      points.data_ptr<float>(), field->prim_pool_.data_ptr<int32_t>(),
      field->bias_pool_.data_ptr<float>(), field->level_mul_.data_ptr<float>(),
      grad_in.data_ptr<float>(), ld_point, ld_chan, embeds_grad.data_ptr<float>(), n, L, F,
      (uint32_t)field->local_size_, field->level_stride_, grad_scale, ws.data_ptr(), ws_bytes,
      stream);
    if (st != F2N_E_UNSUPPORTED) f2n::check(st, "f2n_hash_bwd_binned");
    done = (st == F2N_OK);  // unsupported = the (shrunk) workspace holds not even one tile
  }
  if (!done) {
    f2n::ScopedKernelTimer timer("hash_bwd", stream, (double)n);
    f2n::check(
      f2n_hash_bwd(
        points.data_ptr<float>(), reinterpret_cast<const uint16_t *>(table16.data_ptr()),
        field->prim_pool_.data_ptr<int32_t>(), field->bias_pool_.data_ptr<float>(),
        field->level_mul_.data_ptr<float>(), grad_in.data_ptr<float>(), ld_point, ld_chan,
        embeds_grad.data_ptr<float>(), want_points ? points_grad.data_ptr<float>() : nullptr, n, L,
        F, (uint32_t)field->local_size_, field->level_stride_, grad_scale, stream),
      "f2n_hash_bwd");
  }
  return {points_grad, in_place ? Tensor() : embeds_grad, Tensor()};
}

}  // namespace torch::autograd

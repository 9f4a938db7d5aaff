// hash_3d_anchored.hpp -- Hash3DAnchored: the multi-resolution hash-grid scene field.
//
// Same public surface as the reference class (reference src/hash_3d_anchored.hpp:13-52): a
// torch::nn::Module with feat_pool_ / prim_pool_ / bias_pool_ / mlp_, query(points),
// optim_param_groups(lr), the Hash3DAnchoredInfo custom-class holder and
// torch::autograd::Hash3DAnchoredFunction(points, feat_pool, IValue info).  Parameter names and
// shapes are kept ("feat_pool", "prim_pool", "bias_pool", "mlp.*") so a reference checkpoint loads.
//
// What is different, by design:
//   * N_LEVELS / N_CHANNELS / table size are runtime options (defaults = the reference's
//     compile-time constants, hash_3d_anchored.hpp:10-11, hash_3d_anchored.cpp:21);
//   * the f16 working copy of the table is a persistent shadow refreshed only when feat_pool_
//     changes (the reference re-casts all 2^24 elements on every forward AND backward,
//     hash_3d_anchored.cu:169,198); the shadow is the same RNE cast of the same f32 master;
//   * the per-level scale table mul[l] is computed once on the host (exp2f) and passed to the kernels.
#pragma once

#include "common.hpp"

struct Hash3DAnchoredOptions
{
  int64_t n_levels = 16;     // N_LEVELS
  int64_t n_channels = 2;    // N_CHANNELS (features per level): 1, 2, 4 or 8
  int64_t log2_table = 19;   // rows per level = 2^log2_table (reference: pool = 2^19 * N_LEVELS)
  int64_t level_stride = 0;  // elements between level bases; 0 = reference behaviour (= rows per
                             // level, so adjacent levels overlap)
  int64_t mlp_out_dim = 16;
  bool binned_backward = true;  // large batches: f2n_hash_bwd_binned (needs a scratch workspace)
  // The table's gradient is 64 MiB.  Handed to autograd per backward call it costs a zero fill and,
  // from the second call of an iteration on (a view rendered in chunks), an accumulation pass over
  // all of it -- 48 us per call for a kernel that may take 180.  When feat_pool already has a
  // gradient, the kernels add into it directly and autograd is told "no gradient" for that input.
  // Tensor hooks on feat_pool do not see those calls; switch it off if you rely on them.
  bool accumulate_in_place = true;
  torch::Device device = f2n::default_device();
};

class Hash3DAnchored : public torch::nn::Module
{
  using Tensor = torch::Tensor;

public:
  explicit Hash3DAnchored(const Hash3DAnchoredOptions & opt = {});

  Tensor query(const Tensor & points);

  // contraction + hash encode only (no Linear): [n, L*F], stored channel-major.  The Renderer's
  // fused path feeds this straight into the fused per-sample network kernel.
  // samples_per_ray > 0 declares `points` a dense ray-major [n_rays, samples_per_ray] grid: the
  // forward then walks neighbouring rays per wavefront (f2n_hash_fwd_raytile), same results.
  // contracted_out, if given, receives the contracted points (to hand back to encode_cached).
  Tensor encode(const Tensor & points, int64_t samples_per_ray = 0, Tensor * contracted_out = nullptr);

  // Same autograd node as encode(), but the forward result is supplied: `enc_cm` [L*F, n]
  // channel-major, the encoding of exactly these points computed earlier (the Renderer's first pass).
  // Only the backward (table gradient) runs a kernel.
  // `contracted`, if defined, is the contraction of exactly these points (from encode()); it is
  // used as is when the points carry no gradient.
  Tensor encode_cached(
    const Tensor & points, const Tensor & enc_cm, const Tensor & contracted = Tensor());

  // Row 0 of mlp_ (weight [L*F], bias [1]) as contiguous device tensors: the density head the fused
  // ray march evaluates in-kernel.
  std::pair<Tensor, Tensor> density_head() const;

  // f16 working copy of feat_pool_, refreshed if feat_pool_ was modified since the last call.
  Tensor table_f16();
  // f16 copy of an arbitrary table tensor (the shadow when it is feat_pool_ itself).
  Tensor table_for(const Tensor & feat_pool);

  // For an optimiser that rewrites the shadow itself while updating feat_pool_ (FusedAdam): the
  // shadow buffer (allocated on demand), and the call that declares it current.
  Tensor shadow_storage();
  void mark_shadow_fresh();

  std::vector<torch::optim::OptimizerParamGroup> optim_param_groups(float lr);

  Hash3DAnchoredOptions options_;
  int pool_size_;
  int local_size_;
  int64_t level_stride_;

  Tensor feat_pool_;  // [ pool_size_, n_channels ]
  Tensor prim_pool_;  // [ n_levels, 3 ] int32
  Tensor bias_pool_;  // [ n_levels, 3 ]
  Tensor level_mul_;  // [ n_levels ] f32, exp2f(7 l/(L-1) + 3)  (not a parameter)

  torch::nn::Linear mlp_ = nullptr;

private:
  Tensor feat_pool_f16_;
  const void * shadow_src_ = nullptr;
  uint32_t shadow_version_ = 0;
};

class Hash3DAnchoredInfo : public torch::CustomClassHolder
{
public:
  Hash3DAnchored * hash3d_ = nullptr;
  torch::Tensor precomputed_cm_;  // optional [L*F, n] encoding to return instead of computing it
  int64_t samples_per_ray_ = 0;   // > 0: points are a dense [n_rays, samples_per_ray] grid
};

namespace torch::autograd
{

class Hash3DAnchoredFunction : public Function<Hash3DAnchoredFunction>
{
public:
  static variable_list forward(
    AutogradContext * ctx, Tensor points, Tensor feat_pool_, IValue hash3d_info);

  static variable_list backward(AutogradContext * ctx, variable_list grad_output);
};

}  // namespace torch::autograd

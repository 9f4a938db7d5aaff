// renderer.hpp -- Renderer: sampler + scene field + shader + per-image appearance embedding, and
// the volumetric render of a batch of rays.
//
// Public surface of reference src/renderer.hpp:15-49 (RenderResult, Renderer::render /
// render_all_rays / render_image / optim_param_groups); registered names "scene_field", "shader",
// "app_emb" kept so a reference checkpoint (renderer.pt) loads.
//
// render() has two implementations of the same arithmetic (reference src/renderer.cpp:33-123):
//   fused (default)   one wavefront-per-ray march finds each ray's kept prefix (early termination
//                     in-kernel), a scan turns counts into bounds, one kernel emits the compacted
//                     samples, and compositing is one kernel per direction;
//   op-by-op          the reference's own sequence (sample all, query all, AccumulateSum, where,
//                     4x index, query survivors, Sum...) on the drop-in operators.  Used when rays
//                     carry gradients (pose optimisation) and as the cross-check of the fused path.
#pragma once

#include <memory>
#include <tuple>
#include <vector>

#include "hash_3d_anchored.hpp"
#include "points_sampler.hpp"
#include "sh_shader.hpp"

namespace f2n
{
// One int32 read back from the device without stalling the enqueue side: request() starts the copy
// into pinned memory behind everything already on the stream, wait() blocks on just that copy.
class HostCount
{
public:
  HostCount() = default;
  HostCount(const HostCount &) = delete;
  HostCount & operator=(const HostCount &) = delete;
  ~HostCount();
  void request(const torch::Tensor & device_int32, void * stream);
  int64_t wait();

private:
  int32_t * pinned_ = nullptr;
  void * event_ = nullptr;
  int device_ = -1;  // device the event belongs to (recreated when a request comes from another one)
  bool pending_ = false;
};
}  // namespace f2n

struct RenderResult
{
  using Tensor = torch::Tensor;
  Tensor colors;
  Tensor depths;
  Tensor weights;
  Tensor idx_start_end;
};

struct RendererOptions
{
  Hash3DAnchoredOptions field;
  PtsSamplerOptions sampler;
  bool fused = true;
  bool fused_shade = true;         // per-sample network as one kernel (needs L*F in {8,16,32,64})
  // First-pass strategy of the fused path: -1 = adaptive (dense when the previous call kept more
  // than 40 % of its samples), 0 = always the early-terminating march, 1 = always dense.
  // dense: encode ALL samples once with the level-major kernel, derive the keep-prefix from that
  // encoding and hand the (compacted) encoding to the shading pass, so the field is evaluated once
  // instead of twice when little terminates; march: stop rays in-kernel, re-encode survivors.
  int dense_first_pass = -1;
  float early_stop_trans = 1e-4f;  // renderer.cpp:68
  int pixel_tiles = 8;             // render_image traverses the view in tiles of this many pixels squared (0: rows)
  // Dense first pass: when the previous chunk kept every sample, shade all samples first and accept
  // that as the result if no ray comes near the early-stop threshold (see render_fused); results are
  // identical either way.
  bool speculate_dense = true;
  // ... chunks of at least this many samples accept the guess on the density-margin flag (no exact
  // scan at all), smaller ones run the scan and only hide its read-back behind the guess
  int64_t margin_min_samples = 2 << 20;
  // No host read at all (what a hipGraph capture of a training iteration needs; the reference
  // syncs at src/renderer.cpp:39-40,69,85-87,120): the dense first pass shades ALL samples and
  // returns that as the result without waiting for the survivor count; whether a ray would have
  // terminated early is ACCUMULATED on the device instead (deferred_bad_: += 1 for every render whose
  // exact scan kept fewer samples than it shaded) and the caller asks deferred_check_ok() whenever it
  // next synchronises anyway -- before it lets an optimiser consume the gradients.  A render whose
  // check fails produced colours that include samples the reference would have dropped (their
  // weights are below 1e-4 of the ray's, but not zero) and must be repeated without this option.
  bool deferred_check = false;
  bool check_finite = false;       // the reference's CHECK(isfinite(colors.mean())) host sync
};

class Renderer : public torch::nn::Module
{
  using Tensor = torch::Tensor;

public:
  explicit Renderer(int n_images, const RendererOptions & opt = {});

  RenderResult render(
    const Tensor & rays_o, const Tensor & rays_d, const Tensor & emb_idx, RunningMode mode);

  // Same, with the device-side randomness supplied: `noise` [n_rays, max_samples] (undefined =
  // draw as the mode dictates), `bg_color` [n_rays, 3] (undefined = rand in TRAIN, 0.5 otherwise).
  RenderResult render(
    const Tensor & rays_o, const Tensor & rays_d, const Tensor & emb_idx, RunningMode mode,
    const Tensor & noise, const Tensor & bg_color);

  std::tuple<Tensor, Tensor> render_all_rays(
    const Tensor & rays_o, const Tensor & rays_d, const int batch_size);

  std::tuple<Tensor, Tensor> render_image(
    const torch::Tensor & pose, const torch::Tensor & intrinsic, const int h, const int w,
    const int batch_size);

  std::vector<torch::optim::OptimizerParamGroup> optim_param_groups(float lr);

  RendererOptions options_;
  std::shared_ptr<PtsSampler> pts_sampler_;
  std::shared_ptr<Hash3DAnchored> scene_field_;
  std::shared_ptr<SHShader> shader_;
  Tensor app_emb_;

  int64_t last_n_samples_ = 0;  // survivors of the most recent render() (bench bookkeeping)
  float last_kept_fraction_ = 0.f;
  f2n::HostCount survivors_;

  // options_.deferred_check: renders since the last reset whose guess "nothing terminates" was wrong
  // (device int32, created on first use; stays valid across hipGraph replays).  deferred_check_ok()
  // reads it (a host sync) and resets it.
  Tensor deferred_bad_;
  bool deferred_check_ok();

private:
  RenderResult render_fused(
    const Tensor & rays_o, const Tensor & rays_d, const Tensor & emb_idx, RunningMode mode,
    const Tensor & noise, const Tensor & bg_color);
  RenderResult render_op_by_op(
    const Tensor & rays_o, const Tensor & rays_d, const Tensor & emb_idx, RunningMode mode,
    const Tensor & noise, const Tensor & bg_color);
  RenderResult shade_and_composite(
    const SampleResultFlex & kept, const Tensor & emb_idx, RunningMode mode,
    const Tensor & bg_color, const Tensor & enc_cm = Tensor(), const Tensor & contracted = Tensor(),
    Tensor * near_threshold = nullptr, int64_t grid_samples = 0);
};

namespace f2n
{

// The loss / backward segment of the reference's training loop
// (reference src/main_functions/train_manager.cpp:76-107 minus the optimiser step):
// render(TRAIN), Charbonnier colour loss, scheduled weight-variance loss, backward().
struct TrainStepResult
{
  Tensor loss;        // scalar
  Tensor sq_err_sum;  // scalar: sum over rays and channels of (pred - gt)^2
  int64_t n_values;   // n_rays * 3
  int64_t n_samples;  // surviving samples in this step
};

TrainStepResult train_step(
  Renderer & renderer, const Tensor & rays_o, const Tensor & rays_d, const Tensor & emb_idx,
  const Tensor & gt_colors, float var_loss_weight, const Tensor & noise, const Tensor & bg_color,
  bool run_backward);

}  // namespace f2n

// points_sampler.hpp -- PtsSampler: uniform jittered ray marching samples.
// Public surface of reference src/points_sampler.hpp:15-40 (SampleResultFlex, RunningMode,
// PtsSampler::get_samples); MAX_SAMPLE_PER_RAY / SAMPLE_L become runtime options whose defaults are
// the reference's constants (1024, 1/256).
#pragma once

#include "common.hpp"

constexpr int MAX_SAMPLE_PER_RAY = 1024;  // reference default (points_sampler.hpp:15)

struct SampleResultFlex
{
  using Tensor = torch::Tensor;
  Tensor pts;             // [ n_all_pts, 3 ]
  Tensor dirs;            // [ n_all_pts, 3 ]
  Tensor dt;              // [ n_all_pts ]
  Tensor t;               // [ n_all_pts ]
  Tensor pts_idx_bounds;  // [ n_rays, 2 ] start, end
};

enum RunningMode { TRAIN, VALIDATE };

struct PtsSamplerOptions
{
  int max_samples = MAX_SAMPLE_PER_RAY;
  float step = 1.0f / 256;  // SAMPLE_L (points_sampler.hpp:39)
};

class PtsSampler
{
  using Tensor = torch::Tensor;

public:
  explicit PtsSampler(const PtsSamplerOptions & opt = {});

  // One kernel instead of the reference's ~20 ATen launches (points_sampler.cpp:20-64).  Not
  // differentiable in the rays: use get_samples_aten() when rays carry gradients.
  SampleResultFlex get_samples(const Tensor & rays_o, const Tensor & rays_d, RunningMode mode);
  // Same, with the step-noise tensor [n_rays, max_samples] supplied (undefined = all ones).
  SampleResultFlex get_samples(const Tensor & rays_o, const Tensor & rays_d, const Tensor & noise);

  // The reference's own ATen formulation (differentiable in rays_o / rays_d through autograd).
  SampleResultFlex get_samples_aten(
    const Tensor & rays_o, const Tensor & rays_d, const Tensor & noise);

  // TRAIN: U[0,1) - 0.5 + 1 per sample (points_sampler.cpp:35); VALIDATE: undefined tensor (= ones).
  Tensor draw_noise(int64_t n_rays, RunningMode mode, const torch::Device & device) const;

  PtsSamplerOptions options_;
};

// rays.hpp -- pixel -> world-space ray generation (reference src/rays.hpp:7-13, src/rays.cpp:7-28)
// and the two callers that feed the renderer: a view's pixel grid (src/dataset.cpp:128-146,
// src/renderer.cpp:153-172) and the random training batch (src/dataset.cpp:150-171).
#pragma once

#include <torch/torch.h>

#include <tuple>

struct alignas(32) Rays
{
  torch::Tensor origins;
  torch::Tensor dirs;
};

// pose [B,3,4] (or [B,4,4]), intrinsic [B,3,3], ij [N,2] = (row, col), float or integer; B == 1 or
// B == N.  Pixel centres (+0.5), camera looks down -z, y up.  One kernel (f2n_gen_rays).
Rays get_rays_from_pose(
  const torch::Tensor & pose, const torch::Tensor & intrinsic, const torch::Tensor & ij);

// All h*w pixels of one view in row-major order, without materialising the pixel grid
// (Dataset::get_rays_from_pose(idx) / Renderer::render_image of the reference).
Rays get_view_rays(const torch::Tensor & pose, const torch::Tensor & intrinsic, int h, int w);

// Dataset::sample_random_rays (src/dataset.cpp:150-171) with everything on the device: camera and
// pixel indices are drawn there, each ray reads its own camera from the pose / intrinsic tables
// (no index_select, no host randint + copy).  images: optional [E, h, w, 3] for the ground truth.
// Returns {rays, gt_colors [n,3] (undefined without images), cam_indices [n] i32}.
std::tuple<Rays, torch::Tensor, torch::Tensor> sample_random_rays(
  const torch::Tensor & poses, const torch::Tensor & intrinsics, int h, int w, int64_t batch_size,
  const torch::Tensor & images = {});

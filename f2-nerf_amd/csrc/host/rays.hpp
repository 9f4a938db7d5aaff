// rays.hpp -- pixel -> world-space ray generation (reference src/rays.hpp:7-13, src/rays.cpp:7-28).
#pragma once

#include <torch/torch.h>

struct alignas(32) Rays
{
  torch::Tensor origins;
  torch::Tensor dirs;
};

// pose [B,3,4] (or [B,4,4]), intrinsic [B,3,3], ij [N,2] = (row, col); B == 1 or B == N.
// Pixel centres (+0.5), camera looks down -z, y up.
Rays get_rays_from_pose(
  const torch::Tensor & pose, const torch::Tensor & intrinsic, const torch::Tensor & ij);

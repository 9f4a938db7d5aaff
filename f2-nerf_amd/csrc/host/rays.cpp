// rays.cpp -- see rays.hpp.  Pure ATen, as in the reference (src/rays.cpp:7-28).
#include "rays.hpp"

#include "common.hpp"

using Tensor = torch::Tensor;

Rays get_rays_from_pose(const Tensor & pose, const Tensor & intrinsic, const Tensor & ij)
{
  Tensor row = ij.index({"...", 0}).to(torch::kFloat32) + .5f;
  Tensor col = ij.index({"...", 1}).to(torch::kFloat32) + .5f;
  Tensor fx = intrinsic.index({Slc(), 0, 0}), fy = intrinsic.index({Slc(), 1, 1});
  Tensor cx = intrinsic.index({Slc(), 0, 2}), cy = intrinsic.index({Slc(), 1, 2});
  Tensor u = ((col - cx) / fx).unsqueeze(-1);
  Tensor v = -((row - cy) / fy).unsqueeze(-1);
  Tensor cam_dir = torch::cat({u, v, -torch::ones_like(u)}, 1).unsqueeze(-1);  // [N,3,1]
  Tensor rot = pose.index({Slc(), Slc(0, 3), Slc(0, 3)});
  Tensor pos = pose.index({Slc(), Slc(0, 3), 3});
  Tensor rays_d = torch::matmul(rot, cam_dir).squeeze(-1);
  Tensor rays_o = pos.expand({rays_d.size(0), 3}).contiguous();
  return {rays_o, rays_d};
}

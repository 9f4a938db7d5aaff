// rays.cpp -- see rays.hpp.  The reference's get_rays_from_pose is an ATen chain ending in a batched
// 3x3 GEMM with one problem per ray (src/rays.cpp:7-28; 8.4 ms for the 640 000 rays of an 800x800
// view through rocBLAS); here it is one launch of f2n_gen_rays.
#include "rays.hpp"

#include "common.hpp"

using Tensor = torch::Tensor;

namespace
{

// poses as [B, 3|4, 4] contiguous f32 on the GPU -> (pointer, floats per pose)
std::pair<Tensor, int> pose_blocks(const Tensor & pose)
{
  TORCH_CHECK(
    pose.dim() == 3 && pose.size(2) == 4 && (pose.size(1) == 3 || pose.size(1) == 4),
    "pose must be [B,3,4] or [B,4,4]");
  return {f2n::dev_f32(pose, "pose"), (int)(pose.size(1) * 4)};
}

Rays launch_gen_rays(
  const Tensor & pose, const Tensor & intrinsic, const Tensor & cam_idx, const Tensor & ij,
  int64_t first_pixel, int width, int64_t n)
{
  auto [poses, pose_ld] = pose_blocks(pose);
  Tensor K = f2n::dev_f32(intrinsic, "intrinsic");
  TORCH_CHECK(
    K.dim() == 3 && K.size(1) == 3 && K.size(2) == 3 && K.size(0) == poses.size(0),
    "intrinsic must be [B,3,3] with the poses' B");
  Rays rays{torch::empty({n, 3}, poses.options()), torch::empty({n, 3}, poses.options())};
  f2n::check(
    f2n_gen_rays(
      poses.data_ptr<float>(), pose_ld, K.data_ptr<float>(), poses.size(0), f2n::iptr(cam_idx),
      f2n::iptr(ij), first_pixel, width, rays.origins.data_ptr<float>(),
      rays.dirs.data_ptr<float>(), n, f2n::current_stream(poses)),
    "f2n_gen_rays");
  return rays;
}

}  // namespace

Rays get_rays_from_pose(const Tensor & pose, const Tensor & intrinsic, const Tensor & ij)
{
  TORCH_CHECK(ij.dim() == 2 && ij.size(1) == 2, "ij must be [N,2]");
  const int64_t n = ij.size(0);
  TORCH_CHECK(
    pose.size(0) == 1 || pose.size(0) == n, "pose batch must be 1 or N (src/rays.cpp broadcast)");
  // the reference converts whatever ij holds with .to(kFloat32); pixel indices are exact either way
  Tensor ij32 = f2n::dev_i32(ij.to(torch::kInt32), "ij");
  return launch_gen_rays(pose, intrinsic, Tensor(), ij32, 0, 1, n);
}

Rays get_view_rays(const Tensor & pose, const Tensor & intrinsic, int h, int w)
{
  Tensor p = pose.dim() == 2 ? pose.unsqueeze(0) : pose;
  Tensor k = intrinsic.dim() == 2 ? intrinsic.unsqueeze(0) : intrinsic;
  TORCH_CHECK(p.size(0) == 1 && h > 0 && w > 0, "get_view_rays: one pose, a positive image size");
  return launch_gen_rays(p, k, Tensor(), Tensor(), 0, w, (int64_t)h * w);
}

std::tuple<Rays, Tensor, Tensor> sample_random_rays(
  const Tensor & poses, const Tensor & intrinsics, int h, int w, int64_t batch_size,
  const Tensor & images)
{
  const auto iopt = torch::TensorOptions().dtype(torch::kInt32).device(poses.device());
  const int64_t n_images = poses.size(0);
  Tensor cam = torch::randint(n_images, {batch_size}, iopt);
  Tensor i = torch::randint(0, h, {batch_size}, iopt);
  Tensor j = torch::randint(0, w, {batch_size}, iopt);
  Tensor ij = torch::stack({i, j}, -1).contiguous();
  Rays rays = launch_gen_rays(poses, intrinsics, cam, ij, 0, 1, batch_size);
  Tensor gt;
  if (images.defined()) {
    Tensor flat = (cam.to(torch::kLong) * h + i.to(torch::kLong)) * w + j.to(torch::kLong);
    gt = images.view({-1, 3}).index({flat}).to(poses.device()).contiguous();
  }
  return {rays, gt, cam};
}

// rays.hip -- pixel -> world-space ray (the caller just above the sampler).
// Replaces the ATen chain of get_rays_from_pose (reference src/rays.cpp:7-28: five index launches,
// a cat, and a batched 3x3 GEMM with one problem per ray) for both of its callers: the full pixel
// grid of a view (src/renderer.cpp:153-172, src/dataset.cpp:128-146) and the random training batch
// whose rays each pick their own camera (src/dataset.cpp:150-171) -- there the reference first
// gathers a [n,3,4] pose and a [n,3,3] intrinsic tensor; here the kernel indexes the camera tables.
//
// One thread per ray, 24 bytes out; camera constants are per-lane loads from tables of a few KB
// (L1/L2 resident; wave-uniform when every ray shares the camera).
#include "common.hiph"

namespace
{

__global__ __launch_bounds__(F2N_BLOCK) void gen_rays_kernel(
  const float * __restrict__ poses, int pose_ld, const float * __restrict__ intrinsics,
  int64_t n_cams, const int32_t * __restrict__ cam_idx, const int32_t * __restrict__ ij,
  int64_t first_pixel, int width, float * __restrict__ rays_o, float * __restrict__ rays_d,
  int64_t n)
{
  const int64_t r = (int64_t)blockIdx.x * F2N_BLOCK + threadIdx.x;
  if (r >= n) return;
  const int64_t cam = cam_idx ? (int64_t)cam_idx[r] : (n_cams == n && n_cams > 1 ? r : 0);
  float row, col;
  if (ij) {
    row = (float)ij[2 * r];
    col = (float)ij[2 * r + 1];
  } else {
    const int64_t px = first_pixel + r;
    row = (float)(px / width);
    col = (float)(px % width);
  }
  const float * K = intrinsics + cam * 9;
  const float * P = poses + cam * pose_ld;  // rows of 4: [R | t]
  // pixel centre (+0.5), camera looks down -z, y up (src/rays.cpp:10-21)
  const float u = ((col + .5f) - K[2]) / K[0];
  const float v = -(((row + .5f) - K[5]) / K[4]);
  const float w = -1.f;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    rays_d[3 * r + a] = fmaf(P[4 * a + 2], w, fmaf(P[4 * a + 1], v, P[4 * a] * u));
    rays_o[3 * r + a] = P[4 * a + 3];
  }
}

}  // namespace

extern "C" int f2n_gen_rays(
  const float * poses, int pose_ld, const float * intrinsics, int64_t n_cams,
  const int32_t * cam_idx, const int32_t * ij, int64_t first_pixel, int width, float * rays_o,
  float * rays_d, int64_t n, void * stream)
{
  if (!poses || !intrinsics || !rays_o || !rays_d || n < 0 || n_cams < 1) return F2N_E_INVALID_ARG;
  if (pose_ld != 12 && pose_ld != 16) return F2N_E_INVALID_ARG;  // [3,4] or [4,4] row-major
  if (!ij && width <= 0) return F2N_E_INVALID_ARG;
  if (!cam_idx && n_cams != 1 && n_cams != n) return F2N_E_INVALID_ARG;
  if (n == 0) return F2N_OK;
  hipLaunchKernelGGL(
    gen_rays_kernel, dim3(f2n_div_up(n, F2N_BLOCK)), dim3(F2N_BLOCK), 0, (hipStream_t)stream, poses,
    pose_ld, intrinsics, n_cams, cam_idx, ij, first_pixel, width, rays_o, rays_d, n);
  return f2n_launch_status();
}

// ref_shim.cpp -- pybind11 shim around the reference's OWN host-side translation units that are pure
// ATen and therefore build against this image's ROCm LibTorch without any stand-in:
//   src/points_sampler.cpp (PtsSampler::get_samples), src/rays.cpp (get_rays_from_pose),
//   src/CustomOps/CustomOps.cpp (torch::autograd::TruncExp).
// oracle/Makefile.ref compiles those files FROM /root/reference (never copied into this repo) together
// with this shim into oracle/_ref/_f2nerf_ref.so.  TEST INFRASTRUCTURE ONLY: the -m gpu test
// tests/test_gpu_ref.py uses it to pin the sampler / TruncExp / ray-generation rows of the oracle and
// of the HIP path against the real reference code running on the MI355X (the reference creates every
// tensor on kCUDA, src/common.hpp:11, so it can only run on a GPU box).
// The reference's kernels (*.cu) and the classes that need them are NOT buildable here (nvcc,
// cuda_runtime.h, atomicAdd(__half2*)), so Hash3DAnchored / SHShader / Renderer stay unpinned.
#include <torch/extension.h>

#include "CustomOps/CustomOps.hpp"
#include "points_sampler.hpp"
#include "rays.hpp"

namespace py = pybind11;

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
  m.def("max_sample_per_ray", []() { return MAX_SAMPLE_PER_RAY; });
  m.def("get_samples", [](const torch::Tensor & rays_o, const torch::Tensor & rays_d, bool train) {
    PtsSampler sampler;
    SampleResultFlex r =
      sampler.get_samples(rays_o, rays_d, train ? RunningMode::TRAIN : RunningMode::VALIDATE);
    return py::make_tuple(r.pts, r.dirs, r.dt, r.t, r.pts_idx_bounds);
  });
  m.def("trunc_exp", [](const torch::Tensor & x) { return torch::autograd::TruncExp::apply(x)[0]; });
  m.def(
    "get_rays_from_pose",
    [](const torch::Tensor & pose, const torch::Tensor & intrinsic, const torch::Tensor & ij) {
      Rays r = get_rays_from_pose(pose, intrinsic, ij);
      return py::make_tuple(r.origins, r.dirs);
    });
}

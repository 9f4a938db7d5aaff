// ref_host_shim.cpp -- pybind11 view of the reference's OWN Renderer (src/renderer.{hpp,cpp} and the
// host sources behind it, compiled unmodified from /root/reference) running on the MI355X through
// oracle/ref_cuda_side.cpp, i.e. through the C ABI of libf2nerf_hip.so.  Built by oracle/build_ref.py
// into oracle/_ref/_f2nerf_ref_host.so.  TEST INFRASTRUCTURE ONLY (tests/test_gpu_ref_host.py, run in
// a process of its own by oracle/ref_host_runner.py: the reference registers the same TORCH_LIBRARY
// namespace as this repository's host library, so the two cannot share a process).
#include <torch/extension.h>

#include <map>
#include <string>

#include "CustomOps/CustomOps.hpp"
#include "rays.hpp"
#include "renderer.hpp"

namespace py = pybind11;
using torch::Tensor;

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
  m.def("max_sample_per_ray", []() { return MAX_SAMPLE_PER_RAY; });
  m.def("n_levels", []() { return (int)N_LEVELS; });
  m.def("n_channels", []() { return (int)N_CHANNELS; });
  m.def("weight_var", [](Tensor w, Tensor idx) { return CustomOps::WeightVar(w, idx); });
  py::class_<Renderer, std::shared_ptr<Renderer>>(m, "Renderer")
    .def(py::init([](int n_images) {
      auto r = std::make_shared<Renderer>(n_images);
      r->to(torch::kCUDA);  // src/main_functions/train_manager.cpp:53
      return r;
    }))
    .def(
      "named_parameters",
      [](Renderer & r) {
        std::map<std::string, Tensor> out;
        for (auto & kv : r.named_parameters()) out[kv.key()] = kv.value();
        return out;
      })
    .def("zero_grad", [](Renderer & r) { r.zero_grad(); })
    // src/main_functions/train_manager.cpp:132-136 / src/localizer.cpp:37-39
    .def("save", [](std::shared_ptr<Renderer> r, const std::string & path) { torch::save(r, path); })
    .def("load", [](std::shared_ptr<Renderer> r, const std::string & path) { torch::load(r, path); })
    .def(
      "render",
      [](Renderer & r, const Tensor & o, const Tensor & d, const Tensor & emb, bool train) {
        RenderResult res = r.render(o, d, emb, train ? RunningMode::TRAIN : RunningMode::VALIDATE);
        return py::make_tuple(res.colors, res.depths, res.weights, res.idx_start_end);
      })
    .def(
      "render_all_rays",
      [](Renderer & r, const Tensor & o, const Tensor & d, int batch) {
        auto [c, z] = r.render_all_rays(o, d, batch);
        return py::make_tuple(c, z);
      })
    .def(
      "render_image",
      [](Renderer & r, const Tensor & pose, const Tensor & intrinsic, int h, int w, int batch) {
        auto [c, z] = r.render_image(pose, intrinsic, h, w, batch);
        return py::make_tuple(c, z);
      });
}

// bindings.cpp -- pybind11 view of the C++/LibTorch operator surface, for bench.py and tests/.
// Nothing here computes: every function forwards to the classes in this directory, which in turn
// call the C ABI of libf2nerf_hip.so.
#include <torch/extension.h>

#include "fused_adam.hpp"
#include "hash_3d_anchored.hpp"
#include "kernel_timer.hpp"
#include "points_sampler.hpp"
#include "ragged_ops.hpp"
#include "rays.hpp"
#include "scene_files.hpp"
#include "renderer.hpp"
#include "sh_shader.hpp"

namespace py = pybind11;
using torch::Tensor;

namespace
{

torch::Device parse_device(const std::string & s)
{
  return s.empty() ? f2n::default_device() : torch::Device(s);
}

Hash3DAnchoredOptions field_options(
  int64_t n_levels, int64_t n_channels, int64_t log2_table, int64_t level_stride,
  const std::string & device)
{
  Hash3DAnchoredOptions o;
  o.n_levels = n_levels;
  o.n_channels = n_channels;
  o.log2_table = log2_table;
  o.level_stride = level_stride;
  o.device = parse_device(device);
  return o;
}

py::dict named_params(torch::nn::Module & m)
{
  py::dict d;
  for (auto & kv : m.named_parameters()) d[py::str(kv.key())] = kv.value();
  return d;
}

c10::optional<Tensor> grad_or_none(const Tensor & t)
{
  return t.grad().defined() ? c10::optional<Tensor>(t.grad()) : c10::nullopt;
}

Tensor opt_tensor(const c10::optional<Tensor> & t) { return t.has_value() ? *t : Tensor(); }

RunningMode parse_mode(const std::string & m)
{
  if (m == "train" || m == "TRAIN") return RunningMode::TRAIN;
  if (m == "validate" || m == "VALIDATE") return RunningMode::VALIDATE;
  throw std::invalid_argument("mode must be 'train' or 'validate'");
}

py::tuple result_tuple(const RenderResult & r)
{
  return py::make_tuple(r.colors, r.depths, r.weights, r.idx_start_end);
}

struct AdamHandle
{
  std::shared_ptr<torch::optim::Optimizer> opt;
};

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
  m.doc() = "F2-NeRF rendering hot path: LibTorch C++ operator surface over libf2nerf_hip.so";
  m.attr("MAX_SAMPLE_PER_RAY") = MAX_SAMPLE_PER_RAY;

  // ---- free operators (FlexOps / CustomOps / TruncExp) ------------------------------------------
  m.def("flex_sum", &FlexOps::Sum, "FlexOps::Sum(val, idx_start_end)");
  m.def("flex_accumulate_sum", &FlexOps::AccumulateSum, "FlexOps::AccumulateSum");
  m.def("weight_var", &CustomOps::WeightVar, "CustomOps::WeightVar");
  m.def("scatter_add", &CustomOps::ScatterAdd, "CustomOps::ScatterAdd(emb, idx, to_add)");
  m.def("scatter_idx", &CustomOps::ScatterIdx, "CustomOps::ScatterIdx(n_all, idx_start_end, emb_idx)");
  m.def("trunc_exp", [](const Tensor & x) { return torch::autograd::TruncExp::apply(x)[0]; });
  m.def(
    "composite",
    [](const Tensor & field_out, const Tensor & rgb, const Tensor & dt, const Tensor & t,
       const Tensor & idx, const Tensor & bg) {
      auto o = f2n::composite(field_out, rgb, dt, t, idx, bg);
      return py::make_tuple(o.colors, o.depths, o.weights);
    });
  m.def("get_rays_from_pose", [](const Tensor & pose, const Tensor & intr, const Tensor & ij) {
    Rays r = get_rays_from_pose(pose, intr, ij);
    return py::make_tuple(r.origins, r.dirs);
  });
  m.def("get_view_rays", [](const Tensor & pose, const Tensor & intr, int h, int w) {
    Rays r = get_view_rays(pose, intr, h, w);
    return py::make_tuple(r.origins, r.dirs);
  });
  m.def(
    "sample_random_rays",
    [](const Tensor & poses, const Tensor & intr, int h, int w, int64_t n, const Tensor & images) {
      auto [r, gt, cam] = sample_random_rays(poses, intr, h, w, n, images);
      return py::make_tuple(r.origins, r.dirs, gt, cam);
    },
    py::arg("poses"), py::arg("intrinsics"), py::arg("h"), py::arg("w"), py::arg("batch_size"),
    py::arg("images") = Tensor());
  m.def("train_loss", &f2n::train_loss, py::arg("colors"), py::arg("gt_colors"), py::arg("var"),
        py::arg("var_loss_weight"));
  m.def("manual_seed", [](uint64_t s) { torch::manual_seed(s); });
  m.def("kernel_timer_enable", &f2n::kernel_timer_enable);
  m.def("kernel_timer_collect", []() {
    py::dict d;
    for (auto & t : f2n::kernel_timer_collect())
      d[py::str(t.name)] = py::make_tuple(t.launches, t.total_ms, t.units);
    return d;
  });

  // ---- Hash3DAnchored ----------------------------------------------------------------------------
  py::class_<Hash3DAnchored, std::shared_ptr<Hash3DAnchored>>(m, "Hash3DAnchored")
    .def(
      py::init([](int64_t L, int64_t F, int64_t log2_T, int64_t stride, const std::string & dev) {
        return std::make_shared<Hash3DAnchored>(field_options(L, F, log2_T, stride, dev));
      }),
      py::arg("n_levels") = 16, py::arg("n_channels") = 2, py::arg("log2_table") = 19,
      py::arg("level_stride") = 0, py::arg("device") = "")
    .def("query", &Hash3DAnchored::query)
    .def("table_f16", &Hash3DAnchored::table_f16)
    .def("named_parameters", [](Hash3DAnchored & s) { return named_params(s); })
    .def(
      "encode",
      [](Hash3DAnchored & s, const Tensor & x) {
        auto info = torch::make_intrusive<Hash3DAnchoredInfo>();
        info->hash3d_ = &s;
        return torch::autograd::Hash3DAnchoredFunction::apply(x, s.feat_pool_, torch::IValue(info))[0];
      },
      "Hash3DAnchoredFunction::apply(points, feat_pool, info)[0] on already-contracted points")
    .def(
      "set_accumulate_in_place", [](Hash3DAnchored & s, bool on) { s.options_.accumulate_in_place = on; },
      "Hash3DAnchoredOptions::accumulate_in_place: add into feat_pool.grad directly once it exists")
    .def_readonly("pool_size", &Hash3DAnchored::pool_size_)
    .def_readonly("local_size", &Hash3DAnchored::local_size_)
    .def_readonly("level_stride", &Hash3DAnchored::level_stride_)
    .def_readonly("feat_pool", &Hash3DAnchored::feat_pool_)
    .def_readonly("prim_pool", &Hash3DAnchored::prim_pool_)
    .def_readonly("bias_pool", &Hash3DAnchored::bias_pool_)
    .def_readonly("level_mul", &Hash3DAnchored::level_mul_);

  // ---- PtsSampler --------------------------------------------------------------------------------
  py::class_<PtsSampler, std::shared_ptr<PtsSampler>>(m, "PtsSampler")
    .def(
      py::init([](int max_samples, float step) {
        PtsSamplerOptions o;
        o.max_samples = max_samples;
        o.step = step;
        return std::make_shared<PtsSampler>(o);
      }),
      py::arg("max_samples") = MAX_SAMPLE_PER_RAY, py::arg("step") = 1.0f / 256)
    .def(
      "get_samples",
      [](PtsSampler & s, const Tensor & o, const Tensor & d, const std::string & mode,
         const c10::optional<Tensor> & noise) {
        SampleResultFlex r = noise.has_value() ? s.get_samples(o, d, *noise)
                                               : s.get_samples(o, d, parse_mode(mode));
        return py::make_tuple(r.pts, r.dirs, r.dt, r.t, r.pts_idx_bounds);
      },
      py::arg("rays_o"), py::arg("rays_d"), py::arg("mode") = "validate",
      py::arg("noise") = py::none())
    .def(
      "get_samples_aten",
      [](PtsSampler & s, const Tensor & o, const Tensor & d, const c10::optional<Tensor> & noise) {
        SampleResultFlex r = s.get_samples_aten(o, d, opt_tensor(noise));
        return py::make_tuple(r.pts, r.dirs, r.dt, r.t, r.pts_idx_bounds);
      },
      py::arg("rays_o"), py::arg("rays_d"), py::arg("noise") = py::none());

  // ---- SHShader ----------------------------------------------------------------------------------
  py::class_<SHShader, std::shared_ptr<SHShader>>(m, "SHShader")
    .def(
      py::init([](const std::string & dev) { return std::make_shared<SHShader>(parse_device(dev)); }),
      py::arg("device") = "")
    .def("query", &SHShader::query)
    .def("encode", &SHShader::encode)
    .def("named_parameters", [](SHShader & s) { return named_params(s); });

  // ---- Renderer ----------------------------------------------------------------------------------
  py::class_<Renderer, std::shared_ptr<Renderer>>(m, "Renderer")
    .def(
      py::init([](int n_images, int64_t L, int64_t F, int64_t log2_T, int64_t stride,
                  int max_samples, float step, bool fused, const std::string & dev) {
        RendererOptions o;
        o.field = field_options(L, F, log2_T, stride, dev);
        o.sampler.max_samples = max_samples;
        o.sampler.step = step;
        o.fused = fused;
        return std::make_shared<Renderer>(n_images, o);
      }),
      py::arg("n_images"), py::arg("n_levels") = 16, py::arg("n_channels") = 2,
      py::arg("log2_table") = 19, py::arg("level_stride") = 0,
      py::arg("max_samples") = MAX_SAMPLE_PER_RAY, py::arg("step") = 1.0f / 256,
      py::arg("fused") = true, py::arg("device") = "")
    .def(
      "render",
      [](Renderer & r, const Tensor & o, const Tensor & d, const c10::optional<Tensor> & emb_idx,
         const std::string & mode, const c10::optional<Tensor> & noise,
         const c10::optional<Tensor> & bg) {
        RenderResult res;
        {
          py::gil_scoped_release no_gil;
          res = r.render(
            o, d, opt_tensor(emb_idx), parse_mode(mode), opt_tensor(noise), opt_tensor(bg));
        }
        return result_tuple(res);
      },
      py::arg("rays_o"), py::arg("rays_d"), py::arg("emb_idx") = py::none(),
      py::arg("mode") = "validate", py::arg("noise") = py::none(), py::arg("bg_color") = py::none())
    .def("render_all_rays", &Renderer::render_all_rays, py::call_guard<py::gil_scoped_release>())
    .def("render_image", &Renderer::render_image, py::call_guard<py::gil_scoped_release>())
    .def(
      "train_step",
      [](Renderer & r, const Tensor & o, const Tensor & d, const Tensor & emb_idx, const Tensor & gt,
         float var_loss_weight, const c10::optional<Tensor> & noise,
         const c10::optional<Tensor> & bg, bool backward) {
        f2n::TrainStepResult out;
        {
          // loss.backward() runs the autograd engine, which must not be entered holding the GIL
          py::gil_scoped_release no_gil;
          out = f2n::train_step(
            r, o, d, emb_idx, gt, var_loss_weight, opt_tensor(noise), opt_tensor(bg), backward);
        }
        return py::make_tuple(out.loss, out.sq_err_sum, out.n_values, out.n_samples);
      },
      py::arg("rays_o"), py::arg("rays_d"), py::arg("emb_idx"), py::arg("gt_colors"),
      py::arg("var_loss_weight") = 0.f, py::arg("noise") = py::none(),
      py::arg("bg_color") = py::none(), py::arg("backward") = true,
      "reference train_manager.cpp:76-107 without the optimiser step")
    .def("named_parameters", [](Renderer & r) { return named_params(r); })
    .def("zero_grad", [](Renderer & r) { r.zero_grad(); })
    .def(
      "grads",
      [](Renderer & r) {
        py::dict d;
        for (auto & kv : r.named_parameters()) d[py::str(kv.key())] = grad_or_none(kv.value());
        return d;
      })
    .def("set_fused", [](Renderer & r, bool f) { r.options_.fused = f; })
    .def("set_fused_shade", [](Renderer & r, bool f) { r.options_.fused_shade = f; })
    .def("set_dense_first_pass", [](Renderer & r, int m) { r.options_.dense_first_pass = m; })
    .def("set_speculate_dense", [](Renderer & r, bool f) { r.options_.speculate_dense = f; })
    .def("set_pixel_tiles", [](Renderer & r, int b) { r.options_.pixel_tiles = b; })
    .def("set_margin_min_samples", [](Renderer & r, int64_t n) { r.options_.margin_min_samples = n; })
    .def("set_deferred_check", [](Renderer & r, bool f) { r.options_.deferred_check = f; },
         "no host read in render(): see RendererOptions::deferred_check")
    .def("deferred_check_ok", &Renderer::deferred_check_ok)
    .def_readonly("last_kept_fraction", &Renderer::last_kept_fraction_)
    .def_readonly("last_n_samples", &Renderer::last_n_samples_)
    .def_property_readonly("scene_field", [](Renderer & r) { return r.scene_field_; })
    .def_property_readonly("shader", [](Renderer & r) { return r.shader_; })
    .def_property_readonly("pts_sampler", [](Renderer & r) { return r.pts_sampler_; })
    .def(
      "save", [](std::shared_ptr<Renderer> r, const std::string & path) { torch::save(r, path); })
    .def("load", [](std::shared_ptr<Renderer> r, const std::string & path) { torch::load(r, path); })
    .def(
      "make_adam",
      [](Renderer & r, float lr) {
        AdamHandle h;
        h.opt = std::make_shared<torch::optim::Adam>(r.optim_param_groups(lr));
        return h;
      },
      "torch::optim::Adam over optim_param_groups(lr) (reference train_manager.cpp:55)")
    .def(
      "make_fused_adam",
      [](Renderer & r, float lr) {
        AdamHandle h;
        h.opt = std::make_shared<FusedAdam>(r.optim_param_groups(lr), r.scene_field_);
        return h;
      },
      "FusedAdam over the same groups: one kernel per parameter, emits the table's f16 shadow");

  // ---- scene files: cams_meta.tsv, scene normalisation, inference_params.yaml
  m.def("read_cams_meta", [](const std::string & path) {
    f2n::CamsMeta c = f2n::read_cams_meta(path);
    return py::make_tuple(c.poses, c.intrinsics, c.dist_params, c.bounds);
  });
  m.def("normalize_scene", [](const torch::Tensor & poses) {
    f2n::SceneNormalisation n = f2n::normalize_scene(poses);
    return py::make_tuple(n.poses, n.center, n.radius);
  });
  m.def(
    "save_inference_params",
    [](const std::string & dir, int n_images, int height, int width, const torch::Tensor & intrinsic,
       const torch::Tensor & center, float radius) {
      f2n::InferenceParams p;
      p.n_images = n_images;
      p.height = height;
      p.width = width;
      p.intrinsic = intrinsic;
      p.normalizing_center = center;
      p.normalizing_radius = radius;
      f2n::save_inference_params(dir, p);
    });
  m.def("load_inference_params", [](const std::string & dir) {
    f2n::InferenceParams p = f2n::load_inference_params(dir);
    return py::make_tuple(
      p.n_images, p.height, p.width, p.intrinsic, p.normalizing_center, p.normalizing_radius);
  });

  py::class_<AdamHandle>(m, "Adam")
    .def("step", [](AdamHandle & h) { h.opt->step(); }, py::call_guard<py::gil_scoped_release>())
    .def("zero_grad", [](AdamHandle & h) { h.opt->zero_grad(); })
    .def("n_groups", [](AdamHandle & h) { return h.opt->param_groups().size(); })
    .def("set_lr", [](AdamHandle & h, double lr) {
      for (auto & g : h.opt->param_groups())
        static_cast<torch::optim::AdamOptions &>(g.options()).lr(lr);
    });
}

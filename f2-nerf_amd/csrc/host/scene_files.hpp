// scene_files.hpp -- the two text files either side of the render path:
//   cams_meta.tsv          read by Dataset::Dataset            (reference src/dataset.cpp:27-75)
//   inference_params.yaml  written by Dataset::save_inference_params (src/dataset.cpp:106-133),
//                          read by Localizer::Localizer through cv::FileStorage (src/localizer.cpp:15-36)
// plus the scene normalisation between them (src/dataset.cpp:77-86).  Same columns, same field
// names, same number formatting, so files of either implementation are interchangeable.
// Host-only code (no GPU work): images / PNG I/O stay out of scope.
#pragma once

#include <torch/torch.h>

#include <string>

namespace f2n
{

struct CamsMeta
{
  torch::Tensor poses;        // [n, 3, 4] f32, as stored (not normalised)
  torch::Tensor intrinsics;   // [n, 3, 3]
  torch::Tensor dist_params;  // [n, 4]  k1, k2, p1, p2
  torch::Tensor bounds;       // [n, 2]  near, far
};

// One header line, then one tab-separated row of 12 + 9 + 4 + 2 floats per image.
CamsMeta read_cams_meta(const std::string & path);

struct SceneNormalisation
{
  torch::Tensor poses;   // [n, 3, 4] with the camera positions mapped into the unit ball
  torch::Tensor center;  // [3] mean camera position
  float radius;          // largest distance of a camera from the centre
};
SceneNormalisation normalize_scene(const torch::Tensor & poses);

struct InferenceParams
{
  int n_images = 0, height = 0, width = 0;
  torch::Tensor intrinsic;           // [3, 3]
  torch::Tensor normalizing_center;  // [3]
  float normalizing_radius = 0.f;
};

// <dir>/inference_params.yaml, byte for byte the reference's stream output.
void save_inference_params(const std::string & train_result_dir, const InferenceParams & p);
InferenceParams load_inference_params(const std::string & train_result_dir);

}  // namespace f2n

// scene_files.cpp -- see scene_files.hpp.
#include "scene_files.hpp"

#include <fstream>
#include <sstream>
#include <vector>

using torch::Tensor;

namespace f2n
{

CamsMeta read_cams_meta(const std::string & path)
{
  std::ifstream ifs(path);
  TORCH_CHECK(ifs.good(), "cannot open ", path);
  std::string line;
  std::getline(ifs, line);  // header (src/dataset.cpp:31)
  constexpr int kPose = 12, kIntr = 9, kDist = 4, kBounds = 2, kAll = kPose + kIntr + kDist + kBounds;
  std::vector<float> values;
  int64_t n = 0;
  while (std::getline(ifs, line)) {
    if (line.empty()) continue;
    std::istringstream iss(line);
    std::string token;
    int n_tok = 0;
    while (std::getline(iss, token, '\t')) {
      values.push_back(std::stof(token));
      n_tok++;
    }
    TORCH_CHECK(n_tok == kAll, path, ": row ", n, " has ", n_tok, " columns, expected ", kAll);
    n++;
  }
  TORCH_CHECK(n > 0, path, ": no camera rows");
  Tensor all = torch::from_blob(values.data(), {n, kAll}, torch::kFloat32).clone();
  CamsMeta m;
  m.poses = all.slice(1, 0, kPose).reshape({n, 3, 4}).contiguous();
  m.intrinsics = all.slice(1, kPose, kPose + kIntr).reshape({n, 3, 3}).contiguous();
  m.dist_params = all.slice(1, kPose + kIntr, kPose + kIntr + kDist).contiguous();
  m.bounds = all.slice(1, kPose + kIntr + kDist, kAll).contiguous();
  return m;
}

SceneNormalisation normalize_scene(const Tensor & poses_in)
{
  // src/dataset.cpp:77-86
  Tensor poses = poses_in.clone();
  Tensor cam_pos = poses.index({torch::indexing::Slice(), torch::indexing::Slice(0, 3), 3}).clone();
  Tensor center = cam_pos.mean(0, false);
  Tensor bias = cam_pos - center.unsqueeze(0);
  const float radius = torch::linalg_norm(bias, 2, -1, false).max().item<float>();
  cam_pos = (cam_pos - center.unsqueeze(0)) / radius;
  poses.index_put_({torch::indexing::Slice(), torch::indexing::Slice(0, 3), 3}, cam_pos);
  return {poses.contiguous(), center, radius};
}

void save_inference_params(const std::string & dir, const InferenceParams & p)
{
  // the statements of src/dataset.cpp:106-133, on the same stream state
  std::ofstream ofs(dir + "/inference_params.yaml");
  TORCH_CHECK(ofs.good(), "cannot write ", dir, "/inference_params.yaml");
  Tensor K = p.intrinsic.to(torch::kCPU).to(torch::kFloat32), c = p.normalizing_center.to(torch::kCPU);
  ofs << std::fixed;
  ofs << "%YAML 1.2" << std::endl;
  ofs << "---" << std::endl;
  ofs << "n_images: " << p.n_images << std::endl;
  ofs << "height: " << p.height << std::endl;
  ofs << "width: " << p.width << std::endl;
  ofs << "intrinsic: [";
  ofs << K[0][0].item() << ", " << K[0][1].item() << ", " << K[0][2].item() << "," << std::endl;
  ofs << "            ";
  ofs << K[1][0].item() << ", " << K[1][1].item() << ", " << K[1][2].item() << "," << std::endl;
  ofs << "            ";
  ofs << K[2][0].item() << ", " << K[2][1].item() << ", " << K[2][2].item() << "]" << std::endl;
  ofs << "normalizing_center: [" << c[0].item();
  ofs << ", " << c[1].item();
  ofs << ", " << c[2].item() << "]" << std::endl;
  ofs << "normalizing_radius: " << p.normalizing_radius << std::endl;
}

InferenceParams load_inference_params(const std::string & dir)
{
  // the fields Localizer reads (src/localizer.cpp:23-36); a flow sequence may span lines
  std::ifstream ifs(dir + "/inference_params.yaml");
  TORCH_CHECK(ifs.good(), "Failed to open ", dir, "/inference_params.yaml");
  std::stringstream buf;
  buf << ifs.rdbuf();
  const std::string text = buf.str();
  auto scalar = [&](const std::string & key) {
    const size_t pos = text.find("\n" + key + ":");
    TORCH_CHECK(pos != std::string::npos, "inference_params.yaml: missing ", key);
    const size_t b = pos + key.size() + 2, e = text.find('\n', b);
    return std::stod(text.substr(b, e - b));
  };
  auto sequence = [&](const std::string & key, int64_t n) {
    const size_t pos = text.find("\n" + key + ":");
    TORCH_CHECK(pos != std::string::npos, "inference_params.yaml: missing ", key);
    const size_t b = text.find('[', pos), e = text.find(']', b);
    TORCH_CHECK(b != std::string::npos && e != std::string::npos, key, ": expected [ ... ]");
    std::string body = text.substr(b + 1, e - b - 1);
    for (char & ch : body)
      if (ch == ',' || ch == '\n') ch = ' ';
    std::istringstream iss(body);
    std::vector<float> v;
    float x;
    while (iss >> x) v.push_back(x);
    TORCH_CHECK((int64_t)v.size() == n, key, ": expected ", n, " values, found ", v.size());
    return torch::tensor(v, torch::kFloat);
  };
  InferenceParams p;
  p.n_images = (int)scalar("n_images");
  p.height = (int)scalar("height");
  p.width = (int)scalar("width");
  p.intrinsic = sequence("intrinsic", 9).view({3, 3});
  p.normalizing_center = sequence("normalizing_center", 3);
  p.normalizing_radius = (float)scalar("normalizing_radius");
  return p;
}

}  // namespace f2n

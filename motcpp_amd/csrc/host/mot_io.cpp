// MOT17 dataset indexing + detection files (src/data/mot17_dataset.cpp:12-241 restated without OpenCV). Host-only.
#include <algorithm>
#include <fstream>
#include <regex>
#include <sstream>
#include <stdexcept>

#include "motcpp/data/mot17_dataset.hpp"

namespace motcpp::data {

MOT17Dataset::MOT17Dataset(const std::string& mot_root, const std::string& det_emb_root, const std::string& model_name,
                           const std::string& reid_name)
    : mot_root_(mot_root) {
  if (!det_emb_root.empty() && !model_name.empty()) {  // :18-28
    const std::filesystem::path direct = std::filesystem::path(det_emb_root) / "dets";
    det_path_ = std::filesystem::exists(direct) ? direct : std::filesystem::path(det_emb_root) / model_name / "dets";
    if (!reid_name.empty()) emb_dir_ = std::filesystem::path(det_emb_root) / model_name / "embs" / reid_name;  // tools/motcpp_eval.cpp:77
  }
  index_sequences();
}

namespace {
// "MOT17-02-FRCNN" -> "MOT17-02.txt" (:49-59)
std::string short_det_name(const std::string& seq_name) {
  const size_t first = seq_name.find('-');
  const size_t second = (first == std::string::npos) ? std::string::npos : seq_name.find('-', first + 1);
  if (second != std::string::npos) return "MOT17-" + seq_name.substr(first + 1, second - first - 1) + ".txt";
  return seq_name + ".txt";
}
int ini_int(const std::filesystem::path& cfg, const char* key, int dflt) {  // :112-131
  if (!std::filesystem::exists(cfg)) return dflt;
  std::ifstream file(cfg);
  std::string line;
  const std::regex re(std::string(key) + R"(\s*=\s*(\d+))");
  while (std::getline(file, line)) {
    std::smatch m;
    if (line.find(key) != std::string::npos && std::regex_search(line, m, re)) return std::stoi(m[1].str());
  }
  return dflt;
}
}  // namespace

void MOT17Dataset::index_sequences() {
  if (!std::filesystem::exists(mot_root_)) throw std::runtime_error("MOT root directory does not exist: " + mot_root_.string());
  for (const auto& entry : std::filesystem::directory_iterator(mot_root_)) {
    if (!entry.is_directory()) continue;
    SequenceInfo s;
    s.name = entry.path().filename().string();
    s.seq_dir = entry.path();
    s.img_dir = s.seq_dir / "img1";
    if (det_path_.empty()) {
      s.det_path = s.seq_dir / "det" / "det.txt";
    } else {
      s.det_path = det_path_ / short_det_name(s.name);
      if (!std::filesystem::exists(s.det_path)) s.det_path = det_path_ / (s.name + ".txt");
    }
    s.gt_path = s.seq_dir / "gt" / "gt.txt";
    const bool has_img = std::filesystem::exists(s.img_dir);
    if (!has_img && !std::filesystem::exists(s.det_path)) continue;
    if (has_img) {
      for (const auto& img : std::filesystem::directory_iterator(s.img_dir)) {
        const auto ext = img.path().extension();
        if (ext != ".jpg" && ext != ".png") continue;
        try { s.frame_ids.push_back(std::stoi(img.path().stem().string())); } catch (...) { continue; }
      }
      std::sort(s.frame_ids.begin(), s.frame_ids.end());
    }
    const auto cfg = s.seq_dir / "seqinfo.ini";
    s.fps = ini_int(cfg, "frameRate", 30);
    s.im_width = ini_int(cfg, "imWidth", 1920);
    s.im_height = ini_int(cfg, "imHeight", 1080);
    sequences_.push_back(std::move(s));
  }
  std::sort(sequences_.begin(), sequences_.end(), [](const SequenceInfo& a, const SequenceInfo& b) { return a.name < b.name; });
}

std::vector<std::string> MOT17Dataset::sequence_names() const {
  std::vector<std::string> names;
  for (const auto& s : sequences_) names.push_back(s.name);
  return names;
}

SequenceInfo MOT17Dataset::get_sequence_info(const std::string& seq_name) const {
  for (const auto& s : sequences_)
    if (s.name == seq_name) return s;
  throw std::runtime_error("Sequence not found: " + seq_name);
}

std::map<int, Eigen::MatrixXf> MOT17Dataset::load_detections(const std::filesystem::path& det_path) const {
  std::map<int, std::vector<float>> rows;  // frame -> flattened [x1,y1,x2,y2,conf,cls] rows in file order
  std::map<int, Eigen::MatrixXf> out;
  if (!std::filesystem::exists(det_path)) return out;
  std::ifstream file(det_path);
  std::string line;
  bool comma = false;
  if (std::getline(file, line)) {  // the first line decides the format (:163-168)
    comma = line.find(',') != std::string::npos;
    file.clear();
    file.seekg(0);
  }
  while (std::getline(file, line)) {
    if (line.empty() || line[0] == '#') continue;
    std::vector<float> v;
    std::istringstream iss(line);
    if (comma) {
      std::string token;
      while (std::getline(iss, token, ',')) {
        try { v.push_back(std::stof(token)); } catch (...) { break; }
      }
      if (v.size() < 7) continue;
      const float x1 = v[2], y1 = v[3], w = v[4], h = v[5];
      const float cls = (v.size() > 7) ? v[7] : 0.0f;
      auto& r = rows[static_cast<int>(v[0])];
      r.insert(r.end(), {x1, y1, x1 + w, y1 + h, v[6], cls});
    } else {
      float val;
      while (iss >> val) v.push_back(val);
      if (v.size() < 7) continue;
      auto& r = rows[static_cast<int>(v[0])];
      r.insert(r.end(), {v[1], v[2], v[3], v[4], v[5], v[6]});
    }
  }
  for (const auto& [frame, flat] : rows) {
    const int n = static_cast<int>(flat.size() / 6);
    Eigen::MatrixXf m(n, 6);
    for (int i = 0; i < n; ++i)
      for (int k = 0; k < 6; ++k) m(i, k) = flat[static_cast<size_t>(i) * 6 + k];
    out.emplace(frame, std::move(m));
  }
  return out;
}

std::filesystem::path MOT17Dataset::embedding_path(const std::string& seq_name) const {
  return emb_dir_.empty() ? std::filesystem::path() : emb_dir_ / short_det_name(seq_name);
}

// :243-294. Line k = the k-th detection; detections counted frame by frame in ascending frame order (see the header).
std::map<int, Eigen::MatrixXf> MOT17Dataset::load_embeddings(const std::filesystem::path& emb_path,
                                                             const std::map<int, Eigen::MatrixXf>& detections) const {
  std::map<int, Eigen::MatrixXf> out;
  if (emb_path.empty() || !std::filesystem::exists(emb_path)) return out;
  std::ifstream file(emb_path);
  auto frame = detections.begin();
  int row = 0;
  std::map<int, std::vector<std::vector<float>>> rows;
  for (std::string line; std::getline(file, line);) {
    if (line.empty() || line[0] == '#') continue;
    while (frame != detections.end() && row >= frame->second.rows()) { ++frame; row = 0; }
    if (frame == detections.end()) break;  // more lines than detections (:265)
    std::vector<float> v;
    std::istringstream iss(line);
    for (float x; iss >> x;) v.push_back(x);
    if (v.empty()) continue;  // (:278: the line is dropped without consuming a detection)
    rows[frame->first].push_back(std::move(v));
    ++row;
  }
  for (auto& [f, rr] : rows) {
    const int d = static_cast<int>(rr.back().size());  // the reference resizes to the newest row's width (:287)
    Eigen::MatrixXf m(static_cast<int>(rr.size()), d);
    for (int i = 0; i < m.rows(); ++i)
      for (int k = 0; k < d; ++k) m(i, k) = k < static_cast<int>(rr[i].size()) ? rr[i][k] : 0.0f;
    out.emplace(f, std::move(m));
  }
  return out;
}

}  // namespace motcpp::data

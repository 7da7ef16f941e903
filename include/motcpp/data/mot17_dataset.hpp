// MOT17-style dataset indexing and detection loading (reference: include/motcpp/data/mot17_dataset.hpp,
// src/data/mot17_dataset.cpp:12-241): same class and method names, path rules and parsing rules. Host-only.
// Differences, both forced by the absence of OpenCV: images are never decoded (get_frame is not provided; the frame
// size comes from seqinfo.ini's imWidth/imHeight, 1920x1080 if absent), and a sequence directory is indexed when it
// has img1/ OR det/det.txt (the reference requires img1/). Embedding files are out of scope: the reference maps
// their lines to frames in unordered_map iteration order (mot17_dataset.cpp:254-260), which pins nothing.
#pragma once
#include <filesystem>
#include <map>
#include <string>
#include <vector>

#include "../compat/eigen.hpp"

namespace motcpp::data {

struct SequenceInfo {
  std::string name;
  std::filesystem::path seq_dir, img_dir, det_path, gt_path;
  std::vector<int> frame_ids;  // frames that have an image (may be empty)
  int fps = 30;
  int im_width = 1920, im_height = 1080;
};

class MOT17Dataset {
 public:
  MOT17Dataset(const std::string& mot_root, const std::string& det_emb_root = "", const std::string& model_name = "",
               const std::string& reid_name = "");
  std::vector<std::string> sequence_names() const;                  // sorted by name
  SequenceInfo get_sequence_info(const std::string& seq_name) const;  // throws std::runtime_error("Sequence not found: ...")
  // frame id -> N x 6 [x1,y1,x2,y2,conf,cls], rows in file order. Comma-separated: frame,-1,x,y,w,h,conf[,cls];
  // whitespace-separated (pre-generated): frame x1 y1 x2 y2 conf cls. Missing file -> empty map.
  std::map<int, Eigen::MatrixXf> load_detections(const std::filesystem::path& det_path) const;

 private:
  void index_sequences();
  std::filesystem::path mot_root_, det_path_;
  std::vector<SequenceInfo> sequences_;
};

}  // namespace motcpp::data

// MOT17-style dataset indexing and detection loading (reference: include/motcpp/data/mot17_dataset.hpp,
// src/data/mot17_dataset.cpp:12-241): same class and method names, path rules and parsing rules. Host-only.
// Differences, both forced by the absence of OpenCV: images are never decoded (get_frame is not provided; the frame
// size comes from seqinfo.ini's imWidth/imHeight, 1920x1080 if absent), and a sequence directory is indexed when it
// has img1/ OR det/det.txt (the reference requires img1/). Embedding files (mot17_dataset.cpp:243-294): line k of the file
// is the feature of the k-th detection. The reference walks its detections in unordered_map iteration order (:254-260), which is
// unspecified; here the k-th detection is counted the way the files are written: frames ascending, file order within a frame.
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
  // frame id -> N x D features, row i = the feature of detection row i of that frame. One feature per line (whitespace-separated
  // floats; empty lines and lines starting with '#' are skipped), assigned to the detections frame by frame in ascending frame order;
  // lines beyond the last detection are ignored, a frame the file ends in the middle of gets only the rows it has (as in the
  // reference, where such a frame then fails the tracker's row check). Missing file -> empty map.
  std::map<int, Eigen::MatrixXf> load_embeddings(const std::filesystem::path& emb_path, const std::map<int, Eigen::MatrixXf>& detections) const;
  // <det_emb_root>/<model_name>/embs/<reid_name>/MOT17-<NN>.txt for a sequence named MOT17-<NN>-<DET> (else <name>.txt), the rule of
  // the reference's tool (tools/motcpp_eval.cpp:74-92); empty when the dataset was built without det_emb_root / model / reid names
  std::filesystem::path embedding_path(const std::string& seq_name) const;

 private:
  void index_sequences();
  std::filesystem::path mot_root_, det_path_, emb_dir_;
  std::vector<SequenceInfo> sequences_;
};

}  // namespace motcpp::data

// MOT evaluation driver for the MI355X-backed trackers. Command-line contract (positional arguments, defaults, one
// <sequence>.txt per sequence in MOT format) follows the reference's tools/motcpp_eval.cpp:19-468; the program itself is
// organised differently: a tracker table, a frame plan per sequence, then one loop. Trackers built here: sort, ucmc, bytetrack,
// ocsort, botsort, deepocsort, strongsort, boosttrack (motion only), hybridsort. Images are never decoded: trackers get a blank frame of the sequence's size.
#include <algorithm>
#include <filesystem>
#include <fstream>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "motcpp/data/mot17_dataset.hpp"
#include "motcpp/motcpp.hpp"
#include "motcpp/utils/mot_format.hpp"

namespace {

namespace fs = std::filesystem;
using TrackerPtr = std::unique_ptr<motcpp::BaseTracker>;

bool g_have_embeddings = false;  // this sequence comes with a feature per detection (with_reid for BoT-SORT)

// Evaluation presets per tracker (the values the reference's tool passes, motcpp_eval.cpp:99-246); fps comes from seqinfo.ini.
const std::map<std::string, std::function<TrackerPtr(int)>>& tracker_table() {
  namespace T = motcpp::trackers;
  static const std::map<std::string, std::function<TrackerPtr(int)>> table = {
      {"sort", [](int) { return TrackerPtr(new T::Sort(0.3f, 1, 50, 3, 0.3f, false, 80, "iou", false)); }},
      {"bytetrack",
       [](int fps) { return TrackerPtr(new T::ByteTrack(0.3f, 30, 50, 3, 0.3f, false, 80, "iou", false, 0.1f, 0.45f, 0.8f, 30, fps)); }},
      {"ocsort",
       [](int) { return TrackerPtr(new T::OCSort(0.2f, 30, 50, 3, 0.3f, false, 80, "iou", false, 0.1f, 3, 0.2f, false, 0.01f, 0.0001f)); }},
      // no ReID model on this path: BoT-SORT / DeepOC-SORT take the features of a pre-generated embedding file when the command line
      // names one (det_emb_root, model_name, reid_name), else they run motion-only; camera-motion compensation is not applied
      {"botsort",
       [](int fps) {
         return TrackerPtr(new T::BotSort("", false, false, 0.3f, 30, 50, 3, 0.3f, false, 80, "iou", false, 0.6f, 0.1f, 0.7f, 30, 0.8f, 0.5f,
                                          0.25f, "ecc", fps, false, g_have_embeddings));
       }},
      {"deepocsort",
       [](int) {
         return TrackerPtr(new T::DeepOCSort("", false, false, 0.3f, 30, 50, 3, 0.3f, false, 80, "iou", false, 3, 0.2f, 0.5f, 0.95f, 0.5f,
                                             true, true, true));
       }},
      // strongsort.yaml's values (motcpp_eval.cpp:196-219); the features of a pre-generated embedding file when the command line names one
      {"strongsort",
       [](int) {
         return TrackerPtr(new T::StrongSORT("", false, false, 0.3f, 30, 50, 3, 0.3f, false, 80, "iou", false, 0.6f, 0.4f, 0.7f, 3, 100, 0.98f, 0.9f));
       }},
      // hybridsort.yaml's values (motcpp_eval.cpp:279-316); no ReID weights: with_reid = false
      {"hybridsort",
       [](int) {
         return TrackerPtr(new T::HybridSort("", false, false, 0.5f, 30, 50, 3, 0.3f, false, 80, "hmiou", false, 0.1f, 3, 0.05f, true, true, 30, 0.9f, false,
                                             0.5f, 4.6f, 1.3f, true, true, 1.0f, 0.7f, true, 0.0f, true, 0.4f, 0.4f, "ecc", false));
       }},
      // boosttrack.yaml's values (motcpp_eval.cpp:247-278: BoostTrack++ switches use_rich_s / use_sb / use_vt on); no ReID weights: motion only
      {"boosttrack",
       [](int) {
         return TrackerPtr(new T::BoostTrackTracker("", false, false, 0.6f, 60, 50, 3, 0.3f, false, 80, "iou", false, true, 10, 1.6f, "ecc", 0.5f, 0.25f,
                                                    0.25f, true, true, 0.65f, false, true, true, true, false));
       }},
      // the reference tool's values (motcpp_eval.cpp:112-131): dt from the sequence's frame rate, no camera file (image-space fallback)
      {"ucmc",
       [](int fps) {
         return TrackerPtr(new T::UCMCTrack(0.3f, 30, 50, 3, 0.3f, false, 80, "iou", false, 100.0, 100.0, 5.0, 5.0, 10.0, 1.0 / fps, 0.5f));
       }},
  };
  return table;
}

// Highest frame number in a MOT ground-truth file (first comma-separated field of each line); 0 when there is none.
int last_annotated_frame(const fs::path& gt_file) {
  std::ifstream in(gt_file);
  int last = 0;
  for (std::string row; std::getline(in, row);) {
    if (row.empty() || row.front() == '#') continue;
    last = std::max(last, std::atoi(row.substr(0, row.find(',')).c_str()));
  }
  return last;
}

// Which detection frames to step through and how to number them in the result file. Ablation splits ship detections for the
// whole sequence but annotations for its second half only, renumbered from 1: when the detections reach more than 1.5x past the
// last annotated frame, the first (max_det - last_gt) frames are skipped and the rest are shifted down by that amount
// (the rule of the reference's tool, motcpp_eval.cpp:321-375).
struct FramePlan {
  std::vector<int> frames;  // ascending
  int shift = 0;            // written frame = frame - shift
};

template <class DetMap>
FramePlan plan_frames(const DetMap& detections, const fs::path& gt_file) {
  FramePlan plan;
  plan.frames.reserve(detections.size());
  for (const auto& entry : detections) plan.frames.push_back(entry.first);
  std::sort(plan.frames.begin(), plan.frames.end());
  if (plan.frames.empty() || !fs::exists(gt_file)) return plan;
  const int last_gt = last_annotated_frame(gt_file);
  const int last_det = plan.frames.back();
  if (last_gt <= 0 || 2L * last_det <= 3L * last_gt) return plan;
  plan.shift = last_det - last_gt;
  plan.frames.erase(plan.frames.begin(), std::upper_bound(plan.frames.begin(), plan.frames.end(), plan.shift));
  return plan;
}

int run_sequence(motcpp::data::MOT17Dataset& dataset, const std::string& name, const std::string& method, const fs::path& out_dir) {
  const auto seq = dataset.get_sequence_info(name);
  std::cout << "[" << name << "] detections: " << seq.det_path << (fs::exists(seq.det_path) ? "" : " (missing)") << "\n";
  const auto detections = dataset.load_detections(seq.det_path);
  const fs::path emb_file = dataset.embedding_path(name);
  const auto embeddings = dataset.load_embeddings(emb_file, detections);
  if (!emb_file.empty()) std::cout << "[" << name << "] embeddings: " << emb_file << " (" << embeddings.size() << " frames)\n";
  g_have_embeddings = !embeddings.empty();
  TrackerPtr tracker = tracker_table().at(method)(seq.fps);

  const fs::path result = out_dir / (name + ".txt");
  fs::remove(result);
  const FramePlan plan = plan_frames(detections, seq.gt_path);
  if (plan.shift > 0)
    std::cout << "[" << name << "] ablation split: skipping the first " << plan.shift << " frames, writing frame f as f-" << plan.shift << "\n";

  const cv::Mat blank = cv::Mat::zeros(seq.im_height, seq.im_width, 0);
  int done = 0;
  for (const int f : plan.frames) {
    try {
      const Eigen::MatrixXf& dets = detections.at(f);
      const auto e = embeddings.find(f);
      const Eigen::MatrixXf tracks = tracker->update(dets, blank, e != embeddings.end() ? e->second : Eigen::MatrixXf(dets.rows(), 0));
      if (tracks.rows() > 0) motcpp::utils::write_mot_results(result, motcpp::utils::convert_to_mot_format(tracks, f - plan.shift));
      ++done;
    } catch (const std::exception& e) {
      std::cerr << "[" << name << "] frame " << f << ": " << e.what() << "\n";
    }
  }
  std::cout << "[" << name << "] " << done << " of " << detections.size() << " detection frames tracked -> " << result << "\n";
  return done;
}

}  // namespace

int main(int argc, char* argv[]) {
  if (argc < 3) {
    std::cerr << "usage: " << argv[0] << " <mot_root> <output_dir> [tracking_method=bytetrack] [det_emb_root] [model_name] [reid_name]\n"
              << "  e.g. " << argv[0] << " tests/golden/MOT17-mini/train ./results ocsort\n"
              << "  tracking methods:";
    for (const auto& kv : tracker_table()) std::cerr << " " << kv.first;
    std::cerr << "\n";
    return 1;
  }
  const auto arg = [&](int i, const char* fallback) { return std::string(argc > i ? argv[i] : fallback); };
  const std::string mot_root = arg(1, ""), method = arg(3, "bytetrack");
  const fs::path out_dir = arg(2, "");
  if (!tracker_table().count(method)) {
    std::cerr << "unknown tracking method '" << method << "'\n";
    return 1;
  }
  std::cout << "motcpp_eval (MI355X hot path): " << method << " over " << mot_root << " -> " << out_dir.string() << "\n";
  try {
    motcpp::data::MOT17Dataset dataset(mot_root, arg(4, ""), arg(5, ""), arg(6, ""));
    fs::create_directories(out_dir);
    const auto names = dataset.sequence_names();
    std::cout << names.size() << " sequence(s)\n";
    for (const auto& name : names) {
      try {
        run_sequence(dataset, name, method, out_dir);
      } catch (const std::exception& e) {
        std::cerr << "[" << name << "] skipped: " << e.what() << "\n";
      }
    }
  } catch (const std::exception& e) {
    std::cerr << "error: " << e.what() << "\n";
    return 1;
  }
  std::cout << "done: results in " << out_dir.string() << "\n";
  return 0;
}

// MOT evaluation driver for the MI355X-backed trackers: command line, defaults, per-sequence loop, ablation-offset
// rule and output files of the reference's tools/motcpp_eval.cpp:19-468, for the four trackers built here
// (sort, bytetrack, ocsort, botsort). Images are never decoded: trackers get a blank frame of the sequence's size.
#include <algorithm>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>

#include "motcpp/data/mot17_dataset.hpp"
#include "motcpp/motcpp.hpp"
#include "motcpp/utils/mot_format.hpp"

int main(int argc, char* argv[]) {
  if (argc < 3) {
    std::cerr << "Usage: " << argv[0] << " <mot_root> <output_dir> [tracking_method] [det_emb_root] [model_name] [reid_name]\n";
    std::cerr << "Example: " << argv[0] << " ../assets/MOT17-mini/train ./results bytetrack\n";
    std::cerr << "Example with pre-generated dets: " << argv[0] << " <mot_root> <output_dir> bytetrack ../assets/yolox_x_ablation yolox_x_ablation\n";
    return 1;
  }
  const std::string mot_root = argv[1], output_dir = argv[2];
  const std::string tracking_method = (argc > 3) ? argv[3] : "bytetrack";
  const std::string det_emb_root = (argc > 4) ? argv[4] : "";
  const std::string model_name = (argc > 5) ? argv[5] : "";
  const std::string reid_name = (argc > 6) ? argv[6] : "";

  std::cout << "motcpp - MOT Evaluation Tool v1.0.0 (MI355X hot path)\n";
  std::cout << "==========================\n\n";
  std::cout << "MOT Root: " << mot_root << "\nOutput Dir: " << output_dir << "\nTracking Method: " << tracking_method << "\n";
  if (!det_emb_root.empty()) std::cout << "Det/Emb Root: " << det_emb_root << "\nModel Name: " << model_name << "\n";
  std::cout << "\n";

  try {
    motcpp::data::MOT17Dataset dataset(mot_root, det_emb_root, model_name, reid_name);
    std::filesystem::create_directories(output_dir);
    const auto seq_names = dataset.sequence_names();
    std::cout << "Found " << seq_names.size() << " sequences\n\n";

    for (const auto& seq_name : seq_names) {
      std::cout << "Processing sequence: " << seq_name << "\n";
      try {
        const auto seq = dataset.get_sequence_info(seq_name);
        std::cout << "  Detection file path: " << seq.det_path << "\n";
        std::cout << "  File exists: " << (std::filesystem::exists(seq.det_path) ? "yes" : "no") << "\n";
        const auto detections = dataset.load_detections(seq.det_path);
        std::cout << "  Loaded detections for " << detections.size() << " frames\n";

        std::unique_ptr<motcpp::BaseTracker> tracker;  // parameters: motcpp_eval.cpp:99-246
        if (tracking_method == "sort") {
          tracker = std::make_unique<motcpp::trackers::Sort>(0.3f, 1, 50, 3, 0.3f, false, 80, "iou", false);
        } else if (tracking_method == "bytetrack") {
          tracker = std::make_unique<motcpp::trackers::ByteTrack>(0.3f, 30, 50, 3, 0.3f, false, 80, "iou", false, 0.1f, 0.45f, 0.8f, 30,
                                                                  seq.fps);
        } else if (tracking_method == "ocsort") {
          tracker = std::make_unique<motcpp::trackers::OCSort>(0.2f, 30, 50, 3, 0.3f, false, 80, "iou", false, 0.1f, 3, 0.2f, false, 0.01f,
                                                               0.0001f);
        } else if (tracking_method == "botsort") {  // no ReID weights on this path: motion-only BoT-SORT, CMC not applied
          tracker = std::make_unique<motcpp::trackers::BotSort>("", false, false, 0.3f, 30, 50, 3, 0.3f, false, 80, "iou", false, 0.6f, 0.1f,
                                                                0.7f, 30, 0.8f, 0.5f, 0.25f, "ecc", seq.fps, false, false);
        } else {
          std::cerr << "Unknown tracking method: " << tracking_method << "\n";
          std::cerr << "Supported methods here: sort, bytetrack, ocsort, botsort\n";
          return 1;
        }

        const std::filesystem::path output_file = std::filesystem::path(output_dir) / (seq_name + ".txt");
        if (std::filesystem::exists(output_file)) std::filesystem::remove(output_file);

        std::vector<int> frames;  // every frame that has detections, ascending (:314-319)
        for (const auto& kv : detections) frames.push_back(kv.first);
        std::sort(frames.begin(), frames.end());

        // ablation datasets: detections cover frames 1..2K, ground truth only 1..K -> det frame K+f is GT frame f (:321-375)
        int frame_offset = 0;
        if (!frames.empty() && std::filesystem::exists(seq.gt_path)) {
          std::ifstream gt(seq.gt_path);
          std::string line;
          int max_gt_frame = 0;
          while (std::getline(gt, line)) {
            if (line.empty() || line[0] == '#') continue;
            std::istringstream iss(line);
            std::string token;
            std::getline(iss, token, ',');
            max_gt_frame = std::max(max_gt_frame, std::stoi(token));
          }
          const int max_det = frames.back(), min_det = frames.front();
          if (max_det > max_gt_frame * 1.5 && max_gt_frame > 0) {
            frame_offset = max_det - max_gt_frame;
            std::cout << "  Detected ablation offset: " << frame_offset << " (det " << min_det << "-" << max_det << " -> GT 1-"
                      << max_gt_frame << ")\n";
            std::vector<int> kept;
            for (int f : frames)
              if (f > frame_offset) kept.push_back(f);
            frames = kept;
            if (!frames.empty())
              std::cout << "  Processing frames " << frames.front() << "-" << frames.back() << " (" << frames.size() << " frames)\n";
          }
        }

        const cv::Mat img = cv::Mat::zeros(seq.im_height, seq.im_width, 0);
        int processed = 0;
        for (int frame_id : frames) {
          try {
            const Eigen::MatrixXf& dets = detections.at(frame_id);
            const Eigen::MatrixXf tracks = tracker->update(dets, img, Eigen::MatrixXf(dets.rows(), 0));
            const int out_frame = (frame_offset > 0) ? (frame_id - frame_offset) : frame_id;
            if (tracks.rows() > 0) motcpp::utils::write_mot_results(output_file, motcpp::utils::convert_to_mot_format(tracks, out_frame));
            ++processed;
          } catch (const std::exception& e) {
            std::cerr << "  Error processing frame " << frame_id << ": " << e.what() << "\n";
          }
        }
        std::cout << "  Processed " << processed << " frames\n";
        std::cout << "  Results saved to: " << output_file << "\n\n";
        tracker->reset();
      } catch (const std::exception& e) {
        std::cerr << "Error processing sequence " << seq_name << ": " << e.what() << "\n\n";
      }
    }
  } catch (const std::exception& e) {
    std::cerr << "Error: " << e.what() << "\n";
    return 1;
  }
  std::cout << "Evaluation completed!\nResults saved to: " << output_dir << "\n";
  return 0;
}

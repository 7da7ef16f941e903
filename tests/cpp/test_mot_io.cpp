// Host-only check program for the MOT I/O mirror (include/motcpp/data/mot17_dataset.hpp, utils/mot_format.hpp):
// prints what it parsed so that tests/test_mot_io.py can compare with its own reading of the same files.
#include <cstdio>
#include <cstring>
#include <iostream>

#include "motcpp/data/mot17_dataset.hpp"
#include "motcpp/utils/mot_format.hpp"

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::string mode = argv[1];
  if (mode == "index") {  // index <mot_root> [det_emb_root model_name]
    motcpp::data::MOT17Dataset ds(argv[2], argc > 3 ? argv[3] : "", argc > 4 ? argv[4] : "");
    for (const auto& name : ds.sequence_names()) {
      const auto s = ds.get_sequence_info(name);
      const auto dets = ds.load_detections(s.det_path);
      long rows = 0;
      double sum = 0.0;
      int max_rows = 0, first = 0, last = 0;
      for (const auto& [f, m] : dets) {
        if (!first) first = f;
        last = f;
        rows += m.rows();
        if (m.rows() > max_rows) max_rows = static_cast<int>(m.rows());
        for (int i = 0; i < m.rows(); ++i)
          for (int k = 0; k < 6; ++k) sum += static_cast<double>(m(i, k)) * (k + 1);
      }
      std::printf("SEQ %s fps=%d size=%dx%d frames=%zu first=%d last=%d rows=%ld max_rows=%d sum=%.6f det=%s\n", name.c_str(), s.fps,
                  s.im_width, s.im_height, dets.size(), first, last, rows, max_rows, sum, s.det_path.filename().string().c_str());
    }
    bool threw = false;
    try { ds.get_sequence_info("no-such-sequence"); } catch (const std::runtime_error&) { threw = true; }
    std::printf("NOTFOUND %d\n", threw ? 1 : 0);
    return 0;
  }
  if (mode == "embs") {  // embs <mot_root> <det_emb_root> <model_name> <reid_name>: per sequence and frame the rows, width and a checksum
    motcpp::data::MOT17Dataset ds(argv[2], argv[3], argv[4], argv[5]);
    for (const auto& name : ds.sequence_names()) {
      const auto s = ds.get_sequence_info(name);
      const auto dets = ds.load_detections(s.det_path);
      const auto embs = ds.load_embeddings(ds.embedding_path(name), dets);
      std::printf("EMB %s file=%s frames=%zu\n", name.c_str(), ds.embedding_path(name).filename().string().c_str(), embs.size());
      for (const auto& [f, m] : embs) {
        double sum = 0.0;
        for (int i = 0; i < m.rows(); ++i)
          for (int k = 0; k < m.cols(); ++k) sum += static_cast<double>(m(i, k)) * (i + 1) * (k + 1);
        std::printf("F %d rows=%d dets=%d d=%d sum=%.4f\n", f, static_cast<int>(m.rows()), static_cast<int>(dets.at(f).rows()), static_cast<int>(m.cols()), sum);
      }
    }
    return 0;
  }
  if (mode == "write") {  // write <out_file>: a fixed table through convert_to_mot_format + write_mot_results (twice: it appends)
    Eigen::MatrixXf t(3, 8);
    t << 100.7f, 50.2f, 180.9f, 260.4f, 7, 0.912345678f, 0, 3,
         -5.5f, 10.99f, 20.25f, 99.999f, 12, 0.5f, 1, 0,
         1919.6f, 1000.4f, 1925.1f, 1085.9f, 3, 1.0f, 0, 1;
    const Eigen::MatrixXf m = motcpp::utils::convert_to_mot_format(t, 42);
    if (m.rows() != 3 || m.cols() != 10) return 3;
    motcpp::utils::write_mot_results(argv[2], m);
    motcpp::utils::write_mot_results(argv[2], motcpp::utils::convert_to_mot_format(t, 43));
    const Eigen::MatrixXf e = motcpp::utils::convert_to_mot_format(Eigen::MatrixXf(0, 8), 1);
    return (e.rows() == 0 && e.cols() == 10) ? 0 : 4;
  }
  return 2;
}

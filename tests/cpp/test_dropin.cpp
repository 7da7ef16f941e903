// Drop-in check of the C++ surface: user code written against the reference's headers (motcpp::BaseTracker,
// trackers::Sort/ByteTrack/OCSort/BotSort, utils::linear_assignment/iou_batch) compiles against include/motcpp/ and
// behaves as the reference's own gtest files expect (tests/test_sort.cpp, test_bytetrack.cpp, test_trackers.cpp,
// test_matching.cpp, test_iou.cpp, plus BaseTracker::check_inputs' exception rules, src/tracker.cpp:108-125).
// Needs an MI355X at run time (the classes have no CPU path); compiling it needs none.
#include <cmath>
#include <cstdio>
#include <memory>
#include <set>
#include <stdexcept>
#include <thread>
#include <vector>

#include <motcpp/motcpp.hpp>

static int g_fail = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #c); ++g_fail; } } while (0)

using motcpp::trackers::BotSort;
using motcpp::trackers::ByteTrack;
using motcpp::trackers::DeepOCSort;
using motcpp::trackers::OCSort;
using motcpp::trackers::Sort;
using motcpp::trackers::StrongSORT;
using motcpp::trackers::UCMCTrack;
using motcpp::trackers::BoostTrackTracker;
using motcpp::trackers::HybridSort;

static Eigen::MatrixXf dets1(float x1, float y1, float x2, float y2, float c, float cls) {
  Eigen::MatrixXf d(1, 6);
  d << x1, y1, x2, y2, c, cls;
  return d;
}

int main() {
  cv::Mat img = cv::Mat::zeros(480, 640, CV_8UC3);
  Eigen::MatrixXf single = dets1(100, 100, 200, 200, 0.9f, 0);
  Eigen::MatrixXf multi(3, 6);
  multi << 100, 100, 200, 200, 0.9f, 0, 300, 300, 400, 400, 0.8f, 0, 500, 100, 600, 200, 0.7f, 1;
  Eigen::MatrixXf empty(0, 6);

  {  // test_sort.cpp:36-47
    Sort t(0.3f, 1, 50, 1);
    auto tr = t.update(single, img);
    CHECK(tr.cols() == 8 && tr.rows() == 1);
    CHECK(tr(0, 2) > tr(0, 0) && tr(0, 3) > tr(0, 1));
  }
  {  // test_sort.cpp:49-67 (ids are per instance here: the first id of any tracker is 1)
    Sort t(0.3f, 3, 50, 1);
    t.update(single, img);
    t.update(single, img);
    auto tr = t.update(dets1(110, 110, 210, 210, 0.9f, 0), img);
    CHECK(tr.rows() == 1 && static_cast<int>(tr(0, 4)) == 1);
  }
  {  // test_sort.cpp:69-84
    Sort t(0.3f, 2, 50, 1);
    t.update(single, img);
    t.update(empty, img);
    CHECK(t.update(empty, img).rows() == 0);
  }
  {  // test_sort.cpp:128-148
    Sort t(0.3f, 5, 50, 1);
    for (int i = 0; i < 5; ++i) t.update(dets1(100 + i * 10, 100 + i * 10, 200 + i * 10, 200 + i * 10, 0.9f, 0), img);
    t.update(empty, img);
    auto tr = t.update(dets1(160, 160, 260, 260, 0.9f, 0), img);
    CHECK(tr.rows() == 1 && static_cast<int>(tr(0, 4)) == 1);
  }
  {  // test_bytetrack.cpp:125-149 / test_trackers.cpp:39-107 through the base-class pointer
    std::unique_ptr<motcpp::BaseTracker> ts[3] = {std::make_unique<ByteTrack>(), std::make_unique<OCSort>(), std::make_unique<BotSort>()};
    for (auto& t : ts) {
      std::set<int> a, b;
      for (int f = 0; f < 3; ++f) {
        auto tr = t->update(multi, img);
        for (int i = 0; i < tr.rows(); ++i) {
          CHECK(tr.cols() == 8 && tr(i, 0) < tr(i, 2) && tr(i, 1) < tr(i, 3) && tr(i, 4) > 0 && tr(i, 5) >= 0 && tr(i, 5) <= 1);
          (f == 1 ? a : b).insert(static_cast<int>(tr(i, 4)));
        }
      }
      bool persisted = false;
      for (int id : a) persisted = persisted || b.count(id);
      CHECK(persisted);
      t->reset();
      CHECK(t->update(empty, img).rows() == 0);  // test_trackers.cpp:100-107 (fresh state)
      CHECK(t->update(multi, img).cols() == 8);
    }
  }
  {  // src/tracker.cpp:108-125
    ByteTrack t;
    bool threw = false;
    try { t.update(Eigen::MatrixXf(2, 5), img); } catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);
    threw = false;
    try { t.update(single, cv::Mat()); } catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);
    threw = false;
    try { t.update(single, img, Eigen::MatrixXf(3, 8)); } catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);
    Sort s;  // SORT never calls check_inputs (sort.cpp:102-110)
    CHECK(s.update(single, cv::Mat()).cols() == 8);
  }
  {  // a StreamBatch frame that one stream rejects leaves NO tracker a frame ahead: every stream is validated before any is committed
    ByteTrack a, b, ref;
    motcpp::StreamBatch sb({&a, &b});
    Eigen::MatrixXf bad(2, 5);  // 5 columns: check_inputs throws
    bad.setZero();
    bool threw = false;
    try { sb.update({multi, bad}, img); } catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);
    auto out = sb.update({multi, single}, img);  // the first frame either tracker really sees: ByteTrack activates its tracks on frame 1 only
    Eigen::MatrixXf r = ref.update(multi, img);
    CHECK(out.size() == 2 && out[0].rows() == r.rows() && r.rows() == multi.rows());
    for (int i = 0; i < r.rows(); ++i)
      for (int k = 0; k < 8; ++k) CHECK(out[0](i, k) == r(i, k));
  }
  {  // the device-lifecycle batch gives each stream exactly what a ByteTrack instance gives it
    motcpp::ByteTrackDeviceBatch batch(2, 64, 16);
    ByteTrack a, b;
    for (int f = 0; f < 6; ++f) {
      Eigen::MatrixXf da = multi, db = single;
      for (int i = 0; i < da.rows(); ++i) { da(i, 0) += 2.f * f; da(i, 2) += 2.f * f; }
      auto out = batch.update({da, db});
      Eigen::MatrixXf ra = a.update(da, img), rb = b.update(db, img);
      CHECK(out.size() == 2 && out[0].rows() == ra.rows() && out[1].rows() == rb.rows());
      for (int i = 0; i < ra.rows(); ++i)
        for (int k = 0; k < 8; ++k) CHECK(out[0](i, k) == ra(i, k));
      for (int i = 0; i < rb.rows(); ++i)
        for (int k = 0; k < 8; ++k) CHECK(out[1](i, k) == rb(i, k));
    }
    bool threw = false;
    try { batch.update({multi}); } catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);
    motcpp::SortDeviceBatch sbatch(1, 64, 16);
    Sort ref;
    for (int f = 0; f < 6; ++f) {
      Eigen::MatrixXf d = multi;
      for (int i = 0; i < d.rows(); ++i) { d(i, 0) += 3.f * f; d(i, 2) += 3.f * f; }
      auto out = sbatch.update({d});
      Eigen::MatrixXf r = ref.update(d, img);
      CHECK(out[0].rows() == r.rows());
      for (int i = 0; i < r.rows(); ++i)
        for (int k = 0; k < 8; ++k) CHECK(out[0](i, k) == r(i, k));
    }
  }
  {  // OC-SORT and BoT-SORT with the lifecycle on the device: each stream gets what a tracker instance gives it
    motcpp::OCSortDeviceBatch ob(2, 64, 16);
    OCSort oa, obb;
    motcpp::BotSortDeviceBatch bb(1, 64, 16, 8);
    BotSort bref;
    Eigen::MatrixXf warp(2, 3);
    warp << 1, 0, 5, 0, 1, 0;
    for (int f = 0; f < 6; ++f) {
      Eigen::MatrixXf da = multi, db = single;
      for (int i = 0; i < da.rows(); ++i) { da(i, 0) += 5.f * f; da(i, 2) += 5.f * f; }
      auto out = ob.update({da, db});
      Eigen::MatrixXf ra = oa.update(da, img), rb = obb.update(db, img);
      CHECK(out.size() == 2 && out[0].rows() == ra.rows() && out[1].rows() == rb.rows());
      for (int i = 0; i < ra.rows() && i < out[0].rows(); ++i)
        for (int k = 0; k < 8; ++k) CHECK(out[0](i, k) == ra(i, k));
      Eigen::MatrixXf e(da.rows(), 8);
      for (int i = 0; i < da.rows(); ++i) for (int k = 0; k < 8; ++k) e(i, k) = (k == i % 8) ? 1.0f : 0.1f * static_cast<float>(f + 1);
      std::vector<Eigen::MatrixXf> ws;
      if (f > 0) { ws.push_back(warp); bref.set_camera_motion(warp); }
      auto bo = bb.update({da}, {e}, ws);
      Eigen::MatrixXf br = bref.update(da, img, e);
      CHECK(bo[0].rows() == br.rows());
      for (int i = 0; i < br.rows() && i < bo[0].rows(); ++i)
        for (int k = 0; k < 8; ++k) CHECK(bo[0](i, k) == br(i, k));
    }
    bool threw = false;
    try { ob.update({multi, single}, {multi}); } catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);  // only BoT-SORT batches take embeddings
    threw = false;
    try { motcpp::OCSortDeviceBatch bad(1, 64, 16, 0.2f, 30, 3, 0.3f, 0.1f, 3, 0.2f, false, 0.01f, 0.0001f, "iou_obb"); } catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);
  }
  {  // camera motion (botsort.cpp:317-324 with the warp supplied by the caller): a camera jumping 60 px per frame
    BotSort cmc, plain;
    Eigen::MatrixXf warp(2, 3);
    warp << 1, 0, 60, 0, 1, 0;
    for (int f = 0; f < 6; ++f) {
      Eigen::MatrixXf d = multi;
      for (int i = 0; i < d.rows(); ++i) { d(i, 0) += 60.f * f; d(i, 2) += 60.f * f; }
      if (f > 0) cmc.set_camera_motion(warp);
      Eigen::MatrixXf r = cmc.update(d, img);
      plain.update(d, img);
      CHECK(r.rows() == multi.rows());
      for (int i = 0; i < r.rows(); ++i) {  // each id stays on its own detection, at the detection's place
        const int det = static_cast<int>(r(i, 7));
        CHECK(static_cast<int>(r(i, 4)) == det + 1 && std::fabs(r(i, 0) - d(det, 0)) < 0.5f);
      }
    }
    bool threw = false;
    try { cmc.set_camera_motion(Eigen::MatrixXf(3, 3)); } catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);
    // two cameras in one StreamBatch, each with its own warp per frame (one predict+warp launch carries both), against
    // the same trackers stepped one by one
    BotSort b0, b1, s0, s1;
    motcpp::StreamBatch batch({&b0, &b1});
    Eigen::MatrixXf w0(2, 3), w1(2, 3);
    w0 << 1, 0, 25, 0, 1, -10;
    w1 << 1.01f, 0, -15, 0, 1.01f, 5;
    for (int f = 0; f < 5; ++f) {
      Eigen::MatrixXf d0 = multi, d1 = multi;
      for (int i = 0; i < multi.rows(); ++i) {
        d0(i, 0) += 25.f * f; d0(i, 2) += 25.f * f; d0(i, 1) -= 10.f * f; d0(i, 3) -= 10.f * f;
        d1(i, 0) -= 15.f * f; d1(i, 2) -= 15.f * f; d1(i, 1) += 5.f * f; d1(i, 3) += 5.f * f;
      }
      if (f > 0) { b0.set_camera_motion(w0); b1.set_camera_motion(w1); s0.set_camera_motion(w0); s1.set_camera_motion(w1); }
      auto out = batch.update({d0, d1}, img);
      Eigen::MatrixXf r0 = s0.update(d0, img), r1 = s1.update(d1, img);
      CHECK(out.size() == 2 && out[0].rows() == r0.rows() && out[1].rows() == r1.rows() && r0.rows() == multi.rows());
      for (int i = 0; i < r0.rows(); ++i)
        for (int k = 0; k < 8; ++k) CHECK(out[0](i, k) == r0(i, k) && out[1](i, k) == r1(i, k));
    }
  }
  {  // asso_func: stored by every tracker, read only by OC-SORT, at update time (ocsort.cpp:413; iou.hpp:385-408)
    ByteTrack bt(0.3f, 30, 50, 3, 0.3f, false, 80, "no-such-measure");
    CHECK(bt.update(multi, img).cols() == 8);
    OCSort good(0.2f, 30, 50, 3, 0.3f, false, 80, "giou");
    CHECK(good.update(multi, img).cols() == 8);
    OCSort bad(0.2f, 30, 50, 3, 0.3f, false, 80, "no-such-measure");  // the constructor accepts it, like the reference
    bool threw = false;
    try { bad.update(multi, img); } catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);
    // tests/test_iou.cpp:75-116
    Eigen::MatrixXf b1(1, 4), b2(1, 4), b3(1, 4);
    b1 << 0, 0, 100, 100;
    b2 << 50, 50, 150, 150;
    b3 << 200, 200, 300, 300;
    using namespace motcpp::utils;
    const float g = giou_batch(b1, b2)(0, 0), d = diou_batch(b1, b2)(0, 0), c = ciou_batch(b1, b2)(0, 0);
    CHECK(g >= 0.f && g <= 1.f && d >= 0.f && d <= 1.f && c >= 0.f && c <= 1.f);
    const float ce = centroid_batch(b1, b3, 640, 480)(0, 0);
    CHECK(ce > 0.f && ce < 1.f);
    AssociationFunction asso(640, 480, "iou");
    Eigen::MatrixXf r = asso(b1, b2);
    CHECK(r.rows() == 1 && r.cols() == 1 && std::fabs(r(0, 0) - 0.143f) < 0.01f);
    threw = false;
    try { AssociationFunction nope(640, 480, "nope"); } catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);
  }
  {  // test_matching.cpp:24-110
    using motcpp::utils::linear_assignment;
    Eigen::MatrixXf c1(1, 1);
    c1 << 0.1f;
    auto r = linear_assignment(c1, 0.5f);
    CHECK(r.matches.size() == 1 && r.matches[0][0] == 0 && r.matches[0][1] == 0 && r.unmatched_a.empty() && r.unmatched_b.empty());
    c1 << 0.9f;
    r = linear_assignment(c1, 0.5f);
    CHECK(r.matches.empty() && r.unmatched_a.size() == 1 && r.unmatched_b.size() == 1);
    Eigen::MatrixXf c32(3, 2);
    c32 << 0.1f, 0.9f, 0.9f, 0.1f, 0.9f, 0.9f;
    r = linear_assignment(c32, 0.5f);
    CHECK(r.matches.size() == 2 && r.unmatched_a.size() == 1 && r.unmatched_a[0] == 2 && r.unmatched_b.empty());
    Eigen::MatrixXf c22(2, 2);
    c22 << 0.1f, 0.2f, 0.3f, 0.1f;
    r = linear_assignment(c22, 0.5f);
    CHECK(r.matches.size() == 2 && r.matches[0][1] == 0 && r.matches[1][1] == 1);
    r = linear_assignment(Eigen::MatrixXf(0, 0), 0.5f);
    CHECK(r.matches.empty() && r.unmatched_a.empty() && r.unmatched_b.empty());
  }
  {  // test_iou.cpp:29-75
    Eigen::MatrixXf b1(1, 4), b2(1, 4), b3(1, 4);
    b1 << 0, 0, 100, 100;
    b2 << 50, 50, 150, 150;
    b3 << 200, 200, 300, 300;
    CHECK(motcpp::utils::iou_batch(b1, b1)(0, 0) == 1.0f);
    CHECK(motcpp::utils::iou_batch(b1, b3)(0, 0) == 0.0f);
    CHECK(std::fabs(motcpp::utils::iou_batch(b1, b2)(0, 0) - 0.143f) < 0.01f);
    auto e = motcpp::utils::iou_batch(Eigen::MatrixXf(0, 4), b1);
    CHECK(e.rows() == 0 && e.cols() == 1);
  }
  {  // StreamBatch == the same trackers stepped one by one
    ByteTrack a1, a2, b1, b2;
    motcpp::StreamBatch batch({&a1, &a2});
    for (int f = 0; f < 5; ++f) {
      Eigen::MatrixXf d1 = multi, d2 = single;
      d1(0, 0) += f; d1(0, 2) += f;
      auto out = batch.update({d1, d2}, img);
      auto r1 = b1.update(d1, img), r2 = b2.update(d2, img);
      CHECK(out.size() == 2 && out[0].rows() == r1.rows() && out[1].rows() == r2.rows());
      for (int i = 0; i < r1.rows(); ++i) for (int k = 0; k < 8; ++k) CHECK(out[0](i, k) == r1(i, k));
    }
  }
  {  // DeepOCSort (deepocsort.hpp:97-119): embeddings come with update(); without them it throws unless embedding_off
    DeepOCSort t("", false, false, 0.3f, 30, 50, 1);
    Eigen::MatrixXf e(3, 8);
    for (int i = 0; i < 3; ++i) for (int k = 0; k < 8; ++k) e(i, k) = (i == k % 3) ? 1.0f : 0.1f;
    CHECK(t.update(multi, img, e).rows() == 0);  // first frame: tracks are born, nothing is emitted (deepocsort.cpp:652-664)
    auto tr = t.update(multi, img, e);
    CHECK(tr.rows() == 3 && tr.cols() == 8);
    bool threw = false;
    try { DeepOCSort u; u.update(multi, img); } catch (const std::exception&) { threw = true; }
    CHECK(threw);
    DeepOCSort off("", false, false, 0.3f, 30, 50, 1, 0.3f, false, 80, "iou", false, 3, 0.2f, 0.5f, 0.95f, 0.5f, true);
    off.update(multi, img);
    CHECK(off.update(multi, img).rows() == 3);
    threw = false;
    try { Eigen::MatrixXf w(3, 3); off.set_camera_motion(w); } catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);
  }
  {  // the remaining cases of the reference's gtest files (tests/test_sort.cpp:87-126, tests/test_trackers.cpp:107-135, tests/test_bytetrack.cpp:38-123)
    Sort hi(0.3f, 3, 50, 1, 0.9f);  // IoUThreshold: a far detection does not match, a new track (id 2) is the one reported
    hi.update(single, img);
    auto far = hi.update(dets1(300, 300, 400, 400, 0.9f, 0), img);
    CHECK(far.rows() == 1 && static_cast<int>(far(0, 4)) == 2);
    Sort pc(0.3f, 3, 50, 1, 0.3f, true, 80);  // MultiClassTracking
    CHECK(pc.update(multi, img).cols() == 8);
    Eigen::MatrixXf low(2, 6);
    low << 100, 100, 200, 200, 0.3f, 0, 300, 300, 400, 400, 0.6f, 0;
    Sort cf(0.5f, 3, 50, 1);  // ConfidenceFiltering
    CHECK(cf.update(low, img).rows() <= 1);
    ByteTrack bl(0.5f);  // TrackerWithLowConfidenceDetections
    CHECK(bl.update(low, img).rows() <= 1);
    OCSort oc;  // OCSortUpdateReturnsValidOutput
    CHECK(oc.update(multi, img).cols() == 8);
    ByteTrack two(0.1f, 30, 50, 1, 0.3f, false, 80, "iou", false, 0.1f, 0.45f, 0.8f, 30, 30);  // TwoStageAssociation
    Eigen::MatrixXf mixed(3, 6);
    mixed << 100, 100, 200, 200, 0.9f, 0, 300, 300, 400, 400, 0.3f, 0, 500, 100, 600, 200, 0.6f, 1;
    CHECK(two.update(mixed, img).cols() == 8);
    ByteTrack thr(0.1f, 30, 50, 1, 0.3f, false, 80, "iou", false, 0.1f, 0.6f);  // TrackThresholdFiltering
    CHECK(thr.update(mixed, img).cols() == 8);
    ByteTrack rec(0.3f, 30, 50, 1, 0.3f, false, 80, "iou", false, 0.1f, 0.45f, 0.8f, 30, 30);  // LostTrackRecovery: kept through the second stage
    auto r1 = rec.update(single, img);
    auto r2 = rec.update(dets1(100, 100, 200, 200, 0.3f, 0), img);
    CHECK(r1.rows() == 1 && r2.rows() == 1 && r1(0, 4) == r2(0, 4) && r2(0, 5) == 0.3f);
    ByteTrack f60(0.3f, 30, 50, 3, 0.3f, false, 80, "iou", false, 0.1f, 0.45f, 0.8f, 30, 60);  // FrameRateAwareness
    CHECK(f60.update(single, img).cols() == 8);
  }
  {  // StrongSORT (strongsort.hpp:294-312): the reference's constructor signature; embeddings come with update(). From a cold start the
     // tracks stay tentative (the reference's matching as written, see csrc/host/strongsort.cpp): tables have 8 columns and may be empty.
    StrongSORT t;
    Eigen::MatrixXf e(3, 8);
    for (int i = 0; i < 3; ++i) for (int k = 0; k < 8; ++k) e(i, k) = (i == k % 3) ? 1.0f : 0.1f;
    for (int f = 0; f < 4; ++f) CHECK(t.update(multi, img, e).cols() == 8);
    CHECK(t.update(empty, img).rows() == 0);
    t.reset();
    CHECK(t.update(multi, img).cols() == 8);  // without embeddings the IoU stage decides (strongsort.cpp:688-690)
    bool threw = false;
    try { t.update(single, cv::Mat()); } catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);  // check_inputs (:853)
    threw = false;
    try { StrongSORT w("osnet.onnx"); } catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);  // ReID inference is outside the hot path
  }
  {  // UCMCTrack (ucmc.hpp:140-159): a detection seen in every frame is reported from the third frame on (tentative, birth count 1,
     // then confirmed: ucmc.cpp:497-500) with the detection's own box, confidence, class and row; low-confidence detections never
     // start a track; a calibrated camera (Ki 3 x 4, Ko 4 x 4) is accepted like the reference's vectors
    UCMCTrack t;
    CHECK(t.update(multi, img).rows() == 0);
    CHECK(t.update(multi, img).rows() == 0);
    auto tr = t.update(multi, img);
    CHECK(tr.rows() == 3 && tr.cols() == 8);
    CHECK(tr(0, 0) == 100.0f && tr(0, 3) == 200.0f && static_cast<int>(tr(0, 4)) == 1 && tr(0, 5) == 0.9f && static_cast<int>(tr(0, 7)) == 0);
    CHECK(static_cast<int>(tr(2, 4)) == 3 && static_cast<int>(tr(2, 6)) == 1);
    CHECK(t.update(empty, img).rows() == 0);
    CHECK(t.update(multi, img).rows() == 3);  // coasted tracks are picked up again
    t.reset();
    CHECK(t.update(multi, img).rows() == 0);
    UCMCTrack low(0.3f, 30, 50, 3, 0.3f, false, 80, "iou", false, 100.0, 100.0, 5.0, 5.0, 10.0, 1.0 / 25.0, 0.95f);
    for (int f = 0; f < 4; ++f) CHECK(low.update(multi, img).rows() == 0);  // nothing reaches high_score: no track is ever started
    const std::vector<double> Ki = {1000, 0, 320, 0, 0, 1000, 240, 0, 0, 0, 1, 0};
    const std::vector<double> Ko = {1, 0, 0, 0, 0, -0.5, -0.8660254037844386, 0, 0, 0.8660254037844386, -0.5, 5, 0, 0, 0, 1};
    UCMCTrack cam(0.3f, 30, 50, 3, 0.3f, false, 80, "iou", false, 100.0, 100.0, 5.0, 5.0, 10.0, 1.0 / 30.0, 0.5f, Ki, Ko);
    cam.update(multi, img); cam.update(multi, img);
    CHECK(cam.update(multi, img).rows() == 3);
  }
  {  // BoostTrackTracker (boosttrack.hpp:95-125), motion only: new tracks are reported at once while frame_count <= min_hits
     // (boosttrack.cpp:663-680); a detection below det_thresh that sits on a predicted track comes back with the boosted confidence
     // max_iou * dlo_boost_coef (:393-400); ReID is outside the path
    BoostTrackTracker t;
    auto tr = t.update(multi, img);
    CHECK(tr.rows() == 3 && tr.cols() == 8 && static_cast<int>(tr(0, 4)) == 1 && static_cast<int>(tr(2, 7)) == 2);
    Eigen::MatrixXf weak = multi;
    weak(1, 4) = 0.5f;
    tr = t.update(weak, img);
    CHECK(tr.rows() == 3);
    CHECK(tr(1, 5) > 0.6f && tr(1, 5) < 0.66f);  // 0.5 lifted to IoU x 0.65 (the prediction has not moved)
    CHECK(t.update(empty, img).rows() == 0);
    t.reset();
    CHECK(static_cast<int>(t.update(single, img)(0, 4)) == 1);
    bool threw = false;
    try { BoostTrackTracker r("osnet.onnx"); } catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);
  }
  {  // HybridSort (hybridsort.hpp:126-164): ids are reported + 1 and start at 2 (++next_id_, then id + 1: hybridsort.cpp:21-23, :1225), the
     // table is in reverse track order (:1213); a matched track reports the detection's own box (its last observation, :364-369);
     // embeddings are refused (the ReID branch is not built)
    HybridSort t;
    auto tr = t.update(multi, img);
    CHECK(tr.rows() == 2 && tr.cols() == 8);  // conf > det_thresh 0.7: the first two detections
    CHECK(static_cast<int>(tr(0, 4)) == 3 && static_cast<int>(tr(1, 4)) == 2 && static_cast<int>(tr(0, 7)) == 1 && static_cast<int>(tr(1, 7)) == 0);
    tr = t.update(multi, img);
    CHECK(tr.rows() == 2 && tr(1, 0) == 100.0f && tr(1, 3) == 200.0f && tr(0, 0) == 300.0f);
    CHECK(t.update(empty, img).rows() == 0);
    t.reset();
    CHECK(static_cast<int>(t.update(single, img)(0, 4)) == 2);
    bool threw = false;
    Eigen::MatrixXf e(3, 8);
    for (int i = 0; i < 3; ++i) for (int k = 0; k < 8; ++k) e(i, k) = 1.0f;
    try { t.update(multi, img, e); } catch (const std::exception&) { threw = true; }
    CHECK(threw);
  }
  {  // an embedding matrix with the wrong number of rows is rejected, alone and inside a StreamBatch
    BotSort t;
    Eigen::MatrixXf e(2, 8);
    for (int i = 0; i < 2; ++i) for (int k = 0; k < 8; ++k) e(i, k) = 1.0f;
    bool threw = false;
    try { t.update(multi, img, e); } catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);
    BotSort a, b;
    motcpp::StreamBatch batch({&a, &b});
    threw = false;
    try { batch.update({multi, multi}, img, {e, e}); } catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);
    OCSort bad(0.2f, 30, 50, 3, 0.3f, false, 80, "not_a_measure");
    motcpp::StreamBatch batch2({&bad});
    threw = false;
    try { batch2.update({multi}, img); } catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);  // the batch runs each tracker's own checks
  }
  {  // the motion gate and fuse_iou seams (matching.hpp:60-103): shapes, the chi-square gate, argument errors
    namespace U = motcpp::utils;
    Eigen::MatrixXf means(2, 8), covs(2, 64), meas(3, 4), cost(2, 3);
    for (int i = 0; i < 2; ++i) {
      for (int k = 0; k < 8; ++k) means(i, k) = 0.0f;
      for (int k = 0; k < 64; ++k) covs(i, k) = 0.0f;
      for (int k = 0; k < 8; ++k) covs(i, 9 * k) = 4.0f;
      means(i, 0) = 100.0f * (i + 1); means(i, 1) = 50.0f; means(i, 2) = 0.5f; means(i, 3) = 80.0f;
      for (int j = 0; j < 3; ++j) cost(i, j) = 0.5f;
    }
    const float z[3][4] = {{100.f, 50.f, 0.5f, 80.f}, {200.f, 50.f, 0.5f, 80.f}, {900.f, 700.f, 0.5f, 80.f}};
    for (int j = 0; j < 3; ++j) for (int k = 0; k < 4; ++k) meas(j, k) = z[j][k];
    const Eigen::MatrixXf g = U::gating_distance("xyah", means, covs, meas);
    CHECK(g.rows() == 2 && g.cols() == 3);
    CHECK(g(0, 0) == 0.0f && g(1, 1) == 0.0f && g(0, 2) > 9.4877f);
    const Eigen::MatrixXf f = U::fuse_motion("xyah", cost, means, covs, meas);
    CHECK(f(0, 0) == 0.98f * 0.5f && std::isinf(f(0, 2)) && std::isinf(f(1, 0)));
    const Eigen::MatrixXf s2 = U::gate_cost_matrix("xyah", cost, means, covs, meas, 0.995f, 1e5f);
    CHECK(s2(0, 0) == 0.995f * 0.5f && s2(0, 2) > 1e4f);
    CHECK(U::gating_distance("xywh", means, covs, meas, true)(1, 1) == 0.0f);
    bool threw = false;
    try { U::gating_distance("xyah", means, covs, meas, false, "euclid"); } catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);
    threw = false;
    try { U::gating_distance("xysr", means, covs, meas); } catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);
    Eigen::MatrixXf a(2, 4), b(3, 4), reid(2, 3);
    const float ab[3][4] = {{0, 0, 10, 10}, {20, 20, 40, 40}, {100, 100, 110, 110}};
    for (int k = 0; k < 4; ++k) { a(0, k) = ab[0][k]; a(1, k) = ab[1][k]; }
    for (int j = 0; j < 3; ++j) for (int k = 0; k < 4; ++k) b(j, k) = ab[j][k];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) reid(i, j) = 0.25f;
    const Eigen::MatrixXf fi = U::fuse_iou(reid, a, b);
    CHECK(fi(0, 0) == 0.25f && fi(1, 1) == 0.25f);  // iou 1: 1 - 0.75 * (1 + 1) / 2
    CHECK(fi(0, 2) == 1.0f - 0.75f * 0.5f);         // iou 0: 1 - 0.75 * (1 + 0) / 2
  }
  {  // The reference's threading model (include/motcpp/tracker.hpp:67-69, docs/guides/architecture.md:242-255): one tracker object per
     // camera, each updated from its own thread. Here the objects are streams of a shared device batch and concurrent update() calls run
     // as one launch sequence: every object's tables must be bit-identical to the same object's run on one thread.
    const int T = 16, F = 24;
    auto frame_of = [&](int t, int f) {  // a small moving crowd per camera, deterministic
      const int n = 12 + (t * 7 + f * 3) % 9;
      Eigen::MatrixXf d((f % 10 == 6) ? 0 : n, 6);
      for (int i = 0; i < d.rows(); ++i) {
        const float x = 40.f + 55.f * i + 2.5f * f + 3.f * t, y = 60.f + 17.f * ((i * 5 + t) % 11) + 1.5f * f;
        d(i, 0) = x; d(i, 1) = y; d(i, 2) = x + 38.f + (i % 3); d(i, 3) = y + 90.f + (i % 4);
        d(i, 4) = ((i + f + t) % 7 == 0) ? 0.3f : 0.55f + 0.04f * ((i + t) % 10); d(i, 5) = static_cast<float>(i % 2);
      }
      return d;
    };
    auto run_one = [&](motcpp::BaseTracker& trk, int t, std::vector<Eigen::MatrixXf>* out) {
      for (int f = 0; f < F; ++f) out->push_back(trk.update(frame_of(t, f), img));
    };
    auto same = [](const Eigen::MatrixXf& a, const Eigen::MatrixXf& b) {
      if (a.rows() != b.rows() || a.cols() != b.cols()) return false;
      for (Eigen::Index i = 0; i < a.rows(); ++i) for (Eigen::Index k = 0; k < a.cols(); ++k) if (a(i, k) != b(i, k)) return false;
      return true;
    };
    for (int kind = 0; kind < 3; ++kind) {
      auto make = [&]() -> std::unique_ptr<motcpp::BaseTracker> {
        if (kind == 0) return std::make_unique<ByteTrack>();
        if (kind == 1) return std::make_unique<Sort>(0.3f, 3, 50, 2);
        return std::make_unique<OCSort>();
      };
      std::vector<std::vector<Eigen::MatrixXf>> alone(T), together(T);
      for (int t = 0; t < T; ++t) { auto trk = make(); run_one(*trk, t, &alone[t]); }
      std::vector<std::unique_ptr<motcpp::BaseTracker>> trk(T);
      for (int t = 0; t < T; ++t) trk[t] = make();
      std::vector<std::thread> th;
      for (int t = 0; t < T; ++t) th.emplace_back([&, t] { run_one(*trk[t], t, &together[t]); });
      for (auto& x : th) x.join();
      long rows = 0;
      for (int t = 0; t < T; ++t) {
        CHECK(together[t].size() == alone[t].size());
        for (size_t f = 0; f < alone[t].size() && f < together[t].size(); ++f) { CHECK(same(alone[t][f], together[t][f])); rows += alone[t][f].rows(); }
      }
      CHECK(rows > 0);
    }
  }
  std::printf(g_fail ? "%d check(s) failed\n" : "drop-in ok\n", g_fail);
  return g_fail ? 1 : 0;
}

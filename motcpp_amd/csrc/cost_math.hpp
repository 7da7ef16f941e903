// Pairwise box-cost arithmetic shared by the N x M cost kernel (cost_kernels.hip) and the assignment kernel's
// on-the-fly cost functor (lap_kernel.hip): one definition, so a cost recomputed inside the solver is bit-identical
// to the materialised matrix. Reference order of operations: include/motcpp/utils/iou.hpp:85-95,
// src/utils/matching.cpp:62-65,130-143, src/trackers/botsort.cpp:439-465.
#pragma once
#include <cmath>
#include "grp.hpp"
#include "../../include/motcpp_amd.h"

namespace mot {

MOT_HD float smax(float a, float b) { return (a < b) ? b : a; }  // std::max(a,b), NaN behaviour included
MOT_HD float smin(float a, float b) { return (b < a) ? b : a; }  // std::min(a,b)

MOT_HD float iou_pair(const float a[4], float area_a, const float b[4], float area_b) {
  const float xx1 = smax(a[0], b[0]);
  const float yy1 = smax(a[1], b[1]);
  const float xx2 = smin(a[2], b[2]);
  const float yy2 = smin(a[3], b[3]);
  const float w = smax(0.0f, xx2 - xx1);
  const float h = smax(0.0f, yy2 - yy1);
  const float inter = w * h;
#if defined(__HIP_DEVICE_COMPILE__)
  // Disjoint boxes are the common case in tracking. inter == +0 gives IoU +0 whatever the union is, so when no lane of
  // the wavefront has an intersection the (correctly rounded, ~20-instruction) division is skipped for all of them.
  if (__builtin_amdgcn_ballot_w64(inter != 0.0f) == 0) return 0.0f;
#endif
  const float uni = area_a + area_b - inter;
  return (uni > 0.0f) ? (inter / uni) : 0.0f;
}

// The other association measures of AssociationFunction (include/motcpp/utils/iou.hpp:122-334), elementwise in the
// reference's operation order. a = first argument (detections in OC-SORT), b = second. atan: correctly rounded float
// through double (the reference's libm atanf is < 1 ulp and version dependent; same convention as OC-SORT's acos).
MOT_HD float assoc_pair(int assoc, float frame_diag, const float a[4], float area_a, const float b[4], float area_b) {
  if (assoc == MOT_ASSOC_CENTROID) {  // :303-334
    const float dx = (a[0] + a[2]) / 2.0f - (b[0] + b[2]) / 2.0f;
    const float dy = (a[1] + a[3]) / 2.0f - (b[1] + b[3]) / 2.0f;
    const float dist = sqrtf(dx * dx + dy * dy);
    return 1.0f - dist / frame_diag;
  }
  const float iou = iou_pair(a, area_a, b, area_b);
  if (assoc == MOT_ASSOC_IOU) return iou;
  if (assoc == MOT_ASSOC_HMIOU) {  // :122-150
    const float ih = smax(smin(a[3], b[3]) - smax(a[1], b[1]), 0.0f);
    const float uh = smax(smax(a[3], b[3]) - smin(a[1], b[1]), 1e-10f);
    return iou * (ih / uh);
  }
  const float ox = smax(a[2], b[2]) - smin(a[0], b[0]);  // smallest enclosing box
  const float oy = smax(a[3], b[3]) - smin(a[1], b[1]);
  if (assoc == MOT_ASSOC_GIOU) {  // :155-193
    const float area_enclose = ox * oy;
    const float intersection = iou * (area_a + area_b) / (iou + 1e-10f);
    const float union_area = area_a + area_b - intersection;
    const float g = iou - (area_enclose - union_area) / (area_enclose + 1e-10f);
    return (g + 1.0f) / 2.0f;
  }
  const float ddx = (a[0] + a[2]) / 2.0f - (b[0] + b[2]) / 2.0f;
  const float ddy = (a[1] + a[3]) / 2.0f - (b[1] + b[3]) / 2.0f;
  const float inner = ddx * ddx + ddy * ddy;
  if (assoc == MOT_ASSOC_DIOU) {  // :261-298
    const float outer = ox * ox + oy * oy;
    const float d = iou - inner / (outer + 1e-10f);
    return (d + 1.0f) / 2.0f;
  }
  // CIoU :198-256
  const float epsilon = 1e-7f;
  const float outer = ox * ox + oy * oy + epsilon;
  const float w1 = a[2] - a[0], h1 = a[3] - a[1], w2 = b[2] - b[0], h2 = b[3] - b[1];
  const float ad = static_cast<float>(atan(static_cast<double>(w2 / (h2 + epsilon)))) -
                   static_cast<float>(atan(static_cast<double>(w1 / (h1 + epsilon))));
  const float k = 4.0f / static_cast<float>(3.14159265358979323846 * 3.14159265358979323846);
  const float v = k * (ad * ad);
  const float S = 1.0f - iou;
  const float alpha = v / (S + v + epsilon);
  const float c = iou - inner / outer + alpha * v;
  return (c + 1.0f) / 2.0f;
}

// cost of one pair given its IoU. `emb_at()` is only evaluated when the appearance term can matter.
struct CostParams {
  int mode;
  float prox, app;
  int fuse;
  bool has_emb;    // an embedding-distance matrix exists
  bool const_emb;  // no features at all: the cosine distance is the constant 1
  int assoc = MOT_ASSOC_IOU;  // similarity the cost is built from
  float frame_diag = 1.0f;
};
// WITH_APPEARANCE = false: the caller guarantees mode != MOT_COST_BOTSORT and the gated-appearance branch is not compiled
// (the assignment kernel's variants for ByteTrack/SORT/OC-SORT launches: fewer scalar registers and branches per pair)
template <bool WITH_APPEARANCE = true, class EmbFn>
MOT_HD float cost_from_iou(const CostParams& p, float iou, float conf, EmbFn emb_at) {
  float d = 1.0f - iou;  // iou_distance
  if (WITH_APPEARANCE && p.mode == MOT_COST_BOTSORT) {
    const bool far = d > p.prox;  // mask from the un-fused distance (botsort.cpp:439)
    if (p.fuse) { const float sim = 1.0f - d; d = 1.0f - sim * conf; }
    if (p.has_emb || p.const_emb) {
      float e = 1.0f;
      if (!far) {  // a masked pair is forced to 1 whatever its embedding distance is
        if (p.has_emb) e = emb_at();
        e = e / 2.0f;
        if (e > p.app) e = 1.0f;
      }
      d = smin(d, e);
    }
    return d;
  }
  if (WITH_APPEARANCE && p.mode == MOT_COST_FUSE_IOU) {  // fuse_iou (matching.cpp:109-128): 1 - (1 - reid) * (1 + iou_sim) / 2
    const float reid_sim = 1.0f - emb_at();
    const float iou_sim = 1.0f - d;
    const float fuse_sim = reid_sim * ((1.0f + iou_sim) / 2.0f);
    return 1.0f - fuse_sim;
  }
  // the four plain modes as selects on the (uniform) mode rather than branches: inside the assignment solver this runs
  // once per visited pair, where a taken scalar branch costs more than the two spare multiplies
  const float fused = 1.0f - (1.0f - d) * conf;  // fuse_score: 1 - (1 - d) * conf
  const float dist = (p.mode == MOT_COST_IOU_DIST_FUSE) ? fused : d;
  const float sim = (p.mode == MOT_COST_NEG_IOU) ? -iou : iou;
  return (p.mode == MOT_COST_IOU || p.mode == MOT_COST_NEG_IOU) ? sim : dist;
}

}  // namespace mot

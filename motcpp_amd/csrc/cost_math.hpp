// Pairwise box-cost arithmetic shared by the N x M cost kernel (cost_kernels.hip) and the assignment kernel's
// on-the-fly cost functor (lap_kernel.hip): one definition, so a cost recomputed inside the solver is bit-identical
// to the materialised matrix. Reference order of operations: include/motcpp/utils/iou.hpp:85-95,
// src/utils/matching.cpp:62-65,130-143, src/trackers/botsort.cpp:439-465.
#pragma once
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

// cost of one pair given its IoU. `emb_at()` is only evaluated when the appearance term can matter.
struct CostParams {
  int mode;
  float prox, app;
  int fuse;
  bool has_emb;    // an embedding-distance matrix exists
  bool const_emb;  // no features at all: the cosine distance is the constant 1
};
template <class EmbFn>
MOT_HD float cost_from_iou(const CostParams& p, float iou, float conf, EmbFn emb_at) {
  float d = 1.0f - iou;  // iou_distance
  if (p.mode == MOT_COST_BOTSORT) {
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
  // the four plain modes as selects on the (uniform) mode rather than branches: inside the assignment solver this runs
  // once per visited pair, where a taken scalar branch costs more than the two spare multiplies
  const float fused = 1.0f - (1.0f - d) * conf;  // fuse_score: 1 - (1 - d) * conf
  const float dist = (p.mode == MOT_COST_IOU_DIST_FUSE) ? fused : d;
  const float sim = (p.mode == MOT_COST_NEG_IOU) ? -iou : iou;
  return (p.mode == MOT_COST_IOU || p.mode == MOT_COST_NEG_IOU) ? sim : dist;
}

}  // namespace mot

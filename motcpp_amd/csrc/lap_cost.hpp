// On-the-fly IoU-family cost for the assignment kernel: the solver's row/column scans recompute cost(i,j) from
// the row box (held in registers for a whole row pass) and the column box (LDS), with exactly the arithmetic of
// the N x M cost kernel (cost_math.hpp) — so the assignment is solved without the matrix ever being written to or
// read from HBM. The BoT-SORT appearance term reads its (materialised) cosine distance only for the few pairs
// that overlap enough for it to matter.
#pragma once
#include "cost_math.hpp"
#include "mem.hpp"

namespace mot {

// Boxes staged for one problem: planes x1,y1,x2,y2,area of length ld each (+ per-column confidence).
template <int AS>
struct BoxPlanes {
  const float* p;
  int ld;
  MOT_DEV void load(int i, float b[4], float* area) const {
    b[0] = mem_load<AS>(p + i); b[1] = mem_load<AS>(p + ld + i); b[2] = mem_load<AS>(p + 2 * ld + i);
    b[3] = mem_load<AS>(p + 3 * ld + i);
    *area = mem_load<AS>(p + 4 * ld + i);
  }
};

// RPL = lane-owned real columns kept in registers (column j = t + k*T belongs to lane t, k < RPL); 0 = none.
// The solver's row passes only ever touch a lane's own columns, so with RPL > 0 a pass reads no box from memory at
// all: the row box sits in registers for the pass, the column boxes for the whole solve. `cols` (memory) still backs the
// rare arbitrary-column accesses (general path search, result read-out).
// ROWS = address space of the staged row boxes (LDS or global scratch); column boxes, confidences and the appearance
// distances are always global.
// GENERAL = the similarity may be any mot_assoc measure (cost_math.hpp::assoc_pair: fp64 atan, sqrt, ...). Inlined into
// every unrolled evaluation site that costs ~100 VGPRs, i.e. half the resident wavefronts — so the hot variants are
// compiled for plain IoU only and tasks with another measure run the GENERAL variants.
// PLAIN = no task of the launch uses MOT_COST_BOTSORT: the gated appearance term is compiled out as well.
template <int RPL, int ROWS = kMemAny, bool GENERAL = false, bool PLAIN = false>
struct IouCostT {
  static constexpr int kRPL = RPL;
  // plain IoU-family costs: a pair whose boxes do not intersect costs the same whatever the row is (cost_from_iou of +0),
  // which is what lets the solver's first phase look at the intersecting pairs only (lap_core.hpp, "sparse column minima")
  // ... offered only while the row boxes are not in global scratch: the candidate rows are gathered box by box, which
  // pays from LDS (C2-sized problems, +7 %) and loses to the streaming sweep from global memory (north-star size, -4 %)
  static constexpr bool kPlain = PLAIN && !GENERAL && ROWS != kMemGlobal;
  BoxPlanes<ROWS> rows;
  BoxPlanes<kMemGlobal> cols;
  const float* conf;  // [nc] or nullptr
  CostParams prm;
  const float* emb;   // nr x nc cosine distances (global memory) or nullptr
  int lde;
  struct Owned { float b[4], area, conf; };
  Owned own[RPL > 0 ? RPL : 1];
  MOT_DEV void load_owned(int t, int T, int nc) {
#pragma unroll
    for (int k = 0; k < RPL; ++k) {
      const int j = t + k * T;
      Owned o{{0.f, 0.f, 0.f, 0.f}, 0.f, 0.f};
      if (j < nc) { cols.load(j, o.b, &o.area); o.conf = conf ? gld(conf, j) : 0.0f; }
      own[k] = o;
    }
  }
  struct Row {  // plain data: the row's box
    int i;
    float a[4], area;
  };
  MOT_DEV float eval_f(const Row& r, const float b[4], float barea, float cf, int j) const {
    float iou;
    if constexpr (GENERAL) iou = assoc_pair(prm.assoc, prm.frame_diag, r.a, r.area, b, barea);
    else iou = iou_pair(r.a, r.area, b, barea);
    const float* e = emb;
    const size_t off = static_cast<size_t>(r.i) * lde + j;
    return cost_from_iou<!PLAIN>(prm, iou, cf, [&]() { return gld(e, off); });
  }
  MOT_DEV double eval(const Row& r, const float b[4], float barea, float cf, int j) const {
    return static_cast<double>(eval_f(r, b, barea, cf, j));
  }
  MOT_DEV float at_owned_f(const Row& r, int k, int j) const {  // the cost as the float it is (phase 1 works in float)
    const Owned& o = own[k];
    return eval_f(r, o.b, o.area, o.conf, j);
  }
  MOT_DEV double at(const Row& r, int j) const {
    float b[4], barea;
    cols.load(j, b, &barea);
    return eval(r, b, barea, conf ? gld(conf, j) : 0.0f, j);
  }
  MOT_DEV double at_owned(const Row& r, int k, int j) const {  // k must be a compile-time constant after unrolling
    const Owned& o = own[k];
    return eval(r, o.b, o.area, o.conf, j);
  }
  MOT_DEV float row_x1(int i) const { return mem_load<ROWS>(rows.p + i); }
  MOT_DEV float row_x2(int i) const { return mem_load<ROWS>(rows.p + 2 * rows.ld + i); }
  MOT_DEV bool owned_has_nan(int k) const {
    const Owned& o = own[k];
    return (o.b[0] != o.b[0]) || (o.b[1] != o.b[1]) || (o.b[2] != o.b[2]) || (o.b[3] != o.b[3]);
  }
  MOT_DEV float owned_x1(int k) const { return own[k].b[0]; }
  MOT_DEV float owned_x2(int k) const { return own[k].b[2]; }
  // cost of owned column k against any row it does not intersect: iou_pair gives +0 there (inter == 0)
  MOT_DEV float zero_cost_f(int k) const {
    return cost_from_iou<!PLAIN>(prm, 0.0f, own[k].conf, []() { return 0.0f; });
  }
  MOT_DEV Row row(int i) const {
    Row r;
    r.i = i;
    rows.load(i, r.a, &r.area);
    return r;
  }
  MOT_DEV double at(int i, int j) const { return at(row(i), j); }
};
using IouCost = IouCostT<0>;

}  // namespace mot

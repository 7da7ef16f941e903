// On-the-fly IoU-family cost for the assignment kernel: the solver's row/column scans recompute cost(i,j) from
// the row box (held in registers for a whole row pass) and the column box (LDS), with exactly the arithmetic of
// the N x M cost kernel (cost_math.hpp) — so the assignment is solved without the matrix ever being written to or
// read from HBM. The BoT-SORT appearance term reads its (materialised) cosine distance only for the few pairs
// that overlap enough for it to matter.
#pragma once
#include "cost_math.hpp"

namespace mot {

// Boxes staged for one problem: planes x1,y1,x2,y2,area of length ld each (+ per-column confidence).
struct BoxPlanes {
  const float* p;
  int ld;
  MOT_DEV void load(int i, float b[4], float* area) const {
    b[0] = p[i]; b[1] = p[ld + i]; b[2] = p[2 * ld + i]; b[3] = p[3 * ld + i];
    *area = p[4 * ld + i];
  }
};

// RPL = lane-owned real columns kept in registers (column j = t + k*T belongs to lane t, k < RPL); 0 = none.
// The solver's row passes only ever touch a lane's own columns, so with RPL > 0 a pass reads no box from memory at
// all: the row box sits in registers for the pass, the column boxes for the whole solve. `cols` (memory) still backs the
// rare arbitrary-column accesses (general path search, result read-out).
template <int RPL>
struct IouCostT {
  static constexpr int kRPL = RPL;
  BoxPlanes rows, cols;
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
      if (j < nc) { cols.load(j, o.b, &o.area); o.conf = conf ? conf[j] : 0.0f; }
      own[k] = o;
    }
  }
  struct Row {
    const IouCostT* c;
    int i;
    float a[4], area;
    MOT_DEV double eval(const float b[4], float barea, float cf, int j) const {
      const float iou = iou_pair(a, area, b, barea);
      const float* e = c->emb;
      const size_t off = static_cast<size_t>(i) * c->lde + j;
      return static_cast<double>(cost_from_iou(c->prm, iou, cf, [&]() { return e[off]; }));
    }
    MOT_DEV double at(int j) const {
      float b[4], barea;
      c->cols.load(j, b, &barea);
      return eval(b, barea, c->conf ? c->conf[j] : 0.0f, j);
    }
    MOT_DEV double at_owned(int k, int j) const {  // k must be a compile-time constant after unrolling
      const Owned& o = c->own[k];
      return eval(o.b, o.area, o.conf, j);
    }
  };
  MOT_DEV Row row(int i) const {
    Row r;
    r.c = this; r.i = i;
    rows.load(i, r.a, &r.area);
    return r;
  }
  MOT_DEV double at(int i, int j) const { return row(i).at(j); }
};
using IouCost = IouCostT<0>;

}  // namespace mot

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

struct IouCost {
  BoxPlanes rows, cols;
  const float* conf;  // [nc] or nullptr
  CostParams prm;
  const float* emb;   // nr x nc cosine distances (global memory) or nullptr
  int lde;
  struct Row {
    const IouCost* c;
    int i;
    float a[4], area;
    MOT_DEV double at(int j) const {
      float b[4], barea;
      c->cols.load(j, b, &barea);
      const float iou = iou_pair(a, area, b, barea);
      const float cf = c->conf ? c->conf[j] : 0.0f;
      const float* e = c->emb;
      const size_t off = static_cast<size_t>(i) * c->lde + j;
      return static_cast<double>(cost_from_iou(c->prm, iou, cf, [&]() { return e[off]; }));
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

}  // namespace mot

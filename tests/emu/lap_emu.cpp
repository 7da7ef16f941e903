// TEST HARNESS ONLY — runs motcpp_amd/csrc/lap_core.hpp (the exact algorithm the gfx950 kernel
// executes) on T host threads so its decisions can be compared with the oracle without a GPU.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "emu_group.hpp"
#include "../../motcpp_amd/csrc/lap_core.hpp"
#include "../../motcpp_amd/csrc/lap_cost.hpp"
#include "../../motcpp_amd/csrc/lap_sparse.hpp"

// rowlists != 0: with the optional per-row lists of the entries below thresh/2 (mot_lap_task.rowlist) and the fast scratch
// the kernel keeps in LDS — the parallel scan steps and sparse real-row sweeps of the shortest-path search
extern "C" int emu_lap_rl(const float* cost, int nr, int nc, int ld, float thresh, int T, int rowlists, int* x, int* y) {
  using namespace mot;
  const int n = nr + nc;
  std::vector<char> mem(lap_work_bytes(n) + 64), rl(lap_rowlist_bytes(nr) + 64);
  std::vector<int> fsw(kFsWsInts);
  LapWork W = lap_carve(mem.data(), n);
  if (rowlists) { lap_carve_rowlist(W, rl.data(), nr); W.fsw.p = fsw.data(); }
  long long cyc[16] = {0};
  W.cyc = cyc;
  const MatrixCost C{cost, ld};
  const LapDims P{nr, nc, static_cast<double>(thresh) / 2.0};
  EmuShared sh(T);
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t]() { EmuGroup g(&sh, t); lap_solve(g, C, P, W); });
  for (auto& t : th) t.join();
  for (int i = 0; i < nr; ++i) x[i] = (W.x[i] >= nc) ? -1 : W.x[i];
  for (int j = 0; j < nc; ++j) y[j] = (W.y[j] >= nr) ? -1 : W.y[j];
  if (std::getenv("MOT_EMU_LAP_STATS"))
    std::fprintf(stderr, "emu_lap %dx%d T %d: paths %lld finds %lld | steps %lld members %lld real %lld events %lld | one-at-a-time %lld refused %lld lists %lld\n", nr, nc, T,
                 cyc[6], cyc[14], cyc[8], cyc[9], cyc[10], cyc[11], cyc[12], cyc[13], cyc[15]);
  return 0;
}
// the same with the state the wide matrix launches keep in LDS since round 5: 16-bit y / cols / inv, byte-sized list lengths, the matched costs
extern "C" int emu_lap_rl16(const float* cost, int nr, int nc, int ld, float thresh, int T, int* x, int* y) {
  using namespace mot;
  using Work16 = LapWorkT<kMemAny, kMemAny, kMemAny, kMemAny, kMemAny, kMemAny, short, kMemAny, kMemAny>;
  const int n = nr + nc;
  std::vector<char> mem(lap_work_bytes(n) + 64), rl(lap_rowlist_bytes(nr) + 64);
  std::vector<int> fsw(kFsWsInts);
  std::vector<unsigned char> rn(nr + 1);
  std::vector<float> yc(nc + 1);
  Work16 W;
  lap_carve_hot(W, mem.data(), n);
  lap_carve_cold(W, mem.data() + ((lap_hot_bytes(n) + 7) & ~size_t(7)), n);
  lap_carve_rowlist(W, rl.data(), nr);
  W.fsw.p = fsw.data();
  W.lst16.p = reinterpret_cast<unsigned short*>(fsw.data());  // as in the kernel: over the fast scratch's tables
  W.rl_n.p = rn.data();
  W.ycost.p = yc.data();
  long long cyc[16] = {0};
  W.cyc = cyc;
  const MatrixCost C{cost, ld};
  const LapDims P{nr, nc, static_cast<double>(thresh) / 2.0};
  EmuShared sh(T);
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t]() { EmuGroup g(&sh, t); lap_solve(g, C, P, W); });
  for (auto& t : th) t.join();
  for (int i = 0; i < nr; ++i) x[i] = (W.x[i] >= nc) ? -1 : W.x[i];
  for (int j = 0; j < nc; ++j) y[j] = (W.y[j] >= nr) ? -1 : static_cast<int>(W.y[j]);
  if (std::getenv("MOT_EMU_LAP_STATS"))
    std::fprintf(stderr, "emu_lap16 %dx%d T %d: paths %lld finds %lld | steps %lld members %lld real %lld events %lld | one-at-a-time %lld conflict-steps %lld lists %lld\n", nr, nc, T,
                 cyc[6], cyc[14], cyc[8], cyc[9], cyc[10], cyc[11], cyc[12], cyc[13], cyc[15]);
  // the matched costs must be current at the end: every real pair's entry is the matrix element
  for (int j = 0; j < nc; ++j) {
    const int i = W.y[j];
    if (i >= 0 && i < nr && cyc[15] && yc[j] != cost[static_cast<size_t>(i) * ld + j]) return -7;
  }
  return 0;
}
extern "C" int emu_lap(const float* cost, int nr, int nc, int ld, float thresh, int T, int* x, int* y) {
  return emu_lap_rl(cost, nr, nc, ld, thresh, T, 0, x, y);
}

// same solver with the on-the-fly IoU-family cost functor (row boxes a: nr x 4, column boxes b: nc x 4, row-major);
// rpl > 0 additionally exercises the lane-owned register cache of column boxes (unrolled path + leftover columns)
template <int RPL, bool PLAIN = false>
static int run_iou(const float* a, int nr, const float* b, int nc, const float* conf, int mode, float thresh, int T, int* x, int* y) {
  using namespace mot;
  const int n = nr + nc;
  std::vector<char> mem(lap_work_bytes(n) + 64);
  LapWork W = lap_carve(mem.data(), n);
  std::vector<float> rp(5 * nr), cp(5 * nc);
  for (int i = 0; i < nr; ++i) { for (int k = 0; k < 4; ++k) rp[k * nr + i] = a[i * 4 + k]; rp[4 * nr + i] = (a[i * 4 + 2] - a[i * 4]) * (a[i * 4 + 3] - a[i * 4 + 1]); }
  for (int j = 0; j < nc; ++j) { for (int k = 0; k < 4; ++k) cp[k * nc + j] = b[j * 4 + k]; cp[4 * nc + j] = (b[j * 4 + 2] - b[j * 4]) * (b[j * 4 + 3] - b[j * 4 + 1]); }
  const LapDims P{nr, nc, static_cast<double>(thresh) / 2.0};
  EmuShared sh(T);
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t]() {
      IouCostT<RPL, kMemAny, false, PLAIN> C;  // per lane, like the kernel: the owned-column cache is lane-private
      C.rows = BoxPlanes<kMemAny>{rp.data(), nr};
      C.cols = BoxPlanes<kMemGlobal>{cp.data(), nc};
      C.conf = conf;
      C.prm = CostParams{mode, 0.f, 0.f, 0, false, false};
      C.emb = nullptr; C.lde = 0;
      C.load_owned(t, T, nc);
      EmuGroup g(&sh, t);
      lap_solve(g, C, P, W);
    });
  for (auto& t : th) t.join();
  for (int i = 0; i < nr; ++i) x[i] = (W.x[i] >= nc) ? -1 : W.x[i];
  for (int j = 0; j < nc; ++j) y[j] = (W.y[j] >= nr) ? -1 : W.y[j];
  return 0;
}
extern "C" long emu_void_real_sweeps(int reset) { long& c = mot::lap_dbg_void_real(); const long v = c; if (reset) c = 0; return v; }
extern "C" int emu_lap_iou(const float* a, int nr, const float* b, int nc, const float* conf, int mode, float thresh, int T,
                           int rpl, int* x, int* y) {
  if (rpl == 104) return run_iou<4, true>(a, nr, b, nc, conf, mode, thresh, T, x, y);  // 100 + rpl: the plain-cost variants
  if (rpl == 108) return run_iou<8, true>(a, nr, b, nc, conf, mode, thresh, T, x, y);
  if (rpl == 2) return run_iou<2>(a, nr, b, nc, conf, mode, thresh, T, x, y);   // (the four-wavefront launches behind the fast path: two columns per lane)
  if (rpl == 102) return run_iou<2, true>(a, nr, b, nc, conf, mode, thresh, T, x, y);
  if (rpl == 4) return run_iou<4>(a, nr, b, nc, conf, mode, thresh, T, x, y);
  if (rpl == 8) return run_iou<8>(a, nr, b, nc, conf, mode, thresh, T, x, y);
  return run_iou<0>(a, nr, b, nc, conf, mode, thresh, T, x, y);
}

// ---- sparse fast path (lap_sparse.hpp) on host threads: returns 1 when it certified the unique optimum (x, y filled),
// 0 when it hands the problem to the exact emulation. boxes == nullptr: matrix source.
static int run_sparse(const float* cost, int ld, const float* a, const float* b, const float* conf, int mode, int nr, int nc,
                      float thresh, int T, int* x, int* y, double* mincost) {
  using namespace mot;
  using W = SparseWorkT<kMemAny>;
  const int ecap = kSpK * nc + 16;
  std::vector<char> hotv(sparse_hot_bytes(nr, nc, ecap) + 64), cold(sparse_cold_bytes(nr, nc) + 64);
  char* hot = hotv.data() + ((16 - (reinterpret_cast<size_t>(hotv.data()) & 15)) & 15);  // SpBox is 16-byte aligned
  W w;
  sparse_carve_hot(w, hot, nr, nc, ecap);
  sparse_carve_cold(w, cold.data(), nr, nc);
  std::vector<float> ap(4 * (nr > 0 ? nr : 1)), bp(4 * (nc > 0 ? nc : 1));  // planes [4][n]
  if (a) {
    for (int i = 0; i < nr; ++i) for (int k = 0; k < 4; ++k) ap[k * nr + i] = a[i * 4 + k];
    for (int j = 0; j < nc; ++j) for (int k = 0; k < 4; ++k) bp[k * nc + j] = b[j * 4 + k];
  }
  EmuShared sh(T);
  std::vector<int> res(T, 0);
  std::vector<double> mins(T, 0.0);
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t]() {
      EmuGroup g(&sh, t);
      SparseEnum e;
      if (a) {
        IouCostT<0, kMemAny, false, false> C;
        C.prm = CostParams{mode, 0.f, 0.f, 0, false, false};
        C.emb = nullptr; C.lde = 0; C.conf = nullptr;
        using Row = typename IouCostT<0, kMemAny, false, false>::Row;
        auto eval = [&](int i, const float ra[4], float raa, const float cb[4], float cba, float cf, int j) {
          Row r; r.i = i; for (int k = 0; k < 4; ++k) r.a[k] = ra[k]; r.area = raa;
          return C.eval_f(r, cb, cba, cf, j);
        };
        auto zc = [&](float cf) { return cost_from_iou<true>(C.prm, 0.0f, cf, []() { return 0.0f; }); };
        auto min_iou = [=](float cf) {  // as lap_sparse_kernel.hip does for the plain modes
          float need = 0.0f;
          if (mode == MOT_COST_IOU_DIST) need = 1.0f - thresh;
          else if (mode == MOT_COST_NEG_IOU) need = -thresh;
          else if (mode == MOT_COST_IOU_DIST_FUSE) need = (cf > 0.0f) ? (1.0f - thresh) / cf : 2.0f;
          if (!(need > 1.0e-3f)) return 0.0f;
          return (need < 1.0f) ? 0.99f * need : 0.99f;
        };
        e = sparse_enumerate_boxes(g, w, nr, nc, SparseBoxes{ap.data(), nr, nullptr}, SparseBoxes{bp.data(), nc, nullptr}, conf, nullptr,
                                   thresh, eval, zc, min_iou);
      } else {
        e = sparse_enumerate_matrix(g, w, nr, nc, cost, ld, thresh);
      }
      mins[t] = e.mincost;
      res[t] = (e.ok == 1) ? sparse_solve(g, w, nr, nc, thresh) : ((e.ok == 0 || e.ok <= -12) ? -1 : e.ok);
    });
  for (auto& t : th) t.join();
  for (int t = 1; t < T; ++t) if (res[t] != res[0]) return -99;  // must be uniform
  if (mincost) *mincost = mins[0];
  if (res[0] == 1) {
    for (int i = 0; i < nr; ++i) x[i] = w.x[i];
    for (int j = 0; j < nc; ++j) y[j] = w.y[j];
  }
  return res[0];
}
extern "C" int emu_sparse_matrix(const float* cost, int nr, int nc, int ld, float thresh, int T, int* x, int* y, double* mincost) {
  return run_sparse(cost, ld, nullptr, nullptr, nullptr, 0, nr, nc, thresh, T, x, y, mincost);
}
extern "C" int emu_sparse_boxes(const float* a, int nr, const float* b, int nc, const float* conf, int mode, float thresh, int T,
                                int* x, int* y, double* mincost) {
  return run_sparse(nullptr, 0, a, b, conf, mode, nr, nc, thresh, T, x, y, mincost);
}

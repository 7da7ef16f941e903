// TEST HARNESS ONLY — runs motcpp_amd/csrc/lap_core.hpp (the exact algorithm the gfx950 kernel
// executes) on T host threads so its decisions can be compared with the oracle without a GPU.
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "emu_group.hpp"
#include "../../motcpp_amd/csrc/lap_core.hpp"
#include "../../motcpp_amd/csrc/lap_cost.hpp"

extern "C" int emu_lap(const float* cost, int nr, int nc, int ld, float thresh, int T, int* x, int* y) {
  using namespace mot;
  const int n = nr + nc;
  std::vector<char> mem(lap_work_bytes(n) + 64);
  LapWork W = lap_carve(mem.data(), n);
  const MatrixCost C{cost, ld};
  const LapDims P{nr, nc, static_cast<double>(thresh) / 2.0};
  EmuShared sh(T);
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t]() { EmuGroup g(&sh, t); lap_solve(g, C, P, W); });
  for (auto& t : th) t.join();
  for (int i = 0; i < nr; ++i) x[i] = (W.x[i] >= nc) ? -1 : W.x[i];
  for (int j = 0; j < nc; ++j) y[j] = (W.y[j] >= nr) ? -1 : W.y[j];
  return 0;
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
extern "C" int emu_lap_iou(const float* a, int nr, const float* b, int nc, const float* conf, int mode, float thresh, int T,
                           int rpl, int* x, int* y) {
  if (rpl == 104) return run_iou<4, true>(a, nr, b, nc, conf, mode, thresh, T, x, y);  // 100 + rpl: the plain-cost variants
  if (rpl == 108) return run_iou<8, true>(a, nr, b, nc, conf, mode, thresh, T, x, y);
  if (rpl == 4) return run_iou<4>(a, nr, b, nc, conf, mode, thresh, T, x, y);
  if (rpl == 8) return run_iou<8>(a, nr, b, nc, conf, mode, thresh, T, x, y);
  return run_iou<0>(a, nr, b, nc, conf, mode, thresh, T, x, y);
}

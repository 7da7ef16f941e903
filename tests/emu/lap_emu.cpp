// TEST HARNESS ONLY — runs motcpp_amd/csrc/lap_core.hpp (the exact algorithm the gfx950 kernel
// executes) on T host threads so its decisions can be compared with the oracle without a GPU.
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "emu_group.hpp"
#include "../../motcpp_amd/csrc/lap_core.hpp"

extern "C" int emu_lap(const float* cost, int nr, int nc, int ld, float thresh, int T, int* x, int* y) {
  using namespace mot;
  const int n = nr + nc;
  std::vector<char> mem(lap_work_bytes(n) + 64);
  LapWork W = lap_carve(mem.data(), n);
  LapProblem P{cost, ld, nr, nc, static_cast<double>(thresh) / 2.0};
  EmuShared sh(T);
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t]() { EmuGroup g(&sh, t); lap_solve(g, P, W); });
  for (auto& t : th) t.join();
  for (int i = 0; i < nr; ++i) x[i] = (W.x[i] >= nc) ? -1 : W.x[i];
  for (int j = 0; j < nc; ++j) y[j] = (W.y[j] >= nr) ? -1 : W.y[j];
  return 0;
}

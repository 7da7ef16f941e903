// Host-only check of the blocking worker team (motcpp_amd/csrc/host/team.hpp): static contiguous split, worker ids,
// reuse across many phases, more workers than items, pinning is best-effort.
#include <atomic>
#include <cstdio>
#include <vector>

#include "team.hpp"

using motcpp::rt::Team;

int main() {
  int fails = 0;
  auto check = [&](bool ok, const char* what) { if (!ok) { std::printf("FAIL %s\n", what); ++fails; } };
  for (int threads : {1, 2, 7, 16}) {
    Team team(threads);
    check(team.size() == threads, "size");
    for (int count : {0, 1, 5, 64, 1000}) {
      std::vector<int> owner(count, -1);
      std::atomic<int> calls{0};
      team.parallel_for(count, [&](int i) { owner[i] = Team::worker_id(); ++calls; });
      check(calls == count, "every item exactly once");
      bool contiguous = true, in_range = true;
      for (int i = 0; i < count; ++i) {
        if (owner[i] < 0 || owner[i] >= threads) in_range = false;
        if (i > 0 && owner[i] < owner[i - 1]) contiguous = false;  // static split: worker ids never decrease along the range
      }
      check(in_range && contiguous, "static contiguous split");
      if (count >= threads && threads > 1 && count > 1) check(owner[0] == 0 && owner[count - 1] == threads - 1, "all workers take part");
    }
    // the same item always goes to the same worker (cache / malloc-arena locality of a stream)
    std::vector<int> a(257), b(257);
    team.parallel_for(257, [&](int i) { a[i] = Team::worker_id(); });
    for (int rep = 0; rep < 50; ++rep) team.parallel_for(257, [&](int i) { b[i] = Team::worker_id(); });
    check(a == b, "stable item -> worker mapping");
    team.pin(0);
    std::atomic<int> after{0};
    team.parallel_for(100, [&](int) { ++after; });
    check(after == 100, "works after pinning");
  }
  check(Team::worker_id() == 0, "caller is worker 0 outside a team");
  std::printf(fails ? "team FAILED\n" : "team ok\n");
  return fails ? 1 : 0;
}

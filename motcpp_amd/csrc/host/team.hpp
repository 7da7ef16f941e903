// Host worker team for the per-stream lifecycle work of a batch. A persistent set of threads that BLOCK between
// parallel phases: a frame alternates short bursts of host work (every stream advances its stage machine) with waits
// for the GPU, and on a box with a CPU-time quota (cgroup cpu.max) workers that spin through those waits — what an
// OpenMP runtime does by default — eat the quota the bursts need. The split of a loop is static and contiguous, so a
// stream is always stepped by the same worker: its track tables stay in that core's cache and its heap blocks in that
// thread's malloc arena.
#pragma once
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace motcpp::rt {

class Team {
 public:
  explicit Team(int threads);
  ~Team();
  Team(const Team&) = delete;
  Team& operator=(const Team&) = delete;
  int size() const { return n_; }
  // fn(i) for every i in [0, count); worker w gets the contiguous slice [count*w/n, count*(w+1)/n). The caller is
  // worker 0. fn must not throw.
  void parallel_for(int count, const std::function<void(int)>& fn);
  // pins worker w to the (first_cpu + w)-th CPU of the process's allowed set (best effort)
  void pin(int first_cpu);
  // index of the calling thread inside its team (0 for the driving thread and for threads outside any team)
  static int worker_id();

 private:
  void worker_main(int id);
  void run_slice(int id);
  int n_;
  std::vector<std::thread> thr_;
  std::mutex mu_;
  std::condition_variable cv_start_, cv_done_;
  unsigned long gen_ = 0;
  int pending_ = 0;
  bool stop_ = false;
  const std::function<void(int)>* fn_ = nullptr;
  int count_ = 0;
};

}  // namespace motcpp::rt

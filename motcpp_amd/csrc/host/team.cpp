#include "team.hpp"

#include <sched.h>
#include <unistd.h>

namespace motcpp::rt {

namespace {
thread_local int tls_worker = 0;

const std::vector<int>& allowed_cpus() {  // captured once, before any thread of ours narrows its own mask
  static const std::vector<int> cpus = [] {
    std::vector<int> v;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(getpid(), sizeof(set), &set) == 0)
      for (int c = 0; c < CPU_SETSIZE; ++c)
        if (CPU_ISSET(c, &set)) v.push_back(c);
    return v;
  }();
  return cpus;
}
}  // namespace

int Team::worker_id() { return tls_worker; }

Team::Team(int threads) : n_(threads < 1 ? 1 : threads) {
  allowed_cpus();
  for (int w = 1; w < n_; ++w) thr_.emplace_back([this, w] { worker_main(w); });
}

Team::~Team() {
  {
    std::lock_guard<std::mutex> g(mu_);
    stop_ = true;
  }
  cv_start_.notify_all();
  for (auto& t : thr_) t.join();
}

void Team::run_slice(int id) {
  const long c = count_;
  const int b = static_cast<int>(c * id / n_), e = static_cast<int>(c * (id + 1) / n_);
  for (int i = b; i < e; ++i) (*fn_)(i);
}

void Team::worker_main(int id) {
  tls_worker = id;
  unsigned long seen = 0;
  while (true) {
    {
      std::unique_lock<std::mutex> lk(mu_);
      cv_start_.wait(lk, [&] { return stop_ || gen_ != seen; });
      if (stop_) return;
      seen = gen_;
    }
    run_slice(id);
    {
      std::lock_guard<std::mutex> g(mu_);
      if (--pending_ == 0) cv_done_.notify_one();
    }
  }
}

void Team::parallel_for(int count, const std::function<void(int)>& fn) {
  if (count <= 0) return;
  if (n_ == 1 || count == 1) {
    for (int i = 0; i < count; ++i) fn(i);
    return;
  }
  {
    std::lock_guard<std::mutex> g(mu_);
    fn_ = &fn;
    count_ = count;
    pending_ = n_ - 1;
    ++gen_;
  }
  cv_start_.notify_all();
  const int saved = tls_worker;
  tls_worker = 0;
  run_slice(0);
  tls_worker = saved;
  std::unique_lock<std::mutex> lk(mu_);
  cv_done_.wait(lk, [&] { return pending_ == 0; });
}

void Team::pin(int first_cpu) {
  const std::vector<int>& cpus = allowed_cpus();
  if (cpus.empty() || first_cpu < 0) return;
  const std::function<void(int)> f = [&](int w) {
    cpu_set_t one;
    CPU_ZERO(&one);
    CPU_SET(cpus[static_cast<size_t>(first_cpu + w) % cpus.size()], &one);
    sched_setaffinity(0, sizeof(one), &one);
  };
  if (n_ == 1) { f(0); return; }
  parallel_for(n_, f);  // count == team size: worker w gets exactly item w
}

}  // namespace motcpp::rt

// Pooled tracker streams and the combiner that merges concurrent update() calls into one launch sequence (pool.hpp).
#include "pool.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <iterator>
#include <map>
#include <stdexcept>
#include <thread>

#include <linux/futex.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include "runtime.hpp"

namespace motcpp::rt {
namespace {

// ---- the four device lifecycles behind one table ------------------------------------------------------------------------
struct Ops {
  int (*create)(mot_ctx*, int S, int cap, int maxd, int emb, const float* p, void** out);
  void (*destroy)(void*);
  int (*enqueue)(void*, const mot_frame_in*, int rows_cap);
  int (*collect)(void*, mot_frame_view*);
  int (*reset_stream)(void*, int, int fresh);
  int (*move_stream)(void*, int, void*, int);
  int births_per_det;  // tracks one detection can add in a frame (OC-SORT's quirk Q4: a leftover detection can spawn twice)
};
const Ops kOps[4] = {
    {[](mot_ctx* c, int S, int cap, int d, int, const float* p, void** o) { return mot_bt_create(c, S, cap, d, p, reinterpret_cast<mot_bt_batch**>(o)); },
     [](void* b) { mot_bt_destroy(static_cast<mot_bt_batch*>(b)); },
     [](void* b, const mot_frame_in* in, int rc) { return mot_bt_enqueue_frame(static_cast<mot_bt_batch*>(b), in, rc); },
     [](void* b, mot_frame_view* v) { return mot_bt_collect_view(static_cast<mot_bt_batch*>(b), v); },
     [](void* b, int s, int fresh) { return mot_bt_reset_stream(static_cast<mot_bt_batch*>(b), s, fresh); },
     [](void* a, int s, void* b, int s2) { return mot_bt_move_stream(static_cast<mot_bt_batch*>(a), s, static_cast<mot_bt_batch*>(b), s2); }, 1},
    {[](mot_ctx* c, int S, int cap, int d, int, const float* p, void** o) { return mot_sort_create(c, S, cap, d, p, reinterpret_cast<mot_sort_batch**>(o)); },
     [](void* b) { mot_sort_destroy(static_cast<mot_sort_batch*>(b)); },
     [](void* b, const mot_frame_in* in, int rc) { return mot_sort_enqueue_frame(static_cast<mot_sort_batch*>(b), in, rc); },
     [](void* b, mot_frame_view* v) { return mot_sort_collect_view(static_cast<mot_sort_batch*>(b), v); },
     [](void* b, int s, int fresh) { return mot_sort_reset_stream(static_cast<mot_sort_batch*>(b), s, fresh); },
     [](void* a, int s, void* b, int s2) { return mot_sort_move_stream(static_cast<mot_sort_batch*>(a), s, static_cast<mot_sort_batch*>(b), s2); }, 1},
    {[](mot_ctx* c, int S, int cap, int d, int, const float* p, void** o) { return mot_oc_create(c, S, cap, d, p, reinterpret_cast<mot_oc_batch**>(o)); },
     [](void* b) { mot_oc_destroy(static_cast<mot_oc_batch*>(b)); },
     [](void* b, const mot_frame_in* in, int rc) { return mot_oc_enqueue_frame(static_cast<mot_oc_batch*>(b), in, rc); },
     [](void* b, mot_frame_view* v) { return mot_oc_collect_view(static_cast<mot_oc_batch*>(b), v); },
     [](void* b, int s, int fresh) { return mot_oc_reset_stream(static_cast<mot_oc_batch*>(b), s, fresh); },
     [](void* a, int s, void* b, int s2) { return mot_oc_move_stream(static_cast<mot_oc_batch*>(a), s, static_cast<mot_oc_batch*>(b), s2); }, 2},
    {[](mot_ctx* c, int S, int cap, int d, int e, const float* p, void** o) { return mot_bot_create(c, S, cap, d, e, p, reinterpret_cast<mot_bot_batch**>(o)); },
     [](void* b) { mot_bot_destroy(static_cast<mot_bot_batch*>(b)); },
     [](void* b, const mot_frame_in* in, int rc) { return mot_bot_enqueue_frame(static_cast<mot_bot_batch*>(b), in, rc); },
     [](void* b, mot_frame_view* v) { return mot_bot_collect_view(static_cast<mot_bot_batch*>(b), v); },
     [](void* b, int s, int fresh) { return mot_bot_reset_stream(static_cast<mot_bot_batch*>(b), s, fresh); },
     [](void* a, int s, void* b, int s2) { return mot_bot_move_stream(static_cast<mot_bot_batch*>(a), s, static_cast<mot_bot_batch*>(b), s2); }, 1},
};

constexpr int kLevels = 4;
constexpr int kLevelCap[kLevels] = {512, 2048, 8192, 32768};
constexpr int kLevelDets[kLevels] = {256, 1024, 4096, 16384};

long env_long(const char* name, long dflt) {
  const char* e = std::getenv(name);
  if (!e || !*e) return dflt;
  char* end = nullptr;
  const long v = std::strtol(e, &end, 10);
  return (end && *end == '\0') ? v : dflt;
}

std::mutex g_stats_mu;
PoolStats g_stats;

// diagnostics (MOTCPP_POOL_CPU_DEBUG=1): CPU time (CLOCK_THREAD_CPUTIME_ID) the calling threads spend inside update(), and the leaders inside lead() / run()
struct CpuDbg {
  std::atomic<long long> ns_update{0}, ns_lead{0}, ns_run{0}, ns_wait{0}, n_update{0}, ns_prepare{0}, ns_lock{0}, ns_stage{0}, ns_finish{0};
  bool on = std::getenv("MOTCPP_POOL_CPU_DEBUG") != nullptr;
  ~CpuDbg() {
    if (on && n_update.load() > 0)
      std::fprintf(stderr, "[pool cpu] updates %lld: CPU us per update %.1f (of which lead() %.1f, of which run() %.1f); CPU us per update inside the futex wait call %.1f\n",
                   n_update.load(), ns_update.load() / 1e3 / n_update.load(), ns_lead.load() / 1e3 / n_update.load(), ns_run.load() / 1e3 / n_update.load(),
                   ns_wait.load() / 1e3 / n_update.load());
    if (on && n_update.load() > 0)
      std::fprintf(stderr, "[pool cpu]   prepare %.1f  join: lock %.1f stage %.1f  finish %.1f\n", ns_prepare.load() / 1e3 / n_update.load(), ns_lock.load() / 1e3 / n_update.load(),
                   ns_stage.load() / 1e3 / n_update.load(), ns_finish.load() / 1e3 / n_update.load());
  }
} g_cpu;
long long thread_cpu_ns() { struct timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return static_cast<long long>(ts.tv_sec) * 1000000000ll + ts.tv_nsec; }

struct Request {
  const PooledFrame* in = nullptr;
  int s = -1;
  int n = 0, ld = 0;
  long long det_off = 0, emb_off = -1;
  const float* warp6 = nullptr;
  int pre = 0;  // 0 none, 1 the slot starts fresh (new object), 2 move the stream in from (from, from_s), 3 BaseTracker::reset()
  Segment* from = nullptr;
  int from_s = -1;
  bool also_reset = false;  // pre == 2 with a reset() pending: the moved stream is reset right behind the move
  // results (written by the round's leader before it publishes the round as completed)
  std::atomic<int> copy_state{0};  // 0 not staged yet, 1 being copied, 2 in the staging buffer
  std::string error;
  const float* rows = nullptr;
  int count = 0, alive = 0;
  uint64_t round = 0;  // the round that carried it (its parity names the page-locked table the rows sit in)
};

}  // namespace

// One device batch + its combiner.
class Segment {
 public:
  Segment(int device, int kind, int level, int nstreams, int emb_dim, const std::vector<float>& params)
      : kind_(kind), level_(level), S(nstreams), CAP(kLevelCap[level]), D(kLevelDets[level]), E(emb_dim), ops_(kOps[kind]) {
    if (mot_ctx_create(device, nullptr, &ctx_) != MOT_OK) throw Error("motcpp_amd: no usable gfx950 (MI355X) device " + std::to_string(device));
    if (ops_.create(ctx_, S, CAP, D, E, params.data(), &batch_) != MOT_OK) {
      const std::string why = mot_ctx_last_error(ctx_);
      mot_ctx_destroy(ctx_);
      throw Error("motcpp_amd: could not create a pooled device batch (" + std::to_string(S) + " streams of " + std::to_string(CAP) + " x " +
                  std::to_string(D) + "): " + why);
    }
    ldmax_ = (D + 3) & ~3;
    det_floats_ = static_cast<size_t>(S) * 6 * ldmax_;
    emb_floats_ = static_cast<size_t>(S) * D * (E > 0 ? E : 0);
    bool ok = mot_malloc(ctx_, det_floats_ * sizeof(float), reinterpret_cast<void**>(&d_dets_)) == MOT_OK;
    for (int k = 0; k < 2 && ok; ++k) ok = mot_host_alloc(ctx_, det_floats_ * sizeof(float), reinterpret_cast<void**>(&h_dets_[k])) == MOT_OK;
    if (E > 0) {
      ok = ok && mot_malloc(ctx_, emb_floats_ * sizeof(float), reinterpret_cast<void**>(&d_embs_)) == MOT_OK;
      for (int k = 0; k < 2 && ok; ++k) ok = mot_host_alloc(ctx_, emb_floats_ * sizeof(float), reinterpret_cast<void**>(&h_embs_[k])) == MOT_OK;
    }
    if (!ok) { release(); throw Error("motcpp_amd: out of memory for a pooled segment's staging buffers"); }
    free_.reserve(S);
    for (int s = S - 1; s >= 0; --s) free_.push_back(s);
    counts_.assign(S, -1); ld_.assign(S, 0); det_off_.assign(S, 0); emb_off_.assign(S, -1);
    warps_.assign(static_cast<size_t>(S) * 6, 0.f); has_warp_.assign(S, 0);
    seen_round_.assign(static_cast<size_t>(S), 0);
    for (Round& R : rounds_) { R.slots = std::vector<std::atomic<Request*>>(static_cast<size_t>(S)); for (auto& a : R.slots) a.store(nullptr, std::memory_order_relaxed); }
    window_us_ = env_long("MOTCPP_BATCH_WINDOW_US", 60);
    gap_us_ = env_long("MOTCPP_BATCH_GAP_US", 20);
  }
  ~Segment() { release(); }

  int acquire() {  // a free stream, -1: full (StreamPool's lock held)
    if (free_.empty()) return -1;
    const int s = free_.back();
    free_.pop_back();
    return s;
  }
  void give_back(int s) { free_.push_back(s); }
  int used() const { return S - static_cast<int>(free_.size()); }

  // Joins the open round with one frame of each of the k streams reqs[i]->s (all of this segment) and returns when the round has run.
  // Round 5: the FIRST caller of a round is its leader (no hand-over: it waits for the previous round to finish, closes its own and runs it);
  // everybody else sleeps on the round's futex word, which the leader bumps once when the results are in place — one system call wakes all of
  // them and none of them has a lock to take on the way out. (Round 4: a condition variable per request, all on the segment's mutex — the leader
  // made one notify call per caller and every woken caller queued for the mutex again: with 256 objects on 16 CPUs a round spent twice as
  // long handing out its results as running.)
  void join(Request* const* reqs, int k) {
    // Round 6: joining takes no lock. One atomic word holds the open round and the number of requests in it (round << 24 | joined): a
    // fetch-add puts the caller into whatever round is open and hands it its place there; the leader closes its round by exchanging the word
    // for (round + 1) << 24 — every fetch-add lands before the exchange (this round) or after it (the next). The staging space is reserved with
    // two more fetch-adds on the round's tops. (Round 5's mutex around the request list: with 256 objects woken together by the previous
    // round's completion, 116 us of CPU per update() went into that lock — 87 % of what an update() cost the host — and on the GPU boxes'
    // 16-CPU quota that is what made the cgroup throttle: p99 of update() 51 ms against a median of 0.9.)
    const long long jc0 = g_cpu.on ? thread_cpu_ns() : 0;
    const uint64_t old = gate_.fetch_add(static_cast<uint64_t>(k), std::memory_order_acq_rel);
    const uint64_t r = old >> kGateShift;
    const size_t idx0 = static_cast<size_t>(old & kGateMask);
    const bool first = idx0 == 0;
    {
      Round& R = rounds_[r & 1];
      if (idx0 + static_cast<size_t>(k) > R.slots.size()) std::abort();  // (a stream joins a round once: cannot happen)
      for (int i = 0; i < k; ++i) {
        Request& q = *reqs[i];
        const int n = q.in->n;
        q.n = n;
        q.ld = (n + 3) & ~3;
        if (q.ld < 4) q.ld = 4;
        q.det_off = static_cast<long long>(R.det_top.fetch_add(static_cast<size_t>(6) * q.ld, std::memory_order_relaxed));
        const bool with_emb = q.in->embs != nullptr && E > 0 && n > 0;
        q.emb_off = -1;
        if (with_emb) q.emb_off = static_cast<long long>(R.emb_top.fetch_add(static_cast<size_t>(n) * E, std::memory_order_relaxed));
        q.copy_state.store(0, std::memory_order_relaxed);
        R.slots[idx0 + i].store(&q, std::memory_order_release);  // published: offsets reserved
      }
    }
    // the caller's own copy into the round's page-locked staging, in parallel with the other callers' (a caller that loses the CPU
    // before it gets here is helped out by the round's leader: whoever flips copy_state first does the copy)
    const long long jc1 = g_cpu.on ? thread_cpu_ns() : 0;
    for (int i = 0; i < k; ++i) stage(*reqs[i], static_cast<int>(r & 1));
    if (g_cpu.on) { const long long jc2 = thread_cpu_ns(); g_cpu.ns_lock += jc1 - jc0; g_cpu.ns_stage += jc2 - jc1; }
    if (first) {
      const bool idle = completed_.load(std::memory_order_acquire) >= r;  // nothing in flight: peers in lockstep may be a few microseconds behind
      wait_completed(r, static_cast<int>((r + 1) & 1));                   // (round r - 1 carries the other parity)
      const long long c0 = g_cpu.on ? thread_cpu_ns() : 0;
      lead(r, idle);
      if (g_cpu.on) g_cpu.ns_lead += thread_cpu_ns() - c0;
    } else {
      const long long c0 = g_cpu.on ? thread_cpu_ns() : 0;
      wait_completed(r + 1, static_cast<int>(r & 1));
      if (g_cpu.on) g_cpu.ns_wait += thread_cpu_ns() - c0;
    }
  }
  // copies q's detections (and features) into the staging buffer of its round unless somebody else already does / did
  void stage(Request& q, int parity) {
    int expect = 0;
    if (!q.copy_state.compare_exchange_strong(expect, 1, std::memory_order_acq_rel)) return;
    const PooledFrame& f = *q.in;
    const int n = q.n;
    float* hd = h_dets_[parity] + q.det_off;
    for (int c = 0; c < 6; ++c)
      if (n > 0) std::memcpy(hd + static_cast<size_t>(c) * q.ld, f.dets + static_cast<size_t>(c) * f.ld, sizeof(float) * n);
    if (q.emb_off >= 0) {
      float* he = h_embs_[parity] + q.emb_off;
      if (f.embs_rowmajor) {
        for (int j = 0; j < n; ++j) std::memcpy(he + static_cast<size_t>(j) * E, f.embs + static_cast<size_t>(j) * f.emb_ld, sizeof(float) * E);
      } else {  // column-major n x E -> rows (in blocks of 32 rows so that both sides stay in cache)
        for (int j0 = 0; j0 < n; j0 += 32) {
          const int j1 = (j0 + 32 < n) ? j0 + 32 : n;
          for (int c = 0; c < E; ++c) {
            const float* col = f.embs + static_cast<size_t>(c) * f.emb_ld;
            for (int j = j0; j < j1; ++j) he[static_cast<size_t>(j) * E + c] = col[j];
          }
        }
      }
    }
    q.copy_state.store(2, std::memory_order_release);
  }
  // (after the caller has taken its rows: the round's page-locked table may be rewritten two rounds later)
  void rows_taken(int parity, int k = 1) { outstanding_[parity].fetch_sub(k, std::memory_order_acq_rel); }
  int kind() const { return kind_; }
  int level() const { return level_; }
  void* batch() const { return batch_; }
  // runs f(batch) between rounds, with this segment's context bound: what reads the batch from outside a round (PooledStream::dump) must not
  // overlap a leader that is enqueuing on it, nor leave another device current
  template <class F>
  auto quiesced(F f) {
    std::lock_guard<std::mutex> lk(run_mu_);
    check(mot_ctx_bind(ctx_), "mot_ctx_bind");
    return f(batch_);
  }
  mot_ctx* ctx() const { return ctx_; }
  const int kind_, level_;
  const int S, CAP, D, E;

 private:
  struct Round {
    std::atomic<size_t> det_top{0}, emb_top{0};
    std::vector<std::atomic<Request*>> slots;  // [S] the requests of the round, in joining order
  };
  static constexpr int kGateShift = 24;
  static constexpr uint64_t kGateMask = (uint64_t(1) << kGateShift) - 1;
  int joined_in(uint64_t r) const {  // requests in round r so far (0 once it is closed)
    const uint64_t g = gate_.load(std::memory_order_acquire);
    return (g >> kGateShift) == r ? static_cast<int>(g & kGateMask) : 0;
  }
  static void futex_wait(std::atomic<uint32_t>* w, uint32_t seen) {
    syscall(SYS_futex, reinterpret_cast<uint32_t*>(w), FUTEX_WAIT_PRIVATE, seen, nullptr, nullptr, 0);
  }
  static void futex_wake_all(std::atomic<uint32_t>* w) { syscall(SYS_futex, reinterpret_cast<uint32_t*>(w), FUTEX_WAKE_PRIVATE, 0x7fffffff, nullptr, nullptr, 0); }
  // returns once `want` rounds have completed; sleeps on the word of the round it is waiting for (parity `p`)
  void wait_completed(uint64_t want, int p) {
    for (;;) {  // (a lone object never sleeps here: it leads every round and the previous one is its own)
      const uint32_t seen = word_[p].load(std::memory_order_acquire);
      if (completed_.load(std::memory_order_acquire) >= want) return;
      futex_wait(&word_[p], seen);
    }
  }

  // The calling thread (the round's first caller) runs round r; the previous round has completed.
  void lead(uint64_t r, bool idle) {
    const int p = static_cast<int>(r & 1);
    using clk = std::chrono::steady_clock;
    auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    const auto t_a = clk::now();
    // Batching window, only when the leader found the GPU idle (had it to wait for the previous round, everybody who arrived meanwhile is in
    // already and closing at once is the group commit): objects stepped in lockstep come back within microseconds of each other — the round
    // stays open while they keep arriving (no arrival for `gap` microseconds closes it), until the streams of the last rounds are all in,
    // at most window_us_ + 40 us.
    // how many callers to expect: the streams that took part in one of the last two rounds (round 5 took the larger of the last two batches — a
    // population that once split into two alternating groups, each closing its round on the idle gap before the other arrived, then stayed
    // split: every object waited for two rounds per frame; seen with 16 objects once joining stopped queueing on a lock)
    int expected = 0;
    for (int s2 = 0; s2 < S; ++s2) expected += (seen_round_[s2] + 2 > r) ? 1 : 0;  // (seen_round_ = 1 + the stream's last round)
    if (idle && window_us_ > 0 && joined_in(r) < expected) {
      // (round 4 waited up to window + 2 us per expected stream: 570 us of idle GPU per round with 256 objects on 16 CPUs, whose threads take that
      // long to get a CPU each. Whoever is late simply rides the next round, which fills while this one runs.)
      const auto hard = t_a + std::chrono::microseconds(window_us_ + (expected < 20 ? 2 * expected : 40));
      const auto gap = std::chrono::microseconds(gap_us_);
      int seen = joined_in(r);
      auto last_arrival = t_a;
      for (;;) {
        std::this_thread::yield();
        const auto now = clk::now();
        const int j = joined_in(r);
        if (j >= expected || now >= hard) break;
        if (j != seen) { seen = j; last_arrival = now; }
        else if (now - last_arrival >= gap) break;
      }
    }
    //  From here on the round MUST complete whatever happens on this thread — an allocation that fails, an exception out of the
    // run — or every caller asleep on its futex word, and every later caller of the segment, would wait for good: the guard publishes the round
    // (with the error in every request that has none of its own) on every way out.
    std::vector<Request*> reqs;
    struct Publish {
      Segment* self; uint64_t r; int p; std::vector<Request*>* reqs; std::string err; bool failed = false;
      ~Publish() {
        for (Request* q : *reqs) {
          q->round = r;
          if (failed && q->error.empty()) { try { q->error = err.empty() ? std::string("motcpp_amd: the round's leader failed") : err; } catch (...) {} }
        }
        self->outstanding_[p].store(static_cast<int>(reqs->size()), std::memory_order_release);
        self->completed_.store(r + 1, std::memory_order_release);
        self->word_[p].fetch_add(1, std::memory_order_release);
        futex_wake_all(&self->word_[p]);  // this round's callers, and the leader of the next round if it is waiting already
      }
    } publish{this, r, p, &reqs, {}};
    size_t det_top = 0, emb_top = 0;
    {
      Round& R = rounds_[p];
      const uint64_t closed = gate_.exchange((r + 1) << kGateShift, std::memory_order_acq_rel);  // closed: later arrivals fill the other round
      const size_t count = static_cast<size_t>(closed & kGateMask);
      reqs.reserve(count);  // (the one allocation of the round; if it throws nobody has been taken out of the round yet... and the guard completes it)
      for (size_t i = 0; i < count; ++i) {
        Request* q;
        while ((q = R.slots[i].load(std::memory_order_acquire)) == nullptr) std::this_thread::yield();  // (between its fetch-add and its store)
        R.slots[i].store(nullptr, std::memory_order_relaxed);
        reqs.push_back(q);
      }
      det_top = R.det_top.exchange(0, std::memory_order_relaxed);
      emb_top = R.emb_top.exchange(0, std::memory_order_relaxed);
    }
    try {
      const auto t_b = clk::now();
      for (Request* q : reqs) stage(*q, p);  // the callers that have not got to their copy yet
      for (Request* q : reqs)                // copies in progress on their owners' threads
        spin_then_sleep([&] { return q->copy_state.load(std::memory_order_acquire) == 2; });
      spin_then_sleep([&] { return outstanding_[p].load(std::memory_order_acquire) == 0; });  // readers of the table two rounds back
      const auto t_c = clk::now();
      const long long rc0 = g_cpu.on ? thread_cpu_ns() : 0;
      run(reqs, p, det_top, emb_top);
      if (g_cpu.on) g_cpu.ns_run += thread_cpu_ns() - rc0;
      const auto t_d = clk::now();
      {
        std::lock_guard<std::mutex> g(g_stats_mu);
        g_stats.us_window += us(t_a, t_b); g_stats.us_gather += us(t_b, t_c); g_stats.us_run += us(t_c, t_d);
        g_stats.us_enqueue += last_enqueue_us_;
      }
      for (Request* q : reqs) seen_round_[q->s] = r + 1;
    } catch (const std::exception& e) {
      publish.failed = true;
      try { publish.err = e.what(); } catch (...) {}
    } catch (...) {
      publish.failed = true;
    }
  }
  // waits for a condition another thread is about to establish: a short spin (the common case: microseconds), then sleeps of 50 us — a
  // waiter that lost its peer to a failure does not burn a CPU for good
  template <class Cond>
  static void spin_then_sleep(Cond cond) {
    for (int i = 0; i < 200; ++i) {
      if (cond()) return;
      std::this_thread::yield();
    }
    while (!cond()) std::this_thread::sleep_for(std::chrono::microseconds(20));
  }

  std::mutex run_mu_;  // held while a round runs on the batch (uncontended except against quiesced())
  void run(const std::vector<Request*>& reqs, int parity, size_t det_top, size_t emb_top) {
    std::lock_guard<std::mutex> run_lock(run_mu_);
    check(mot_ctx_bind(ctx_), "mot_ctx_bind");
    std::fill(counts_.begin(), counts_.end(), -1);
    bool any_warp = false, any_emb = false;
    for (Request* q : reqs) {
      if (q->pre != 0 && std::getenv("MOTCPP_POOL_DEBUG")) std::fprintf(stderr, "[pool] round leader: stream %d pre %d (level %d)\n", q->s, q->pre, level_);
      if (q->pre == 1 || q->pre == 3) check(ops_.reset_stream(batch_, q->s, q->pre == 1 ? 1 : 0), "reset_stream");
      else if (q->pre == 2) {
        check(ops_.move_stream(q->from->batch_, q->from_s, batch_, q->s), "move_stream");
        if (q->also_reset) check(ops_.reset_stream(batch_, q->s, 0), "reset_stream");
        std::lock_guard<std::mutex> g(g_stats_mu);
        g_stats.moves += 1;
      }
      counts_[q->s] = q->n; ld_[q->s] = q->ld; det_off_[q->s] = q->det_off; emb_off_[q->s] = q->emb_off;
      any_emb = any_emb || q->emb_off >= 0;
      has_warp_[q->s] = q->warp6 ? 1 : 0;
      if (q->warp6) { std::memcpy(&warps_[static_cast<size_t>(q->s) * 6], q->warp6, sizeof(float) * 6); any_warp = true; }
    }
    // (tried in round 4: letting the kernels read one camera's few KB of detections in place from the page-locked staging buffer saves the
    // copy launch and costs more than that in the three kernels that read the raw detections over PCIe: bt_begin 8 -> 13 us,
    // bt_after_second 18 -> 30 us)
    if (det_top) check(mot_memcpy_h2d(ctx_, d_dets_, h_dets_[parity], det_top * sizeof(float)), "detections upload");
    if (any_emb && emb_top) check(mot_memcpy_h2d(ctx_, d_embs_, h_embs_[parity], emb_top * sizeof(float)), "embeddings upload");
    mot_frame_in in;
    std::memset(&in, 0, sizeof(in));
    in.d_dets = d_dets_; in.h_counts = counts_.data(); in.h_det_ld = ld_.data(); in.h_det_off = det_off_.data();
    if (any_emb) { in.d_embs = d_embs_; in.h_emb_off = emb_off_.data(); }
    if (any_warp) { in.h_warps6 = warps_.data(); in.h_has_warp = has_warp_.data(); }
    const long long rows_cap_ll = static_cast<long long>(S) * CAP;
    const int rows_cap = rows_cap_ll > (1 << 24) ? (1 << 24) : static_cast<int>(rows_cap_ll);
    const auto t_e0 = std::chrono::steady_clock::now();
    check(ops_.enqueue(batch_, &in, rows_cap), "enqueue_frame");
    last_enqueue_us_ = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_e0).count();
    mot_frame_view v;
    std::memset(&v, 0, sizeof(v));
    const int crc = ops_.collect(batch_, &v);
    // A stream that overflowed its level's capacities raises the batch's error word; the view is filled all the same and names the streams
    // (alive[s] = -(error code)): only THEIR callers see the error — the reference's exceptions are per tracker object (src/tracker.cpp:108-125)
    // — everybody else's rows are valid. Anything else (a HIP error, a full row table) is the whole round's.
    bool per_stream = false;
    if (crc == MOT_ERR_CAPACITY && v.alive && v.counts && v.rows) {
      long long total = 0;
      for (int s = 0; s < S; ++s) total += v.counts[s] > 0 ? v.counts[s] : 0;
      per_stream = total <= static_cast<long long>(rows_cap);
      if (per_stream) {  // (the stream in error need not be in this round: one that sits the frame out keeps raising the batch's word until it is reset)
        bool named = false;
        for (int s = 0; s < S && !named; ++s) named = v.alive[s] < 0;
        per_stream = named;
      }
    }
    if (!per_stream) check(crc, "collect_view");
    // stream s's rows start at the sum of the counts before it
    offs_.resize(static_cast<size_t>(S) + 1);
    int acc = 0;
    for (int s = 0; s < S; ++s) { offs_[s] = acc; acc += v.counts[s] > 0 ? v.counts[s] : 0; }
    for (Request* q : reqs) {
      q->count = v.counts[q->s];
      q->rows = v.rows + static_cast<size_t>(offs_[q->s]) * 8;
      q->alive = v.alive ? v.alive[q->s] : 0;
      if (q->alive < 0) {
        q->error = "motcpp_amd: this tracker's stream exceeded the capacities of its pooled level (" + std::to_string(CAP) + " tracks x " + std::to_string(D) +
                   " detections; device error " + std::to_string(-q->alive) + "): call reset() before using the object again";
        q->count = 0; q->alive = 0;
      }
    }
    std::lock_guard<std::mutex> g(g_stats_mu);
    g_stats.rounds += 1;
    g_stats.frames += static_cast<long>(reqs.size());
    if (static_cast<long>(reqs.size()) > g_stats.max_round) g_stats.max_round = static_cast<long>(reqs.size());
  }
  void check(int rc, const char* what) {
    if (rc != MOT_OK) throw Error(std::string("motcpp_amd: ") + what + " failed: " + mot_ctx_last_error(ctx_));
  }
  void release() {
    if (batch_) { ops_.destroy(batch_); batch_ = nullptr; }
    if (ctx_) {
      if (d_dets_) mot_free(ctx_, d_dets_);
      if (d_embs_) mot_free(ctx_, d_embs_);
      for (int k = 0; k < 2; ++k) { if (h_dets_[k]) mot_host_free(ctx_, h_dets_[k]); if (h_embs_[k]) mot_host_free(ctx_, h_embs_[k]); }
      mot_ctx_destroy(ctx_);
      ctx_ = nullptr;
    }
  }

  const Ops& ops_;
  mot_ctx* ctx_ = nullptr;
  void* batch_ = nullptr;
  int ldmax_ = 0;
  size_t det_floats_ = 0, emb_floats_ = 0;
  float* d_dets_ = nullptr; float* h_dets_[2] = {nullptr, nullptr};
  float* d_embs_ = nullptr; float* h_embs_[2] = {nullptr, nullptr};
  std::vector<int> free_;
  // combiner
  Round rounds_[2];
  std::atomic<uint64_t> gate_{0};          // open round << 24 | requests joined (see join())
  std::atomic<uint64_t> completed_{0};     // rounds whose results are delivered (rounds run strictly one after the other)
  std::atomic<uint32_t> word_[2] = {{0}, {0}};  // futex words, by round parity: bumped when a round of that parity completes
  std::atomic<int> outstanding_[2] = {{0}, {0}};  // callers that have not copied their rows out of that parity's table yet
  std::vector<uint64_t> seen_round_;       // [S] 1 + the last round a stream took part in, 0: never (leader only)
  long window_us_ = 60, gap_us_ = 20;
  double last_enqueue_us_ = 0.0;
  // leader's scratch
  std::vector<int> counts_, ld_, offs_;
  std::vector<long long> det_off_, emb_off_;
  std::vector<float> warps_;
  std::vector<unsigned char> has_warp_;
};

// Segments of one (device, kind, parameters, emb_dim), by level.
class StreamPool {
 public:
  StreamPool(int device, int kind, std::vector<float> params, int emb_dim) : device_(device), kind_(kind), params_(std::move(params)), emb_dim_(emb_dim) {}
  static std::shared_ptr<StreamPool> get(int device, int kind, const std::vector<float>& params, int emb_dim) {
    static std::mutex m;
    static std::map<std::vector<float>, std::weak_ptr<StreamPool>> pools;
    std::vector<float> key = params;
    key.push_back(static_cast<float>(device)); key.push_back(static_cast<float>(kind)); key.push_back(static_cast<float>(emb_dim));
    std::lock_guard<std::mutex> g(m);
    for (auto it = pools.begin(); it != pools.end();) it = it->second.expired() ? pools.erase(it) : std::next(it);  // (pools that went away)
    auto sp = pools[key].lock();
    if (!sp) { sp = std::make_shared<StreamPool>(device, kind, params, emb_dim); pools[key] = sp; }
    return sp;
  }
  // a free stream on `level` (a new segment when the existing ones are full)
  void acquire(int level, Segment** seg, int* s) {
    std::lock_guard<std::mutex> g(mu_);
    for (auto& sg : segs_[level]) {
      const int k = sg->acquire();
      if (k >= 0) { *seg = sg.get(); *s = k; return; }
    }
    segs_[level].push_back(std::make_unique<Segment>(device_, kind_, level, segment_streams(level), emb_dim_, params_));
    *seg = segs_[level].back().get();
    *s = (*seg)->acquire();
  }
  void release(Segment* seg, int s) {
    std::lock_guard<std::mutex> g(mu_);
    seg->give_back(s);
  }
  int emb_dim() const { return emb_dim_; }

 private:
  // streams per segment: as many as fit a memory budget (MOTCPP_POOL_SEGMENT_MB, default 1536), at most MOTCPP_POOL_SEGMENT_STREAMS (256)
  int segment_streams(int level) const {
    const double cap = kLevelCap[level], d = kLevelDets[level], e = emb_dim_;
    double per = cap * 600.0 + d * 100.0 + 3.0 * static_cast<double>(mot_lap_work_bytes(kLevelCap[level], kLevelDets[level])) + d * 6 * 4 * 3;
    if (kind_ == kPoolOCSort) per += 2.0 * d * cap * 4.0 + d * 520.0 + cap * 200.0;
    if (kind_ == kPoolBotSort && emb_dim_ > 0) per += cap * e * 4.0 + 4.0 * d * e * 4.0 + cap * d * 4.0;
    per += cap * 32.0 * 2;  // page-locked row tables of the two flights
    const double budget = static_cast<double>(env_long("MOTCPP_POOL_SEGMENT_MB", 1536)) * 1048576.0;
    long n = static_cast<long>(budget / per);
    const long top = env_long("MOTCPP_POOL_SEGMENT_STREAMS", 256);
    if (n > top) n = top;
    if (n < 2) n = 2;
    return static_cast<int>(n);
  }
  int device_, kind_;
  std::vector<float> params_;
  int emb_dim_;
  std::mutex mu_;
  std::vector<std::unique_ptr<Segment>> segs_[kLevels];
};

PoolStats pool_stats(bool reset) {
  std::lock_guard<std::mutex> g(g_stats_mu);
  const PoolStats s = g_stats;
  if (reset) g_stats = PoolStats{};
  return s;
}

bool pooling_enabled() {
  static const bool on = [] { const char* e = std::getenv("MOTCPP_LIFECYCLE"); return !(e && std::string(e) == "host"); }();
  return on;
}

int pooled_params(int c_kind, const float* p, int np, std::vector<float>* out) {
  auto P = [&](int i, float d) { return (p && i < np) ? p[i] : d; };
  switch (c_kind) {
    case 0: *out = {P(0, 0.3f), P(1, 1.f), P(2, 50.f), P(3, 3.f), P(4, 0.3f)}; return kPoolSort;
    case 1: *out = {P(0, 0.1f), P(1, 0.45f), P(2, 0.8f), P(3, 25.f), P(4, 30.f)}; return kPoolByteTrack;
    case 2: *out = {P(0, 0.2f), P(1, 30.f), P(2, 50.f), P(3, 3.f), P(4, 0.3f), P(5, 0.1f), P(6, 3.f), P(7, 0.2f), P(8, 0.f), P(9, 0.01f), P(10, 0.0001f),
                    P(11, 0.f), 1920.f, 1080.f}; return kPoolOCSort;
    case 3: *out = {P(0, 0.5f), P(1, 0.1f), P(2, 0.6f), P(3, 30.f), P(4, 0.8f), P(5, 0.5f), P(6, 0.25f), P(7, 30.f), P(8, 0.f), P(9, 1.f)}; return kPoolBotSort;
  }
  throw Error("pooled trackers: unknown tracker kind");
}

// ---- PooledStream -------------------------------------------------------------------------------------------------------
PooledStream::PooledStream(int device, int kind, const float* params, int nparams) : device_(device), kind_(kind), params_(params, params + nparams) {
  if (kind < 0 || kind > 3) throw Error("PooledStream: unknown tracker kind");
}
PooledStream::~PooledStream() {
  if (seg_) pool_->release(seg_, s_);
}
int PooledStream::level() const { return seg_ ? seg_->level() : -1; }

void PooledStream::attach(int level, int emb_dim) {
  if (!pool_) pool_ = StreamPool::get(device_, kind_, params_, emb_dim);
  pool_->acquire(level, &seg_, &s_);
  fresh_ = true;
}

// level / slot decisions of one frame, before it joins a round: fills *req (pre-operation, slot, warp)
void PooledStream::prepare(const PooledFrame& f, void* req_) {
  Request& req = *static_cast<Request*>(req_);
  const int n = f.n;
  const int bpd = kOps[kind_].births_per_det;
  // the level this frame needs: room for its detections and for every track it can add to the live ones
  int need = 0;
  while (need < kLevels && (kLevelDets[need] < n || kLevelCap[need] < alive_ + bpd * n || kLevelCap[need] < 2 * n)) ++need;
  // (tests: MOTCPP_POOL_TEST_PIN_LEVEL=1 keeps every object on the level of its first frame, so that the device's own capacity check — which
  // the level logic otherwise never lets fire — can be exercised: tests/test_gpu_pooled.py)
  static const bool pin = env_long("MOTCPP_POOL_TEST_PIN_LEVEL", 0) != 0;
  if (pin && seg_) need = seg_->level();
  if (need >= kLevels)
    throw Error("motcpp_amd: " + std::to_string(n) + " detections with " + std::to_string(alive_) + " live tracks exceed the largest pooled level (" +
                std::to_string(kLevelCap[kLevels - 1]) + " tracks x " + std::to_string(kLevelDets[kLevels - 1]) + " detections)");
  const bool with_emb = kind_ == kPoolBotSort && f.embs && f.emb_dim > 0;
  if (seg_ && with_emb && f.emb_dim != pool_->emb_dim())
    throw Error("motcpp_amd: embeddings of dimension " + std::to_string(f.emb_dim) + " given to a BoT-SORT object that started with dimension " +
                std::to_string(pool_->emb_dim()) + " (the dimension is fixed by the first frame that carries detections)");
  req.in = &f;
  if (!seg_) {
    if (kind_ == kPoolOCSort && params_.size() >= 14) { params_[12] = static_cast<float>(f.img_w); params_[13] = static_cast<float>(f.img_h); }  // centroid measure: the frame's diagonal (iou.hpp:329)
    const long floor_level = env_long("MOTCPP_POOL_MIN_LEVEL", 0);
    if (need < floor_level && floor_level < kLevels) need = static_cast<int>(floor_level);
    attach(need, with_emb ? f.emb_dim : 0);
  } else if (need > seg_->level() && fresh_) {
    //  the slot was never reset (a first frame that was prepared and withdrawn): nothing to move — whatever its old owner left there
    // must not travel along; the stream starts fresh on the larger level
    pool_->release(seg_, s_);
    pool_->acquire(need, &seg_, &s_);
  } else if (need > seg_->level()) {  // outgrown: the stream moves into a slot of a larger level at the head of this round
    req.pre = 2; req.from = seg_; req.from_s = s_;
    pool_->acquire(need, &seg_, &s_);
  }
  if (fresh_) { req.pre = 1; fresh_ = false; reset_pending_ = false; }
  else if (reset_pending_) { if (req.pre == 2) req.also_reset = true; else req.pre = 3; reset_pending_ = false; }
  req.s = s_;
  req.warp6 = have_warp_ ? warp_ : nullptr;
}
void PooledStream::unprepare(void* req_) {
  Request& req = *static_cast<Request*>(req_);
  if (req.pre == 1) fresh_ = true;
  if (req.pre == 3 || req.also_reset) reset_pending_ = true;
  if (req.pre == 2) { pool_->release(seg_, s_); seg_ = req.from; s_ = req.from_s; }
}
// after the round: this stream's rows out of the round's page-locked table (which is then free to be rewritten)
int PooledStream::finish(void* req_, const float** rows) {
  Request& req = *static_cast<Request*>(req_);
  have_warp_ = false;
  const int parity = static_cast<int>(req.round & 1);
  if (!req.error.empty()) {
    seg_->rows_taken(parity);
    if (req.from) pool_->release(req.from, req.from_s);  // (the old slot goes either way: its state was read before the failure or is lost with it)
    //  the round failed: whether this stream's reset ran is unknown — the next frame asks for it again
    if (req.pre == 1) fresh_ = true;
    if (req.pre == 3 || req.also_reset) reset_pending_ = true;
    throw Error(req.error);
  }
  const int m = req.count > 0 ? req.count : 0;
  struct Taken {  //  the round's table is released on every way out: a resize that throws must not leave the leader two rounds on waiting
    Segment* seg; int parity;
    ~Taken() { seg->rows_taken(parity); }
  } taken{seg_, parity};
  rows_.resize(static_cast<size_t>(m) * 8);
  if (m) std::memcpy(rows_.data(), req.rows, sizeof(float) * rows_.size());
  alive_ = req.alive;
  if (req.from) pool_->release(req.from, req.from_s);
  *rows = rows_.data();
  return m;
}

int PooledStream::update(const PooledFrame& f, const float** rows) {
  if (kind_ == kPoolBotSort && f.n == 0) {  // botsort.cpp:267-269: returns before anything is touched, this frame's warp included
    have_warp_ = false;
    rows_.clear();
    *rows = rows_.data();
    return 0;
  }
  const long long c0 = g_cpu.on ? thread_cpu_ns() : 0;
  Request req;
  prepare(f, &req);
  const long long c1 = g_cpu.on ? thread_cpu_ns() : 0;
  Request* q = &req;
  seg_->join(&q, 1);
  const long long c2 = g_cpu.on ? thread_cpu_ns() : 0;
  const int m = finish(&req, rows);
  if (g_cpu.on) { const long long c3 = thread_cpu_ns(); g_cpu.ns_update += c3 - c0; g_cpu.n_update += 1; g_cpu.ns_prepare += c1 - c0; g_cpu.ns_finish += c3 - c2; }
  return m;
}

void PooledStream::update_many(PooledStream* const* streams, const PooledFrame* frames, int k, const float** rows, int* counts) {
  std::vector<Request> reqs(k);
  std::vector<char> seen(k, 0);
  for (int i = 0; i < k; ++i) {
    if (streams[i]->kind_ == kPoolBotSort && frames[i].n == 0) {  // (see update())
      streams[i]->have_warp_ = false; streams[i]->rows_.clear();
      rows[i] = streams[i]->rows_.data(); counts[i] = 0; seen[i] = 2;
      continue;
    }
    try { streams[i]->prepare(frames[i], &reqs[i]); }
    catch (...) {  // nothing has run yet: the streams prepared so far go back to where they were
      for (int j = 0; j < i; ++j) if (seen[j] != 2) streams[j]->unprepare(&reqs[j]);
      throw;
    }
  }
  // one round per segment involved (objects with the same parameters and level share one). A group's rows are taken out of its round's
  // table RIGHT AFTER its join, before the next segment is joined: a thread that held untaken rows of one segment while it waited in another
  // could, with a second thread visiting the two segments in the opposite order, stop both segments' leaders for good (each waits for the
  // previous round's table to be released: Segment::lead, outstanding_[r & 1]).
  std::vector<Request*> group;
  std::vector<int> members;
  std::string first_error;
  for (int i = 0; i < k; ++i) {
    if (seen[i]) continue;
    group.clear();
    members.clear();
    for (int j = i; j < k; ++j)
      if (!seen[j] && streams[j]->seg_ == streams[i]->seg_) { seen[j] = 1; group.push_back(&reqs[j]); members.push_back(j); }
    streams[i]->seg_->join(group.data(), static_cast<int>(group.size()));
    for (int j : members) {
      try { counts[j] = streams[j]->finish(&reqs[j], &rows[j]); }
      catch (const std::exception& e) { if (first_error.empty()) first_error = e.what(); counts[j] = 0; rows[j] = nullptr; }
    }
  }
  if (!first_error.empty()) throw Error(first_error);
}

void PooledStream::reset() {
  // the stream's device state is reset at the head of its next round, as the reference's reset() does it (mot_*_reset_stream, fresh = 0)
  if (seg_ && !fresh_) reset_pending_ = true;
  alive_ = 0;
  have_warp_ = false;
}

void PooledStream::set_camera_motion(const float* w) {
  have_warp_ = w != nullptr;
  if (w) std::memcpy(warp_, w, sizeof(warp_));
}

int PooledStream::dump(std::vector<int>* ids, std::vector<float>* mean, std::vector<float>* cov, std::vector<float>* feats,
                       std::vector<unsigned char>* has_feat) {
  if (!seg_ || fresh_ || reset_pending_) { ids->clear(); mean->clear(); cov->clear(); if (feats) feats->clear(); if (has_feat) has_feat->clear(); return 0; }
  const int d = state_dim(), cap = seg_->CAP, e = seg_->E;
  ids->assign(cap, 0); mean->assign(static_cast<size_t>(cap) * d, 0.f); cov->assign(static_cast<size_t>(cap) * d * d, 0.f);
  if (feats) feats->assign(static_cast<size_t>(cap) * (e > 0 ? e : 1), 0.f);
  if (has_feat) has_feat->assign(cap, 0);
  const int n = seg_->quiesced([&](void* b) {  // (between rounds, this segment's device current)
    if (kind_ == kPoolByteTrack) return mot_bt_dump(static_cast<mot_bt_batch*>(b), s_, ids->data(), mean->data(), cov->data(), cap);
    if (kind_ == kPoolSort) return mot_sort_dump(static_cast<mot_sort_batch*>(b), s_, ids->data(), mean->data(), cov->data(), cap);
    if (kind_ == kPoolOCSort) return mot_oc_dump(static_cast<mot_oc_batch*>(b), s_, ids->data(), mean->data(), cov->data(), cap);
    return mot_bot_dump(static_cast<mot_bot_batch*>(b), s_, ids->data(), mean->data(), cov->data(), (feats && e > 0) ? feats->data() : nullptr,
                        has_feat ? has_feat->data() : nullptr, cap);
  });
  if (n < 0) throw Error("motcpp_amd: state dump failed");
  ids->resize(n); mean->resize(static_cast<size_t>(n) * d); cov->resize(static_cast<size_t>(n) * d * d);
  if (feats) feats->resize(static_cast<size_t>(n) * (e > 0 ? e : 0));
  if (has_feat) has_feat->resize(n);
  return n;
}

}  // namespace motcpp::rt

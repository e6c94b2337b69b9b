// A copy of a few hundred kilobytes by several host threads: the staging copies of the host-vector entry point
// (cask_hip_spmv: the caller's x into pinned memory, the result out of it).  One core moves such a block at 15-25 GB/s
// (the destination -- or, for a result the GPU wrote, the source -- is not in its caches), i.e. 25-35 us per 500 KB
// vector, which was most of the 77-83 us a call took; four cores do it in under 10.
//
// The helpers are created on the first large copy and then SPIN for a while after every job (a loop of spmv() calls --
// the case worth having this for -- finds them hot; a futex wake-up costs 5-10 us, as much as the copy it would help
// with) before they go to sleep on a condition variable.
//
// Two things a pool of spinning threads must survive (r6: bench.py's process measured 1 440 us per call with the first
// version, 70 without any helpers): (1) a caller whose own thread is PINNED -- OMP_PROC_BIND binds the main thread to one
// core when an OpenMP runtime starts, and threads inherit their creator's mask, so every helper landed on the caller's
// core and the spin loops fought over it: a helper asks for every CPU at start (the kernel intersects with what the
// process may use); (2) a machine on which helpers simply do not pay (a busy box; a virtual machine whose cores are
// far apart: this container copies 500 KB in 13 us with one core and 60-80 with helpers): whoever waits yields after a
// short spin, and the pool MEASURES -- every copy is timed, both ways of copying a size class are tried three times,
// the faster one is used from then on, and the other is tried again every 256th call.
// No HIP here: compiled and run under ASan / UBSan and under ThreadSanitizer on the CPU (`make asan`,
// tests/cpp/test_host_copy.cpp).
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include <sched.h>
#include <unistd.h>

namespace caskhip {

class HostCopyPool {
 public:
  static constexpr size_t MIN_PARALLEL_BYTES = 128 * 1024;    // below this one core is as fast as waking anybody
  static constexpr int SPIN_US = 300;                         // a helper spins this long after its last job

  // helpers: threads besides the caller (0 = plain memcpy)
  explicit HostCopyPool(int helpers) : jobs_((size_t)(helpers > 0 ? helpers : 0)) {
    pid_ = getpid();
    for (size_t i = 0; i < jobs_.size(); i++) threads_.emplace_back([this, i] { run(i); });
  }
  HostCopyPool(const HostCopyPool &) = delete;
  HostCopyPool &operator=(const HostCopyPool &) = delete;
  ~HostCopyPool() {
    if (getpid() != pid_) {                                   // a forked child: the threads do not exist here
      for (auto &t : threads_) t.detach();
      return;
    }
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_.store(true, std::memory_order_release);
    }
    cv_.notify_all();
    for (auto &t : threads_) t.join();
  }

  int helpers() const { return (int)jobs_.size(); }

  // memcpy(dst, src, bytes) by the caller and the helpers; returns when every byte is in place.  One copy at a time
  // (callers are serialised by a mutex: the pool serves a latency path, not a throughput one).
  // kind: copies that behave alike share their measurements (0: a caller's vector into staging -- the source is warm in
  // the caller's caches; 1: a result out of staging -- the source was written by the GPU and comes from memory)
  void copy(void *dst, const void *src, size_t bytes, int kind = 0) {
    if (bytes < MIN_PARALLEL_BYTES || jobs_.empty() || getpid() != pid_) {
      std::memcpy(dst, src, bytes);
      return;
    }
    std::lock_guard<std::mutex> one(call_mu_);
    // which way for this size class (log2 of the size): whichever measured faster, the other one every 256th call
    int cls = 0;
    while (cls + 1 < CLASSES && (MIN_PARALLEL_BYTES << (cls + 1)) <= bytes) cls++;
    Stat &st = stats_[kind & 1][cls];
    int way;                                                  // 0: this thread alone, 1: with the helpers
    if (st.n[0] < 3 || st.n[1] < 3) way = st.n[1] <= st.n[0] ? 1 : 0;
    else {
      way = st.us[1] < st.us[0] ? 1 : 0;
      if ((++st.calls & 255) == 0) way ^= 1;
    }
    const auto t0 = std::chrono::steady_clock::now();
    auto record = [&] {
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() * (double)(MIN_PARALLEL_BYTES << cls) / (double)bytes;
      st.us[way] = st.n[way] ? 0.75 * st.us[way] + 0.25 * us : us;      // (normalised to the class's smallest size)
      st.n[way]++;
    };
    if (way == 0) {
      std::memcpy(dst, src, bytes);
      record();
      return;
    }
    const size_t parts = jobs_.size() + 1;
    // shares are multiples of 4 KiB so that no two threads write the same cache line
    const size_t share = (((bytes + parts - 1) / parts) + 4095) & ~(size_t)4095;
    size_t at = share < bytes ? share : bytes;                // the caller takes [0, at)
    int given = 0;
    for (size_t i = 0; i < jobs_.size() && at < bytes; i++) {
      const size_t n = share < bytes - at ? share : bytes - at;
      jobs_[i].dst = static_cast<char *>(dst) + at;
      jobs_[i].src = static_cast<const char *>(src) + at;
      jobs_[i].bytes = n;
      at += n;
      given++;
    }
    pending_.store(given, std::memory_order_relaxed);
    const uint64_t gen = ++generation_;
    for (int i = 0; i < given; i++) jobs_[(size_t)i].gen.store(gen, std::memory_order_seq_cst);   // publishes job i (its fields are written)
    if (sleepers_.load(std::memory_order_seq_cst) > 0) {     // somebody went to sleep: wake them (the rare path)
      std::lock_guard<std::mutex> lk(mu_);
      cv_.notify_all();
    }
    std::memcpy(dst, src, share < bytes ? share : bytes);
    for (unsigned spins = 0; pending_.load(std::memory_order_acquire) > 0; spins++) {
      if (spins < 2000) cpu_relax();
      else std::this_thread::yield();                         // a helper may need this very core
    }
    record();
  }
  // (reporting) does the pool currently copy a block of `bytes` with its helpers?
  bool parallel_for(size_t bytes, int kind = 0) const {
    if (bytes < MIN_PARALLEL_BYTES || jobs_.empty()) return false;
    int cls = 0;
    while (cls + 1 < CLASSES && (MIN_PARALLEL_BYTES << (cls + 1)) <= bytes) cls++;
    const Stat &st = stats_[kind & 1][cls];
    return st.n[0] < 3 || st.n[1] < 3 || st.us[1] < st.us[0];
  }

 private:
  struct alignas(64) Job {                                    // one per helper, on its own cache line: the helper spins on `gen`
    char *dst = nullptr;
    const char *src = nullptr;
    size_t bytes = 0;
    std::atomic<uint64_t> gen{0};                             // the copy() call this job belongs to; stored last (release)
  };
  static void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
  }
  // A helper only ever looks at ITS job: a job is written by copy() before its `gen` is stored and not touched again
  // until the helper has counted itself out of `pending_`, so neither side reads what the other is writing.
  void run(size_t me) {
    {                                                         // every CPU the process may use, not the creator's (possibly pinned) mask
      cpu_set_t all;
      CPU_ZERO(&all);
      for (int c = 0; c < CPU_SETSIZE; c++) CPU_SET(c, &all);
      (void)sched_setaffinity(0, sizeof(all), &all);
    }
    Job &job = jobs_[me];
    uint64_t seen = 0;
    unsigned polls = 0;
    auto idle_since = std::chrono::steady_clock::now();
    while (true) {
      const uint64_t gen = job.gen.load(std::memory_order_acquire);
      if (gen != seen) {
        seen = gen;
        std::memcpy(job.dst, job.src, job.bytes);
        pending_.fetch_sub(1, std::memory_order_release);
        idle_since = std::chrono::steady_clock::now();
        continue;
      }
      if (stop_.load(std::memory_order_acquire)) return;
      cpu_relax();
      if ((++polls & 255) == 0) std::this_thread::yield();    // (shares a core gracefully if it has to)
      if (std::chrono::steady_clock::now() - idle_since > std::chrono::microseconds(SPIN_US)) {
        std::unique_lock<std::mutex> lk(mu_);
        sleepers_.fetch_add(1, std::memory_order_seq_cst);   // (seq_cst with copy()'s store of `gen` / load of `sleepers_`: one of the two sees the other)
        cv_.wait(lk, [&] { return stop_.load(std::memory_order_acquire) || job.gen.load(std::memory_order_seq_cst) != seen; });
        sleepers_.fetch_sub(1, std::memory_order_seq_cst);
        idle_since = std::chrono::steady_clock::now();
      }
    }
  }

  std::vector<Job> jobs_;
  std::vector<std::thread> threads_;
  uint64_t generation_ = 0;                                   // (copy() calls are serialised by call_mu_)
  std::atomic<int> pending_{0}, sleepers_{0};
  std::atomic<bool> stop_{false};
  static constexpr int CLASSES = 8;                           // 128 KiB, 256 KiB, ... >= 16 MiB
  struct Stat {
    double us[2] = {0.0, 0.0};                                // smoothed time of a copy, alone / with the helpers
    int n[2] = {0, 0};
    unsigned calls = 0;
  };
  Stat stats_[2][CLASSES];                                    // [kind][size class] (under call_mu_)
  std::mutex mu_, call_mu_;
  std::condition_variable cv_;
  pid_t pid_ = 0;
};

}  // namespace caskhip

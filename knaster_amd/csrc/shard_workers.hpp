// shard_workers.hpp -- the worker threads of a host-sharded bank (host_shards.hpp).  No HIP in here: the CPU test
// tests/cpp/shard_workers_test.cpp runs it under ThreadSanitizer.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace {

inline void cpu_relax() {
#if defined(__x86_64__)
  asm volatile("pause");
#endif
}

// K-1 persistent workers; run(fn) executes fn(0) on the caller and fn(1..K-1) on the workers, and returns when all are done.
// Workers spin briefly for the next job (back-to-back batched calls arrive microseconds apart) before they sleep.
class ShardWorkers {
 public:
  explicit ShardWorkers(int n) : n_(n) {
    for (int k = 1; k < n; ++k) threads_.emplace_back([this, k] { loop(k); });
  }
  ~ShardWorkers() {
    {
      std::lock_guard<std::mutex> l(m_);
      stop_ = true;
      epoch_.fetch_add(1, std::memory_order_release);
    }
    cv_job_.notify_all();
    for (std::thread& t : threads_) t.join();
  }
  void run(const std::function<void(int)>& fn) {
    if (n_ > 1) {
      {
        std::lock_guard<std::mutex> l(m_);
        job_ = &fn;
        remaining_.store(n_ - 1, std::memory_order_relaxed);
        epoch_.fetch_add(1, std::memory_order_release);
      }
      cv_job_.notify_all();
    }
    // An exception (std::bad_alloc from a queue that grows) must neither end a worker thread -- std::terminate -- nor
    // unwind the caller while workers still run a job that lives on the caller's stack: every thread catches its own,
    // run() waits for all of them and then rethrows the first on the calling thread, where the C ABI's guard turns it
    // into a status.
    std::exception_ptr mine;
    try {
      fn(0);
    } catch (...) {
      mine = std::current_exception();
    }
    if (n_ > 1) {
      for (int spin = 0; spin < 20000 && remaining_.load(std::memory_order_acquire) != 0; ++spin) cpu_relax();
      if (remaining_.load(std::memory_order_acquire) != 0) {
        std::unique_lock<std::mutex> l(m_);
        cv_done_.wait(l, [this] { return remaining_.load(std::memory_order_acquire) == 0; });
      }
    }
    std::exception_ptr theirs;
    {
      std::lock_guard<std::mutex> l(m_);
      theirs = failed_;
      failed_ = nullptr;
    }
    if (mine) std::rethrow_exception(mine);
    if (theirs) std::rethrow_exception(theirs);
  }

 private:
  void loop(int k) {
    uint64_t seen = 0;
    while (true) {
      for (int spin = 0; spin < 20000 && epoch_.load(std::memory_order_acquire) == seen; ++spin) cpu_relax();
      if (epoch_.load(std::memory_order_acquire) == seen) {
        std::unique_lock<std::mutex> l(m_);
        cv_job_.wait(l, [&] { return epoch_.load(std::memory_order_acquire) != seen; });
      }
      seen = epoch_.load(std::memory_order_acquire);
      const std::function<void(int)>* job;
      {
        std::lock_guard<std::mutex> l(m_);
        if (stop_) return;
        job = job_;
      }
      try {
        (*job)(k);
      } catch (...) {
        std::lock_guard<std::mutex> l(m_);
        if (!failed_) failed_ = std::current_exception();
      }
      if (remaining_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
        std::lock_guard<std::mutex> l(m_);
        cv_done_.notify_one();
      }
    }
  }
  int n_;
  std::vector<std::thread> threads_;
  std::mutex m_;
  std::condition_variable cv_job_, cv_done_;
  const std::function<void(int)>* job_ = nullptr;
  std::atomic<uint64_t> epoch_{0};
  std::atomic<int> remaining_{0};
  bool stop_ = false;
  std::exception_ptr failed_;  // the first exception a worker's job threw (under m_)
};


}  // namespace

// Minimal scheduling-only stand-in for Taskflow v3.8.0 (not vendored in the reference,
// no network here). TEST INFRASTRUCTURE ONLY: lets the *unmodified* reference CPU backend
// (/root/reference/icicle/backend/cpu) compile into oracle/_ref/. It carries no arithmetic:
// the CPU backend only uses Taskflow::{emplace,clear} and Executor(n).run(tf).wait()
// (cpu_msm.hpp:175-176,228-229,246,321; ntt_cpu.h:79-118). Tasks touch disjoint data, so
// any schedule yields identical results.
#pragma once
#include <atomic>
#include <cstdlib>
#include <functional>
#include <thread>
#include <vector>

namespace tf {

  class Taskflow
  {
  public:
    template <typename F>
    void emplace(F&& f)
    {
      m_tasks.emplace_back(std::forward<F>(f));
    }
    void clear() { m_tasks.clear(); }
    std::vector<std::function<void()>> m_tasks;
  };

  // ICICLE_TASKFLOW_SHIM_MAX_THREADS=<n>: upper bound on the threads of one run(). The reference asks for hardware_concurrency()
  // workers; a container whose cgroup grants 16 cores of CPU time while 256 are visible (the GPU boxes of this build) then runs 256
  // threads on 16 cores' worth of quota. tests/conftest.py and bench.py set the bound to what the cgroup grants. Scheduling only:
  // tasks touch disjoint data, any thread count gives identical results.
  inline unsigned shim_max_threads()
  {
    static const unsigned cap = [] {
      const char* e = std::getenv("ICICLE_TASKFLOW_SHIM_MAX_THREADS");
      const long v = e ? std::atol(e) : 0;
      return v > 0 ? (unsigned)v : 0u;
    }();
    return cap;
  }

  class Executor
  {
  public:
    explicit Executor(unsigned n = std::thread::hardware_concurrency()) : m_n(n ? n : 1)
    {
      if (shim_max_threads() && m_n > shim_max_threads()) m_n = shim_max_threads();
    }

    struct Done {
      void wait() {}
      void get() {}
    };

    Done run(Taskflow& tf)
    {
      const size_t nt = tf.m_tasks.size();
      if (nt == 0) return {};
      std::atomic<size_t> next{0};
      auto body = [&]() {
        for (;;) {
          size_t i = next.fetch_add(1);
          if (i >= nt) break;
          tf.m_tasks[i]();
        }
      };
      const unsigned nthr = (unsigned)std::min<size_t>(m_n, nt);
      std::vector<std::thread> pool;
      for (unsigned t = 1; t < nthr; ++t)
        pool.emplace_back(body);
      body();
      for (auto& t : pool)
        t.join();
      return {};
    }
    size_t num_workers() const { return m_n; }

  private:
    unsigned m_n;
  };

} // namespace tf

// Pageable host memory <-> device through a pinned staging ring ("host-resident operands as a
// first-class path"). The wrappers' default operands are pageable (`HostSlice` over a Vec): hipMemcpyAsync then goes
// through the runtime's own staging at ~16 GB/s each way (profiles/r03_host_operands.txt: a 2^24 x 64 BabyBear batch,
// 4 GiB in + 4 GiB out, spends 0.53 s in copies around 11 ms of transforms). Here a large pageable copy is cut into
// chunks; a few host threads move chunk i between the caller's memory and a pinned ring slot while the DMA engine moves
// chunk i - 1 between the slot and the device, so the copy runs at what the slower of host memcpy bandwidth and the
// PCIe link allows. Pinned or small operands take the plain hipMemcpyAsync.
//
//   stage_h2d: returns when every chunk has been handed to the DMA engine on `st` (stream-ordered like hipMemcpyAsync).
//   stage_d2h: returns when the data is in the caller's buffer (hipMemcpyAsync to pageable memory is synchronous as well).
#include "common.h"
#include <condition_variable>
#include <cstring>
#include <functional>
#include <thread>

namespace icicle_hip {
  namespace {

    constexpr size_t CHUNK = (size_t)32 << 20; // ring slot
    constexpr int SLOTS = 4;
    constexpr size_t MIN_STAGED = (size_t)16 << 20; // smaller copies are not worth the hand-over

    // ---- a small pool of host threads for parallel memcpy ----
    class CopyPool
    {
    public:
      static CopyPool& get()
      {
        static CopyPool* p = new CopyPool(); // lives as long as the process (threads are detached in spirit: never joined at exit)
        return *p;
      }
      // dst[0, n) <- src[0, n), split over the pool; returns when done
      void copy(void* dst, const void* src, size_t n)
      {
        const int T = (int)m_threads.size();
        if (T == 0 || n < ((size_t)4 << 20)) {
          memcpy(dst, src, n);
          return;
        }
        std::lock_guard<std::mutex> one_at_a_time(m_call); // (callers on different devices share the host's memory bandwidth anyway)
        std::unique_lock<std::mutex> lk(m_mu);
        m_dst = (char*)dst, m_src = (const char*)src, m_n = n;
        m_pending = T;
        m_epoch++;
        m_cv.notify_all();
        m_done.wait(lk, [&] { return m_pending == 0; });
      }

    private:
      CopyPool()
      {
        int T = 8;
        if (const char* e = getenv("ICICLE_HIP_HOST_COPY_THREADS")) T = atoi(e);
        T = std::max(0, std::min(T, 64));
        for (int t = 0; t < T; t++)
          m_threads.emplace_back([this, t, T] { worker(t, T); });
        for (auto& th : m_threads)
          th.detach();
      }
      void worker(int t, int T)
      {
        uint64_t seen = 0;
        for (;;) {
          char* d;
          const char* s;
          size_t n;
          {
            std::unique_lock<std::mutex> lk(m_mu);
            m_cv.wait(lk, [&] { return m_epoch != seen; });
            seen = m_epoch;
            d = m_dst, s = m_src, n = m_n;
          }
          const size_t part = ((n / T) + 4095) & ~(size_t)4095;
          const size_t lo = std::min(n, part * t), hi = std::min(n, lo + part);
          if (hi > lo) memcpy(d + lo, s + lo, hi - lo);
          {
            std::lock_guard<std::mutex> lk(m_mu);
            if (--m_pending == 0) m_done.notify_all();
          }
        }
      }
      std::vector<std::thread> m_threads;
      std::mutex m_mu, m_call;
      std::condition_variable m_cv, m_done;
      char* m_dst = nullptr;
      const char* m_src = nullptr;
      size_t m_n = 0;
      int m_pending = 0;
      uint64_t m_epoch = 0;
    };

    struct Ring { // one per device; a staged copy holds `mu` from its first to its last chunk
      std::mutex mu;
      void* slot[SLOTS] = {nullptr};
      hipEvent_t ev[SLOTS] = {nullptr};
      bool busy[SLOTS] = {false};
      bool ok = false, tried = false;
      bool init()
      {
        if (tried) return ok;
        tried = true;
        for (int i = 0; i < SLOTS; i++) {
          if (hipHostMalloc(&slot[i], CHUNK, hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            return false;
          }
        }
        ok = true;
        return true;
      }
    };
    Ring& ring_of(int dev)
    {
      static std::mutex mu;
      static std::map<int, Ring*> rings;
      std::lock_guard<std::mutex> g(mu);
      auto it = rings.find(dev);
      if (it == rings.end()) it = rings.emplace(dev, new Ring()).first;
      return *it->second;
    }

    bool staging_enabled()
    {
      static const bool on = !(getenv("ICICLE_HIP_HOST_STAGING") && atoi(getenv("ICICLE_HIP_HOST_STAGING")) == 0);
      return on;
    }
    // true for ordinary (pageable, unregistered) host memory
    bool is_pageable_host(const void* p)
    {
      hipPointerAttribute_t a;
      if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return true; // unknown to the runtime: plain malloc / mmap memory
      }
      return a.type == hipMemoryTypeUnregistered;
    }

  } // namespace

  hipError_t stage_h2d(void* dst, const void* src, size_t bytes, hipStream_t st)
  {
    if (bytes < MIN_STAGED || !staging_enabled() || !is_pageable_host(src)) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, st);
    Ring& r = ring_of(current_device_id());
    std::lock_guard<std::mutex> g(r.mu);
    if (!r.init()) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, st);
    CopyPool& pool = CopyPool::get();
    int i = 0;
    for (size_t off = 0; off < bytes; off += CHUNK, i = (i + 1) % SLOTS) {
      const size_t n = std::min(CHUNK, bytes - off);
      if (r.busy[i]) { // the DMA that last read this slot
        hipError_t e = hipEventSynchronize(r.ev[i]);
        if (e != hipSuccess) return e;
      }
      pool.copy(r.slot[i], (const char*)src + off, n);
      hipError_t e = hipMemcpyAsync((char*)dst + off, r.slot[i], n, hipMemcpyHostToDevice, st);
      if (e != hipSuccess) return e;
      e = hipEventRecord(r.ev[i], st);
      if (e != hipSuccess) return e;
      r.busy[i] = true;
    }
    return hipSuccess;
  }

  hipError_t stage_d2h(void* dst, const void* src, size_t bytes, hipStream_t st)
  {
    if (bytes < MIN_STAGED || !staging_enabled() || !is_pageable_host(dst)) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, st);
    Ring& r = ring_of(current_device_id());
    std::lock_guard<std::mutex> g(r.mu);
    if (!r.init()) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, st);
    CopyPool& pool = CopyPool::get();
    for (int i = 0; i < SLOTS; i++) // slots may still be feeding an upload of an earlier call
      if (r.busy[i]) {
        hipError_t e = hipEventSynchronize(r.ev[i]);
        if (e != hipSuccess) return e;
        r.busy[i] = false;
      }
    const size_t nchunks = (bytes + CHUNK - 1) / CHUNK;
    auto issue = [&](size_t c) -> hipError_t {
      const size_t off = c * CHUNK, n = std::min(CHUNK, bytes - off);
      const int i = (int)(c % SLOTS);
      hipError_t e = hipMemcpyAsync(r.slot[i], (const char*)src + off, n, hipMemcpyDeviceToHost, st);
      if (e != hipSuccess) return e;
      return hipEventRecord(r.ev[i], st);
    };
    const size_t ahead = SLOTS - 1; // DMAs in flight in front of the host copy
    for (size_t c = 0; c < std::min(ahead, nchunks); c++) {
      hipError_t e = issue(c);
      if (e != hipSuccess) return e;
    }
    for (size_t c = 0; c < nchunks; c++) {
      if (c + ahead < nchunks) {
        hipError_t e = issue(c + ahead); // its slot was drained by the host copy of chunk c - 1
        if (e != hipSuccess) return e;
      }
      const size_t off = c * CHUNK, n = std::min(CHUNK, bytes - off);
      const int i = (int)(c % SLOTS);
      hipError_t e = hipEventSynchronize(r.ev[i]);
      if (e != hipSuccess) return e;
      pool.copy((char*)dst + off, r.slot[i], n);
    }
    return hipSuccess;
  }

} // namespace icicle_hip

// Shared host-side plumbing for libicicle_hip.so: error translation, per-thread active device,
// stream-ordered temporaries, kernel timing.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <unordered_map>
#include <variant>
#include <vector>
#include "../../include/icicle_hip.h"

namespace icicle_hip {

  // hipError_t -> eIcicleError, modelled on the reference's CUDA table
  // (icicle/backend/cuda_pqc/include/gpu-utils/error_translation.h:6-33)
  inline icicle_error_t translate(hipError_t e, icicle_error_t dflt)
  {
    switch (e) {
    case hipSuccess: return ICICLE_SUCCESS;
    case hipErrorInvalidDevice: return ICICLE_INVALID_DEVICE;
    case hipErrorOutOfMemory: return ICICLE_OUT_OF_MEMORY;
    case hipErrorInvalidDevicePointer: return ICICLE_INVALID_POINTER;
    case hipErrorInvalidValue: return dflt == ICICLE_SUCCESS ? ICICLE_INVALID_ARGUMENT : dflt;
    default: return dflt == ICICLE_SUCCESS ? ICICLE_INVALID_ARGUMENT : dflt;
    }
  }

#define HIP_TRY(expr, dflt)                                                                                            \
  do {                                                                                                                 \
    hipError_t _e = (expr);                                                                                            \
    if (_e != hipSuccess) {                                                                                            \
      (void)hipGetLastError();                                                                                         \
      if (icicle_hip::verbose())                                                                                       \
        fprintf(stderr, "[icicle_hip] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(_e), __FILE__, __LINE__);     \
      return icicle_hip::translate(_e, dflt);                                                                          \
    }                                                                                                                  \
  } while (0)

#define ICICLE_TRY(expr)                                                                                               \
  do {                                                                                                                 \
    icicle_error_t _ie = (expr);                                                                                       \
    if (_ie != ICICLE_SUCCESS) return _ie;                                                                             \
  } while (0)

  bool verbose();
  bool sync_debug(); // ICICLE_HIP_SYNC_DEBUG=1: synchronise + check after every kernel launch

// after every kernel launch: catches launch-time errors always, execution errors in debug mode
#define LAUNCH_CHECK(name, stream)                                                                                     \
  do {                                                                                                                 \
    hipError_t _le = hipGetLastError();                                                                                \
    if (_le == hipSuccess && icicle_hip::sync_debug()) _le = hipStreamSynchronize(stream);                             \
    if (_le != hipSuccess) {                                                                                           \
      fprintf(stderr, "[icicle_hip] kernel %s failed: %s (%s:%d)\n", name, hipGetErrorString(_le), __FILE__, __LINE__); \
      return icicle_hip::translate(_le, ICICLE_INVALID_ARGUMENT);                                                      \
    }                                                                                                                  \
  } while (0)

  // thread-local active device (reference: thread_local sCurDevice, src/device_api.cpp:86-102).
  int current_device_id();
  // makes the calling thread's HIP context match its icicle device; call at the top of every API
  icicle_error_t bind_current_device();

  // true when `p` lies in a device allocation of the HIP runtime (hipMalloc memory of any device), false for pageable or
  // pinned host memory. For the few entry points where a wrapper of the reference leaves a location flag at its default
  // although the buffer's location is fixed by its type (msm_precompute_bases, below).
  inline bool points_to_device_memory(const void* p)
  {
    hipPointerAttribute_t a;
    if (!p || hipPointerGetAttributes(&a, p) != hipSuccess) {
      (void)hipGetLastError(); // plain malloc memory: "invalid value", not an error of ours
      return false;
    }
    return a.type == hipMemoryTypeDevice;
  }

  // true when [p, p + bytes) provably runs past the END of the device allocation that holds p (hipMemGetAddressRange). Host memory
  // and pointers the runtime cannot place give false: only a provable overrun is refused. (A sub-allocating caller -- a caching
  // allocator -- is bounded by ITS block, so this is a lower bound on safety, not a guarantee.)
  inline bool overruns_device_allocation(const void* p, size_t bytes)
  {
    if (!points_to_device_memory(p)) return false;
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    const uintptr_t off = (uintptr_t)p - (uintptr_t)base;
    return off > size || bytes > size - off;
  }

  // opaque ConfigExtension (reference: include/icicle/config_extension.h:12-46)
  struct ConfigExt {
    std::unordered_map<std::string, std::variant<int, bool>> kv;
    bool has(const char* k) const { return kv.find(k) != kv.end(); }
    int get_int(const char* k, int dflt) const
    {
      auto it = kv.find(k);
      if (it == kv.end()) return dflt;
      if (auto p = std::get_if<int>(&it->second)) return *p;
      return dflt;
    }
    bool get_bool(const char* k, bool dflt) const
    {
      auto it = kv.find(k);
      if (it == kv.end()) return dflt;
      if (auto p = std::get_if<bool>(&it->second)) return *p;
      return dflt;
    }
  };

  // Workspace arenas for temporaries. hipMallocAsync/hipFreeAsync was the first choice, but on this
  // ROCm 7.2 / gfx950 stack large pool allocations came back with multi-MiB ranges whose contents
  // were lost between two kernels of the same stream (profiles/r01_notes.md), so temporaries now come
  // from cached hipMalloc arenas, PyTorch-caching-allocator style:
  //   * acquire(bytes, stream): reuse a free arena of this device that is big enough (or grow one);
  //     if its previous user was another stream, the new stream waits on the arena's last-use event
  //     (hipStreamWaitEvent) -- the host never blocks, so is_async calls stay asynchronous;
  //   * release(stream): record the last-use event on the stream.
  struct Arena {
    void* base = nullptr;
    size_t cap = 0;
    size_t used = 0;
    int device = 0;
    bool busy = false;
    hipEvent_t last_use = nullptr;
    hipStream_t last_stream = nullptr;
    double released_at = 0; // host clock (seconds) of the last release: idle arenas decay (runtime.hip arena_decay_locked)
  };
  Arena* arena_acquire(size_t bytes, hipStream_t st); // nullptr on allocation failure
  void arena_release(Arena* a, hipStream_t st);
  void arena_trim(int device); // frees all idle arenas of a device (icicle_hip_release_workspace, release_domain)
  void arena_decay(int device); // frees the arenas of a device that have been idle for ICICLE_HIP_WORKSPACE_DECAY_S seconds
  size_t arena_cached_bytes(int device);

  // One temporary = one arena lease (released, i.e. made reusable in stream order, on destruction).
  class TempBuf
  {
  public:
    TempBuf() = default;
    TempBuf(const TempBuf&) = delete;
    TempBuf& operator=(const TempBuf&) = delete;
    ~TempBuf() { release(); }
    hipError_t alloc(size_t bytes, hipStream_t s)
    {
      release();
      m_stream = s;
      m_arena = arena_acquire(bytes ? bytes : 16, s);
      return m_arena ? hipSuccess : hipErrorOutOfMemory;
    }
    void release()
    {
      if (m_arena) {
        arena_release(m_arena, m_stream);
        m_arena = nullptr;
      }
    }
    template <class T>
    T* as() const
    {
      return reinterpret_cast<T*>(ptr());
    }
    void* ptr() const { return m_arena ? m_arena->base : nullptr; }

  private:
    Arena* m_arena = nullptr;
    hipStream_t m_stream = nullptr;
  };

  // ---- RCCL, bound lazily (multi-device calls only): the library is dlopen'ed on first use so that single-GPU
  // users never load it and a process that already holds an RCCL (PyTorch ships its own librccl.so) shares that copy.
  constexpr int RCCL_UINT32 = 3; // ncclDataType_t ncclUint32 (rccl.h)
  struct RcclApi {
    int (*CommInitAll)(void** comms, int ndev, const int* devlist);
    int (*CommDestroy)(void* comm);
    int (*AllGather)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t st);
    int (*Send)(const void* send, size_t count, int dtype, int peer, void* comm, hipStream_t st);
    int (*Recv)(void* recv, size_t count, int dtype, int peer, void* comm, hipStream_t st);
    int (*GroupStart)();
    int (*GroupEnd)();
    const char* (*GetErrorString)(int);
    // optional (diagnostics only: what the communicator layer saw, icicle_hip_collectives_info)
    int (*CommCount)(void* comm, int* count);
    int (*CommUserRank)(void* comm, int* rank);
    int (*GetVersion)(int* version);
  };
  const RcclApi* rccl_api(); // entry points of the selected NCCL-ABI library (default librccl.so); nullptr when it cannot be loaded
  // One communicator per device slot of `devs` (ncclCommInitAll), created once per device list and cached. Collectives
  // on one communicator set must not interleave between two host-side calls, so the set comes with a mutex that the
  // multi-device entry points hold from their first to their last collective (two host threads calling a multi-device
  // msm() on the same devices are serialised instead of deadlocking inside RCCL).
  struct RcclCommSet {
    std::vector<void*> comms;
    std::mutex call_mtx;
    const RcclApi* api = nullptr; // the library the communicators were created by: every collective on them goes through IT
                                  // (a concurrent icicle_hip_set_collectives_library() must not pair them with another one)
  };
  icicle_error_t rccl_comms_for(const std::vector<int>& devs, RcclCommSet** set);

  // ---- multi-device plumbing shared by msm_multi.hpp and the NTT row-shard paths ----
  // Device slots of a multi-device call: `P` = min(G, slots) host threads, slot p on physical device devs[p]. Normally
  // a slot IS a visible GPU. icicle_hip_test_set_virtual_devices(K) (tests only) makes K slots exist whatever the box
  // has, mapped round-robin onto the physical GPUs -- slots then share a device, each with its own host thread, stream
  // and (loopback) communicator rank, and slots other than 0 treat the caller's buffers as remote (staged by copy).
  int virtual_device_slots(); // 0 = off
  struct DeviceSlots {
    int home = 0;          // the calling thread's device
    int P = 1;             // slots used
    bool is_virtual = false;
    std::vector<int> devs; // physical device per slot
    bool local(int p) const { return devs[p] == home && !(is_virtual && p > 0); } // may slot p use the caller's device buffers in place?
  };
  icicle_error_t make_device_slots(int G, DeviceSlots* out);
  // One-shot failure injection for the rehearsal tests: returns true exactly once for the (slot, stage) armed with
  // icicle_hip_test_inject_failure. Stages: 1 = worker set-up, 2 = right before the bucket-exchange gate,
  // 3 = right before the result-gather gate.
  bool test_failure_armed(int slot, int stage);
  // per-process counters of what the multi-device / pipelined paths moved (icicle_hip_multi_stats)
  struct MultiStats {
    std::atomic<uint64_t> staged_base_bytes{0}, staged_scalar_bytes{0}, exchanged_bucket_bytes{0}, resident_base_hits{0}, threaded_calls{0}, exchange_messages{0};
    std::atomic<uint64_t> peer_staged_copies{0}; // cross-device copies that took the no-peer-access route (hipMemcpyPeerAsync)
    std::atomic<uint64_t> plan_fallbacks{0};     // MSMs re-run on the uniform window plan after the mixed-width one did not get its memory
  };
  MultiStats& multi_stats();
  // ---- copies between the caller's device and a worker's device (in-process multi-GPU paths) ----------------------------------
  // A worker thread (device `self`) moves operands from / results to memory of the calling thread's device (`home`). Where the
  // platform lets `self` address `home` (hipDeviceEnablePeerAccess succeeds or was on already: xGMI inside one MI355X node) the
  // copy is ONE hipMemcpy*Async(hipMemcpyDefault). Where it is refused -- IOMMU / ACS settings, a container without the peer's
  // render node, more than the supported peers -- or when icicle_hip_test_set_no_peer_access(true) says so, every copy whose
  // other side is DEVICE memory goes through hipMemcpyPeerAsync row by row, which the HIP runtime stages through host memory
  // when the two devices are not peers (slower, never wrong). Host-side operands take the plain copy either way.
  struct PeerRoute {
    int self = 0, home = 0;
    bool direct = true; // self may address home's memory
  };
  PeerRoute peer_route(int self, int home); // call on the worker thread after binding `self`; enables peer access on first use
  // rows x width bytes; `other_is_src`: src lies on `home` (or the host), dst on `self`; otherwise the reverse
  hipError_t peer_copy2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t rows, const PeerRoute& r, bool other_is_src, hipStream_t st);
  // a long-lived non-blocking side stream of the calling thread's device (operand staging runs beside the compute stream)
  hipStream_t side_stream(int which);

  // A few events per host thread and device, created once and reused round-robin: a call that stays asynchronous may
  // return while its stream waits are still pending, so events cannot be destroyed at the end of a call (a wait refers
  // to the record that was current when it was issued; re-recording later is harmless).
  hipEvent_t ring_event();
  void ring_events_release(); // a short-lived worker thread gives its events back before it ends (its streams are idle then)

  // ---- bases kept on the devices between calls (MSMConfig.ext "hip_bases_resident", msm_multi.hpp) ----
  struct ResidentKey {
    const void* bases;
    size_t row_bytes_total; // bytes of one row of the caller's base array (n * pf * point bytes)
    int rows, G, g, device, slot;
    int generation = 0; // caller's content id ("hip_bases_generation"): bump it when the bytes at `bases` change
    bool operator<(const ResidentKey& o) const
    {
      return std::tie(bases, row_bytes_total, rows, G, g, device, slot, generation) < std::tie(o.bases, o.row_bytes_total, o.rows, o.G, o.g, o.device, o.slot, o.generation);
    }
  };
  struct ResidentShard {
    void* ptr = nullptr;
    size_t bytes = 0;
    hipEvent_t ready = nullptr; // recorded behind the copy that filled the shard
  };
  std::mutex& resident_mtx();
  std::map<ResidentKey, ResidentShard>& resident_map();
  // frees the cached shards of `bases` (nullptr: of every pointer) on every device; returns the bytes given back.
  // Also called when the caller frees a device allocation through icicle_free / icicle_free_async: copies made for a
  // pointer that is gone must not be served to whatever is allocated at that address next.
  size_t resident_release(const void* bases);

  // ---- precomputed base tables (msm_precompute_bases) ---------------------------------------------
  // A table fixes the doubling shift c * wpf between the copies of a base, so msm() must run with the window size the table
  // was built for. Both calls derive c from the size of ONE MSM (like cpu_msm.hpp:466 vs :207), which goes silently wrong
  // as soon as the sizes differ -- an MSM over a prefix of a table, a per-MSM table precomputed with another batch_size
  // (ADVICE r04). msm_precompute_bases therefore REMEMBERS where it wrote a table and with which c; msm() with
  // precompute_factor > 1 and config.c <= 0 asks here first and only derives c from its own size for a table this process
  // has not seen (one the caller copied or loaded: the reference's own contract then). Entries go when the allocation is
  // freed through icicle_free / the plugin (table_forget_range); a host table is recognised by 32 bytes of its second entry.
  void table_register(const void* table, size_t bytes, size_t entry_bytes, int pf, int c);
  int table_lookup_c(const void* bases, size_t entry_bytes, int pf); // 0: unknown table
  void table_forget_overlap(const void* dst, size_t size); // a runtime-API write of [dst, dst + size): tables recorded there are stale
  void table_forget_range(const void* ptr); // every table that starts inside the device allocation `ptr` belongs to

  // ---- host-thread rendezvous of the multi-device entry points (msm_multi.hpp, ntt_split.hpp) ----
  // Rendezvous of the per-device host threads in front of a collective: a thread that failed earlier (allocation,
  // copy, launch) must not leave its peers blocked inside ncclSend/ncclRecv/ncclAllGather forever. Every thread
  // arrives exactly once per gate, with its status; all of them leave with "everybody was fine" or all with "someone
  // failed" and then skip the collective.
  struct PhaseGate {
    std::mutex mu;
    std::condition_variable cv;
    int expected = 1, arrived = 0;
    bool failed = false;
    bool arrive(bool ok)
    {
      std::unique_lock<std::mutex> lk(mu);
      if (!ok) failed = true;
      if (++arrived == expected) {
        cv.notify_all();
      } else {
        cv.wait(lk, [&] { return arrived == expected; });
      }
      return !failed;
    }
    // registers an arrival without waiting for the others: a worker that is on its way OUT (error return, exception)
    // owes one arrival to every gate it has not passed, possibly to several -- if it waited at the first of them while
    // its peers still wait for it at another, nobody would ever move (found by the rehearsal test: set-up failure of one
    // slot with the bucket exchange on)
    void leave(bool ok)
    {
      std::lock_guard<std::mutex> lk(mu);
      if (!ok) failed = true;
      if (++arrived == expected) cv.notify_all();
    }
  };
  // A worker's obligation to arrive at a gate exactly once. If the worker leaves early -- error return or exception --
  // the destructor arrives for it with "failed", so the peers skip the collective instead of waiting for ever.
  struct GateTicket {
    PhaseGate* gate = nullptr;
    bool used = false;
    GateTicket() = default;
    explicit GateTicket(PhaseGate* g) : gate(g) {}
    GateTicket(const GateTicket&) = delete;
    GateTicket& operator=(const GateTicket&) = delete;
    bool arrive(bool ok)
    {
      if (!gate || used) return ok;
      used = true;
      return gate->arrive(ok);
    }
    ~GateTicket()
    {
      if (gate && !used) gate->leave(false);
    }
  };

  // ---- dominant-kernel timing with hipEvents on the launch stream (bench.py roofline figure) ----
  struct KernelTimer {
    static bool enabled();
    // records start/stop events around a launch region on `s` and queues them for later readout
    static void begin(int which, hipStream_t s);
    static void end(int which, hipStream_t s);
  };

} // namespace icicle_hip

// Multi-device and pipelined MSM behind the unchanged <curve>_msm symbol (VERDICT r01 items e4 / e5,
// VERDICT r02 items 2, 3, 6).
//
// The reference has no multi-device MSM: its model is "one host thread per device, the caller splits the work"
// (docs/docs/start/architecture/multi-device.md:34-36,76). Backend-specific knobs travel in MSMConfig.ext
// (include/icicle/msm.h:52, include/icicle/backend/msm_config.h:4-16); this backend reads
//
//   "hip_num_devices"          int   G >= 1: cut the (scalar, base) pairs into G contiguous shards (the same cut as
//                                    icicle_amd/dist.py shard_range) and run them on min(G, visible GPUs) device slots
//                                    starting at the calling thread's active device; with fewer GPUs than shards the
//                                    extra "logical" shards share a device and run back to back on its stream.
//   "hip_msm_exchange_buckets" bool  false (E1, default): every device finishes its shard; the 3*L-word partial
//                                    results are all-gathered (RCCL) and summed with k_proj_sum.
//                                    true (E2): the exchange happens in BUCKET space -- after accumulation device d
//                                    receives slice d of every peer's bucket array (grouped ncclSend / ncclRecv =
//                                    all-to-all over xGMI), adds the slices and reduces only its own slice; the
//                                    per-slice results then take the E1 exchange. E2 strong-scales the bucket
//                                    reduction at the price of moving W*2^(c-1)*sizeof(projective)/G bytes per peer.
//   "hip_bases_resident"       bool  the caller promises that the bases at this pointer do not change until
//                                    icicle_hip_msm_release_resident_bases(bases): every device keeps the shards it
//                                    staged (keyed on pointer, size, precompute factor, shard cut), so the second and
//                                    later calls move scalars only -- SURVEY.md 8(e) "bases stay resident per GPU;
//                                    scalars streamed" (the reference's own vehicle for it is are_points_on_device,
//                                    include/icicle/msm.h:39-47, which names ONE device). The copies are keyed on the
//                                    POINTER: memory from icicle_malloc releases them when it is freed; for any other
//                                    memory (host buffers, another allocator's device memory) the release call is
//                                    MANDATORY before the address can be reused, or the contents are versioned with
//   "hip_bases_generation"     int   a caller-chosen id that is part of the cache key: a new value = new copies.
//   "hip_force_rccl"           bool  use the RCCL exchange even with one device slot (size-1 communicator): test hook.
//
// Operand staging is a two-stage pipeline per device: while shard j runs on the compute stream, the operands of shard
// j + 1 are copied on a side stream into the other half of a two-slot ring (host -> device, or peer -> device from the
// calling device), so a device never waits for a copy except the first one. The same pipeline serves a single GPU with
// HOST-resident scalars (the wrappers' default HostSlice): msm_run() cuts such a call into chunks so that the
// host-to-device copy of chunk j + 1 hides behind the MSM of chunk j.
//
// One device slot: everything is enqueued on the caller's stream and is_async is honoured. Several slots: one host
// thread + one stream per slot (the reference's own model, applied inside the call); the call then returns when every
// device has finished, whatever is_async says.
#pragma once
#include <condition_variable>
#include <mutex>
#include <thread>

namespace icicle_hip {

  inline void shard_range(int n, int g, int G, int* lo, int* hi)
  { // contiguous shard g of n items, remainder spread over the first shards (== icicle_amd/dist.py shard_range)
    const int base = n / G, rem = n % G;
    *lo = g * base + std::min(g, rem);
    *hi = *lo + base + (g < rem ? 1 : 0);
  }

  // E2 hook of one device slot: sums the bucket arrays of its logical shards, then (last shard) exchanges bucket
  // slices with the peers and leaves this device's slice, summed over all devices, in the caller's bucket array.
  template <class C>
  struct BucketExchange : MsmBucketHook<C> {
    using Proj = typename EC<C>::Proj;
    int nshards = 1, seen = 0, P = 1, p = 0;
    TempBuf acc, recv, send;
    bool self_exchange = false; // "hip_force_rccl" with one device slot
    void* comm = nullptr;
    const RcclApi* api_of_comm = nullptr; // the library `comm` belongs to (RcclCommSet::api)
    GateTicket* ticket = nullptr; // used (once) right before the slice exchange

    icicle_error_t after_accumulate(Proj* buckets, size_t tw, uint32_t nb, uint32_t nseg, uint32_t m, hipStream_t st, bool* skip_reduce, uint32_t* seg_lo, uint32_t* nsegr) override
    {
      const size_t nbk = tw * nb;
      const bool last = (++seen == nshards);
      if (nshards > 1) {
        if (seen == 1) {
          HIP_TRY(acc.alloc(nbk * sizeof(Proj), st), ICICLE_ALLOCATION_FAILED);
          HIP_TRY(hipMemcpyAsync(acc.ptr(), buckets, nbk * sizeof(Proj), hipMemcpyDeviceToDevice, st), ICICLE_COPY_FAILED);
        } else {
          k_bucket_add<C><<<(unsigned)((nbk + 127) / 128), 128, 0, st>>>(acc.as<Proj>(), buckets, nbk);
          LAUNCH_CHECK("k_bucket_add", st);
        }
        if (!last) {
          *skip_reduce = true;
          return ICICLE_SUCCESS;
        }
        HIP_TRY(hipMemcpyAsync(buckets, acc.ptr(), nbk * sizeof(Proj), hipMemcpyDeviceToDevice, st), ICICLE_COPY_FAILED);
      }
      // slot q's slice of every window: reduction chunks [nseg*q/P, nseg*(q+1)/P), i.e. buckets [blo, bhi). A chunk is
      // m = 64*mrow buckets but a small window (nb < m: tiny shards, a caller-set c <= 6) is ONE chunk of nb buckets,
      // so the bucket range is clamped to the window (the unclamped range ran into the next window).
      auto slice = [&](int q, uint32_t* lo, uint32_t* hi, size_t* blo, size_t* bhi) {
        *lo = (uint32_t)((uint64_t)nseg * q / P);
        *hi = (uint32_t)((uint64_t)nseg * (q + 1) / P);
        *blo = std::min<size_t>((size_t)*lo * m, nb);
        *bhi = std::min<size_t>((size_t)*hi * m, nb);
        if (*hi == nseg) *bhi = nb; // the last chunk of a window owns its tail
      };
      uint32_t lo, hi;
      size_t blo, bhi;
      slice(p, &lo, &hi, &blo, &bhi);
      *seg_lo = lo;
      *nsegr = hi - lo;
      // One device slot: nothing to exchange -- unless "hip_force_rccl" asks for the exchange anyway: the slot then sends its
      // slice to ITSELF through the size-1 communicator (ncclSend / ncclRecv to the own rank inside a group is legal) and
      // REPLACES the slice with what arrived, so that the real librccl point-to-point calls run on a single-GPU box and a
      // mis-delivered byte shows up in the MSM result.
      if (P == 1 && !self_exchange) return ICICLE_SUCCESS;
      const RcclApi* api = api_of_comm;
      if (!api) return ICICLE_API_NOT_IMPLEMENTED;
      const size_t mine = bhi - blo; // buckets of my slice per window
      // ONE message per peer: the tw window slices a peer owns are packed into a contiguous send buffer first (a strided
      // device copy), and arrive contiguously ([peer][window][mine]) -- rounds 2-3 issued tw x (P - 1) Send / Recv pairs per
      // device (91 at c = 20, G = 8)
      std::vector<size_t> soff(P + 1, 0);
      for (int q = 0; q < P; q++) {
        uint32_t qlo, qhi;
        size_t qblo, qbhi;
        slice(q, &qlo, &qhi, &qblo, &qbhi);
        soff[q + 1] = soff[q] + ((q == p && !self_exchange) ? 0 : tw * (qbhi - qblo));
      }
      bool ready = api != nullptr && recv.alloc(std::max<size_t>(1, (size_t)P * tw * mine) * sizeof(Proj), st) == hipSuccess &&
                   send.alloc(std::max<size_t>(1, soff[P]) * sizeof(Proj), st) == hipSuccess;
      for (int q = 0; q < P && ready; q++) {
        uint32_t qlo, qhi;
        size_t qblo, qbhi;
        slice(q, &qlo, &qhi, &qblo, &qbhi);
        if (soff[q + 1] == soff[q]) continue;
        ready = hipMemcpy2DAsync(send.as<Proj>() + soff[q], (qbhi - qblo) * sizeof(Proj), buckets + qblo, (size_t)nb * sizeof(Proj), (qbhi - qblo) * sizeof(Proj), tw, hipMemcpyDeviceToDevice, st) == hipSuccess;
      }
      if (test_failure_armed(p, 2)) ready = false;
      const bool all_ready = ticket ? ticket->arrive(ready) : ready; // a peer (or this device) cannot take part: nobody enters the collective
      if (!ready) return api ? ICICLE_ALLOCATION_FAILED : ICICLE_API_NOT_IMPLEMENTED;
      if (!all_ready) return ICICLE_ALLOCATION_FAILED;
      constexpr size_t PWORDS = sizeof(Proj) / 4;
      if (api->GroupStart() != 0) return ICICLE_COPY_FAILED;
      bool ok = true; // a failed call must not leave the group open: GroupEnd is always reached
      size_t sent = 0;
      for (int q = 0; q < P && ok; q++) {
        if (q == p && !self_exchange) continue;
        const size_t cnt = soff[q + 1] - soff[q];
        if (cnt) {
          ok = api->Send(send.as<Proj>() + soff[q], cnt * PWORDS, RCCL_UINT32, q, comm, st) == 0;
          sent += cnt * sizeof(Proj);
          multi_stats().exchange_messages++;
        }
        if (ok && mine) ok = api->Recv(recv.as<Proj>() + (size_t)q * tw * mine, tw * mine * PWORDS, RCCL_UINT32, q, comm, st) == 0;
      }
      if (api->GroupEnd() != 0 || !ok) return ICICLE_COPY_FAILED;
      multi_stats().exchanged_bucket_bytes += sent;
      if (mine) {
        for (int q = 0; q < P; q++) {
          if (q == p) {
            if (self_exchange) // my own slice, back from the round trip through the communicator
              HIP_TRY(hipMemcpy2DAsync(buckets + blo, (size_t)nb * sizeof(Proj), recv.as<Proj>() + (size_t)q * tw * mine, mine * sizeof(Proj), mine * sizeof(Proj), tw, hipMemcpyDeviceToDevice, st), ICICLE_COPY_FAILED);
            continue;
          }
          for (size_t w = 0; w < tw; w++) {
            k_bucket_add<C><<<(unsigned)((mine + 127) / 128), 128, 0, st>>>(buckets + w * nb + blo, recv.as<Proj>() + ((size_t)q * tw + w) * mine, mine);
            LAUNCH_CHECK("k_bucket_add(slice)", st);
          }
        }
      }
      return ICICLE_SUCCESS;
    }
  };

  struct MsmMultiOpts {
    int G = 1;
    int max_slots = 0; // > 0: use at most this many device slots (1 = pipeline the shards on the calling device only)
    bool exchange_buckets = false, force_rccl = false, bases_resident = false;
    int bases_generation = 0;
  };

  template <class C>
  static icicle_error_t msm_multi_run(const void* scalars_v, const void* bases_v, int n, const icicle_msm_config_t* cfg, void* results_v, MsmMultiOpts opt)
  {
    using E = EC<C>;
    constexpr int PW = 2 * E::N32, RW = 3 * E::N32, SW = 8;
    const int G = opt.G;
    if (!results_v || n < 0 || G < 1) return ICICLE_INVALID_ARGUMENT;
    if (n > 0 && (!scalars_v || !bases_v)) return ICICLE_INVALID_POINTER;
    const int batch = std::max(1, cfg->batch_size);
    const int pf = std::max(1, cfg->precompute_factor);
    const bool shared = cfg->are_points_shared_in_batch || batch == 1;
    const int brows = shared ? 1 : batch;
    DeviceSlots ds;
    ICICLE_TRY(make_device_slots(G, &ds));
    if (opt.max_slots > 0 && ds.P > opt.max_slots) {
      ds.P = opt.max_slots;
      ds.devs.resize(ds.P);
    }
    const int P = ds.P, home = ds.home;
    bool exchange_buckets = opt.exchange_buckets;
    if (exchange_buckets && n < G) exchange_buckets = false; // every shard must reach the accumulation step

    // one window size for every shard: a precomputed base table fixes the doubling shift c*wpf, which
    // msm_precompute_bases derived from the FULL size (cpu_msm.hpp:455-480); and E2 adds bucket arrays element-wise
    icicle_msm_config_t sub = *cfg;
    sub.ext = nullptr;
    {
      const MsmPlan pl = make_plan(std::max(pf > 1 ? n : (n + G - 1) / G, 1), C::fr::NBITS, *cfg, -1);
      sub.c = pl.c;
      // the bucket exchange needs a whole batch in one launch group of msm_run_single; decided here, from the shape
      // alone, so that every device takes the same branch (fallback: partial-sum exchange)
      if (exchange_buckets && (size_t)batch * pl.wpf > 60000) exchange_buckets = false;
      // Partial sums need no common window size: without a base table and without the bucket exchange every shard plans for
      // itself, which lets the large ones take the mixed-width plans of round 5 (msm_plan.h; 2^26-term shards: 12 windows)
      if (pf == 1 && !exchange_buckets && cfg->c <= 0) sub.c = 0;
    }
    sub.are_scalars_on_device = sub.are_points_on_device = sub.are_results_on_device = true;
    sub.is_async = true;

    // "hip_force_rccl": take the RCCL exchange even with ONE device slot (communicator of size 1), so that the
    // loader, ncclCommInitAll and ncclAllGather bindings are exercised on a single-GPU box (tests)
    const bool use_rccl = P > 1 || opt.force_rccl;
    RcclCommSet* cset = nullptr;
    std::unique_lock<std::mutex> comm_lock;
    if (use_rccl) {
      if (!rccl_api()) {
        fprintf(stderr, "[icicle_hip] hip_num_devices > 1 needs librccl.so (not loadable)\n");
        return ICICLE_API_NOT_IMPLEMENTED;
      }
      ICICLE_TRY(rccl_comms_for(ds.devs, &cset));
      ICICLE_TRY(bind_current_device());
      comm_lock = std::unique_lock<std::mutex>(cset->call_mtx); // collectives of two calls must not interleave
    }
    const bool threaded = P > 1;
    if (threaded) {
      // the shards run on this call's own streams: whatever the caller queued on config.stream (input copies) comes first
      HIP_TRY(hipStreamSynchronize((hipStream_t)cfg->stream), ICICLE_SYNCHRONIZATION_FAILED);
      multi_stats().threaded_calls++;
    }

    PhaseGate gate_exchange, gate_gather;
    gate_exchange.expected = gate_gather.expected = P;
    std::vector<icicle_error_t> rcs(P, ICICLE_SUCCESS);

    auto worker = [&](int p) -> icicle_error_t {
      GateTicket t_exchange((exchange_buckets && P > 1) ? &gate_exchange : nullptr);
      GateTicket t_gather(use_rccl ? &gate_gather : nullptr);
      if (test_failure_armed(p, 1)) return ICICLE_ALLOCATION_FAILED; // (tickets arrive with "failed" on the way out)
      if (icicle_hip_set_device(ds.devs[p]) != ICICLE_SUCCESS) return ICICLE_INVALID_DEVICE;
      // operands are pulled from the calling device: direct xGMI copies where the platform allows them, hipMemcpyPeerAsync
      // (host-staged by the runtime) where peer access is refused -- common.h PeerRoute
      const PeerRoute route = peer_route(ds.devs[p], home);
      // one slot: the caller's stream. Several: a long-lived stream per (device, slot) -- never a stream created and
      // destroyed per call: the workspace arenas keep last-use events recorded on whatever stream used them, and an event
      // whose stream is gone makes later hipEventSynchronize / hipStreamWaitEvent calls fail (seen in the rehearsal suite)
      hipStream_t st = threaded ? side_stream(200 + p) : (hipStream_t)cfg->stream;
      if (threaded && !st) return ICICLE_STREAM_CREATION_FAILED;
      hipStream_t cs = side_stream(threaded ? 300 + p : 0); // operand staging; slots that share a device get their own
      icicle_error_t rc;
      { // the hook's buffers are released in stream order: it must die before a stream created here does
        BucketExchange<C> hook;
        rc = [&]() -> icicle_error_t {
          if (!cs) return ICICLE_STREAM_CREATION_FAILED;
          std::vector<int> mine;
          for (int g = p; g < G; g += P)
            mine.push_back(g);
          const int ns_p = (int)mine.size();
          TempBuf partials, devpart, gathered, fin;
          HIP_TRY(partials.alloc((size_t)std::max(1, ns_p) * batch * RW * 4, st), ICICLE_ALLOCATION_FAILED);
          HIP_TRY(devpart.alloc((size_t)batch * RW * 4, st), ICICLE_ALLOCATION_FAILED);
          hook.nshards = ns_p, hook.P = P, hook.p = p, hook.comm = cset ? cset->comms[p] : nullptr;
          hook.api_of_comm = cset ? cset->api : nullptr;
          hook.self_exchange = P == 1 && opt.force_rccl && cset != nullptr;
          hook.ticket = &t_exchange;
          icicle_msm_config_t c2 = sub;
          c2.stream = st;

          // ---- two-slot operand ring: prepare = lease the buffers (compute stream), copy = fill them (side stream)
          struct Stage {
            TempBuf sc, b;
            const uint32_t* scp = nullptr;
            const uint32_t* bp = nullptr;
            int lo = 0, ns = 0;
            bool copy_sc = false, copy_b = false, wait_resident = false;
            hipEvent_t leased = nullptr, filled = nullptr, resident_ready = nullptr;
          } ring[2];
          auto prepare = [&](int j) -> icicle_error_t {
            Stage& s = ring[j & 1];
            int hi;
            shard_range(n, mine[j], G, &s.lo, &hi);
            s.ns = hi - s.lo;
            s.scp = (const uint32_t*)scalars_v + (size_t)s.lo * SW;
            s.bp = (const uint32_t*)bases_v + (size_t)s.lo * pf * PW;
            s.copy_sc = s.copy_b = s.wait_resident = false;
            s.sc.release(); // (leases of shard j - 2: given back in stream order, behind its MSM)
            s.b.release();
            if (s.ns == 0) return ICICLE_SUCCESS;
            const bool sc_direct = cfg->are_scalars_on_device && ds.local(p) && batch == 1;
            if (!sc_direct) { // [batch][ns] rows out of the caller's [batch][n]
              HIP_TRY(s.sc.alloc((size_t)batch * s.ns * SW * 4, st), ICICLE_ALLOCATION_FAILED);
              s.copy_sc = true;
            }
            const bool b_direct = cfg->are_points_on_device && ds.local(p) && brows == 1;
            if (!b_direct) {
              const size_t row = (size_t)s.ns * pf * PW * 4;
              if (opt.bases_resident) {
                // first use: stage the shard into memory that stays (filled on the side stream right here, under the
                // lock, so that a concurrent call never sees an entry whose copy has not been enqueued yet)
                const ResidentKey key{bases_v, (size_t)n * pf * PW * 4, brows, G, mine[j], ds.devs[p], p, opt.bases_generation};
                std::lock_guard<std::mutex> g(resident_mtx());
                auto& m = resident_map();
                auto it = m.find(key);
                if (it == m.end()) {
                  ResidentShard rs;
                  rs.bytes = row * brows;
                  HIP_TRY(hipMalloc(&rs.ptr, rs.bytes), ICICLE_ALLOCATION_FAILED);
                  const uint32_t* src = (const uint32_t*)bases_v + (size_t)s.lo * pf * PW;
                  // the copy runs on the side stream: it must come BEHIND whatever the caller queued on the compute stream
                  // (bases still being written there would otherwise be cached stale, for good -- ADVICE r03)
                  hipEvent_t before = ring_event();
                  if (!before || hipEventRecord(before, st) != hipSuccess || hipStreamWaitEvent(cs, before, 0) != hipSuccess) {
                    (void)hipGetLastError();
                    (void)hipFree(rs.ptr);
                    return ICICLE_SYNCHRONIZATION_FAILED;
                  }
                  if (hipEventCreateWithFlags(&rs.ready, hipEventDisableTiming) != hipSuccess ||
                      peer_copy2d(rs.ptr, row, src, (size_t)n * pf * PW * 4, row, brows, route, true, cs) != hipSuccess ||
                      hipEventRecord(rs.ready, cs) != hipSuccess) {
                    (void)hipGetLastError();
                    if (rs.ready) (void)hipEventDestroy(rs.ready);
                    (void)hipFree(rs.ptr);
                    return ICICLE_COPY_FAILED;
                  }
                  multi_stats().staged_base_bytes += rs.bytes;
                  it = m.emplace(key, rs).first;
                } else {
                  multi_stats().resident_base_hits++;
                }
                s.wait_resident = true;
                s.resident_ready = it->second.ready;
                s.bp = (const uint32_t*)it->second.ptr;
              } else {
                HIP_TRY(s.b.alloc(row * brows, st), ICICLE_ALLOCATION_FAILED);
                s.copy_b = true;
              }
            }
            if (s.copy_sc || s.copy_b) { // the side stream may write the leased ranges once their previous user is done
              s.leased = ring_event();
              if (!s.leased) return ICICLE_ALLOCATION_FAILED;
              HIP_TRY(hipEventRecord(s.leased, st), ICICLE_SYNCHRONIZATION_FAILED);
            } else {
              s.leased = nullptr;
            }
            return ICICLE_SUCCESS;
          };
          auto copy = [&](int j) -> icicle_error_t {
            Stage& s = ring[j & 1];
            s.filled = nullptr;
            if (!s.copy_sc && !s.copy_b) return ICICLE_SUCCESS;
            if (s.leased) HIP_TRY(hipStreamWaitEvent(cs, s.leased, 0), ICICLE_SYNCHRONIZATION_FAILED);
            if (s.copy_sc) {
              const uint32_t* src = (const uint32_t*)scalars_v + (size_t)s.lo * SW;
              HIP_TRY(peer_copy2d(s.sc.ptr(), (size_t)s.ns * SW * 4, src, (size_t)n * SW * 4, (size_t)s.ns * SW * 4, batch, route, true, cs), ICICLE_COPY_FAILED);
              s.scp = (const uint32_t*)s.sc.ptr();
              multi_stats().staged_scalar_bytes += (size_t)batch * s.ns * SW * 4;
            }
            if (s.copy_b) {
              const size_t row = (size_t)s.ns * pf * PW * 4;
              const uint32_t* src = (const uint32_t*)bases_v + (size_t)s.lo * pf * PW;
              HIP_TRY(peer_copy2d(s.b.ptr(), row, src, (size_t)n * pf * PW * 4, row, brows, route, true, cs), ICICLE_COPY_FAILED);
              s.bp = (const uint32_t*)s.b.ptr();
              multi_stats().staged_base_bytes += row * brows;
            }
            s.filled = ring_event();
            if (!s.filled) return ICICLE_ALLOCATION_FAILED;
            HIP_TRY(hipEventRecord(s.filled, cs), ICICLE_SYNCHRONIZATION_FAILED);
            return ICICLE_SUCCESS;
          };

          if (ns_p > 0) {
            ICICLE_TRY(prepare(0));
            ICICLE_TRY(copy(0));
          }
          for (int j = 0; j < ns_p; j++) {
            // lease shard j + 1's buffers BEFORE shard j's kernels are enqueued (the side stream waits for the lease
            // point, not for shard j), fill them AFTER (a copy from pageable host memory blocks this thread: the GPU
            // then already has shard j to work on)
            if (j + 1 < ns_p) ICICLE_TRY(prepare(j + 1));
            Stage& s = ring[j & 1];
            if (s.filled) HIP_TRY(hipStreamWaitEvent(st, s.filled, 0), ICICLE_SYNCHRONIZATION_FAILED);
            if (s.wait_resident) HIP_TRY(hipStreamWaitEvent(st, s.resident_ready, 0), ICICLE_SYNCHRONIZATION_FAILED);
            ICICLE_TRY(msm_run_single<C>(s.scp, s.bp, s.ns, &c2, partials.as<uint32_t>() + (size_t)j * batch * RW, exchange_buckets ? &hook : nullptr));
            if (j + 1 < ns_p) ICICLE_TRY(copy(j + 1));
          }
          // per-device partial: E1 = sum over this device's shards; E2 = the last shard's result (it reduced the summed slice)
          if (exchange_buckets && ns_p > 0) {
            HIP_TRY(hipMemcpyAsync(devpart.ptr(), partials.as<uint32_t>() + (size_t)(ns_p - 1) * batch * RW, (size_t)batch * RW * 4, hipMemcpyDeviceToDevice, st), ICICLE_COPY_FAILED);
          } else {
            k_proj_sum<C><<<batch, 64, 0, st>>>(partials.as<uint32_t>(), ns_p, (size_t)batch * RW, devpart.as<uint32_t>());
            LAUNCH_CHECK("k_proj_sum(shards)", st);
          }
          const uint32_t* result = devpart.as<uint32_t>();
          if (use_rccl) { // "all-reduce" of partial sums: EC addition is not an RCCL reduce op -> all-gather + local projective sum
            const RcclApi* api = cset->api;
            bool have_bufs = gathered.alloc((size_t)P * batch * RW * 4, st) == hipSuccess && fin.alloc((size_t)batch * RW * 4, st) == hipSuccess;
            if (test_failure_armed(p, 3)) have_bufs = false;
            if (!t_gather.arrive(have_bufs) || !have_bufs) return ICICLE_ALLOCATION_FAILED;
            if (api->AllGather(devpart.ptr(), gathered.ptr(), (size_t)batch * RW, RCCL_UINT32, cset->comms[p], st) != 0) return ICICLE_COPY_FAILED;
            k_proj_sum<C><<<batch, 64, 0, st>>>(gathered.as<uint32_t>(), P, (size_t)batch * RW, fin.as<uint32_t>());
            LAUNCH_CHECK("k_proj_sum(devices)", st);
            result = fin.as<uint32_t>();
          }
          if (p == 0) // the calling device delivers the result
            HIP_TRY(hipMemcpyAsync(results_v, result, (size_t)batch * RW * 4, cfg->are_results_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st), ICICLE_COPY_FAILED);
          if (threaded || !cfg->are_results_on_device || !cfg->is_async) HIP_TRY(hipStreamSynchronize(st), ICICLE_SYNCHRONIZATION_FAILED);
          return ICICLE_SUCCESS;
        }();
      }
      if (threaded) {
        (void)hipStreamSynchronize(st);
        (void)hipStreamSynchronize(cs);
        ring_events_release();
      } else if (rc != ICICLE_SUCCESS) {
        (void)hipStreamSynchronize(st); // nothing of a failed call may still be running when its buffers are reused
      }
      return rc;
    };

    if (!threaded) {
      rcs[0] = worker(0);
    } else {
      std::vector<std::thread> th;
      for (int p = 0; p < P; p++)
        th.emplace_back([&, p]() {
          try {
            rcs[p] = worker(p);
          } catch (...) { // (the worker's gate tickets have arrived for it during unwinding)
            rcs[p] = ICICLE_INVALID_ARGUMENT;
          }
        });
      for (auto& t : th)
        t.join();
    }
    ICICLE_TRY(icicle_hip_set_device(home));
    for (int p = 0; p < P; p++)
      if (rcs[p] != ICICLE_SUCCESS) return rcs[p];
    return ICICLE_SUCCESS;
  }

  // entry point used by every <curve>[_g2]_msm symbol
  template <class C>
  static icicle_error_t msm_run(const void* scalars_v, const void* bases_v, int n, const icicle_msm_config_t* cfg, void* results_v)
  {
    if (!cfg) return ICICLE_INVALID_ARGUMENT;
    icicle_msm_config_t with_c;
    if (cfg->precompute_factor > 1 && cfg->c <= 0) {
      // a base table written by msm_precompute_bases in this process: run with the window size it was built for, whatever
      // the size of THIS msm is (a prefix of the table, another batch shape; common.h "precomputed base tables")
      const int c = table_lookup_c(bases_v, (size_t)2 * EC<C>::N32 * 4, cfg->precompute_factor);
      if (c > 0) {
        with_c = *cfg;
        with_c.c = c;
        cfg = &with_c;
      }
    }
    MsmMultiOpts opt;
    opt.G = 0;
    if (cfg->ext) {
      const ConfigExt* e = reinterpret_cast<const ConfigExt*>(cfg->ext);
      opt.G = e->get_int("hip_num_devices", 0);
      opt.exchange_buckets = e->get_bool("hip_msm_exchange_buckets", false);
      opt.force_rccl = e->get_bool("hip_force_rccl", false);
      opt.bases_resident = e->get_bool("hip_bases_resident", false);
      opt.bases_generation = e->get_int("hip_bases_generation", 0);
    }
    if (opt.G >= 1) return msm_multi_run<C>(scalars_v, bases_v, n, cfg, results_v, opt);
    // Host-resident scalars on one GPU (the wrappers' default HostSlice): cut the MSM into chunks so that the
    // host-to-device copy of chunk j + 1 runs behind the MSM of chunk j (the copy of 2^26 scalars is 2 GiB, ~40 ms of
    // PCIe against ~70 ms of MSM). ICICLE_HIP_MSM_HOST_CHUNKS=<n> overrides the chunk count (1 = one plain copy).
    const int batch = std::max(1, cfg->batch_size);
    if (!cfg->are_scalars_on_device && scalars_v && bases_v && results_v && batch == 1 && std::max(1, cfg->precompute_factor) == 1) {
      static const int forced = getenv("ICICLE_HIP_MSM_HOST_CHUNKS") ? atoi(getenv("ICICLE_HIP_MSM_HOST_CHUNKS")) : 0;
      const int chunks = forced > 0 ? forced : (n >= (1 << 24) ? 4 : (n >= (1 << 22) ? 2 : 1));
      if (chunks > 1 && n >= chunks && virtual_device_slots() == 0) {
        opt.G = chunks;
        opt.max_slots = 1; // this is the single-GPU call, pipelined: never spread over the box's other devices
        return msm_multi_run<C>(scalars_v, bases_v, n, cfg, results_v, opt);
      }
    }
    return msm_run_single<C>(scalars_v, bases_v, n, cfg, results_v);
  }

} // namespace icicle_hip

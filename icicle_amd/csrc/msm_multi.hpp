// Multi-device MSM behind the unchanged <curve>_msm symbol.
//
// The reference has no multi-device MSM: its model is "one host thread per device, the caller splits the work"
// (docs/docs/start/architecture/multi-device.md:34-36,76). Backend-specific knobs travel in MSMConfig.ext
// (include/icicle/msm.h:52, include/icicle/backend/msm_config.h:4-16); this backend reads
//
//   "hip_num_devices"          int   G >= 1: cut the (scalar, base) pairs into G contiguous shards (the same cut as
//                                    icicle_amd/dist.py shard_range) and run them on min(G, visible GPUs) devices
//                                    starting at the calling thread's active device; with fewer GPUs than shards the
//                                    extra "logical" shards share a device (that is how the path is rehearsed on one GPU).
//   "hip_msm_exchange_buckets" bool  false (E1, default): every device finishes its shard; the 3*L-word partial
//                                    results are all-gathered (RCCL) and summed with k_proj_sum.
//                                    true (E2): the exchange happens in BUCKET space -- after accumulation device d
//                                    receives slice d of every peer's bucket array (grouped ncclSend / ncclRecv =
//                                    all-to-all over xGMI), adds the slices and reduces only its own slice; the
//                                    per-slice results then take the E1 exchange. E2 strong-scales the bucket
//                                    reduction at the price of moving W*2^(c-1)*sizeof(projective)/G bytes per peer.
//   "hip_force_rccl"           bool  use the RCCL exchange even with one physical device (size-1 communicator): test hook.
//
// One host thread and one stream per physical device (the reference's own model, applied inside the call). The
// call is synchronous with respect to the host whatever is_async says. Inputs may live on the host or on the
// calling device; shards for other devices are staged with hipMemcpy2DAsync (peer or host-to-device).
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
  };

  // E2 hook of one physical device: sums the bucket arrays of its logical shards, then (last shard) exchanges bucket
  // slices with the peers and leaves this device's slice, summed over all devices, in the caller's bucket array.
  template <class C>
  struct BucketExchange : MsmBucketHook<C> {
    using Proj = typename EC<C>::Proj;
    int nshards = 1, seen = 0, P = 1, p = 0;
    TempBuf acc, recv;
    void* comm = nullptr;
    PhaseGate* gate = nullptr; // passed (once) right before the slice exchange
    bool gate_passed = false;

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
      // this device's slice of every window: segments [lo, hi)
      const uint32_t lo = (uint32_t)((uint64_t)nseg * p / P), hi = (uint32_t)((uint64_t)nseg * (p + 1) / P);
      *seg_lo = lo;
      *nsegr = hi - lo;
      if (P == 1) return ICICLE_SUCCESS;
      const RcclApi* api = rccl_api();
      if (!api) return ICICLE_API_NOT_IMPLEMENTED;
      const size_t mine = (size_t)(hi - lo) * m; // buckets of my slice per window
      const bool have_recv = recv.alloc(std::max<size_t>(1, (size_t)P * tw * mine) * sizeof(Proj), st) == hipSuccess;
      gate_passed = true;
      if (gate && !gate->arrive(have_recv)) return ICICLE_ALLOCATION_FAILED; // a peer (or this device) cannot take part
      if (!have_recv) return ICICLE_ALLOCATION_FAILED;
      constexpr size_t PWORDS = sizeof(Proj) / 4;
      if (api->GroupStart() != 0) return ICICLE_COPY_FAILED;
      for (int q = 0; q < P; q++) {
        if (q == p) continue;
        const uint32_t qlo = (uint32_t)((uint64_t)nseg * q / P), qhi = (uint32_t)((uint64_t)nseg * (q + 1) / P);
        for (size_t w = 0; w < tw; w++) {
          if (qhi > qlo && api->Send(buckets + w * nb + (size_t)qlo * m, (size_t)(qhi - qlo) * m * PWORDS, RCCL_UINT32, q, comm, st) != 0) return ICICLE_COPY_FAILED;
          if (mine && api->Recv(recv.as<Proj>() + ((size_t)q * tw + w) * mine, mine * PWORDS, RCCL_UINT32, q, comm, st) != 0) return ICICLE_COPY_FAILED;
        }
      }
      if (api->GroupEnd() != 0) return ICICLE_COPY_FAILED;
      if (mine) {
        for (int q = 0; q < P; q++) {
          if (q == p) continue;
          for (size_t w = 0; w < tw; w++) {
            k_bucket_add<C><<<(unsigned)((mine + 127) / 128), 128, 0, st>>>(buckets + w * nb + (size_t)lo * m, recv.as<Proj>() + ((size_t)q * tw + w) * mine, mine);
            LAUNCH_CHECK("k_bucket_add(slice)", st);
          }
        }
      }
      return ICICLE_SUCCESS;
    }
  };

  template <class C>
  static icicle_error_t msm_multi_run(const void* scalars_v, const void* bases_v, int n, const icicle_msm_config_t* cfg, void* results_v, int G, bool exchange_buckets, bool force_rccl)
  {
    using E = EC<C>;
    constexpr int PW = 2 * E::N32, RW = 3 * E::N32, SW = 8;
    if (!results_v || n < 0 || G < 1) return ICICLE_INVALID_ARGUMENT;
    if (n > 0 && (!scalars_v || !bases_v)) return ICICLE_INVALID_POINTER;
    const int batch = std::max(1, cfg->batch_size);
    const int pf = std::max(1, cfg->precompute_factor);
    const bool shared = cfg->are_points_shared_in_batch || batch == 1;
    ICICLE_TRY(bind_current_device());
    const int home = current_device_id();
    // the shards run on this call's own streams: whatever the caller queued on config.stream (input copies) comes first
    HIP_TRY(hipStreamSynchronize((hipStream_t)cfg->stream), ICICLE_SYNCHRONIZATION_FAILED);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev), ICICLE_INVALID_DEVICE);
    const int P = std::max(1, std::min(G, ndev));
    if (exchange_buckets && n < G) exchange_buckets = false; // every shard must reach the accumulation step
    std::vector<int> devs(P);
    for (int p = 0; p < P; p++)
      devs[p] = (home + p) % ndev;

    // one window size for every shard: a precomputed base table fixes the doubling shift c*wpf, which
    // msm_precompute_bases derived from the FULL size (cpu_msm.hpp:455-480); and E2 adds bucket arrays element-wise
    icicle_msm_config_t sub = *cfg;
    sub.ext = nullptr;
    {
      const MsmPlan pl = make_plan(std::max(pf > 1 ? n : (n + G - 1) / G, 1), C::fr::NBITS, *cfg);
      sub.c = pl.c;
    }
    sub.are_scalars_on_device = sub.are_points_on_device = sub.are_results_on_device = true;
    sub.is_async = true;

    // "hip_force_rccl": take the RCCL exchange even with ONE physical device (communicator of size 1), so that the
    // loader, ncclCommInitAll and ncclAllGather bindings are exercised on a single-GPU box (tests)
    const bool use_rccl = P > 1 || force_rccl;
    std::vector<void*> comms(P, nullptr);
    if (use_rccl) {
      if (!rccl_api()) {
        fprintf(stderr, "[icicle_hip] hip_num_devices > 1 needs librccl.so (not loadable)\n");
        return ICICLE_API_NOT_IMPLEMENTED;
      }
      ICICLE_TRY(rccl_comms_for(devs, comms));
      ICICLE_TRY(bind_current_device());
    }

    PhaseGate gate_exchange, gate_gather;
    gate_exchange.expected = gate_gather.expected = P;
    std::vector<icicle_error_t> rcs(P, ICICLE_SUCCESS);
    auto worker = [&](int p) -> icicle_error_t {
      auto bail = [&](icicle_error_t e) { // nothing started on this device: tell the peers at both gates
        if (exchange_buckets && P > 1) (void)gate_exchange.arrive(false);
        if (use_rccl) (void)gate_gather.arrive(false);
        return e;
      };
      if (icicle_hip_set_device(devs[p]) != ICICLE_SUCCESS) return bail(ICICLE_INVALID_DEVICE);
      hipStream_t st = nullptr;
      if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return bail(ICICLE_STREAM_CREATION_FAILED);
      icicle_error_t rc;
      { // the hook's buffers are released in stream order: it must die before the stream does
      BucketExchange<C> hook;
      bool gather_gate_passed = false;
      rc = [&]() -> icicle_error_t {
        std::vector<int> mine;
        for (int g = p; g < G; g += P)
          mine.push_back(g);
        const int ns_p = (int)mine.size();
        TempBuf partials, devpart, gathered, fin;
        HIP_TRY(partials.alloc((size_t)ns_p * batch * RW * 4, st), ICICLE_ALLOCATION_FAILED);
        HIP_TRY(devpart.alloc((size_t)batch * RW * 4, st), ICICLE_ALLOCATION_FAILED);
        hook.nshards = ns_p, hook.P = P, hook.p = p, hook.comm = comms[p];
        hook.gate = (exchange_buckets && P > 1) ? &gate_exchange : nullptr;
        icicle_msm_config_t c2 = sub;
        c2.stream = st;
        for (int j = 0; j < ns_p; j++) {
          int lo, hi;
          shard_range(n, mine[j], G, &lo, &hi);
          const int ns = hi - lo;
          TempBuf d_sc, d_b;
          const uint32_t* sc = (const uint32_t*)scalars_v + (size_t)lo * SW;
          const uint32_t* bs = (const uint32_t*)bases_v + (size_t)lo * pf * PW;
          if (ns > 0) {
            const bool sc_direct = cfg->are_scalars_on_device && devs[p] == home && batch == 1;
            if (!sc_direct) { // [batch][ns] rows out of the caller's [batch][n]
              HIP_TRY(d_sc.alloc((size_t)batch * ns * SW * 4, st), ICICLE_ALLOCATION_FAILED);
              HIP_TRY(hipMemcpy2DAsync(d_sc.ptr(), (size_t)ns * SW * 4, sc, (size_t)n * SW * 4, (size_t)ns * SW * 4, batch, hipMemcpyDefault, st), ICICLE_COPY_FAILED);
              sc = d_sc.as<uint32_t>();
            }
            const int brows = shared ? 1 : batch;
            const bool b_direct = cfg->are_points_on_device && devs[p] == home && brows == 1;
            if (!b_direct) {
              const size_t row = (size_t)ns * pf * PW * 4;
              HIP_TRY(d_b.alloc(row * brows, st), ICICLE_ALLOCATION_FAILED);
              HIP_TRY(hipMemcpy2DAsync(d_b.ptr(), row, bs, (size_t)n * pf * PW * 4, row, brows, hipMemcpyDefault, st), ICICLE_COPY_FAILED);
              bs = d_b.as<uint32_t>();
            }
          }
          ICICLE_TRY(msm_run_single<C>(sc, bs, ns, &c2, partials.as<uint32_t>() + (size_t)j * batch * RW, exchange_buckets ? &hook : nullptr));
          HIP_TRY(hipStreamSynchronize(st), ICICLE_SYNCHRONIZATION_FAILED); // staged inputs are released in stream order, but keep shards serial
        }
        // per-device partial: E1 = sum over this device's shards; E2 = the last shard's result (it reduced the summed slice)
        if (exchange_buckets) {
          HIP_TRY(hipMemcpyAsync(devpart.ptr(), partials.as<uint32_t>() + (size_t)(ns_p - 1) * batch * RW, (size_t)batch * RW * 4, hipMemcpyDeviceToDevice, st), ICICLE_COPY_FAILED);
        } else {
          k_proj_sum<C><<<batch, 64, 0, st>>>(partials.as<uint32_t>(), ns_p, (size_t)batch * RW, devpart.as<uint32_t>());
          LAUNCH_CHECK("k_proj_sum(shards)", st);
        }
        const uint32_t* result = devpart.as<uint32_t>();
        if (use_rccl) { // "all-reduce" of partial sums: EC addition is not an RCCL reduce op -> all-gather + local projective sum
          const RcclApi* api = rccl_api();
          const bool have_bufs = gathered.alloc((size_t)P * batch * RW * 4, st) == hipSuccess && fin.alloc((size_t)batch * RW * 4, st) == hipSuccess;
          gather_gate_passed = true;
          if (!gate_gather.arrive(have_bufs) || !have_bufs) return ICICLE_ALLOCATION_FAILED;
          if (api->AllGather(devpart.ptr(), gathered.ptr(), (size_t)batch * RW, RCCL_UINT32, comms[p], st) != 0) return ICICLE_COPY_FAILED;
          k_proj_sum<C><<<batch, 64, 0, st>>>(gathered.as<uint32_t>(), P, (size_t)batch * RW, fin.as<uint32_t>());
          LAUNCH_CHECK("k_proj_sum(devices)", st);
          result = fin.as<uint32_t>();
        }
        if (p == 0) // the calling device delivers the result
          HIP_TRY(hipMemcpyAsync(results_v, result, (size_t)batch * RW * 4, cfg->are_results_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st), ICICLE_COPY_FAILED);
        HIP_TRY(hipStreamSynchronize(st), ICICLE_SYNCHRONIZATION_FAILED);
        return ICICLE_SUCCESS;
      }();
      // a thread that bailed out before a collective still reports to its gate, so that the peers skip it too
      if (exchange_buckets && P > 1 && !hook.gate_passed) (void)gate_exchange.arrive(false);
      if (use_rccl && !gather_gate_passed) (void)gate_gather.arrive(false);
      }
      (void)hipStreamSynchronize(st);
      (void)hipStreamDestroy(st);
      return rc;
    };

    if (P == 1) {
      rcs[0] = worker(0);
    } else {
      std::vector<std::thread> th;
      for (int p = 0; p < P; p++)
        th.emplace_back([&, p]() {
          try {
            rcs[p] = worker(p);
          } catch (...) {
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
    if (cfg->ext) {
      const ConfigExt* e = reinterpret_cast<const ConfigExt*>(cfg->ext);
      const int G = e->get_int("hip_num_devices", 0);
      if (G >= 1) return msm_multi_run<C>(scalars_v, bases_v, n, cfg, results_v, G, e->get_bool("hip_msm_exchange_buckets", false), e->get_bool("hip_force_rccl", false));
    }
    return msm_run_single<C>(scalars_v, bases_v, n, cfg, results_v);
  }

} // namespace icicle_hip

// ONE transform split over the device slots of a node behind the unchanged <field>_ntt symbol (north star: "large NTTs
// shard coefficients across the GPUs with all-to-all of NTT chunks over xGMI"). Included by
// ntt.hip inside namespace icicle_hip, after ntt_run / twiddle_rows_run.
//
// NTTConfig.ext {"hip_num_devices": G} with FEWER transforms in the batch than G (configs 2 and 4 have batch >= G and
// take the row-shard path of ntt_multi.hpp, which needs no exchange): every transform of the batch is cut over P = G
// device slots with the four-step (Bailey) decomposition N = N1 * N2, j = j1 N2 + j2, k = k1 + N1 k2 --
//     slot p holds rows j1 of the [N1 x N2] input matrix (a contiguous natural-order chunk)
//     exchange 1   -> columns j2 on the slot                      [N2/P x N1]
//     N1-point transforms over j1 (this backend's batched NTT), twiddle w_N^(j2 k1)
//     exchange 2   -> rows k1 on the slot                          [N1/P x N2]
//     N2-point transforms over j2                                  X[k1 + N1 k2] at [k1][k2]
//     exchange 3   -> natural-order chunks                         [N2/P x N1]
// An exchange is the transpose of a matrix distributed by rows: a local tiled transpose, a grouped ncclSend / ncclRecv with
// every peer at once (xGMI is point-to-point: all 7 links of a GPU carry one block each, where a ring would be bound by
// one link), and strided row copies into place. Same mathematics as icicle_amd/dist.py ntt_distributed (one process per
// GPU); this is the in-process form a Rust / Go / C++ caller of ntt() reaches. Natural order in and out (kNN), no coset;
// every other configuration, or a transform too small for the slots, runs on the calling device alone.

// out[c][r] = in[r][c]: [rows x cols] -> [cols x rows], 32 x 32 tiles through LDS
static __global__ __launch_bounds__(256) void k_transpose_u32(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t rows, uint32_t cols)
{
  __shared__ uint32_t tile[32][33];
  const uint32_t c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
  for (uint32_t i = ty; i < 32; i += 8)
    if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = in[(size_t)(r0 + i) * cols + c0 + tx];
  __syncthreads();
  for (uint32_t i = ty; i < 32; i += 8)
    if (c0 + i < cols && r0 + tx < rows) out[(size_t)(c0 + i) * rows + r0 + tx] = tile[tx][i];
}

template <class PR>
static icicle_error_t ntt_split_run(const uint32_t* input, int size, int dir, const icicle_ntt_config_u32_t* cfg, uint32_t* output, const DeviceSlots& ds, const SplitShape& sh, bool self_exchange = false)
{
  // self_exchange ("hip_force_rccl", one slot): the slot's own block travels through the communicator as well (a send to the
  // own rank inside the group), so the real ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd run on a single-GPU box
  const int P = sh.P, home = ds.home, batch = std::max(1, cfg->batch_size);
  const uint64_t n = (uint64_t)size, n1 = (uint64_t)1 << sh.a, n2 = (uint64_t)1 << sh.b;
  const uint64_t chunk = n / P; // words per slot
  const bool inverse = dir == ICICLE_NTT_INVERSE;
  uint32_t root = 0;
  {
    std::lock_guard<std::mutex> g(DomainStore<PR>::mtx());
    auto it = DomainStore<PR>::map().find(home);
    if (it == DomainStore<PR>::map().end() || !it->second.tw) return ICICLE_INVALID_ARGUMENT; // domain not initialised
    if (sh.logn > it->second.log_max) return ICICLE_INVALID_ARGUMENT;
    root = it->second.root;
  }
  if (!rccl_api()) {
    fprintf(stderr, "[icicle_hip] a transform split over devices needs librccl.so (not loadable)\n");
    return ICICLE_API_NOT_IMPLEMENTED;
  }
  RcclCommSet* cset = nullptr;
  ICICLE_TRY(rccl_comms_for(ds.devs, &cset));
  ICICLE_TRY(bind_current_device());
  std::unique_lock<std::mutex> comm_lock(cset->call_mtx);
  HIP_TRY(hipStreamSynchronize((hipStream_t)cfg->stream), ICICLE_SYNCHRONIZATION_FAILED);
  multi_stats().threaded_calls++;

  PhaseGate gate;
  gate.expected = P;
  // one more gate in front of EVERY exchange (3 per transform): a slot that failed after the first gate -- a launch, a
  // copy, a sub-transform -- leaves through its unused tickets ("failed"), and its peers skip the collective instead of
  // waiting in ncclRecv for a block that will never come
  const size_t ngates = (size_t)3 * batch;
  std::unique_ptr<PhaseGate[]> xgates(new PhaseGate[ngates]);
  for (size_t i = 0; i < ngates; i++)
    xgates[i].expected = P;
  std::vector<icicle_error_t> rcs(P, ICICLE_SUCCESS);
  auto worker = [&](int p) -> icicle_error_t {
    GateTicket ticket(&gate);
    std::vector<std::unique_ptr<GateTicket>> xt;
    for (size_t i = 0; i < ngates; i++)
      xt.emplace_back(new GateTicket(&xgates[i]));
    size_t xi = 0;
    if (test_failure_armed(p, 1)) return ICICLE_ALLOCATION_FAILED;
    ICICLE_TRY(icicle_hip_set_device(ds.devs[p]));
    const PeerRoute route = peer_route(ds.devs[p], home); // (common.h: direct xGMI copies, or hipMemcpyPeerAsync where peer access is refused)
    hipStream_t st = side_stream(200 + p); // long-lived per (device, slot): see msm_multi.hpp
    if (!st) return ICICLE_STREAM_CREATION_FAILED;
    icicle_error_t rc = [&]() -> icicle_error_t {
      const RcclApi* api = cset->api;
      void* comm = cset->comms[p];
      TempBuf bufA, bufB, bufT;
      bool ready = bufA.alloc(chunk * 4, st) == hipSuccess && bufB.alloc(chunk * 4, st) == hipSuccess && bufT.alloc(chunk * 4, st) == hipSuccess;
      if (ready && ds.devs[p] != home) { // the domain of the calling device, brought up here on first use
        icicle_ntt_init_domain_config_t ic{st, false, nullptr};
        ready = ntt_init_domain_run<PR>(&root, &ic) == ICICLE_SUCCESS;
        if (ready) {
          std::lock_guard<std::mutex> g(DomainStore<PR>::mtx());
          auto& d = DomainStore<PR>::map()[current_device_id()];
          ready = d.root == root; // (a peer that already holds a domain of another root: see ntt_multi_run)
          if (ready && d.owner < 0) d.owner = home;
        }
      }
      if (!ticket.arrive(ready) || !ready) return ICICLE_ALLOCATION_FAILED; // all slots or none enter the exchanges
      uint32_t *A = bufA.as<uint32_t>(), *B = bufB.as<uint32_t>(), *T = bufT.as<uint32_t>();
      icicle_ntt_config_u32_t sub = *cfg;
      sub.ext = nullptr;
      sub.stream = st;
      sub.are_inputs_on_device = sub.are_outputs_on_device = true;
      sub.is_async = true;
      sub.ordering = ICICLE_kNN;
      sub.coset_gen = 1;
      sub.columns_batch = false;

      // the transpose of a [R x C] matrix distributed by rows (this slot holds `rl` = R / P of them in `src`): `dst`
      // receives this slot's C / P rows of the [C x R] transpose
      auto exchange = [&](const uint32_t* src, uint32_t* dst, uint64_t R, uint64_t C) -> icicle_error_t {
        const uint64_t rl = R / P, cl = C / P;
        k_transpose_u32<<<dim3((unsigned)((C + 31) / 32), (unsigned)((rl + 31) / 32)), 256, 0, st>>>(src, T, (uint32_t)rl, (uint32_t)C); // T[c][r], block q = rows [q cl, (q + 1) cl)
        LAUNCH_CHECK("k_transpose_u32", st);
        uint32_t* recv = const_cast<uint32_t*>(src); // the source is dead once it is transposed
        const bool inject = xi == 1 && test_failure_armed(p, 2); // (rehearsal: this slot fails between two exchanges)
        if (!xt[xi++]->arrive(!inject) || inject) return ICICLE_COPY_FAILED; // every slot enters this exchange, or none does
        bool ok = api->GroupStart() == 0;
        for (int q = 0; q < P && ok; q++) {
          if (q == p && !self_exchange) continue;
          ok = api->Send(T + (size_t)q * cl * rl, cl * rl, RCCL_UINT32, q, comm, st) == 0;
          ok = ok && api->Recv(recv + (size_t)q * cl * rl, cl * rl, RCCL_UINT32, q, comm, st) == 0;
          multi_stats().exchange_messages++;
        }
        if (api->GroupEnd() != 0 || !ok) return ICICLE_COPY_FAILED;
        multi_stats().exchanged_bucket_bytes += (uint64_t)(self_exchange ? P : P - 1) * cl * rl * 4;
        for (int q = 0; q < P; q++) { // block of sender q: [cl][rl] -> columns [q rl, (q + 1) rl) of dst [cl][R]
          const uint32_t* blk = (q == p && !self_exchange) ? T + (size_t)q * cl * rl : recv + (size_t)q * cl * rl;
          HIP_TRY(hipMemcpy2DAsync(dst + (size_t)q * rl, R * 4, blk, rl * 4, rl * 4, cl, hipMemcpyDeviceToDevice, st), ICICLE_COPY_FAILED);
        }
        return ICICLE_SUCCESS;
      };

      for (int bi = 0; bi < batch; bi++) {
        const uint32_t* in_b = input + (size_t)bi * n + (size_t)p * chunk;
        uint32_t* out_b = output + (size_t)bi * n + (size_t)p * chunk;
        HIP_TRY(peer_copy2d(A, 0, in_b, 0, chunk * 4, 1, route, true, st), ICICLE_COPY_FAILED); // rows j1 of [n1 x n2]
        multi_stats().staged_scalar_bytes += chunk * 4;
        ICICLE_TRY(exchange(A, B, n1, n2)); // B: [n2/P][n1], row = global j2
        sub.batch_size = (int)(n2 / P);
        ICICLE_TRY(ntt_run<PR>(B, (int)n1, dir, &sub, B, 1)); // over j1 -> k1
        ICICLE_TRY(twiddle_rows_run<PR>(B, n2 / P, n1, (uint64_t)p * (n2 / P), (uint32_t)sh.logn, inverse, st));
        ICICLE_TRY(exchange(B, A, n2, n1)); // A: [n1/P][n2], row = global k1
        sub.batch_size = (int)(n1 / P);
        ICICLE_TRY(ntt_run<PR>(A, (int)n2, dir, &sub, A, 1)); // over j2 -> k2: X[k1 + n1 k2] at [k1][k2]
        ICICLE_TRY(exchange(A, B, n1, n2)); // B: [n2/P][n1] = X in natural order, chunk p
        HIP_TRY(peer_copy2d(out_b, 0, B, 0, chunk * 4, 1, route, false, st), ICICLE_COPY_FAILED);
      }
      HIP_TRY(hipStreamSynchronize(st), ICICLE_SYNCHRONIZATION_FAILED);
      return ICICLE_SUCCESS;
    }();
    (void)hipStreamSynchronize(st);
    ring_events_release();
    return rc;
  };
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
  ICICLE_TRY(icicle_hip_set_device(home));
  for (int p = 0; p < P; p++)
    if (rcs[p] != ICICLE_SUCCESS) return rcs[p];
  return ICICLE_SUCCESS;
}

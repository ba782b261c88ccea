// Host check of the rendezvous discipline of the multi-device entry points (icicle_amd/csrc/common.h PhaseGate /
// GateTicket): P workers, two gates in front of two "collectives"; one worker fails at a given stage. Every worker must
// come back (no dead-lock), and either all of them enter a collective or none does.
#include "../icicle_amd/csrc/common.h"
#include <atomic>
#include <thread>
using namespace icicle_hip;

// returns 0 if the run terminated with a consistent outcome; `victim` fails before gate `stage` (0 = nobody fails,
// 1 = at set-up: before both gates, 2 = right before the first gate, 3 = right before the second gate)
static int run(int P, int victim, int stage, bool first_gate_in_use)
{
  PhaseGate g1, g2;
  g1.expected = g2.expected = P;
  std::atomic<int> in1{0}, in2{0}, failed{0};
  auto worker = [&](int p) {
    GateTicket t1(first_gate_in_use ? &g1 : nullptr), t2(&g2);
    if (p == victim && stage == 1) {
      failed++;
      return; // tickets leave for this worker
    }
    if (first_gate_in_use) {
      const bool mine = !(p == victim && stage == 2);
      const bool all = t1.arrive(mine);
      if (!mine || !all) {
        failed++;
        return;
      }
      in1++;
    }
    const bool mine = !(p == victim && stage == 3);
    const bool all = t2.arrive(mine);
    if (!mine || !all) {
      failed++;
      return;
    }
    in2++;
  };
  std::vector<std::thread> th;
  for (int p = 0; p < P; p++)
    th.emplace_back(worker, p);
  for (auto& t : th)
    t.join();
  const bool any_fail = stage != 0;
  if (first_gate_in_use) {
    const int want1 = (stage == 1 || stage == 2) ? 0 : P;
    if (in1 != want1) return 1;
  }
  const int want2 = any_fail ? 0 : P;
  if (in2 != want2) return 2;
  if (any_fail && failed == 0) return 3;
  return 0;
}

extern "C" int gate_check(void)
{
  for (int P : {1, 2, 3, 8})
    for (int first = 0; first < 2; first++)
      for (int stage = 0; stage <= 3; stage++) {
        if (stage == 2 && !first) continue;
        for (int victim = 0; victim < P; victim++)
          for (int rep = 0; rep < 20; rep++)
            if (int rc = run(P, victim, stage, first != 0)) return 1000 * P + 100 * stage + 10 * first + rc;
      }
  return 0;
}

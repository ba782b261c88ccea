// Lane-native instantiations of the 31-bit NTT pass kernel (ntt_fast.hpp, LN = true): transforms interleaved word by
// word -- columns_batch (icicle/backend/cpu/include/ntt_cpu.h:250,274-275: element j of transform b at j * batch + b)
// and the quartic extension field (four base-field transforms per row, icicle/src/ntt.cpp:88-101). A translation unit of
// its own so that it compiles beside ntt.hip.
#include "ntt_fast.hpp"

namespace icicle_hip {
  pass_fn_t pick_pass_lanes_babybear(int s, bool dif, bool inv, bool coset, bool outrev) { return pick_pass_t<babybear_params, true>(s, dif, inv, coset, outrev); }
  pass_fn_t pick_pass_lanes_koalabear(int s, bool dif, bool inv, bool coset, bool outrev) { return pick_pass_t<koalabear_params, true>(s, dif, inv, coset, outrev); }
  pass_fn_t pick_pass_rn_lanes_babybear(int s, bool coset) { return pick_pass_rn_t<babybear_params, true>(s, 1, coset); }
  pass_fn_t pick_pass_rn_lanes_koalabear(int s, bool coset) { return pick_pass_rn_t<koalabear_params, true>(s, 1, coset); }
} // namespace icicle_hip

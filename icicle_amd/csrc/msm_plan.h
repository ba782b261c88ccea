// Host-side planning helpers of the MSM that carry no device code (also compiled by tests/plan_harness.cpp).
#pragma once
#include <algorithm>

namespace icicle_hip {

  constexpr int MSM_MAX_GROUPS = 16;

  // Window groups of the pipelined schedule (msm_impl.hpp): `tw` windows cut into at most `want` groups, group 0 holding
  // the HIGHEST windows. Group g = [glo[g], ghi[g]); the end groups are half as wide as the inner ones (a short first
  // group lets accumulation start early, a short last one leaves little to reduce at the end). Returns the group count.
  static inline int msm_window_groups(int tw, int want, int* glo, int* ghi)
  {
    int NG = std::max(1, std::min(std::min(want, MSM_MAX_GROUPS), tw / 2));
    if (tw < 4) NG = 1;
    const double unit = (double)tw / (NG <= 2 ? NG : NG - 1);
    double acc = 0;
    int hi = tw;
    for (int g = 0; g < NG; g++) {
      acc += (NG <= 2 || (g > 0 && g < NG - 1)) ? unit : unit / 2;
      int lo = g == NG - 1 ? 0 : std::max(0, tw - (int)(acc + 0.5));
      lo = std::min(lo, hi - 1);
      if (g < NG - 1) lo = std::max(lo, NG - 1 - g); // every later group keeps at least one window
      glo[g] = lo, ghi[g] = hi;
      hi = lo;
    }
    return NG;
  }

} // namespace icicle_hip

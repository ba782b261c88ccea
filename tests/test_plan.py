"""CPU: the NTT pass decomposition (icicle_amd/csrc/ntt_plan.h) compiled for the host -- every pass covers a row exactly
once and the last pass's scatter is a permutation, for every size up to 2^22 and both tile-width regimes."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_ntt_plan_covers_every_slot_once():
    so = os.path.join(HERE, "_build", "libplan.so")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    src = os.path.join(HERE, "plan_harness.cpp")
    hdrs = [os.path.join(HERE, "..", "icicle_amd", "csrc", h) for h in ("ntt_plan.h", "msm_plan.h")]
    if not os.path.exists(so) or max([os.path.getmtime(src)] + [os.path.getmtime(h) for h in hdrs]) > os.path.getmtime(so):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", src, "-o", so])
    lib = ctypes.CDLL(so)
    assert lib.plan_check(22) == 0
    assert lib.msm_groups_check() == 0  # window groups of the pipelined MSM schedule (msm_plan.h)
    assert lib.split_shape_check() == 0  # shapes of a transform split over device slots (ntt_plan.h)
    assert lib.msm_plan_check() == 0  # the window plan of msm() (msm_plan.h make_plan)
    # ECNTT: stage widths + the radix-2^r matrix-form index algebra, simulated mod a small prime against the O(n^2) definition
    assert lib.ecntt_plan_check() == 0

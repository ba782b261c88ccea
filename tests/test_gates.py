"""CPU: the rendezvous discipline of the multi-device entry points (PhaseGate / GateTicket, icicle_amd/csrc/common.h)
compiled for the host: a worker that fails before any gate leaves through its tickets' destructors, and nobody may wait
for it -- the first GPU rehearsal of round 3 dead-locked exactly there (the leaving worker waited at the second gate while
its peers waited for it at the first)."""
import ctypes
import os
import subprocess
import threading

HERE = os.path.dirname(os.path.abspath(__file__))


def test_gate_tickets_never_deadlock():
    so = os.path.join(HERE, "_build", "libgates.so")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    src = os.path.join(HERE, "gate_harness.cpp")
    hdr = os.path.join(HERE, "..", "icicle_amd", "csrc", "common.h")
    if not os.path.exists(so) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(so):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-pthread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", src, "-o", so])
    lib = ctypes.CDLL(so)
    out = {}
    t = threading.Thread(target=lambda: out.setdefault("rc", lib.gate_check()), daemon=True)
    t.start()
    t.join(timeout=120)
    assert not t.is_alive(), "dead-lock in the gate discipline"
    assert out["rc"] == 0, out

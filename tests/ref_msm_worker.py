"""TEST INFRASTRUCTURE: worker process of tests/refpool.py -- ONE reference CPU MSM (oracle/_ref, the unmodified reference backend).
`python tests/ref_msm_worker.py <scratch dir> <key>`: reads msm_<key>.json (curve + keyword arguments of oracle.ref.RefCurve.msm) and
in_<key>_{scalars,bases}.npy (mapped, not copied), writes out_<key>.npy and then the marker done_<key> ("<seconds in the reference
call> <wall-clock time of completion>"). A process of its own, so that the reference's bucket arrays are cleared, filled and merged in
an address space no other job faults pages into."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    d, key = sys.argv[1], sys.argv[2]
    from oracle import ref

    spec = json.load(open(os.path.join(d, f"msm_{key}.json")))
    scalars = np.load(os.path.join(d, f"in_{key}_scalars.npy"), mmap_mode="r")
    bases = np.load(os.path.join(d, f"in_{key}_bases.npy"), mmap_mode="r")
    t0 = time.time()
    out = ref.RefCurve(spec["curve"]).msm(scalars, bases, **spec["kw"])
    dt = time.time() - t0
    np.save(os.path.join(d, f"out_{key}.npy"), out)
    with open(os.path.join(d, f"done_{key}.tmp"), "w") as f:
        f.write(f"{dt:.3f} {time.time():.3f}")
    os.replace(os.path.join(d, f"done_{key}.tmp"), os.path.join(d, f"done_{key}"))
    print(f"{key}: {dt:.1f} s", flush=True)


if __name__ == "__main__":
    main()

"""TEST INFRASTRUCTURE: worker process of tests/refpool.py -- runs the reference CPU backend's <field>_ntt on the jobs of one lane,
in order. `python tests/ref_ntt_worker.py <scratch dir> <lane>`: reads lane_<lane>.json (a list of job specs) and in_<key>.npy,
writes out_<key>_<i>.npy and then the marker done_<key> (holding the seconds spent in the reference calls, domain init
excluded). The reference's twiddle domain is a per-process singleton (cpu_ntt_domain.h), which is why these legs cannot share
the pytest process with foreground tests that init / release it at other sizes."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    d, lane = sys.argv[1], sys.argv[2]
    from oracle import ref

    jobs = json.load(open(os.path.join(d, f"lane_{lane}.json")))
    fields, domain = {}, {}
    for j in jobs:
        rf = fields.setdefault(j["field"], ref.RefNttField(j["field"]))
        if domain.get(j["field"]) != j["logn"]:
            if j["field"] in domain:
                rf.release_domain()
            t0 = time.time()
            rf.init_domain(rf.get_root_of_unity(1 << j["logn"]))
            domain[j["field"]] = j["logn"]
            print(f"[{lane}] {j['field']} domain 2^{j['logn']}: {time.time() - t0:.1f} s", flush=True)
        x = np.load(os.path.join(d, f"in_{j['key']}.npy"), mmap_mode="r")
        x = np.ascontiguousarray(x).reshape(-1)
        calls = [(j["direction"], j["ordering"], j["coset_gen"])] + [tuple(c) for c in j["chain"]]
        spent = 0.0
        for i, (direction, ordering, coset) in enumerate(calls):
            t0 = time.time()
            x = rf.ntt(x, 1 << j["logn"], direction, batch=j["batch"], ordering=ordering, coset_gen=coset)
            spent += time.time() - t0
            np.save(os.path.join(d, f"out_{j['key']}_{i}.npy"), x)
        with open(os.path.join(d, f"done_{j['key']}.tmp"), "w") as f:
            f.write(f"{spent:.3f}")
        os.replace(os.path.join(d, f"done_{j['key']}.tmp"), os.path.join(d, f"done_{j['key']}"))
        print(f"[{lane}] {j['key']}: {spent:.1f} s", flush=True)


if __name__ == "__main__":
    main()

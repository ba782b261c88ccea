#!/usr/bin/env python3
"""profiles/rNN_pmc_traffic.txt (tools/pmc_traffic.sh) -> profiles/rNN_pmc_traffic.json, the file bench.py reads `roofline.traffic` from.
Stamps the SHA-256 of the kernel sources the counters were taken on (bench.py only attaches the traffic when the sources it runs
on hash the same: VERDICT r04 item 9).   usage: tools/pmc_json.py profiles/r05_pmc_traffic.txt > profiles/r05_pmc_traffic.json"""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MSM_SOURCES = ["msm_impl.hpp", "msm_plan.h", "ec.hpp", "bigfield.hpp", "mont_asm.hpp"]
NTT_SOURCES = ["ntt_fast.hpp", "ntt.hip", "ntt_plan.h", "smallfield.hpp"]


def sources_sha16(names):
    h = hashlib.sha256()
    for n in names:
        with open(os.path.join(ROOT, "icicle_amd", "csrc", n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def main(path):
    rows = {}
    for ln in open(path):
        m = re.match(r"(FETCH_SIZE|WRITE_SIZE)\s+(.*?)\s+launches\s+(\d+)\s+avg\s+(\d+) KiB", ln)
        if m:
            rows.setdefault(m.group(1), []).append((m.group(2), int(m.group(3)), int(m.group(4))))

    def pick(counter, pred):
        c = [r for r in rows.get(counter, []) if pred(r[0])]
        return max(c, key=lambda r: r[1]) if c else None

    acc_f, acc_w = pick("FETCH_SIZE", lambda n: "k_accumulate" in n), pick("WRITE_SIZE", lambda n: "k_accumulate" in n)
    gat_f = pick("FETCH_SIZE", lambda n: "k_diag_gather" in n)
    col_f = pick("FETCH_SIZE", lambda n: "bear_params, 4, 2, false" in n)
    col_w = pick("WRITE_SIZE", lambda n: "bear_params, 4, 2, false" in n)
    out = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, only --kernel-trace next to them; tools/pmc_traffic.sh), MI355X; raw per-kernel values: {os.path.relpath(path, ROOT)}; converted by tools/pmc_json.py",
           "msm_sources_sha16": sources_sha16(MSM_SOURCES), "ntt_sources_sha16": sources_sha16(NTT_SOURCES)}
    if acc_f and acc_w and gat_f:
        known_kib = (1 << 27) * 64 // 1024  # k_diag_gather: 2^27 random 64-byte gathers
        out["msm_bn254_2^26"] = {"kernel": "k_accumulate<bn254_g1>", "fetch_size_kb_raw": acc_f[2], "write_size_kb": acc_w[2],
                                 "fetch_correction": round(known_kib / gat_f[2], 4),
                                 "note": f"FETCH_SIZE calibrated in the same PMC run on k_diag_gather ({known_kib} KiB known, {gat_f[2]} KiB counted)"}
    if col_f and col_w:
        known = 4 * 1024 * 1024  # a pass of 2^24 x 64 x 4 B reads 4 GiB
        out["ntt_babybear_2^24x64_one_direction"] = {"kernel": "k_ntt_fast<babybear> x 3 passes", "fetch_size_kb_raw_per_pass": col_f[2],
                                                     "write_size_kb_per_pass": col_w[2], "passes": 3, "fetch_correction": round(known / col_f[2], 4),
                                                     "note": "FETCH_SIZE calibrated on the 4 GiB each pass provably loads (gfx950 counts wide coalesced streams at about half); WRITE_SIZE is exact"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])

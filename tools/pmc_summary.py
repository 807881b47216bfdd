#!/usr/bin/env python3
"""rocprofv3 --pmc CSVs (FETCH_SIZE pass, WRITE_SIZE pass) -> HBM-side bytes per launch and kernel.

bytes = 2 * FETCH_SIZE + WRITE_SIZE, both reported in KB; the factor 2 is the gfx950 correction of
MI355X_MICROARCH.md's HBM section (checked on k_moments: 4.0 GB tensor -> FETCH_SIZE 1,953,536 KB)."""
import csv
import json
import re
import sys
from collections import defaultdict


def per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] != counter:
                continue
            m = re.search(r"mcr::(?:pq::)?(k_\w+)(<[^(]*>)?\(", row["Kernel_Name"])
            if not m:
                continue
            name = m.group(1)
            if name == "k_merge":
                name = "k_fold_merge" if "true" in (m.group(2) or "") else "k_merge"
            if name == "k_diag_long_scan":
                name = "k_diag_long_scan"
            if name == "k_acov_seg":
                name = "k_acov_seg" if "true" in (m.group(2) or "") else "k_acov_more"
            if name == "k_diag_combine":
                name = "k_diag"
            tot[name] += float(row["Counter_Value"])
            cnt[name] += 1
    return {k: tot[k] / cnt[k] for k in tot}


def main():
    f = per_kernel(sys.argv[1], "FETCH_SIZE")
    w = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"_doc": "HBM-side bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (KB -> bytes); rocprofv3 --pmc, FETCH_SIZE and "
                   "WRITE_SIZE in separate passes, MCR_LANES=1. FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 "
                   "reports half of a coalesced streaming read; checked on k_moments: 4.0 GB tensor -> FETCH_SIZE "
                   "1,953,536 KB).",
           "4x10000x100-f64-pcn": {k: int(round((2 * f.get(k, 0.0) + w.get(k, 0.0)) * 1024)) for k in sorted(set(f) | set(w))}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""rocprofv3 --pmc CSVs with SQ counters -> per-kernel averages and the ratios DESIGN.md quotes.

    python tools/sq_summary.py pass_a.csv [pass_b.csv ...] > profiles/r02_sq_summary.json

SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md):
WAIT_ANY (parked at s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~ WAVE_CYCLES."""
import csv, json, re, sys
from collections import defaultdict


def short(name):
    m = re.search(r"mcr::(?:pq::|fft::)?(k_\w+)(<[^(]*>)?\(", name)
    if not m:
        return None
    n, t = m.group(1), m.group(2) or ""
    if n == "k_merge":
        n = "k_fold_merge" if "true" in t else "k_merge"
    if n == "k_acov_seg":
        n = "k_acov_seg" if "true" in t else "k_acov_more"
    return n


tot = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for path in sys.argv[1:]:
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            k = short(row["Kernel_Name"])
            if k is None:
                continue
            tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[k][row["Counter_Name"]] += 1
out = {}
for k in sorted(tot):
    a = {c: tot[k][c] / cnt[k][c] for c in tot[k]}
    r = {}
    wc = a.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
            if c in a:
                r[c.replace("SQ_", "").lower() + "_share_of_wave_cycles"] = round(a[c] / wc, 4)
        if "SQ_INSTS_VALU" in a:
            r["valu_insts_per_wave_quad_cycle"] = round(a["SQ_INSTS_VALU"] / wc, 4)
    if a.get("SQ_INSTS_LDS"):
        if "SQ_LDS_BANK_CONFLICT" in a:
            r["lds_bank_conflict_cycles_per_lds_inst"] = round(a["SQ_LDS_BANK_CONFLICT"] / a["SQ_INSTS_LDS"], 3)
        if "SQ_LDS_IDX_ACTIVE" in a:
            r["lds_array_cycles_per_lds_inst"] = round(a["SQ_LDS_IDX_ACTIVE"] / a["SQ_INSTS_LDS"], 3)
    out[k] = {"per_launch": {c: round(v, 1) for c, v in sorted(a.items())}, "ratios": r}
print(json.dumps(out, indent=1))

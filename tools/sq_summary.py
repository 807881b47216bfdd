#!/usr/bin/env python3
"""rocprofv3 --pmc CSVs with SQ counters -> per-kernel averages and the ratios DESIGN.md quotes.

    python tools/sq_summary.py pass_a.csv [pass_b.csv ...] > profiles/r02_sq_summary.json

SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md):
WAIT_ANY (parked at s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~ WAVE_CYCLES.

Utilisation is normalised by the counters themselves, not by the nominal residency (What's weak 2):
  simd_quad_cycles   = SQ_BUSY_CYCLES / 32 (the counter is summed over the 32 shader engines) / 4 * 1024 SIMDs
  avg_resident_waves = SQ_WAVE_CYCLES / simd_quad_cycles          (per SIMD, over the kernel's busy time)
  valu_issue_share   = SQ_INSTS_VALU / simd_quad_cycles           (1.0 = one VALU instruction per SIMD every 4 cycles,
                       which is what ONE wave can issue; the SIMD-32 pipe itself takes a 32-bit instruction every 2
                       cycles when two or more waves have one ready, so 2.0 would be its ceiling for 32-bit work)
  kernel_us_at_2p4GHz = SQ_BUSY_CYCLES / 32 / 2400"""
N_SE, N_SIMD = 32, 1024
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
    if a.get("SQ_BUSY_CYCLES"):
        sqc = a["SQ_BUSY_CYCLES"] / N_SE / 4.0 * N_SIMD
        r["kernel_us_at_2p4GHz"] = round(a["SQ_BUSY_CYCLES"] / N_SE / 2400.0, 2)
        if wc:
            r["avg_resident_waves_per_simd"] = round(wc / sqc, 3)
        if "SQ_INSTS_VALU" in a:
            r["valu_issue_share_of_simd_quad_cycles"] = round(a["SQ_INSTS_VALU"] / sqc, 3)
        if "SQ_INSTS_LDS" in a:
            r["lds_insts_per_simd_quad_cycle"] = round(a["SQ_INSTS_LDS"] / sqc, 4)
    out[k] = {"per_launch": {c: round(v, 1) for c, v in sorted(a.items())}, "ratios": r}
print(json.dumps(out, indent=1))

#!/usr/bin/env python3
"""Copies the artefacts of tools/collect_profiles.sh + tools/collect_sq.sh from gpurun_out/ into profiles/ (SQ counter
CSVs trimmed to two launches per kernel) and prints the numbers DESIGN.md quotes.  usage: install_profiles.py <tag>"""
import collections
import csv
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
tag = sys.argv[1] if len(sys.argv) > 1 else "r02_v4"
prof, sq, out = ROOT / "gpurun_out" / "prof", ROOT / "gpurun_out" / "sq", ROOT / "profiles"
for f in sorted(prof.glob(f"{tag}_*")):
    if f.suffix != ".err":
        shutil.copy(f, out / f.name)
if (prof / "pmc_traffic.json").exists():
    shutil.copy(prof / "pmc_traffic.json", out / "pmc_traffic.json")
for n in ("a", "b"):
    src = sq / f"{tag}_pmc_sq_{n}_c1.csv"
    if not src.exists():
        continue
    rows = list(csv.DictReader(open(src)))
    per = collections.defaultdict(list)
    for r in rows:
        per[r["Kernel_Name"]].append(r["Dispatch_Id"])
    keep = {(k, i) for k, v in per.items() for i in sorted(set(v), key=int)[:2]}
    with open(out / src.name, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()), quoting=csv.QUOTE_ALL)
        w.writeheader()
        w.writerows(r for r in rows if (r["Kernel_Name"], r["Dispatch_Id"]) in keep)
if (sq / f"{tag}_sq_summary.json").exists():
    shutil.copy(sq / f"{tag}_sq_summary.json", out / f"{tag}_sq_summary.json")

for f in (f"{tag}_c1_bench.json", f"{tag}_c1_bench_driver_flags.json", f"{tag}_corpus_bench.json", f"{tag}_stress_16GB_bench.json"):
    d = json.loads((out / f).read_text().strip().splitlines()[-1])
    print(f, round(d["ms_per_step"], 4), "ms/step", round(d["value"] / 1e9, 2), "G pd/s", d.get("validated"))
    if "kernels" in d:
        print("  ", {k: v["avg_us"] for k, v in d["kernels"].items()})
    if d.get("moments_roofline"):
        print("   moments GB/s", round(d["moments_roofline"]["achieved"]), d["hbm_probe"])
    if "roofline" in d:
        print("   roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"], 4), round(d["roofline"]["avg_launch_us"], 1), "us")
print((out / f"{tag}_stress_16GB_pipeline.json").read_text().splitlines()[-1])
print(json.load(open(out / "pmc_traffic.json"))["4x10000x100-f64-pcn"])
d = json.load(open(out / f"{tag}_sq_summary.json"))
for k in ("k_tile_sort", "k_bucket_merge", "k_fold_merge", "k_acov_seg"):
    v = d[k]["per_launch"]
    print(f"{k:16s} VALU/pd={v['SQ_INSTS_VALU'] * 64 / 4e6:6.1f} LDS/pd={v['SQ_INSTS_LDS'] * 64 / 4e6:5.1f} "
          f"act_valu/wave={v['SQ_ACTIVE_INST_VALU'] / v['SQ_WAVE_CYCLES']:.3f} wait_any={v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']:.2f} "
          f"wait_inst={v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES']:.2f} (lds {v['SQ_WAIT_INST_LDS'] / v['SQ_WAVE_CYCLES']:.2f}) "
          f"ldscyc/inst={v['SQ_LDS_IDX_ACTIVE'] / max(v['SQ_INSTS_LDS'], 1):.1f} conf/inst={v['SQ_LDS_BANK_CONFLICT'] / max(v['SQ_INSTS_LDS'], 1):.1f}")

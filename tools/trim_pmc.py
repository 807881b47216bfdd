#!/usr/bin/env python3
"""Keeps the first N dispatches per kernel of a rocprofv3 --pmc counter CSV (the raw file of a bench run is 10 - 100 MB;
gpurun brings back at most 64 MiB).  usage: trim_pmc.py <csv> [N=20]   (in place)"""
import collections
import csv
import sys

path, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = list(csv.DictReader(open(path, newline="")))
if rows:
    per = collections.defaultdict(list)
    for r in rows:
        per[r["Kernel_Name"]].append(int(r["Dispatch_Id"]))
    keep = {(k, i) for k, v in per.items() for i in sorted(set(v))[:n]}
    with open(path, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()), quoting=csv.QUOTE_ALL)
        w.writeheader()
        w.writerows(r for r in rows if (r["Kernel_Name"], int(r["Dispatch_Id"])) in keep)

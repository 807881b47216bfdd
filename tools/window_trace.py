#!/usr/bin/env python3
"""Timeline of a short timed window from a `rocprofv3 --kernel-trace` CSV: per call (11 consecutive launches of a
stream) start / end, and how much of the window the GPU had 1, 2, 3, 4 calls' kernels running.
usage: window_trace.py <kernel_trace.csv> [steps]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40],
              r.get("Stream_Id") or r.get("Queue_Id")) for r in rows), key=lambda x: x[0])
tile = [k for k in ks if "k_tile_sort" in k[2]]
tile = tile[-steps:]                      # the last window's calls
t0 = tile[0][0]
sel = [k for k in ks if k[0] >= t0]
t1 = max(k[1] for k in sel)
print(f"window: {len(tile)} calls, {(t1 - t0) / 1e3:.1f} us, {(t1 - t0) / 1e3 / len(tile):.1f} us/call")
ev = []
for s, e, _, _ in sel:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, last, hist = 0, t0, defaultdict(int)
for t, d in ev:
    hist[depth] += t - last
    last = t
    depth += d
for d in sorted(hist):
    print(f"  {d} kernels running: {hist[d] / 1e3:8.1f} us ({100 * hist[d] / (t1 - t0):.1f} %)")
print("first kernels (us since window start):")
for s, e, n, q in sel[:30]:
    print(f"  {(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f}  q={q}  {n}")
print("last kernels:")
for s, e, n, q in sel[-14:]:
    print(f"  {(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f}  q={q}  {n}")

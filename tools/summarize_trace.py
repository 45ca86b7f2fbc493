#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV by (kernel, grid size): calls, total / average / min / max duration.
usage: tools/summarize_trace.py <kernel_trace.csv> [min_total_us]"""
import csv
import sys
from collections import defaultdict

rows = defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        rows[(name, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["VGPR_Count"]),
              int(r["LDS_Block_Size"]))].append(dur)
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
tot_all = sum(sum(v) for v in rows.values())
print(f"{'kernel':60s} {'blocks':>7s} {'vgpr':>5s} {'lds':>6s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s}")
for (name, blocks, vg, lds), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) < thr:
        continue
    print(f"{name[:60]:60s} {blocks:7d} {vg:5d} {lds:6d} {len(v):6d} {sum(v):12.1f} {sum(v)/len(v):10.1f} {min(v):10.1f} {max(v):10.1f} {100*sum(v)/tot_all:6.2f}")
print(f"total kernel time {tot_all/1e3:.2f} ms")

"""Print the per-stream kernel timeline of one fit from a rocprofv3 kernel trace CSV (diagnostics):
python tools/timeline.py <kernel_trace.csv> [fit_index_from_end=2] [t_lo_us t_hi_us]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
idx = [i for i, r in enumerate(rows) if "gram_fused" in r["Kernel_Name"]]
i0 = idx[-k]
t0 = int(rows[i0]["End_Timestamp"])
lo = float(sys.argv[3]) if len(sys.argv) > 3 else -1e12
hi = float(sys.argv[4]) if len(sys.argv) > 4 else 1e12
# walk back to the start of this fit (previous gram end) so that the pre-Gram side-stream work is shown too
start = idx[-k - 1] + 1 if len(idx) > k else 0
for r in rows[start:]:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3
    e = (int(r["End_Timestamp"]) - t0) / 1e3
    if s < lo or s > hi:
        if s > hi: break
        continue
    name = r["Kernel_Name"].split("(")[0][-44:]
    print(f"{r['Queue_Id']:>3} {s:10.1f} {e:10.1f} {e - s:8.1f} {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):6d} {name}")
    if "gram_fused" in r["Kernel_Name"] and int(r["End_Timestamp"]) > t0:
        break

#!/usr/bin/env python3
"""Copy / condense what tools/r03_collect_{a,b}.sh left under gpurun_out/r03 into profiles/r03_* (the tracked evidence).
Run in the repo after the two collection calls:  python3 tools/r03_publish.py"""
import csv, json, os, shutil, subprocess, sys
from collections import defaultdict
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(R, "gpurun_out", "r03")
P = os.path.join(R, "profiles")


def cp(src, dst):
    if os.path.exists(os.path.join(O, src)):
        shutil.copyfile(os.path.join(O, src), os.path.join(P, dst))
    else:
        print("missing", src)


for src, dst in (("bench_line.json", "r03_bench_line.json"), ("bench_line_profiled.json", "r03_bench_line_profiled.json"),
                 ("bench_prof/run_kernel_stats.csv", "r03_bench_kernel_stats.csv"), ("gpu_tests.log", "r03_gpu_tests.log"),
                 ("gpu_tests_cloth.log", "r03_gpu_tests_cloth.log"), ("cloth_units.txt", "r03_cloth_units.txt"),
                 ("lockstep_bench.log", "r03_cv_lockstep_bench.log"), ("gram_bench.log", "r03_gram_bench.log"),
                 ("smoke.log", "r03_smoke.log")):
    cp(src, dst)
for i in (1, 2, 3):
    cp(f"pmc_gram_{i}/run_counter_collection.csv", f"r03_pmc_gram_{i}.csv")
trace = os.path.join(O, "bench_prof", "run_kernel_trace.csv")
if os.path.exists(trace):
    with open(os.path.join(P, "r03_bench_trace_summary.txt"), "w") as f:
        subprocess.run([sys.executable, os.path.join(R, "tools", "summarize_trace.py"), trace], stdout=f, check=False)
    with open(os.path.join(P, "r03_fit_timeline.txt"), "w") as f:
        subprocess.run([sys.executable, os.path.join(R, "tools", "timeline.py"), trace, "2", "-40000", "16000"], stdout=f, check=False)


def counters(path, kernel_substr, skip_first=True):
    """per-launch averages (and per-launch durations) of the kernels whose name contains kernel_substr"""
    by = defaultdict(lambda: defaultdict(float))
    dur = {}
    if not os.path.exists(path):
        return {}, []
    for r in csv.DictReader(open(path)):
        if kernel_substr in r["Kernel_Name"]:
            by[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
            dur[int(r["Dispatch_Id"])] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6
    out = {}
    for c, d in by.items():
        ids = sorted(d)
        if skip_first and len(ids) > 1:
            ids = ids[1:]
        out[c] = sum(d[i] for i in ids) / len(ids)
    ids = sorted(dur)
    return out, [dur[i] for i in (ids[1:] if skip_first and len(ids) > 1 else ids)]


g1, _ = counters(os.path.join(P, "r03_pmc_gram_1.csv"), "gram_fused")
g2, _ = counters(os.path.join(P, "r03_pmc_gram_2.csv"), "gram_fused")
g3, d3 = counters(os.path.join(P, "r03_pmc_gram_3.csv"), "gram_fused")
if g1 and g2 and g3:
    n, m, d, p = 100000, 2000, 384, 6
    fetch = g1["FETCH_SIZE"] * 1024.0 * 2.0  # gfx950: FETCH_SIZE counts 64 B per 128-B request of wide coalesced reads
    write = g2["WRITE_SIZE"] * 1024.0
    cyc = g3["GRBM_GUI_ACTIVE"] / 8.0        # the counter sums the 8 XCDs
    dur = sum(d3) / len(d3)
    out = {"n": n, "m": m, "d": d, "p": p,
           "kernel": "nk::gram_fused_f64_kernel (fused Gram launch, 4608 workgroups, hand-scheduled assembly k loop)",
           "FETCH_SIZE_kb": g1["FETCH_SIZE"], "WRITE_SIZE_kb": g2["WRITE_SIZE"],
           "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for wide (16 B/lane) coalesced reads -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact",
           "hbm_bytes_per_launch": fetch + write,
           "algorithmic_bytes_per_launch": 8.0 * (n * (2 * m + p + 2) + n * d) + 2 * 4608 * 128 * 128 * 8.0,
           "TCC_HIT_sum": g2.get("TCC_HIT_sum"), "TCC_MISS_sum": g2.get("TCC_MISS_sum"),
           "l2_hit_rate": g2["TCC_HIT_sum"] / (g2["TCC_HIT_sum"] + g2["TCC_MISS_sum"]),
           "SQ_VALU_MFMA_BUSY_CYCLES": g3["SQ_VALU_MFMA_BUSY_CYCLES"], "GRBM_GUI_ACTIVE": g3["GRBM_GUI_ACTIVE"],
           "mfma_busy_fraction": g3["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc),
           "launch_ms_under_profiler": dur, "shader_clock_ghz": cyc / (dur * 1e-3) / 1e9,
           "SQ_LDS_BANK_CONFLICT": g3.get("SQ_LDS_BANK_CONFLICT"),
           "waves_parked_fraction": g3["SQ_WAIT_ANY"] / g3["SQ_WAVE_CYCLES"] if "SQ_WAIT_ANY" in g3 else None,
           "note": "round 3 (profiles/r03_pmc_gram_{1,2,3}.csv; averages over the launches after the cold first one); separate --pmc passes "
                   "(FETCH_SIZE | WRITE_SIZE+TCC | SQ) on tools/gram_bench.py (tools/r03_collect_b.sh); memory-side counters include "
                   "Infinity-Cache hits.  Round 2 (compiler-scheduled k loop): MFMA busy 0.91, same traffic."}
    json.dump(out, open(os.path.join(P, "gram_traffic.json"), "w"), indent=1)
    print("gram: busy %.3f, traffic %.3e B, L2 hit %.3f, clock %.2f GHz, %.2f ms" %
          (out["mfma_busy_fraction"], out["hbm_bytes_per_launch"], out["l2_hit_rate"], out["shader_clock_ghz"], dur))

#!/usr/bin/env python3
"""Copy / condense what tools/r02_collect_{a,b}.sh left under gpurun_out/r02 into profiles/r02_* (the tracked evidence).
Run in the repo after the two collection calls:  python3 tools/r02_publish.py"""
import csv, json, os, shutil, subprocess, sys
from collections import defaultdict
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(R, "gpurun_out", "r02")
P = os.path.join(R, "profiles")

def cp(src, dst):
    if os.path.exists(os.path.join(O, src)):
        shutil.copyfile(os.path.join(O, src), os.path.join(P, dst))
    else:
        print("missing", src)

for src, dst in (("bench_line.json", "r02_bench_line.json"), ("bench_line_profiled.json", "r02_bench_line_profiled.json"),
                 ("bench_prof/run_kernel_stats.csv", "r02_bench_kernel_stats.csv"), ("gpu_tests.log", "r02_gpu_tests.log"),
                 ("lockstep_bench.log", "r02_cv_lockstep_bench.log"), ("cv_bench.log", "r02_cv_bench.log"),
                 ("rollout_bench.log", "r02_rollout_bench.log"), ("chain_mw_probe.log", "r02_chain_mw_probe.log"),
                 ("chain_mw_probe_stepwise.log", "r02_chain_mw_probe_stepwise.log"),
                 ("host_fit_bench.log", "r02_host_fit_bench.log"), ("soak.log", "r02_soak.log"),
                 ("shape_sweep.log", "r02_shape_sweep.log"), ("jacobi_bench.log", "r02_jacobi_bench.log"),
                 ("exact_kernel_bench.log", "r02_exact_kernel_bench.log"), ("kmat_epilogue_probe.log", "r02_kmat_epilogue_probe.log"),
                 ("rollout_fuzz.log", "r02_rollout_fuzz.log")):
    cp(src, dst)
for i in (1, 2, 3):
    cp(f"pmc_gram_{i}/run_counter_collection.csv", f"r02_pmc_gram_{i}.csv")
for w, tag in (("duffing", "fetch"), ("cloth", "fetch")):
    cp(f"kmat_{w}_p1/run_counter_collection.csv", f"r02_pmc_kmat_{w}_fetch.csv")
    cp(f"kmat_{w}_p2/run_counter_collection.csv", f"r02_pmc_kmat_{w}_write.csv")
for w in ("duffing", "duffing_rbf", "duffing_linear", "cloth"):
    cp(f"kmat_{w}_t/run_kernel_stats.csv", f"r02_kmat_{w}_kernel_stats.csv")

trace = os.path.join(O, "bench_prof", "run_kernel_trace.csv")
if os.path.exists(trace):
    with open(os.path.join(P, "r02_bench_trace_summary.txt"), "w") as f:
        subprocess.run([sys.executable, os.path.join(R, "tools", "summarize_trace.py"), trace], stdout=f, check=False)
    with open(os.path.join(P, "r02_fit_timeline.txt"), "w") as f:
        subprocess.run([sys.executable, os.path.join(R, "tools", "timeline.py"), trace, "2", "-40000", "16000"], stdout=f, check=False)

def counters(path, kernel_substr, skip_first=True):
    """average counter values per launch of the kernels whose name contains kernel_substr"""
    by = defaultdict(lambda: defaultdict(float))
    if not os.path.exists(path):
        return {}
    for r in csv.DictReader(open(path)):
        if kernel_substr in r["Kernel_Name"]:
            by[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    out = {}
    for c, d in by.items():
        ids = sorted(d)
        if skip_first and len(ids) > 1:
            ids = ids[1:]
        out[c] = sum(d[i] for i in ids) / len(ids)
    return out

# fused Gram launch
g1 = counters(os.path.join(P, "r02_pmc_gram_1.csv"), "gram_fused")
g2 = counters(os.path.join(P, "r02_pmc_gram_2.csv"), "gram_fused")
g3 = counters(os.path.join(P, "r02_pmc_gram_3.csv"), "gram_fused")
if g1 and g2 and g3:
    n, m, d, p = 100000, 2000, 384, 6
    fetch_kb, write_kb = g1["FETCH_SIZE"], g2["WRITE_SIZE"]
    hbm = (2.0 * fetch_kb + write_kb) * 1024.0
    old = json.load(open(os.path.join(P, "gram_traffic.json")))
    old.update({"FETCH_SIZE_kb": fetch_kb, "WRITE_SIZE_kb": write_kb, "hbm_bytes_per_launch": hbm,
                "TCC_HIT_sum": g2["TCC_HIT_sum"], "TCC_MISS_sum": g2["TCC_MISS_sum"],
                "l2_hit_rate": g2["TCC_HIT_sum"] / (g2["TCC_HIT_sum"] + g2["TCC_MISS_sum"]),
                "SQ_VALU_MFMA_BUSY_CYCLES": g3["SQ_VALU_MFMA_BUSY_CYCLES"], "GRBM_GUI_ACTIVE": g3["GRBM_GUI_ACTIVE"],
                "mfma_busy_fraction": g3["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * g3["GRBM_GUI_ACTIVE"] / 8.0),
                "SQ_LDS_BANK_CONFLICT": g3["SQ_LDS_BANK_CONFLICT"]})
    json.dump(old, open(os.path.join(P, "gram_traffic.json"), "w"), indent=1)
    print("gram: hbm bytes/launch %.3e, L2 hit %.3f, MFMA busy %.3f" % (hbm, old["l2_hit_rate"], old["mfma_busy_fraction"]))

# distance kernel
def avg_us(stats_csv, substr):
    if not os.path.exists(stats_csv):
        return None, None
    for r in csv.DictReader(open(stats_csv)):
        if substr in r["Name"]:
            return r["Name"].split("(")[0], float(r["AverageNs"]) / 1e3
    return None, None
summ = {}
shapes = {"duffing": (69900, 200, 2), "duffing_rbf": (69900, 200, 2), "duffing_linear": (69900, 200, 2), "cloth": (30300, 500, 192)}
for w, (n, m, d) in shapes.items():
    name, us = avg_us(os.path.join(P, f"r02_kmat_{w}_kernel_stats.csv"), "kmat_")
    if us is None:
        continue
    alg = (n * m + n * d + m * d) * 8.0
    e = {"kernel": name, "avg_us": us, "algorithmic_bytes": alg, "achieved_TBps": alg / (us * 1e-6) / 1e12,
         "frac_of_8TBps": alg / (us * 1e-6) / 8e12}
    f = counters(os.path.join(P, f"r02_pmc_kmat_{w}_fetch.csv"), "kmat_", skip_first=False)
    wv = counters(os.path.join(P, f"r02_pmc_kmat_{w}_write.csv"), "kmat_", skip_first=False)
    if f and wv:
        e.update({"FETCH_SIZE_kb": f["FETCH_SIZE"], "WRITE_SIZE_kb": wv["WRITE_SIZE"],
                  "hbm_bytes": (2.0 * f["FETCH_SIZE"] + wv["WRITE_SIZE"]) * 1024.0})
    summ[w] = e
    print(w, "%.1f us  %.2f TB/s" % (us, e["achieved_TBps"]))
summ["note"] = ("tools/kmat_bench.py under rocprofv3 (--kernel-trace --stats for durations; separate --pmc passes for FETCH_SIZE and "
                "WRITE_SIZE; FETCH doubled per the gfx950 correction). duffing*: n=69900 m=200 d=2 (benchmark_lqr_classic.py shape) "
                "with Matern-5/2 / RBF / linear epilogues; cloth: n=30300 m=500 d=192 RBF (fp64 VALU bound: 2.9e9 pair-dims). "
                "Regenerated by tools/r02_publish.py.")
json.dump(summ, open(os.path.join(P, "r02_kmat_summary.json"), "w"), indent=1)

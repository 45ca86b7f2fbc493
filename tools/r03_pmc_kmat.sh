#!/bin/bash
# VALU-side PMC evidence for the write-bound distance kernel (nk::kmat_flat_kernel) at the Duffing shape: instructions per
# output entry and how busy the vector ALUs are (one counter group per run, program directly after `--`)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for w in duffing duffing_rbf duffing_linear; do
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/kmatv_${w} -o run -- python3 $R/tools/kmat_bench.py $w 5 > $R/gpurun_out/kmatv_${w}.log 2>&1 || { tail -5 $R/gpurun_out/kmatv_${w}.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, json, os
R = os.environ['GRAFT_REPO_ROOT']
out = {}
n, m = 69900, 200
for w in ('duffing', 'duffing_rbf', 'duffing_linear'):
    f = glob.glob(f'{R}/gpurun_out/kmatv_{w}/**/*counter_collection.csv', recursive=True)[0]
    per = {}
    for r in csv.DictReader(open(f)):
        if 'kmat_flat' in r['Kernel_Name']:
            d = per.setdefault(r['Dispatch_Id'], {'kernel': r['Kernel_Name'].split('(')[0]})
            d[r['Counter_Name']] = float(r['Counter_Value'])
            d['dur_us'] = (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-3
    last = list(per.values())[-1]
    cyc = last['GRBM_GUI_ACTIVE'] / 8.0                      # shader cycles of the dispatch (the counter sums the 8 XCDs)
    out[w] = dict(kernel=last['kernel'], dur_us_profiled=last['dur_us'], entries=n * m,
                  valu_insts_per_wave_entry=last['SQ_INSTS_VALU'] * 64.0 / (n * m),   # wave instructions x 64 lanes / entries
                  valu_busy=last['SQ_ACTIVE_INST_VALU'] * 4.0 / (1024.0 * cyc),       # quad-cycles -> cycles, 1024 SIMDs
                  wave_cycles_waiting_issue=last['SQ_WAIT_INST_ANY'] / last['SQ_WAVE_CYCLES'],
                  wave_cycles_parked=last['SQ_WAIT_ANY'] / last['SQ_WAVE_CYCLES'],
                  clock_ghz=cyc / (last['dur_us'] * 1e-6) / 1e9, raw={k: v for k, v in last.items() if k not in ('kernel',)})
    print(w, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in out[w].items() if k != 'raw'})
json.dump(out, open(f'{R}/gpurun_out/r03_kmat_valu.json', 'w'), indent=1)
PY

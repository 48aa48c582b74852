#!/bin/bash
# kernel trace of the secondary configurations (tools/bench_configs.py)
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$PWD
python tools/bench_configs.py 2>&1 | grep config
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_cfg -o cfg -- python $REPO/tools/bench_configs.py > $REPO/gpurun_out/prof_cfg.log 2>&1
cd $REPO; python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/prof_cfg/cfg_kernel_trace.csv")))
# last complete config-3 step: kernels after the last k_bin_count (cell mode only in config 3)
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "k_bin_count" in n or "k_cell_count" in n or "k_bin" in n]
last = max(i for i, n in enumerate(names) if "k_nbr_cell" in n)
start = max(i for i in range(last) if "k_nbr_cell" in names[i]) if sum("k_nbr_cell" in n for n in names) > 1 else 0
seg = rows[start + 1:last + 1]
t0 = int(seg[0]["Start_Timestamp"]); t1 = int(seg[-1]["End_Timestamp"])
print(f"one config-3 step: {len(seg)} kernels, span {(t1 - t0) / 1e3:.1f} us, busy {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg) / 1e3:.1f} us")
for r in seg:
    print(f"  {(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} us  +{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f}  {r['Kernel_Name'][:70]}")
PY

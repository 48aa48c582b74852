#!/bin/bash
# kernel timeline of the last eager config-2 step (5248 atoms in 256 molecules): start offsets, durations, gaps
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$PWD
rm -rf gpurun_out/c2
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/c2 -o c2 -- python $REPO/tools/cfg2_trace.py > $REPO/gpurun_out/c2.log 2>&1
cd $REPO
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/c2/**/c2_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last step = from the last k_nbr* kernel on
i0 = max(i for i, r in enumerate(rows) if "k_nbr" in r["Kernel_Name"] or "nbr_batch" in r["Kernel_Name"])
while i0 > 0 and int(rows[i0]["Start_Timestamp"]) - int(rows[i0 - 1]["End_Timestamp"]) < 20000 and "Cijk" not in rows[i0-1]["Kernel_Name"]:
    i0 -= 1
t0 = int(rows[i0]["Start_Timestamp"]); prev = t0; busy = 0
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} us  +gap {(s - prev) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {r['Kernel_Name'][:70]}  grid {r.get('Grid_Size_X','?')}")
    prev = e; busy += e - s
print(f"span {(prev - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us")
PY

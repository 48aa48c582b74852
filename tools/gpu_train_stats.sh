#!/bin/bash
# kernel statistics + per-launch durations of the weight-gradient kernels of the config-5 training step (eager): gpurun_out/train_*
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$PWD
rm -rf gpurun_out/prof_train
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_train -o train -- python $REPO/tools/train_bench.py --kind ani2x --members 8 --steps 10 > $REPO/gpurun_out/prof_train.log 2>&1
echo "rocprof train exit $?"; cd $REPO
f=$(find gpurun_out/prof_train -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-150 && cp "$f" gpurun_out/train_kernel_stats.csv
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_train/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_wgrad" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
g = [int(r["Grid_Size_X"]) if "Grid_Size_X" in r else 0 for r in rows]
print("k_wgrad launches (us), last 9:", [round(x, 1) for x in d[-9:]], "grid", g[-3:])
PY

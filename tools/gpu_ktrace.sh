#!/bin/bash
# kernel-trace stats of one kbench invocation: tools/gpu_ktrace.sh <kbench args...>
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$PWD
rm -rf gpurun_out/kt
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/kt -o kt -- python $REPO/tools/kbench.py "$@" > $REPO/gpurun_out/kt.log 2>&1
cd $REPO; tail -1 gpurun_out/kt.log
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/kt/kt_kernel_stats.csv")))
for r in rows[:14]:
    print(f"{r['Name'][:60]:60s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:10.1f} total_ms={float(r['TotalDurationNs'])/1e6:9.2f} {r['Percentage']}%")
PY

#!/bin/bash
# PMC counters for one kbench invocation: tools/gpu_pmc.sh "<counters>" <kbench args...>
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$PWD
C="$1"; shift
rm -rf gpurun_out/pmc_k
cd /tmp && timeout 600 rocprofv3 --pmc $C --output-format csv -d $REPO/gpurun_out/pmc_k -o pmc -- python $REPO/tools/kbench.py "$@" > $REPO/gpurun_out/pmc_k.log 2>&1
cd $REPO; tail -2 gpurun_out/pmc_k.log
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/pmc_k/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "anihip" not in k: continue
        k = k.split("(")[0][-40:]
        a = acc[k][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for k, d in acc.items():
    print(k, " ".join(f"{c}={v[0] / v[1]:.4g}" for c, v in sorted(d.items())), f"(n={list(d.values())[0][1]})")
PY

#!/bin/bash
# round 5: the software-pipelined weight-gradient kernel against the plain one (same call), training tests on the new default
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_training.py -x -q > gpurun_out/r5d_train_tests.log 2>&1; echo "training tests exit $?"; tail -6 gpurun_out/r5d_train_tests.log; grep "fast train\|flat-optimizer" gpurun_out/parity_report.txt | tail -8
for r in 1 2; do for v in "" wgplain; do
  lib=""; [ -n "$v" ] && lib=$PWD/build_alt/libanihip_$v.so
  echo "== variant ${v:-product (pipelined)}"
  TORCHANI_AMD_LIB=$lib timeout 300 python tools/train_bench.py --kind ani2x --members 8 --graph --steps 40 2>&1 | grep -v amdgpu.ids | tail -1
  TORCHANI_AMD_LIB=$lib timeout 300 python tools/train_bench.py --kind ani1x --members 1 --graph --steps 40 2>&1 | grep -v amdgpu.ids | tail -1
done; done
rm -rf gpurun_out/prof_train
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_train -o train -- python $REPO/tools/train_bench.py --kind ani2x --members 8 --steps 10 > $REPO/gpurun_out/prof_train.log 2>&1
cd $REPO; f=$(find gpurun_out/prof_train -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -6 "$f" | cut -c1-150 && cp "$f" gpurun_out/r05_train_kernel_stats_pipelined.csv
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_train/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_wgrad_b3" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print("k_wgrad_b3p launches (us), last 9:", [round(x) for x in d[-9:]])
PY

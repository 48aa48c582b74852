#!/bin/bash
# round 5: fast training path after the first optimisations (bias sums inside the weight-gradient kernel, fused-only repack, chunk sizing)
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_training.py -x -q > gpurun_out/r5b_train_tests.log 2>&1; echo "training tests exit $?"; tail -12 gpurun_out/r5b_train_tests.log
for v in "" wb1536 wb6144 wb12288; do
  lib=""; [ -n "$v" ] && lib=$PWD/build_alt/libanihip_$v.so
  echo "== variant ${v:-product}"
  TORCHANI_AMD_LIB=$lib timeout 300 python tools/train_bench.py --kind ani2x --members 8 --graph --steps 40 2>&1 | grep -v amdgpu.ids | tail -1
done
echo "== eager"; timeout 300 python tools/train_bench.py --kind ani2x --members 8 2>&1 | grep -v amdgpu.ids | tail -3
echo "== ani1x"; timeout 300 python tools/train_bench.py --kind ani1x --members 1 2>&1 | grep -v amdgpu.ids | tail -3
timeout 300 python tools/train_bench.py --kind ani1x --members 1 --graph 2>&1 | grep -v amdgpu.ids | tail -1
rm -rf gpurun_out/prof_train
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_train -o train -- python $REPO/tools/train_bench.py --kind ani2x --members 8 --steps 10 > $REPO/gpurun_out/prof_train.log 2>&1
echo "rocprof exit $?"; cd $REPO
f=$(find gpurun_out/prof_train -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-160 && cp "$f" gpurun_out/r05_train_kernel_stats.csv
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_train/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_wgrad_b3" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print("k_wgrad_b3 launches (us), last 9:", [round(x) for x in d[-9:]])
PY

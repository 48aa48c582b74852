#!/bin/bash
# round 5, first GPU contact of the fast training path: its tests, the config-5 step old vs new on ONE box, kernel statistics
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_training.py -x -q > gpurun_out/r5a_train_tests.log 2>&1; echo "training tests exit $?"; tail -15 gpurun_out/r5a_train_tests.log
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "partition_skin or autograd_path or rows_updated or grad_helpers" > gpurun_out/r5a_parity_tests.log 2>&1; echo "parity subset exit $?"; tail -4 gpurun_out/r5a_parity_tests.log
for a in "" "--graph" "--optimizer torch --train-precision fp32" "--optimizer torch --train-precision fp32 --graph" "--optimizer torch" "--train-precision fp32"; do
  echo "== train_bench ani2x x8 $a"
  timeout 300 python tools/train_bench.py --kind ani2x --members 8 $a 2>&1 | grep -v amdgpu.ids | tail -4
done
echo "== train_bench ani1x x1"; timeout 300 python tools/train_bench.py --kind ani1x --members 1 2>&1 | grep -v amdgpu.ids | tail -3
timeout 300 python tools/train_bench.py --kind ani1x --members 1 --graph 2>&1 | grep -v amdgpu.ids | tail -1
rm -rf gpurun_out/prof_train
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_train -o train -- python $REPO/tools/train_bench.py --kind ani2x --members 8 --steps 10 > $REPO/gpurun_out/prof_train.log 2>&1
echo "rocprof exit $?"; cd $REPO
f=$(find gpurun_out/prof_train -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -24 "$f" && cp "$f" gpurun_out/r05_train_kernel_stats.csv

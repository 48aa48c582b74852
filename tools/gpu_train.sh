#!/bin/bash
# config-5 training step timings + kernel stats of the same command
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
python tools/train_bench.py --kind ani1x --members 1 2>&1 | tail -4 | tee gpurun_out/train_bench.txt
python tools/train_bench.py --kind ani2x --members 8 --steps 8 2>&1 | tail -4 | tee -a gpurun_out/train_bench.txt
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_train -o train -- python $REPO/tools/train_bench.py --kind ani2x --members 8 --steps 4 --warmup 1 > $REPO/gpurun_out/prof_train.log 2>&1
cd $REPO; f=$(find gpurun_out/prof_train -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f" | cut -c1-150

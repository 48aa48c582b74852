#!/bin/bash
# kernel statistics of one rank's step at W = 8 (emulated on one GPU)
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$PWD
rm -rf gpurun_out/prof_shard
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_shard -o shard -- python $REPO/bench.py --emulate-shard 3/8 --shuffle --steps 20 --warmup 4 > $REPO/gpurun_out/n_shard.log 2>&1
cd $REPO
f=$(find gpurun_out/prof_shard -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/n_shard_kernel_stats.csv && head -14 $f | cut -c1-160

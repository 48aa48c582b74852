#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r3e.log; : > $L
timeout 1500 python -m pytest tests/test_gpu_distributed.py -x -q 2>&1 | tail -15 >> $L
for a in "3/8 --shuffle" "0/8 --shuffle" "3/8" "1/4 --shuffle" "0/2 --shuffle" "0/1"; do
  timeout 600 python bench.py --emulate-shard $a --steps 10 2>&1 | tail -1 >> $L
done
cat $L

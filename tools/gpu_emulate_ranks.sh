#!/bin/bash
# every rank of W = 8 and W = 4 (the step of a job is its slowest rank), shuffled input
mkdir -p gpurun_out
: > gpurun_out/l_emul.log
for r in 0 1 2 3 4 5 6 7; do
  timeout 300 python bench.py --emulate-shard $r/8 --shuffle --steps 20 --warmup 4 2>&1 | grep -v amdgpu.ids >> gpurun_out/l_emul.log
done
for r in 0 1 2 3; do
  timeout 300 python bench.py --emulate-shard $r/4 --shuffle --steps 20 --warmup 4 2>&1 | grep -v amdgpu.ids >> gpurun_out/l_emul.log
done

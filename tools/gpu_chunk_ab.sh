#!/bin/bash
# network stage at the headline size for several chunk sizes (atoms per launch group)
mkdir -p gpurun_out
: > gpurun_out/chunk_ab.log
for rep in 1 2; do
for c in 262144 524288 1048576 4194304; do
  echo "== chunk $c" >> gpurun_out/chunk_ab.log
  timeout 300 python tools/kbench.py --side 92 --reps 5 --stages mlp --mask on --compact --chunk $c 2>&1 | grep atoms >> gpurun_out/chunk_ab.log
done
done
cat gpurun_out/chunk_ab.log

#!/bin/bash
# A/B of two libanihip variants on the AEV stages at the headline size: tools/gpu_ab_aev.sh <tagA> <tagB> [order]
mkdir -p gpurun_out
: > gpurun_out/ab_aev.log
for rep in 1 2 3; do
for v in $1 $2; do
  echo "== $v" >> gpurun_out/ab_aev.log
  TORCHANI_AMD_LIB=build_alt/libanihip_$v.so timeout 300 python tools/kbench.py --side 92 --reps 5 --stages fwd,bwd --order ${3:-lattice} 2>&1 | grep atoms >> gpurun_out/ab_aev.log
done
done
cat gpurun_out/ab_aev.log

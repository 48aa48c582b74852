#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r3b.log; : > $L
for v in ${VARIANTS}; do
  echo "== $v" >> $L
  TORCHANI_AMD_LIB=$PWD/build_alt/libanihip_$v.so timeout 600 python tools/kbench.py --side ${SIDE:-92} --stages ${STAGES:-fwd} --reps 10 2>&1 | tail -1 >> $L
done
cat $L

#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r3c.log; : > $L
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "${TESTS:-aev_forward or neighbor_rows or fused or slab_masks or water_box or solvated or config3 or degenerate or external}" 2>&1 | tail -8 >> $L
for v in ${VARIANTS}; do
  echo "== $v" >> $L
  TORCHANI_AMD_LIB=$PWD/build_alt/libanihip_$v.so timeout 600 python tools/kbench.py --side ${SIDE:-92} --stages ${STAGES:-fwd} --reps 10 2>&1 | tail -1 >> $L
done
cat $L

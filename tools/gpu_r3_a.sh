#!/bin/bash
# round 3, call A: k_aev_fwd3 parity + A/B against k_aev_fwd2 (variants built by tools/build_variants.sh)
mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r3a.log; : > $L
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "aev_forward or neighbor_rows or fused or slab_masks or water_box or solvated or config3 or degenerate or external" 2>&1 | tail -15 >> $L
for v in fwd2 f3w5r1 f3w5r0 f3w4r1 f3w4r0; do
  echo "== $v" >> $L
  TORCHANI_AMD_LIB=$PWD/build_alt/libanihip_$v.so timeout 600 python tools/kbench.py --side 92 --stages fwd --reps 10 2>&1 | tail -2 >> $L
done
echo "== protein-like (cfg3) fwd" >> $L
for v in fwd2 f3w5r1; do
  TORCHANI_AMD_LIB=$PWD/build_alt/libanihip_$v.so timeout 600 python tools/kbench_cfg3.py 2>&1 | tail -2 >> $L
done
cat $L

#!/bin/bash
# A/B of library variants at the headline size: VARIANTS="a b" STAGES=mlp bash tools/gpu_ab_libs.sh   ("" = the product library)
export TMPDIR=/tmp
for r in 1 2; do
for v in "" $VARIANTS; do
  lib=""; [ -n "$v" ] && lib=$PWD/build_alt/libanihip_$v.so
  echo "== variant ${v:-product}"
  TORCHANI_AMD_LIB=$lib timeout 300 python tools/kbench.py --side ${SIDE:-92} --reps 5 --stages ${STAGES:-mlp} --mask on --compact $KARGS 2>&1 | grep -v amdgpu.ids | tail -1
done
done

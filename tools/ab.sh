#!/bin/bash
# interleaved A/B of two builds of the library on one box: tools/_ab/libanihip_base.so vs the in-tree build
STAGES=${STAGES:-fwd,bwd}
for r in 1 2 3; do
  echo "== base"; TORCHANI_AMD_LIB=$PWD/tools/_ab/libanihip_base.so python tools/kbench.py --side ${SIDE:-64} --reps 7 --stages $STAGES --mask on 2>&1 | grep -v amdgpu.ids | tail -4
  echo "== new";  python tools/kbench.py --side ${SIDE:-64} --reps 7 --stages $STAGES --mask on 2>&1 | grep -v amdgpu.ids | tail -4
done

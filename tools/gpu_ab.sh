#!/bin/bash
# A/B of library builds in tools/_ab against the in-tree one: STAGES=fwd,bwd SIDE=92 bash tools/gpu_ab.sh
for lib in "" $(ls $PWD/tools/_ab/*.so 2>/dev/null); do
  echo "== lib=$(basename "$lib")"
  TORCHANI_AMD_LIB=$lib timeout 300 python tools/kbench.py --side ${SIDE:-92} --reps ${REPS:-7} --stages ${STAGES:-fwd,bwd} --mask on 2>&1 | grep -v amdgpu.ids | tail -1
done

#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
for lib in "" $(ls $PWD/tools/_ab/*.so 2>/dev/null); do
  echo "== lib=$lib"
  for s in 64 92; do TORCHANI_AMD_LIB=$lib timeout 300 python tools/kbench.py --side $s --reps 5 --stages ${STAGES:-fwd,bwd} --mask on 2>&1 | grep -v amdgpu.ids | tail -1; done
done | tee gpurun_out/kbench.txt

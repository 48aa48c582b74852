#!/bin/bash
# round-2 dev run: parity tests + stage timings
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
for s in 64 92; do timeout 300 python tools/kbench.py --side $s --reps 5 --stages nbr,fwd,bwd --mask on 2>&1 | grep -v amdgpu.ids | tail -2; done | tee gpurun_out/kbench.txt
timeout 40 tools/_bin/valubench 2>&1 | tail -4 | tee gpurun_out/valubench2.txt

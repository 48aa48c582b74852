#!/bin/bash
# round-4 call A: GPU suite, item-order A/B of the fused network kernel, evidence run on the shipped default
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$PWD
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4a_tests.log 2>&1; echo "tests exit $?"; tail -3 gpurun_out/r4a_tests.log
for r in 1 2; do
  for fl in 0 256; do
    echo "== mlp flags $fl"
    timeout 300 python tools/kbench.py --side 92 --reps 5 --stages mlp --mask on --compact --mlp-flags $fl 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
bash tools/gpu_r3_profile.sh > gpurun_out/r4a_profile.log 2>&1
mv gpurun_out/r03_pmc_aev.json gpurun_out/r04_pmc_default.json
tail -60 gpurun_out/r4a_profile.log

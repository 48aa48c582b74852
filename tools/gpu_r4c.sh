#!/bin/bash
# round-4 call C: phase 5 v2 (8 slabs per pass, both row blocks per wave) -- parity, A/B, phase trace
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "layer0_backward_inside" > gpurun_out/r4c_tests.log 2>&1; echo "tests exit $?"; tail -3 gpurun_out/r4c_tests.log
for r in 1 2; do
  for fl in 0 1024; do
    echo "== mlp flags $fl"
    timeout 300 python tools/kbench.py --side 92 --reps 5 --stages mlp --mask on --compact --mlp-flags $fl 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
for fl in 0 1024; do
  echo "== trace, mlp flags $fl"
  TORCHANI_AMD_LIB=$PWD/build_alt/libanihip_ftrace.so ANIHIP_FUSED_TRACE=/tmp/ft.bin timeout 300 python tools/kbench.py --side 64 --stages mlp --mask on --reps 1 --compact --mlp-flags $fl 2>&1 | grep atoms
  python tools/fused_trace.py /tmp/ft.bin
done > gpurun_out/r4c_trace.log 2>&1
cat gpurun_out/r4c_trace.log

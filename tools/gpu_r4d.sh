#!/bin/bash
# round-4 call D: phase 5 variants -- parity, A/B + fine trace
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "layer0_backward_inside" 2>&1 | tail -3
for r in 1 2; do
  for fl in 0 1024; do
    echo "== mlp flags $fl"
    timeout 300 python tools/kbench.py --side 92 --reps 5 --stages mlp --mask on --compact --mlp-flags $fl 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
echo "== trace"
TORCHANI_AMD_LIB=$PWD/build_alt/libanihip_ftrace.so ANIHIP_FUSED_TRACE=/tmp/ft.bin timeout 300 python tools/kbench.py --side 64 --stages mlp --mask on --reps 1 --compact 2>&1 | grep atoms
python tools/fused_trace.py /tmp/ft.bin

#!/bin/bash
# phase timeline of k_mlp_fused (development copy of the library with the stamps compiled in)
mkdir -p gpurun_out
for c in "" "--compact"; do
  TORCHANI_AMD_LIB=build_alt/libanihip_ftrace.so ANIHIP_FUSED_TRACE=/tmp/ft.bin timeout 300 python tools/kbench.py --side 56 --stages mlp --mask on --reps 1 $c 2>&1 | grep atoms
  python tools/fused_trace.py /tmp/ft.bin | head -16
done > gpurun_out/fused_trace.log 2>&1
cat gpurun_out/fused_trace.log

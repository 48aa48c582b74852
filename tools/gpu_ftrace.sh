#!/bin/bash
# phase + sub-phase timeline of k_mlp_fused (development build with stamps: VARIANT_TU=all tools/build_variants.sh ftrace "-DANIHIP_DEV_TRACE")
mkdir -p gpurun_out; export TMPDIR=/tmp
TORCHANI_AMD_LIB=$PWD/build_alt/libanihip_ftrace.so ANIHIP_FUSED_TRACE=/tmp/ft.bin timeout 600 python tools/kbench.py --side 40 --stages mlp --mask on --compact --reps 1 --mlp-flags ${MLPFLAGS:-0} $KARGS 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/ftrace.txt
python tools/fused_trace.py /tmp/ft.bin >> gpurun_out/ftrace.txt 2>&1
cat gpurun_out/ftrace.txt

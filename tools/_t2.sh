TORCHANI_AMD_LIB=$PWD/build_alt/libanihip_ftrace.so ANIHIP_FUSED_TRACE=/tmp/ft.bin timeout 300 python tools/kbench.py --side 64 --stages mlp --mask on --reps 1 --compact 2>&1 | grep atoms
python tools/fused_trace.py /tmp/ft.bin

#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu -k "nbr or neighbor or cell or golden or half or external" 2>&1 | tail -5 > gpurun_out/g_tests.log
TORCHANI_AMD_LIB=build_alt/libanihip_ntr.so timeout 300 python tools/nbr_trace.py 64 > gpurun_out/g_nbr_trace.log 2>&1
timeout 300 python tools/nbr_trace.py 92 >> gpurun_out/g_nbr_trace.log 2>&1

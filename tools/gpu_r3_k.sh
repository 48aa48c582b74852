#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "general_symmetry or general_grid" 2>&1 | tail -25 > gpurun_out/k_tests.log

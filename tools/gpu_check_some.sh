#!/bin/bash
# run a subset of the GPU tests: tools/gpu_check_some.sh "<pytest -k expression>"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu -k "$1" 2>&1 | tail -12 > gpurun_out/some_tests.log
cat gpurun_out/some_tests.log

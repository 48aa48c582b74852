#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "locality_sort or present_species or shards_add_up" 2>&1 | tail -8 > gpurun_out/j_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --parity-sample 64 --shuffle > gpurun_out/j_bench_shuffle.log 2>&1

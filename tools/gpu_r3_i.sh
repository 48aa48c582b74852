#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "skinny_layer0 or present_species or shards_add_up or partition_skin or headline or mlp_ensemble" 2>&1 | tail -8 > gpurun_out/i_tests.log
timeout 600 python tools/compact_debug.py > gpurun_out/i_debug.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --parity-sample 256 > gpurun_out/i_bench.log 2>&1

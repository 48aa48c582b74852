#!/bin/bash
# round 3: partition cost after the rewrite + skin reuse test
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "partition_skin or shards_add_up" 2>&1 | tail -5 > gpurun_out/f_tests.log
timeout 600 python -m pytest tests/test_gpu_distributed.py -x -q -m gpu 2>&1 | tail -5 >> gpurun_out/f_tests.log
timeout 600 python bench.py --emulate-shard 3/8 --shuffle --steps 20 --warmup 5 > gpurun_out/f_emul.log 2>&1
timeout 600 python bench.py --emulate-shard 0/1 --steps 10 --warmup 3 >> gpurun_out/f_emul.log 2>&1

#!/bin/bash
mkdir -p gpurun_out
timeout 600 python bench.py --emulate-shard 3/8 --shuffle --steps 30 --warmup 5 > gpurun_out/h_emul.log 2>&1
timeout 600 python bench.py --emulate-shard 0/8 --shuffle --steps 30 --warmup 5 >> gpurun_out/h_emul.log 2>&1
timeout 600 python bench.py --emulate-shard 1/2 --steps 20 --warmup 5 >> gpurun_out/h_emul.log 2>&1

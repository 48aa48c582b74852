#!/bin/bash
# the RCCL calls of the N > 1 step on a 1-GPU box (world size 1, collectives forced)
mkdir -p gpurun_out
export TORCHANI_AMD_FORCE_GROUP=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 \
    tools/rccl_world1.py > gpurun_out/rccl_world1.log 2>&1
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 \
    bench.py --gpus 1 --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --parity-sample 128 > gpurun_out/rccl_world1_bench.log 2>&1

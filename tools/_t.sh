timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "fused_kernel_gelu" 2>&1 | tail -3
bash tools/gpu_emulate_ranks.sh 2>&1 | tail -20

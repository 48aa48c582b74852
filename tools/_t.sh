timeout 900 python -m pytest tests/test_gpu_distributed.py -x -q -k "two_ranks_energies" 2>&1 | tail -30

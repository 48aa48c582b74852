timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "general_grid_on_a_large" 2>&1 | tail -30

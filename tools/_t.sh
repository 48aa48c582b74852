timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "layer0_backward_inside or large_systems or locality_sort or general_grid_slab" 2>&1 | tail -2
VARIANTS="base" bash tools/gpu_ab_libs.sh

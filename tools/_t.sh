timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "layer0_backward_inside or large_systems" 2>&1 | tail -2
VARIANTS="base g2 g8" bash tools/gpu_ab_libs.sh

timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "layer0_backward_inside or mlp_ensemble or weight_distribution or energies_and_forces_fused" 2>&1 | tail -3
VARIANTS="base" bash tools/gpu_ab_libs.sh

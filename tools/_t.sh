timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_training.py -x -q -k "padding or fused or training or weight_grads or mlp_ensemble" 2>&1 | tail -4

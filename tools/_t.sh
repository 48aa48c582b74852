timeout 900 python -m pytest tests/test_gpu_training.py -x -q 2>&1 | tail -25
grep "^x2rtrain" gpurun_out/parity_report.txt | tail

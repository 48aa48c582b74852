timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "general or grid or rows_updated" 2>&1 | tail -30
grep "^grid\|^ftrain general" gpurun_out/parity_report.txt | tail -12

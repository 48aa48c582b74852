timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "rows_updated_in_place or does_not_synchronize" 2>&1 | tail -2
for r in 1 2 3; do timeout 300 python tools/kbench.py --side 92 --reps 7 --stages fwd,fwdu --compact 2>&1 | grep -v amdgpu.ids | tail -1; done

#!/bin/bash
# round-4 call F: AEV rows updated in place -- parity, timing against full rows
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "rows_updated_in_place or general_grid_on_a_large or aev_forward_and_backward or energies_and_forces_fused" 2>&1 | tail -3
for r in 1 2; do
timeout 300 python tools/kbench.py --side 92 --reps 7 --stages fwd,fwdu --compact 2>&1 | grep -v amdgpu.ids | tail -1
done
timeout 300 python tools/kbench.py --side 64 --reps 7 --stages fwd,fwdu --compact 2>&1 | grep -v amdgpu.ids | tail -1

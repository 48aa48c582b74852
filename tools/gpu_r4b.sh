#!/bin/bash
# round-4 call B: layer-0 backward inside the fused kernel -- parity, then A/B against the GEMM hand-over at the headline size
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "layer0_backward_inside or skinny_layer0 or large_systems_of_any or locality_sort or fused or mlp_ensemble" > gpurun_out/r4b_tests.log 2>&1; echo "tests exit $?"; tail -5 gpurun_out/r4b_tests.log
grep "^l0b" gpurun_out/parity_report.txt | tail -20
for r in 1 2; do
  for fl in 0 1024; do
    echo "== mlp flags $fl"
    timeout 300 python tools/kbench.py --side 92 --reps 5 --stages mlp --mask on --compact --mlp-flags $fl 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
echo "== one launch group"; timeout 300 python tools/kbench.py --side 92 --reps 5 --stages mlp --mask on --compact --chunk 2400000 2>&1 | grep -v amdgpu.ids | tail -1
echo "== side 64 (786k)"; for fl in 0 1024; do timeout 300 python tools/kbench.py --side 64 --reps 5 --stages mlp --mask on --compact --mlp-flags $fl 2>&1 | grep -v amdgpu.ids | tail -1; done
timeout 600 python bench.py --no-secondary --no-cpu-baseline --parity-sample 256 > gpurun_out/r4b_bench.log 2> gpurun_out/r4b_bench.err; echo "bench exit $?"; tail -3 gpurun_out/r4b_bench.err; python -c "
import json
d=json.loads([l for l in open('gpurun_out/r4b_bench.log') if l.startswith('{')][0])
print(d['ms_per_step'], d['stages_ms'], d['parity_sample'], d['roofline_mfma']['frac'])"

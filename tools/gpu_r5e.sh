#!/bin/bash
# round 5: the optional two-product backward -- its test, its own bench line, and the default's line on the same box
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "two_product or layer0_backward_inside" > gpurun_out/r5e_tests.log 2>&1; echo "tests exit $?"; tail -4 gpurun_out/r5e_tests.log; grep "bwd2" gpurun_out/parity_report.txt | tail -2
timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-dense-stage --two-product-backward > gpurun_out/r5e_bench_bwd2.log 2> gpurun_out/r5e_bench_bwd2.err; echo "bench (two products) exit $?"; tail -2 gpurun_out/r5e_bench_bwd2.err
timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-dense-stage > gpurun_out/r5e_bench_default.log 2> gpurun_out/r5e_bench_default.err; echo "bench (default) exit $?"
python - <<'PY'
import json
for f in ("r5e_bench_bwd2", "r5e_bench_default"):
    d = json.loads([l for l in open(f"gpurun_out/{f}.log") if l.startswith('{')][0])
    print(f, d['ms_per_step'], d['stages_ms'], d['parity_sample']['max_dE_atom'], d['parity_sample']['max_dF'], d['config']['two_product_backward'])
PY

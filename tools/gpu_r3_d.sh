#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r3d.log; : > $L
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "${TESTS}" 2>&1 | tail -12 >> $L
if [ -n "$BENCH" ]; then
timeout 900 python bench.py --steps 5 --warmup 3 --no-secondary > gpurun_out/bench_r3d.json 2> gpurun_out/bench_r3d.err
echo "bench exit $?" >> $L; tail -3 gpurun_out/bench_r3d.err >> $L
python - >> $L <<'PY'
import json
try:
    r = json.loads(open("gpurun_out/bench_r3d.json").read().strip().splitlines()[-1])
    print("value", r["value"], "ms", r["ms_per_step"], "stages", r["stages_ms"])
    print("roofline", r["roofline"]["frac"], "bwd", r["roofline_bwd"]["frac"], "mfma", r["roofline_mfma"]["frac"])
    print("parity", r["parity_sample"]); print("cpu", r.get("cpu_baseline"))
except Exception as e:
    print("no bench line", e)
PY
fi
cat $L; grep -a "stress\|sampled" gpurun_out/parity_report.txt | tail -20

#!/bin/bash
# the whole GPU suite + a short bench line (development check before a commit)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/suite_tests.log 2>&1; echo "tests exit $?"; tail -4 gpurun_out/suite_tests.log
timeout 600 python bench.py --no-secondary --no-cpu-baseline --parity-sample 256 > gpurun_out/suite_bench.log 2> gpurun_out/suite_bench.err; echo "bench exit $?"; tail -3 gpurun_out/suite_bench.err; python -c "
import json
d=json.loads([l for l in open('gpurun_out/suite_bench.log') if l.startswith('{')][0])
print(d['ms_per_step'], d['stages_ms'], d['parity_sample'], d['roofline']['frac'], d['roofline_bwd']['frac'], d['roofline_mfma']['frac'])"

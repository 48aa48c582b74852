#!/bin/bash
# the whole GPU suite + the driver's bench command (development check before a commit)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/suite_tests.log 2>&1; echo "tests exit $?"; tail -4 gpurun_out/suite_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/suite_bench.log 2> gpurun_out/suite_bench.err ) 2>&1 | grep real; echo "bench exit $?"; tail -3 gpurun_out/suite_bench.err; python -c "
import json
d=json.loads([l for l in open('gpurun_out/suite_bench.log') if l.startswith('{')][0])
print(d['ms_per_step'], d['stages_ms'], d['parity_sample']['max_dF'], d['roofline']['frac'], d['roofline_bwd']['frac'], d['roofline_mfma']['frac'], d['roofline_nbr']['frac'])
print(d['secondary']['two_product_backward'])
print({k: v['ms_per_step'] for k, v in d['secondary']['config5'].items() if isinstance(v, dict)})
print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"

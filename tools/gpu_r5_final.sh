#!/bin/bash
# round 5, final check: whole GPU suite, smoke, the driver's bench command, kernel statistics of the training step
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$PWD
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/final_tests.log 2>&1; echo "tests exit $?"; tail -3 gpurun_out/final_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final_bench.log 2> gpurun_out/final_bench.err ) 2>&1 | grep real; echo "bench exit $?"; python -c "
import json
d=json.loads([l for l in open('gpurun_out/final_bench.log') if l.startswith('{')][0])
print(d['ms_per_step'], d['value'], d['stages_ms'], d['parity_sample']['max_dF'], d['roofline']['frac'], d['roofline_bwd']['frac'], d['roofline_mfma']['frac'], d['roofline_nbr']['frac'])
print(d['secondary']['two_product_backward'])
print({k: v['ms_per_step'] for k, v in d['secondary']['config5'].items() if isinstance(v, dict)}, d['secondary']['config5']['ani2x_x8_graph']['roofline']['frac'])
print(d['secondary']['config2'], d['secondary']['config3'])"
rm -rf gpurun_out/prof_train
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_train -o train -- python $REPO/tools/train_bench.py --kind ani2x --members 8 --steps 10 > $REPO/gpurun_out/prof_train.log 2>&1
cd $REPO; f=$(find gpurun_out/prof_train -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-140 && cp "$f" gpurun_out/r05_train_kernel_stats_final.csv

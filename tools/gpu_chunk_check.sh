#!/bin/bash
mkdir -p gpurun_out
for c in 0 4194304; do
  timeout 900 python bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-dense-stage --parity-sample 192 --mlp-chunk $c 2>/dev/null \
   | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('chunk $c: %.3f ms/step  mlp %.3f  parity dE %.2e dF %.2e' % (l['ms_per_step'], l['stages_ms']['mlp_fwd_bwd'], l['parity_sample']['max_dE_atom'], l['parity_sample']['max_dF']))"
done | tee gpurun_out/chunk_check.log

#!/bin/bash
# round 5: whole GPU suite after the pruning + md crossover (Verlet refresh forced vs plain rebuild) + option C costing + bench line
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$PWD
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5c_tests.log 2>&1; echo "tests exit $?"; tail -6 gpurun_out/r5c_tests.log
for side in 40 48 56 64; do
  echo "== md_bench side $side (refresh forced)"; timeout 300 python tools/md_bench.py --side $side --steps 20 --force-verlet 2>&1 | grep -v amdgpu.ids | tail -2
done 2>&1 | tee gpurun_out/r5c_md.txt
echo "== md_bench side 92 (default: rebuild above the crossover)"; timeout 400 python tools/md_bench.py --side 92 --steps 10 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a gpurun_out/r5c_md.txt
for w in 4 8; do
  echo "== emulate shard 1/$w"; timeout 300 python bench.py --emulate-shard 1/$w --steps 5 --shuffle 2>&1 | grep -v amdgpu.ids | tail -3
  echo "== emulate shard 1/$w option C"; timeout 300 python bench.py --emulate-shard 1/$w --steps 5 --shuffle --emulate-option-c 2>&1 | grep -v amdgpu.ids | tail -4
done 2>&1 | tee gpurun_out/r5c_option_c.txt
timeout 900 python bench.py > gpurun_out/r5c_bench.log 2> gpurun_out/r5c_bench.err; echo "bench exit $?"; tail -2 gpurun_out/r5c_bench.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r5c_bench.log') if l.startswith('{')][0])
print(d['ms_per_step'], d['stages_ms'], d['parity_sample'])
print({k: d[k]['frac'] for k in ('roofline', 'roofline_bwd', 'roofline_mfma', 'roofline_nbr')}, d['roofline']['full_rows'])
print(json.dumps(d['secondary']['config5'], indent=None)[:3000])
print(d['secondary']['config2'], d['secondary']['config3'])
PY

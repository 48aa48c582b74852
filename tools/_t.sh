timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "does_not_synchronize" 2>&1 | tail -15
timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-dense-stage --parity-sample 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['ms_per_step_median'], sum(v for k,v in d['stages_ms'].items()))"

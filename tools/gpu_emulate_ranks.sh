#!/bin/bash
# Emulated strong scaling on ONE box: the single-GPU step, then every rank of W = 2, 4, 8 (the step of a job is its slowest
# rank; no collective: the all-gather of ~1.5 MB per rank is not in these numbers), shuffled input.
mkdir -p gpurun_out
: > gpurun_out/emulate_ranks.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-dense-stage --parity-sample 0 --shuffle 2>/dev/null \
  | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('single GPU (shuffled input, cell-sorted copy): %.3f ms/step' % l['ms_per_step'])" >> gpurun_out/emulate_ranks.log
for w in 2 4 8; do
  for ((r=0; r<w; r++)); do
    timeout 300 python bench.py --emulate-shard $r/$w --shuffle --steps 20 --warmup 4 2>&1 | grep "^shard" >> gpurun_out/emulate_ranks.log
  done
done
python - <<'PY' >> gpurun_out/emulate_ranks.log
import re
t1=None; best={}
for line in open('gpurun_out/emulate_ranks.log'):
    m=re.match(r'single GPU.*: ([\d.]+) ms', line)
    if m: t1=float(m.group(1))
    m=re.match(r'shard (\d+)/(\d+).*: ([\d.]+) ms/step', line)
    if m: best.setdefault(int(m.group(2)), []).append(float(m.group(3)))
for w,v in sorted(best.items()):
    print(f"W = {w}: slowest rank {max(v):.3f} ms (fastest {min(v):.3f}); ideal {t1 / w:.3f}; efficiency {t1 / w / max(v):.1%}")
PY
cat gpurun_out/emulate_ranks.log

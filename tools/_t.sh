bash tools/gpu_rccl_world1.sh; cat gpurun_out/rccl_world1.log | grep -v amdgpu.ids | tail -8; tail -c 2500 gpurun_out/rccl_world1_bench.log | grep -v amdgpu.ids | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('world1 bench: ms/step', d['ms_per_step'], 'collective', {k:v for k,v in d['collective'].items() if k!='ranks_seen'}); print(d['collective']['ranks_seen']); print('parity', d.get('parity_sample'))
"
echo "== 2 ranks over gloo sharing the GPU, 192k atoms"
timeout 600 python bench.py --gpus 2 --steps 5 --warmup 3 --waters-side 40 --dist-backend gloo --no-dense-stage 2>&1 | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], d['collective']['op'], d['collective']['bytes_per_step'], [ (r['rank'], r['peers'], r['ms_per_step_this_rank']) for r in d['collective']['ranks_seen']])"

#!/bin/bash
# kernel statistics of one golden config input (eager calls): bash tools/gpu_cfg_stats.sh cfg3_1c17_ani2x cell
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$PWD; NAME=${1:-cfg3_1c17_ani2x}; NL=${2:-cell}
cat > /tmp/cfgx.py <<'PY'
import sys, os, numpy as np, torch
sys.path.insert(0, os.environ["REPO"])
from torchani_amd.models import ANI2x
GOLD = os.path.join(os.environ["REPO"], "tests", "golden")
dev = torch.device("cuda:0")
with np.load(os.path.join(GOLD, sys.argv[1] + ".npz")) as z:
    sp, x = z["species"].astype(np.int64), z["coords"]
    cell = z["cell"] if "cell" in z.files else None
model = ANI2x(seed=0, device=dev, periodic_table_index=False, neighborlist=sys.argv[2])
model.auto_graph_atoms = 0
s, c = torch.from_numpy(sp).to(dev), torch.from_numpy(x).to(dev)
cl = None if cell is None else torch.from_numpy(cell).to(dev)
pbc = None if cell is None else (True, True, True)
for _ in range(30):
    model.energies_and_forces(s, c, cl, pbc, check_overflow=False)
torch.cuda.synchronize()
PY
rm -rf gpurun_out/prof_cfgx
cd /tmp && REPO=$REPO timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_cfgx -o cfg -- python /tmp/cfgx.py $NAME $NL > $REPO/gpurun_out/prof_cfgx.log 2>&1
echo "rocprof exit $?"; cd $REPO
f=$(find gpurun_out/prof_cfgx -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${NAME}_kernel_stats.csv && python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"  {r['Name'][:100]:100s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f} per-step_us {float(r['TotalDurationNs'])/30/1e3:8.1f}")
PY

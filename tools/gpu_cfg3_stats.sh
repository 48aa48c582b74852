#!/bin/bash
# kernel statistics of config 3 (46 357-atom solvated 1hz5, eager calls): gpurun_out/cfg3_kernel_stats.csv
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$PWD
cat > /tmp/cfg3.py <<'PY'
import sys, os, numpy as np, torch
sys.path.insert(0, os.environ["REPO"])
from torchani_amd.models import ANI2x
GOLD = os.path.join(os.environ["REPO"], "tests", "golden")
dev = torch.device("cuda:0")
with np.load(os.path.join(GOLD, "cfg3_1hz5_water_ani2x.npz")) as z:
    sp, x, cell = z["species"].astype(np.int64), z["coords"], z["cell"]
model = ANI2x(seed=0, device=dev, periodic_table_index=False, neighborlist="cell")
model.auto_graph_atoms = 0
s, c, cl = torch.from_numpy(sp).to(dev), torch.from_numpy(x).to(dev), torch.from_numpy(cell).to(dev)
for _ in range(30):
    model.energies_and_forces(s, c, cl, (True, True, True), check_overflow=False)
torch.cuda.synchronize()
print("order", model._engine_species(s.to(torch.int32))[1])
PY
rm -rf gpurun_out/prof_cfg3
cd /tmp && REPO=$REPO timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_cfg3 -o cfg -- python /tmp/cfg3.py > $REPO/gpurun_out/prof_cfg3.log 2>&1
echo "rocprof exit $?"; cd $REPO; tail -2 gpurun_out/prof_cfg3.log
f=$(find gpurun_out/prof_cfg3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-170 && cp "$f" gpurun_out/cfg3_kernel_stats.csv

#!/bin/bash
# round-5 evidence run: kernel statistics of the bench command (headline), of configs 2 / 3 and of the config-5 training step,
# FETCH / WRITE PMC of the headline kernels (separate passes, per the guide) -> gpurun_out/r05_*
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$PWD
rm -rf gpurun_out/prof
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof -o bench -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dense-stage --no-secondary --parity-sample 0 > $REPO/gpurun_out/prof_bench.log 2>&1
echo "rocprof bench exit $?"; cd $REPO
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-150 && cp "$f" gpurun_out/r05_bench_kernel_stats.csv
cat > /tmp/cfg23.py <<'PY'
import sys, os, numpy as np, torch
sys.path.insert(0, os.environ["REPO"]); sys.path.insert(0, os.path.join(os.environ["REPO"], "tools"))
from torchani_amd.models import ANI2x
which = sys.argv[1]
GOLD = os.path.join(os.environ["REPO"], "tests", "golden")
dev = torch.device("cuda:0")
name, nl = ("cfg2_xyz13_28_ani2x", "batch") if which == "2" else ("cfg3_1hz5_water_ani2x", "cell")
with np.load(os.path.join(GOLD, name + ".npz")) as z:
    sp, x = z["species"].astype(np.int64), z["coords"]
    cell = z["cell"] if "cell" in z.files else None
model = ANI2x(seed=0, device=dev, periodic_table_index=False, neighborlist=nl)
model.auto_graph_atoms = 0
s, c = torch.from_numpy(sp).to(dev), torch.from_numpy(x).to(dev)
cl = None if cell is None else torch.from_numpy(cell).to(dev)
pbc = None if cell is None else (True, True, True)
for _ in range(30):
    model.energies_and_forces(s, c, cl, pbc, check_overflow=False)
torch.cuda.synchronize()
PY
for c in 2 3; do
  rm -rf gpurun_out/prof_cfg$c
  cd /tmp && REPO=$REPO timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_cfg$c -o cfg -- python /tmp/cfg23.py $c > $REPO/gpurun_out/prof_cfg$c.log 2>&1
  echo "rocprof config $c exit $?"; cd $REPO
  f=$(find gpurun_out/prof_cfg$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-150 && cp "$f" gpurun_out/r05_cfg${c}_kernel_stats.csv
done
rm -rf gpurun_out/prof_train
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_train -o train -- python $REPO/tools/train_bench.py --kind ani2x --members 8 --steps 10 > $REPO/gpurun_out/prof_train.log 2>&1
echo "rocprof train exit $?"; cd $REPO
f=$(find gpurun_out/prof_train -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -10 "$f" | cut -c1-150 && cp "$f" gpurun_out/r05_train_kernel_stats.csv
timeout 300 python tools/train_bench.py --kind ani2x --members 8 --graph --steps 40 2>&1 | grep -v amdgpu.ids | tail -1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  cd /tmp && timeout 600 rocprofv3 --pmc $c --output-format csv -d $REPO/gpurun_out/pmc_$c -o pmc -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dense-stage --no-secondary --parity-sample 0 > $REPO/gpurun_out/pmc_$c.log 2>&1
  echo "pmc $c exit $?"; cd $REPO
done
python - <<'PY'
import csv, glob, json, collections
out = {"n_atoms": 2336064, "fetch_correction": 2.0, "kernels": {},
       "workload": "bench.py at the headline size (2336064-atom periodic water box)",
       "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, tools/gpu_r5_profile.sh), mean KB per "
                 "dispatch; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md"}
for c, key in (("FETCH_SIZE", "fetch_size_kb"), ("WRITE_SIZE", "write_size_kb")):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != c: continue
            k = row["Kernel_Name"]
            for name in ("k_aev_fwd3", "k_aev_bwd", "k_mlp_fused", "k_gemm_h2", "k_gemm_l0b", "k_nbr_cell2"):
                if name + "<" in k or name + "(" in k:
                    acc[name][0] += float(row["Counter_Value"]); acc[name][1] += 1
    for name, (s, n) in acc.items():
        out["kernels"].setdefault(name, {})[key] = s / n
        out["kernels"][name]["dispatches_" + c] = n
json.dump(out, open("gpurun_out/r05_pmc.json", "w"), indent=1)
print(json.dumps(out["kernels"], indent=None))
PY

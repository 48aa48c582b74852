#!/bin/bash
# round-6 evidence run: kernel statistics of the bench command (headline), of configs 2 / 3 and of the config-5 training step,
# FETCH / WRITE PMC of the headline kernels (separate passes, per the guide) -> gpurun_out/r06_*
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$PWD
rm -rf gpurun_out/prof
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof -o bench -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dense-stage --no-secondary --parity-sample 0 > $REPO/gpurun_out/prof_bench.log 2>&1
echo "rocprof bench exit $?"; cd $REPO
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-150 && cp "$f" gpurun_out/r06_bench_kernel_stats.csv
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
  f=$(find gpurun_out/prof_cfg$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-150 && cp "$f" gpurun_out/r06_cfg${c}_kernel_stats.csv
done
rm -rf gpurun_out/prof_train
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_train -o train -- python $REPO/tools/train_bench.py --kind ani2x --members 8 --steps 10 > $REPO/gpurun_out/prof_train.log 2>&1
echo "rocprof train exit $?"; cd $REPO
f=$(find gpurun_out/prof_train -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -10 "$f" | cut -c1-150 && cp "$f" gpurun_out/r06_train_kernel_stats.csv
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
       "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, tools/gpu_r6_profile.sh), mean KB per "
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
# the AEV forward kernel runs in two roles in a pass: the first call of the engine writes EVERY row (zeros and values: the
# "full rows" traffic), the later ones rewrite only the flagged slabs of the rows the engine keeps -- told apart by dispatch order
per = {}
for c, key in (("FETCH_SIZE", "fetch_size_kb"), ("WRITE_SIZE", "write_size_kb")):
    rows = []
    for f in glob.glob(f"gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True):
        rows += [(int(r["Dispatch_Id"]), float(r["Counter_Value"])) for r in csv.DictReader(open(f))
                 if r["Counter_Name"] == c and "k_aev_fwd3<" in r["Kernel_Name"]]
    rows.sort()
    if len(rows) >= 2:
        per[key] = (rows[0][1], sum(v for _, v in rows[1:]) / (len(rows) - 1))
if len(per) == 2 and per["write_size_kb"][0] > 2.0 * per["write_size_kb"][1]:
    out["kernels"]["k_aev_fwd3"]["full_rows"] = {k: v[0] for k, v in per.items()}
    out["kernels"]["k_aev_fwd3"]["kept_rows"] = {k: v[1] for k, v in per.items()}
json.dump(out, open("gpurun_out/r06_pmc.json", "w"), indent=1)
print(json.dumps(out["kernels"], indent=None))
PY
# SQ counters of the headline kernels (three passes; SQ has eight slots, GRBM two of its own): what bounds each kernel, as counters
# -> gpurun_out/r06_pmc_sq.json  (SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles:
# MI355X_MICROARCH.md, per-instruction constants table)
G1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
G2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_MFMA"
G3="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
i=0
for G in "$G1" "$G2" "$G3"; do
  i=$((i+1)); rm -rf gpurun_out/sq_$i
  cd /tmp && timeout 600 rocprofv3 --pmc $G --output-format csv -d $REPO/gpurun_out/sq_$i -o pmc -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dense-stage --no-secondary --parity-sample 0 > $REPO/gpurun_out/sq_$i.log 2>&1
  echo "sq pass $i exit $?"; cd $REPO
done
python - <<'PY'
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/sq_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        for name in ("k_aev_fwd3", "k_aev_bwd", "k_mlp_fused", "k_nbr_cell2"):
            if name + "<" in k or name + "(" in k:
                a = acc[name][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
out = {"n_atoms": 2336064, "workload": "bench.py at the headline size (2336064-atom periodic water box), --steps 1 --warmup 1",
       "source": "rocprofv3 --pmc, three passes (tools/gpu_r6_profile.sh); summed over the launches of a step, mean over the steps; SQ_WAVE_CYCLES, SQ_WAIT_*, "
                 "SQ_ACTIVE_INST_* are quad-cycles summed over the waves, SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE cycles",
       "kernels": {}}
# (the network kernel runs as one launch per species, some of them empty: counters are summed over a STEP's launches --
# steps of a pass = dispatches of k_aev_bwd, which runs once per step)
steps = {c: v[1] for c, v in acc.get("k_aev_bwd", {}).items()}
for k, d in acc.items():
    # launches of this kernel per step: 1, or 7 for the per-species launches of the network kernel (a pass also holds a few
    # extra single launches of the bench's stage timers: mean per dispatch x launches per step)
    g = {c: v[0] / v[1] * max(1, round(v[1] / max(1, steps.get(c, v[1])))) for c, v in d.items()}
    e = {"counters_per_step": g, "dispatches": list(d.values())[0][1], "steps": max(steps.values()) if steps else None}
    wc = g.get("SQ_WAVE_CYCLES")
    if wc:
        # share of a resident wave's time in which it issues VALU work / waits / issues anything
        for name, c in (("valu_active_over_wave_cycles", "SQ_ACTIVE_INST_VALU"), ("wait_any_over_wave_cycles", "SQ_WAIT_ANY"),
                        ("wait_inst_over_wave_cycles", "SQ_WAIT_INST_ANY"), ("active_any_over_wave_cycles", "SQ_ACTIVE_INST_ANY"),
                        ("lds_active_over_wave_cycles", "SQ_ACTIVE_INST_LDS")):
            if c in g: e[name] = g[c] / wc
    if "GRBM_GUI_ACTIVE" in g:
        # GRBM_GUI_ACTIVE comes back summed over the eight XCDs (checked on kernels whose occupancy is known: with this
        # normalisation the resident waves per SIMD come out as 3.9 for the 16-wave-per-CU AEV kernels and 1.93 for the
        # 8-wave network kernel): cycles of the launch = GRBM_GUI_ACTIVE / 8, SIMD-cycles = that x 256 CUs x 4 SIMDs
        simd_cycles = g["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
        if "SQ_ACTIVE_INST_VALU" in g: e["simd_valu_busy"] = 4.0 * g["SQ_ACTIVE_INST_VALU"] / simd_cycles   # quad-cycles -> cycles
        if "SQ_VALU_MFMA_BUSY_CYCLES" in g: e["simd_mfma_busy"] = g["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles
        if "SQ_WAVE_CYCLES" in g: e["waves_per_simd_resident"] = 4.0 * g["SQ_WAVE_CYCLES"] / simd_cycles
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
        if c in g: e[c.lower() + "_per_atom"] = g[c] / out["n_atoms"]
    out["kernels"][k] = e
json.dump(out, open("gpurun_out/r06_pmc_sq.json", "w"), indent=1)
for k, e in out["kernels"].items():
    print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in e.items() if a != "counters_per_step"})
PY

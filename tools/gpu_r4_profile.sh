#!/bin/bash
# round-4 evidence run (final binaries: layer-0 backward inside the fused kernel, AEV rows updated in place): bench line, rocprofv3 kernel statistics of the same command, FETCH/WRITE PMC of the AEV kernels
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$PWD
timeout 900 python bench.py --no-secondary > gpurun_out/bench.log 2> gpurun_out/bench.err
echo "bench exit $?" >> gpurun_out/bench.err; tail -2 gpurun_out/bench.err; cat gpurun_out/bench.log
rm -rf gpurun_out/prof
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof -o bench -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dense-stage --no-secondary --parity-sample 0 > $REPO/gpurun_out/prof_bench.log 2>&1
echo "rocprof exit $?"; cd $REPO
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -16 "$f"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  cd /tmp && timeout 600 rocprofv3 --pmc $c --output-format csv -d $REPO/gpurun_out/pmc_$c -o pmc -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dense-stage --no-secondary --parity-sample 0 > $REPO/gpurun_out/pmc_$c.log 2>&1
  echo "pmc $c exit $?"; cd $REPO
done
python - <<'PY'
import csv, glob, json, collections
out = {"n_atoms": 2336064, "fetch_correction": 2.0, "kernels": {},
       "workload": "bench.py at the headline size (2336064-atom periodic water box)",
       "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, tools/gpu_r4_profile.sh), mean KB per "
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
json.dump(out, open("gpurun_out/r04_pmc_l0b.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY

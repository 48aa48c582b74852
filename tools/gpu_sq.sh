#!/bin/bash
# SQ counters of the hot kernels (786 432-atom water box), three rocprofv3 --pmc passes; summary -> gpurun_out/r02_pmc_sq_counters.txt
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$PWD
G1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"
G2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU"
G3="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD"
i=0
for G in "$G1" "$G2" "$G3"; do
  i=$((i+1)); rm -rf gpurun_out/sq_$i
  cd /tmp && timeout 300 rocprofv3 --pmc $G --output-format csv -d $REPO/gpurun_out/sq_$i -o pmc -- python $REPO/tools/kbench.py --side 64 --reps 2 --stages fwd,bwd,mlp --mask on > $REPO/gpurun_out/sq_$i.log 2>&1
  echo "pass $i exit $?"; cd $REPO
done
python - <<'PY' > gpurun_out/r02_pmc_sq_counters.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/sq_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        for name in ("k_aev_fwd2", "k_aev_bwd", "k_mlp_fused", "k_gemm_h2", "k_gemm_l0b", "k_nbr_cell2"):
            if name + "<" in k or name + "(" in k:
                a = acc[name][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
print("SQ counters, rocprofv3 --pmc (tools/gpu_sq.sh: three passes of tools/kbench.py --side 64 --stages fwd,bwd,mlp --mask on;")
print("786 432-atom water box; mean per dispatch; k_mlp_fused / k_gemm_h2 per 262 144-atom chunk)")
for k, d in acc.items():
    n = list(d.values())[0][1]
    print(f"\n{k}  (n={n} dispatches)")
    print("  " + " ".join(f"{c}={v[0] / v[1]:.4g}" for c, v in sorted(d.items())))
    g = lambda c: d[c][0] / d[c][1] if c in d else float("nan")
    busy = g("SQ_BUSY_CYCLES")
    if busy == busy and busy > 0:
        # SQ_BUSY_CYCLES is summed over the XCDs' SQs; per-SIMD issue slots ~ SQ_WAVE_CYCLES-independent: report ratios
        print(f"  VALU active / wave-resident: {g('SQ_ACTIVE_INST_VALU') / g('SQ_WAVE_CYCLES'):.3f}   "
              f"MFMA busy / (4 x busy): {g('SQ_VALU_MFMA_BUSY_CYCLES') / (4 * busy):.3f}   "
              f"wait-any / wave-resident: {g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'):.3f}")
PY
cat gpurun_out/r02_pmc_sq_counters.txt

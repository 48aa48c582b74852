#!/bin/bash
# SQ counters of the AEV forward kernel for the given variants (one rocprofv3 --pmc pass each, 8 SQ slots)
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$PWD
G="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
for v in ${VARIANTS}; do
  rm -rf gpurun_out/sqf_$v
  cd /tmp && TORCHANI_AMD_LIB=$REPO/build_alt/libanihip_$v.so timeout 300 rocprofv3 --pmc $G --output-format csv -d $REPO/gpurun_out/sqf_$v -o pmc -- python $REPO/tools/kbench.py --side ${SIDE:-64} --reps 2 --stages ${STAGES:-fwd} > $REPO/gpurun_out/sqf_$v.log 2>&1
  echo "$v exit $?"; cd $REPO
done
python - <<'PY' | tee gpurun_out/sq_fwd_summary.txt
import csv, glob, collections, os
for d in sorted(glob.glob("gpurun_out/sqf_*")):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            for name in ("k_aev_fwd2", "k_aev_fwd3", "k_aev_bwd"):
                if name in k:
                    a = acc[name][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
    for k, dd in acc.items():
        g = lambda c: dd[c][0] / dd[c][1]
        n_atoms = 786432.0
        print(f"{os.path.basename(d)} {k}: per atom: VALU {g('SQ_INSTS_VALU')/n_atoms:.0f} SALU {g('SQ_INSTS_SALU')/n_atoms:.0f} LDS {g('SQ_INSTS_LDS')/n_atoms:.0f} | "
              f"wave quad-cycles/atom {g('SQ_WAVE_CYCLES')/n_atoms:.0f} active_valu {g('SQ_ACTIVE_INST_VALU')/n_atoms:.0f} wait_any {g('SQ_WAIT_ANY')/n_atoms:.0f} wait_inst {g('SQ_WAIT_INST_ANY')/n_atoms:.0f} busy {g('SQ_BUSY_CYCLES'):.3g}")
PY

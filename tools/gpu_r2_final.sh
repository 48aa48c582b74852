#!/bin/bash
# round-2 final evidence: the bench line and the rocprofv3 kernel statistics of the same command (the counters of the large
# kernels, unchanged since, stay in profiles/r02_pmc_*)
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$PWD
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err
echo "bench exit $?" >> gpurun_out/bench.err; tail -2 gpurun_out/bench.err; cat gpurun_out/bench.log
rm -rf gpurun_out/prof
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof -o bench -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dense-stage --no-secondary > $REPO/gpurun_out/prof_bench.log 2>&1
echo "rocprof exit $?"; cd $REPO
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f"

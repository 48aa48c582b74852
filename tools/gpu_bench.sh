#!/bin/bash
# bench + rocprofv3 kernel trace of the same command (summaries go to gpurun_out/, copied to profiles/)
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
if [ -z "$SKIP_BENCH" ]; then
timeout 900 python bench.py --waters-side ${SIDE:-92} --steps ${STEPS:-5} --warmup 2 > gpurun_out/bench.log 2> gpurun_out/bench.err
echo "bench exit $?" >> gpurun_out/bench.err
tail -3 gpurun_out/bench.err; cat gpurun_out/bench.log
fi
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof -o bench -- python $REPO/bench.py --waters-side ${SIDE:-92} --steps 2 --warmup 1 --no-cpu-baseline --no-dense-stage > $REPO/gpurun_out/prof_bench.log 2>&1
echo "rocprof exit $?"
cd $REPO; find gpurun_out/prof -type f | head -20
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f"
if [ -n "$PMC" ]; then
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && timeout 900 rocprofv3 --pmc $c --output-format csv -d $REPO/gpurun_out/pmc_$c -o pmc -- python $REPO/bench.py --waters-side ${PMC_SIDE:-40} --steps 1 --warmup 0 --no-cpu-baseline > $REPO/gpurun_out/pmc_$c.log 2>&1
  echo "pmc $c exit $?"
done
cd $REPO; find gpurun_out/pmc_* -type f | head; python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE | tee gpurun_out/pmc_summary.txt
fi
exit 0

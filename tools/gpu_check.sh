#!/bin/bash
# First-contact GPU run: build check, smoke, parity tests (no -x: collect every failure), short report.
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 "$@" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/smoke.log
tail -40 gpurun_out/pytest_gpu.log
cat gpurun_out/parity_report.txt 2>/dev/null | tail -60

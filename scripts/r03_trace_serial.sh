#!/bin/bash
# Kernel trace of the bench with one launch at a time on the chip.  usage: scripts/r03_trace_serial.sh <tag>
set -o pipefail
tag=$1
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_trace -o t --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-generate-leg --no-fill-ahead --no-lanes > gpurun_out/${tag}_bench_serial_under_rocprof.json 2> gpurun_out/${tag}_trace.err || exit 1
python3 scripts/trace_by_grid.py gpurun_out/${tag}_trace > gpurun_out/${tag}_serial_by_grid.txt || exit 1
cp gpurun_out/${tag}_trace/*kernel_stats.csv gpurun_out/${tag}_serial_kernel_stats.csv || exit 1
rm -rf gpurun_out/${tag}_trace
head -40 gpurun_out/${tag}_serial_by_grid.txt

#!/bin/bash
# Profile set of round 3.  usage: scripts/r03_profile.sh <tag>   (writes gpurun_out/<tag>_*)
#  1. kernel trace of the bench with one launch at a time on the chip (--no-fill-ahead --no-lanes: a launch's duration is its own)
#  2. kernel trace of the default bench (fill beside updates: durations overlap, wall time is what counts)
#  3. two counter passes (FETCH_SIZE, WRITE_SIZE) over scripts/pmc_gemm.py
#  4. the driver's own command
set -o pipefail
tag=$1
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_trace -o t --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-generate-leg --no-fill-ahead --no-lanes > gpurun_out/${tag}_bench_serial_under_rocprof.json 2> gpurun_out/${tag}_trace.err || exit 1
python3 scripts/trace_by_grid.py gpurun_out/${tag}_trace > gpurun_out/${tag}_serial_by_grid.txt || exit 1
cp gpurun_out/${tag}_trace/*kernel_stats.csv gpurun_out/${tag}_serial_kernel_stats.csv || exit 1
rm -rf gpurun_out/${tag}_trace
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_trace -o t --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-generate-leg > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_trace.err || exit 1
python3 scripts/trace_by_grid.py gpurun_out/${tag}_trace > gpurun_out/${tag}_by_grid.txt || exit 1
cp gpurun_out/${tag}_trace/*kernel_stats.csv gpurun_out/${tag}_kernel_stats.csv || exit 1
rm -rf gpurun_out/${tag}_trace
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${tag}_pmc_f -o p --output-format csv -- python3 scripts/pmc_gemm.py > gpurun_out/${tag}_pmc_shapes.json 2> gpurun_out/${tag}_pmc_f.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${tag}_pmc_w -o p --output-format csv -- python3 scripts/pmc_gemm.py > /dev/null 2> gpurun_out/${tag}_pmc_w.err || exit 1
grep '^\[' gpurun_out/${tag}_pmc_shapes.json > gpurun_out/${tag}_shapes_only.json
python3 scripts/pmc_summary.py gpurun_out/${tag}_pmc_f gpurun_out/${tag}_pmc_w gpurun_out/${tag}_shapes_only.json gpurun_out/${tag}_pmc_traffic.json > gpurun_out/${tag}_pmc_gemm.txt || exit 1
rm -rf gpurun_out/${tag}_pmc_f gpurun_out/${tag}_pmc_w
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit 1
head -24 gpurun_out/${tag}_serial_by_grid.txt
cat gpurun_out/${tag}_pmc_gemm.txt
cut -c1-300 gpurun_out/${tag}_bench.json

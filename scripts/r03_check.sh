#!/bin/bash
# GPU tests, then the driver's own bench command.  usage: scripts/r03_check.sh <tag> [pytest args]
set -o pipefail
tag=$1; shift
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread "$@" > gpurun_out/${tag}_gpu_tests.txt 2>&1
rc=$?
tail -5 gpurun_out/${tag}_gpu_tests.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit 1
cut -c1-1500 gpurun_out/${tag}_bench.json

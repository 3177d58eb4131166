#!/bin/bash
# A/B of one bench flag set on one box, alternating.  usage: scripts/r03_ab_bench.sh <tag> "<flags A>" "<flags B>" [rounds]
set -o pipefail
tag=$1; fa=$2; fb=$3; rounds=${4:-2}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for i in $(seq 1 $rounds); do
  for arm in A B; do
    if [ $arm = A ]; then f=$fa; else f=$fb; fi
    timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-generate-leg $f > gpurun_out/${tag}_${arm}${i}.json 2> gpurun_out/${tag}_${arm}${i}.err || { tail -5 gpurun_out/${tag}_${arm}${i}.err; exit 1; }
    python3 - gpurun_out/${tag}_${arm}${i}.json "$arm$i [$f]" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], "ms/step", d["ms_per_step"], "median", d["step_ms"]["median"], "one stream", d["one_after_the_other"]["ms_per_step"], "gemm TF/s", round(d["roofline"]["achieved"], 1), flush=True)
PY
  done
done

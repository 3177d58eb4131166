#!/bin/bash
# Multi-rank code path on one GPU: (1) two ranks over gloo on device 0 (host-staged buckets; balanced dealing of the
# global batch, fill beside updates), (2) one rank over RCCL with the overlapped gradient reduction.
# usage: scripts/r03_rehearse.sh <tag>
set -o pipefail
tag=$1
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
PG_REHEARSE_ONE_GPU=1 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_rehearse2.json 2> gpurun_out/${tag}_rehearse2.err || { tail -5 gpurun_out/${tag}_rehearse2.err; exit 1; }
cut -c1-400 gpurun_out/${tag}_rehearse2.json
PG_REHEARSE_SYNC=1 timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-generate-leg > gpurun_out/${tag}_sync1.json 2> gpurun_out/${tag}_sync1.err || { tail -5 gpurun_out/${tag}_sync1.err; exit 1; }
cut -c1-400 gpurun_out/${tag}_sync1.json

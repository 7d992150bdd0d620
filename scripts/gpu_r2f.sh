#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
timeout 1500 python scripts/stream_bench.py 2>gpurun_out/r2f/err.txt | tail -1 > gpurun_out/r2f/stream_bench_cfg4.json
cat gpurun_out/r2f/stream_bench_cfg4.json; tail -3 gpurun_out/r2f/err.txt

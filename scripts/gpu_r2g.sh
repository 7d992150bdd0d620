#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
timeout 1500 python scripts/stream_bench.py 2>gpurun_out/r2g/err.txt | tail -1 > gpurun_out/r2g/stream_bench_cfg4.json
cat gpurun_out/r2g/stream_bench_cfg4.json; tail -3 gpurun_out/r2g/err.txt

#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
timeout 600 python -m pytest tests/test_gpu_stream.py -m gpu -x -q 2>&1 | tail -12
LV_STREAM_MAP=2000000 LV_STREAM_AZ=1024 LV_STREAM_UPDATES=100 timeout 600 python scripts/stream_bench.py 2>gpurun_out/r2e/err.txt | tail -1 > gpurun_out/r2e/stream_bench_small.json
cat gpurun_out/r2e/stream_bench_small.json; tail -3 gpurun_out/r2e/err.txt

#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2l
make -s -C limo-velo_amd/host 2>&1 | tail -2
LV_STREAM_MAP=${LV_STREAM_MAP:-2000000} LV_STREAM_REVS=${LV_STREAM_REVS:-12} timeout 1200 python scripts/stream_bench_cpp.py 2>gpurun_out/r2l/err.log | tee gpurun_out/r2l/stream_cpp.json | cut -c1-1200
tail -3 gpurun_out/r2l/err.log

#!/bin/bash
# small-batch / small-window paths: parity tests of the rows they touch, then the cfg4 cycle at reduced size
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2k
timeout 1200 python -m pytest tests/test_gpu_deskew.py tests/test_gpu_cloud.py tests/test_gpu_map_add.py tests/test_gpu_pipeline.py tests/test_gpu_stream.py tests/test_gpu_golden.py tests/test_gpu_filter.py -x -q 2>&1 | tail -8
LV_STREAM_MAP=2000000 LV_STREAM_UPDATES=100 timeout 600 python scripts/stream_bench.py 2>gpurun_out/r2k/stream.err | tee gpurun_out/r2k/stream.json | cut -c1-900

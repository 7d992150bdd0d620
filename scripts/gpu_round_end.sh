#!/bin/bash
# end-of-round check: full GPU suite, then configs[4] at scale through the Python harness and through the C++ host program
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/round_end
make -s -C limo-velo_amd/host 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/round_end/pytest.log 2>&1; grep -E "passed|failed|error" gpurun_out/round_end/pytest.log | tail -3
timeout 900 python scripts/stream_bench.py 2>gpurun_out/round_end/stream.err | tee gpurun_out/round_end/stream_bench_cfg4.json | cut -c1-700
LV_STREAM_REVS=30 timeout 1200 python scripts/stream_bench_cpp.py 2>gpurun_out/round_end/cpp.err | tee gpurun_out/round_end/stream_cpp_cfg4.json | cut -c1-1200
tail -2 gpurun_out/round_end/cpp.err

#!/bin/bash
# full GPU suite + default bench
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/full
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/full/pytest.log 2>&1
grep -E "passed|failed|error" gpurun_out/full/pytest.log | tail -5
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/full/bench.json | cut -c1-400

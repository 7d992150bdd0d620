#!/bin/bash
# full GPU suite + default bench
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/full
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/full/bench.json | cut -c1-400

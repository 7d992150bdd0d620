#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/r2b/pytest.txt
cat gpurun_out/r2b/pytest.txt
timeout 600 python bench.py --no-cpu-baseline 2>gpurun_out/r2b/bench.stderr | tail -1 > gpurun_out/r2b/bench.json
cat gpurun_out/r2b/bench.json | python scripts/summ.py
tail -5 gpurun_out/r2b/bench.stderr

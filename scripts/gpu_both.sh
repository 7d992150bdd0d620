#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu" ; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
bash scripts/gpu_explore.sh

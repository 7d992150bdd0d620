#!/bin/bash
# Runs on the GPU box (via gpurun): smoke, GPU parity tests, a short bench.  Logs -> gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m1 -E "gfx9" || true
echo "== smoke" ; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -15
echo "== pytest gpu" ; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25
echo "== bench" ; timeout 600 python bench.py --steps 50 --warmup 5 2>&1 | tail -5 | tee gpurun_out/bench_first.json

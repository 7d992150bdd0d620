#!/bin/bash
# first run of the one-launch-per-pass kernel: smoke, the update parity tests, bench A/B against the three-kernel pass
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2h
export TMPDIR=/tmp
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
echo "== parity (update paths)"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "update or timed or ragged or ring or non_finite or correlated or headline" 2>&1 | tail -15
echo "== bench fused fit_sel=1"; LV_FIT_SEL=1 timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --rotate 0 2>&1 | tail -1 | tee gpurun_out/r2h/bench_fused_sel1.json
echo "== bench fused fit_sel=0"; LV_FIT_SEL=0 timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --rotate 0 2>&1 | tail -1 | tee gpurun_out/r2h/bench_fused_sel0.json
echo "== bench three-kernel"; LV_FUSED_PASS=0 timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --rotate 0 2>&1 | tail -1 | tee gpurun_out/r2h/bench_old.json

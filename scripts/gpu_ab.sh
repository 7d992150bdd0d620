#!/bin/bash
# A/B of environment knobs in one box: usage: gpu_ab.sh "VAR=a" "VAR=b" ...  (each: pass clocks summary + 2 bench lines)
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for kv in "$@"; do
  echo "=== $kv"
  env $kv LV_PASS_CLK=1 timeout 300 python scripts/pass_clocks.py 30 2>&1 | grep -v "books:" | grep "launch\|sum of\|fold\|gauss\|gain"
  for i in 1 2; do env $kv timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --rotate 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value']), 'it/s', round(d['ms_per_step']*1000,1), 'us')"; done
done

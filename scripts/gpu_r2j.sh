#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/trace_tmp
for mode in 1 0; do
  rm -rf $OUT; mkdir -p $OUT
  cd /tmp
  LV_FUSED_PASS=$mode timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o s -- python $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline --rotate 0 > $OUT/log.txt 2>&1
  echo "== LV_FUSED_PASS=$mode"
  if [ $mode = 1 ]; then per=5; else per=12; fi
  python $R/scripts/trace_summ2.py $(ls $OUT/*kernel_trace.csv | head -1) $per
  tail -1 $OUT/log.txt | cut -c1-200
done
rm -rf $OUT

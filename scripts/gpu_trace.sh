#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_tmp
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 150 rocprofv3 --kernel-trace --output-format csv -d $OUT -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/log.txt 2>&1
python $GRAFT_REPO_ROOT/scripts/trace_summ.py $(ls $OUT/*kernel_trace.csv | head -1)
tail -1 $OUT/log.txt | python $GRAFT_REPO_ROOT/scripts/summ.py
rm -f $OUT/*kernel_trace.csv

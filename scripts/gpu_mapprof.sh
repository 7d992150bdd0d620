#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/mapprof
mkdir -p $OUT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o map -- python $GRAFT_REPO_ROOT/scripts/map_add_prof.py > $OUT/stats.log 2>&1
find $OUT -name "*kernel_stats.csv" -exec head -40 {} \;
find $OUT -name "*kernel_trace.csv" -delete
tail -3 $OUT/stats.log

#!/bin/bash
# per-kernel profile of the cfg4 mapping cycle at a reduced stream size (small windows: the cycle is latency-bound)
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/streamprof
mkdir -p $OUT
cd /tmp
export LV_STREAM_MAP=${LV_STREAM_MAP:-2000000} LV_STREAM_UPDATES=${LV_STREAM_UPDATES:-100}
timeout 600 python $GRAFT_REPO_ROOT/scripts/stream_bench.py > $OUT/plain.json 2> $OUT/plain.err
tail -c 1500 $OUT/plain.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stream -- python $GRAFT_REPO_ROOT/scripts/stream_bench.py > $OUT/stats.log 2>&1
find $OUT -name "*kernel_stats.csv" -exec head -70 {} \;
find $OUT -name "*kernel_trace.csv" -delete

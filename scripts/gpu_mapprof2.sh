#!/bin/bash
# rocprofv3 kernel statistics of the map insert, both cases of scripts/map_add_prof2.py
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/mapprof2
mkdir -p $OUT
cd /tmp
for m in same new; do
  MODE=$m timeout 300 python $GRAFT_REPO_ROOT/scripts/map_add_prof2.py 2>/dev/null | tail -1
  MODE=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$m -o map -- python $GRAFT_REPO_ROOT/scripts/map_add_prof2.py > $OUT/$m.log 2>&1
  f=$(find $OUT/$m -name "*kernel_stats.csv" | head -1)
  cp $f $OUT/map_add_${m}_kernel_stats.csv
  echo "== $m"; python $GRAFT_REPO_ROOT/scripts/kstats.py $OUT/map_add_${m}_kernel_stats.csv 24 | grep -v "pass_kernel\|map_bucket\|bucket_\|box_build\|map_insert\|map_bounds\|cell_\|map_count\|map_keys\|map_gather"
done
find $OUT -name "*kernel_trace.csv" -delete

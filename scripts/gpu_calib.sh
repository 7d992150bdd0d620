#!/bin/bash
# FETCH_SIZE calibration on known byte counts (scripts/ubench/fetch_calib.hip) -> gpurun_out/calib/fetch_calib.json
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/calib
mkdir -p $OUT
cd /tmp
$GRAFT_REPO_ROOT/scripts/ubench/fetch_calib > $OUT/known.json
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc -o calib -- $GRAFT_REPO_ROOT/scripts/ubench/fetch_calib > $OUT/pmc.log 2>&1
python $GRAFT_REPO_ROOT/scripts/ubench/fetch_calib_summ.py $OUT/pmc $OUT/known.json | tee $OUT/fetch_calib.json
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o calib -- $GRAFT_REPO_ROOT/scripts/ubench/fetch_calib > $OUT/stats.log 2>&1
find $OUT -name "*kernel_stats.csv" -exec cat {} \;
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete

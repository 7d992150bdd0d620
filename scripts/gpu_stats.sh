#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/stats_tmp
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 5 --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/log.txt 2>&1
python - <<PY
import csv
for r in csv.DictReader(open("$OUT/s_kernel_stats.csv")):
    n = r["Name"]
    if "rocprim" in n or "hipcub" in n: continue
    print("%-46s calls %5s avg %9.1f ns  min %8s max %8s" % (n[:46], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
PY
tail -1 $OUT/log.txt | python $GRAFT_REPO_ROOT/scripts/summ.py
rm -f $OUT/*kernel_trace.csv

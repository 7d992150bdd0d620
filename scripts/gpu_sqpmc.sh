#!/bin/bash
# SQ counters of the search kernel (instruction mix / VALU busy / wait cycles), per pass index
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/sqpmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-}"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o s -- $CMD > $OUT/log$i.txt 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    # order search kernel dispatches -> pass index = order % 4
    disp = sorted({int(r["Dispatch_Id"]) for r in rows if "search_kernel" in r["Kernel_Name"]})
    idx = {d: i % 4 for i, d in enumerate(disp)}
    for r in rows:
        if "search_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]][idx[int(r["Dispatch_Id"])]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print("%-32s" % k, " ".join("%14.0f" % (sum(acc[k][p]) / max(len(acc[k][p]), 1)) for p in range(4)))
PY
find $OUT -name "*.csv" -delete

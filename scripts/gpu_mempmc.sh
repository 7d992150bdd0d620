#!/bin/bash
# memory-pipeline counters of the search kernel (per pass index): where do its loads wait?
# NOTE: a set with TA_* counters made rocprofv3 abort and hang on this pool (cost: the whole gpurun limit); every
# rocprofv3 call in scripts/ now runs under `timeout`.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/mempmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-}"
i=0
for set in "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "TCC_BUSY_avr TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_REQ_sum TCC_CYCLE_sum" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL GRBM_GUI_ACTIVE GRBM_TA_BUSY"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o s -- $CMD > $OUT/log$i.txt 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    disp = sorted({int(r["Dispatch_Id"]) for r in rows if "search_kernel" in r["Kernel_Name"]})
    idx = {d: i % 4 for i, d in enumerate(disp)}
    for r in rows:
        if "search_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]][idx[int(r["Dispatch_Id"])]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print("%-40s" % k, " ".join("%14.1f" % (sum(acc[k][p]) / max(len(acc[k][p]), 1)) for p in range(4)))
PY
find $OUT -name "*.csv" -delete

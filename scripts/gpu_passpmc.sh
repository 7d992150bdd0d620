#!/bin/bash
# SQ counters of pass_kernel on the headline bench (where do the wave cycles go: parked on memory / barriers, issue stalls, VALU)
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/passpmc
mkdir -p $OUT
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --regions 1 --no-cpu-baseline --no-parity --no-cycle --rotate 0 --no-phases"
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/a -o p -- $CMD > $OUT/a.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS --output-format csv -d $OUT/b -o p -- $CMD > $OUT/b.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $OUT/c -o p -- $CMD > $OUT/c.log 2>&1
python - <<PY
import csv, glob, collections
for sub in ("a","b","c"):
    for f in glob.glob("$OUT/"+sub+"/**/*counter_collection.csv", recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            n=r["Kernel_Name"]
            if "pass_kernel" not in n: continue
            g=int(r.get("Grid_Size","0") or 0)
            key="closing" if g<=1024 else "searching"
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for n,d in sorted(acc.items()):
            print(sub, n, {k: round(sum(v)/len(v)) for k,v in d.items()}, "launches", len(next(iter(d.values()))))
PY
grep -il "error" $OUT/*.log | head
find $OUT -name "*.csv" -delete

#!/bin/bash
# PMC pass over the map insert's kernels (MODE=new of scripts/map_add_prof2.py): instructions and memory requests per kernel
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/mappmc
mkdir -p $OUT
cd /tmp
MODE=${MODE:-new} timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/a -o m -- python $GRAFT_REPO_ROOT/scripts/map_add_prof2.py > $OUT/a.log 2>&1
MODE=${MODE:-new} timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --output-format csv -d $OUT/b -o m -- python $GRAFT_REPO_ROOT/scripts/map_add_prof2.py > $OUT/b.log 2>&1
python - <<PY
import csv, glob, collections
for sub in ("a","b"):
    for f in glob.glob("$OUT/"+sub+"/**/*counter_collection.csv", recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            n=r["Kernel_Name"]
            if "inc_" not in n: continue
            n=n.split("(")[0].replace("lv::","")
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for n,d in sorted(acc.items()):
            print(n, {k: round(sum(v)/len(v)) for k,v in d.items()}, "launches", len(next(iter(d.values()))))
PY
tail -2 $OUT/a.log $OUT/b.log | grep -i "error\|fail" | head
find $OUT -name "*.csv" -delete

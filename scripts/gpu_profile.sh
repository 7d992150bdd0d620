#!/bin/bash
# Round artefacts: bench line + rocprofv3 kernel stats + PMC traffic for the same command.
set -u
export TMPDIR=/tmp
R=${ROUND:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_$R
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py 2>$OUT/bench.stderr | tail -1 > $OUT/bench_$R.json
cat $OUT/bench_$R.json | python scripts/summ.py
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline"
cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $R -- $CMD > $OUT/stats.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o $R -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -o $R -- $CMD > $OUT/pmc_tcc.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*.csv" | head -20
python - <<PY
import csv, collections, glob, json
out = "$OUT"
res = {}
for sub, names in (("pmc_fetch", ["FETCH_SIZE"]), ("pmc_tcc", ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"])):
    for f in glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "search_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            res[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
json.dump(res, open(out + "/pmc_search_$R.json", "w"), indent=1)
print(res)
PY
find $OUT -name "*counter_collection.csv" -delete
du -sh $OUT

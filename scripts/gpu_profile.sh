#!/bin/bash
# Round artefacts: bench line + rocprofv3 kernel stats + PMC traffic for the same command.
set -u
export TMPDIR=/tmp
R=${ROUND:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_$R
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py 2>$OUT/bench.stderr | tail -1 > $OUT/bench_$R.json
cat $OUT/bench_$R.json | python scripts/summ.py
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-parity --no-cycle"
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
kinds = set()
for sub, names in (("pmc_fetch", ["FETCH_SIZE"]), ("pmc_tcc", ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"])):
    for f in glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            # the dominant kernel: pass_kernel (one launch per pass) or search_kernel (three-kernel pass: LV_FUSED_PASS=0);
            # the closing launch of an update (one workgroup, no search) is not a searching launch: grid 1 -> skipped
            n = r["Kernel_Name"]
            if "search_kernel" in n or ("pass_kernel" in n and int(r.get("Grid_Size", "0") or 0) > 1024):
                kinds.add("pass" if "pass_kernel" in n else "search")
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            res[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
kind = "pass" if "pass" in kinds else "search"
json.dump(res, open(out + "/pmc_%s_$R.json" % kind, "w"), indent=1)
print(kind, res)
PY
find $OUT -name "*counter_collection.csv" -delete
# phase stamps of pass_kernel (wall-clock timeline of the launches of an update, phases per workgroup)
cd $GRAFT_REPO_ROOT
LV_PASS_CLK=1 timeout 300 python scripts/pass_clocks.py 30 > $OUT/pass_clocks_$R.txt 2>&1
grep -v cycles $OUT/pass_clocks_$R.txt
du -sh $OUT

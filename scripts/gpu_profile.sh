#!/bin/bash
# Round artefacts: bench line + rocprofv3 kernel stats + PMC traffic for the same command.
set -u
export TMPDIR=/tmp
R=${ROUND:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_$R
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py 2>$OUT/bench.stderr | tail -1 > $OUT/bench_$R.json
cat $OUT/bench_$R.json | python scripts/summ.py
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --regions 9 --resident-only --no-cpu-baseline --no-parity --no-cycle --no-phases --rotate 0"
cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $R -- $CMD > $OUT/stats.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o $R -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -o $R -- $CMD > $OUT/pmc_tcc.log 2>&1
# the launch sequence of the timed (resident, pipelined) step from the kernel trace of the stats run: duration of every launch of
# an update by its index, the gaps between them, and the period from one update's first launch to the next one's
python - <<PY
import csv, glob, statistics as st
f = glob.glob("$OUT/stats/**/*kernel_trace.csv", recursive=True)
rows = []
for r in csv.DictReader(open(f[0])):
    if "pass_kernel" in r["Kernel_Name"]:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("<")[1].split(">")[0].split(", ")[1] == "true"))   # pass_kernel<EXT, CLOSING, MULTI>
rows.sort()
ups, cur = [], []
for s, e, closing in rows:
    cur.append((s, e))
    if closing:
        ups.append(cur); cur = []
ups = [u for u in ups if len(u) == 5]
# keep the steady part of the resident regions: updates whose first launch follows the previous closing launch within 4 us
steady = [u for p, u in zip(ups, ups[1:]) if u[0][0] - p[-1][1] < 4000]
with open("$OUT/${R}_resident_sequence.txt", "w") as o:
    print(f"updates in the trace {len(ups)}, of them back to back (first launch < 4 us behind the previous closing launch) {len(steady)}", file=o)
    for i in range(5):
        d = [(u[i][1] - u[i][0]) / 1e3 for u in steady]
        g = [(u[i][0] - u[i - 1][1]) / 1e3 for u in steady] if i else [0.0]
        print(f"launch {i}{' (closing)' if i == 4 else ''}: duration median {st.median(d):.2f} us (p10 {sorted(d)[len(d)//10]:.2f} p90 {sorted(d)[len(d)*9//10]:.2f})" + (f"  gap before it median {st.median(g):.2f} us" if i else ""), file=o)
    per = [(b[0][0] - a[0][0]) / 1e3 for a, b in zip(steady, steady[1:]) if b[0][0] - a[-1][1] < 4000]
    gap0 = [(b[0][0] - a[-1][1]) / 1e3 for a, b in zip(steady, steady[1:]) if b[0][0] - a[-1][1] < 4000]
    print(f"period of an update (first launch to first launch) median {st.median(per):.2f} us; gap closing -> next first launch median {st.median(gap0):.2f} us", file=o)
print(open("$OUT/${R}_resident_sequence.txt").read())
PY
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

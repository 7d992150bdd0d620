#!/bin/bash
# round 5: is the occasional ~4.5 ms cycle beside a background rebuild tied to the paced form?  N repetitions of the configs[4]
# replay per form, every cycle dumped, entry points slower than 2 ms reported (LV_SLOW_CALL_MS).
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=/root/repo/gpurun_out/r05_outlier
mkdir -p $O; rm -f $O/*.txt
F="LV_DEMO_FORCE_REBUILD=80,LV_DEMO_FORCE_REBUILD2=160,LV_SLOW_CALL_MS=2"
SPEC=""
for i in ${REPS:-1 2 3 4}; do
  SPEC="$SPEC;paced32_$i=$F,LV_DEMO_CYCLE_DUMP=$O/paced32_$i.txt;s256_$i=$F,LV_RELIN_PACED_WGS=0,LV_DEMO_CYCLE_DUMP=$O/s256_$i.txt"
done
LV_STREAM_ONLY_AB=1 LV_STREAM_AB="${SPEC#;}" timeout 1500 python scripts/stream_bench_cpp.py 2>$O/err.txt | tail -1 > $O/stream.json
python - <<PY
import glob, json
d = json.load(open("$O/stream.json"))
for k, v in d.items():
    if isinstance(v, dict): print(k, "cycle ms", v.get("cycle_ms"), "second", v["forced_rebuild"]["second_cycle_ms"])
for f in sorted(glob.glob("$O/*_?.txt")):
    rows = [l.split() for l in open(f)]
    ms = [float(r[1]) for r in rows[30:]]
    med = sorted(ms)[len(ms) // 2]
    big = [(r[0], r[1], r[2], r[3], r[4]) for r in rows[30:] if float(r[1]) > 1.0]
    print(f.split("/")[-1], "median", med, "max", max(ms), "cycles > 1 ms (cycle, ms, state, adopted, journal):", big)
PY
grep "slow call" $O/err.txt | sort | uniq -c | sort -rn | head -20

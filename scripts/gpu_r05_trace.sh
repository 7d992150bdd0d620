#!/bin/bash
# kernel trace of the C++ stream replay with a forced background rebuild: where do the main stream's launches wait?
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_trace
mkdir -p $O
cd /tmp
LV_STREAM_ROCPROF=$O/prof LV_STREAM_AB="forced_async=LV_DEMO_FORCE_REBUILD=80,LV_SLOW_CALL_MS=${SLOW_MS:-1.5}${SLICE:+,LV_RELIN_SLICE_WGS=$SLICE}" LV_STREAM_ONLY_AB=1 timeout 1500 python $GRAFT_REPO_ROOT/scripts/stream_bench_cpp.py 2>$O/err.txt | tail -1 > $O/stream.json
grep "slow call" $O/err.txt | head -40
python - <<PY
import csv, glob, collections
f = glob.glob("$O/prof/**/*kernel_trace.csv", recursive=True)
print(f)
rows = list(csv.DictReader(open(f[0])))
print(len(rows), "kernels; columns", list(rows[0].keys()))
q = collections.Counter(r["Queue_Id"] for r in rows)
print("queues", q)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
# per queue: name counts; find the rebuild queue = the one with map_* / bucket_* kernels
byq = collections.defaultdict(list)
for r in rows: byq[r["Queue_Id"]].append(r)
for qid, rs in byq.items():
    names = collections.Counter(r["Kernel_Name"].split("(")[0][:60] for r in rs)
    print("queue", qid, len(rs), names.most_common(6))
# longest kernels overall
top = sorted(rows, key=lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), reverse=True)[:25]
for r in top:
    print("%9.3f ms  start %10.3f ms  q %s  grid %s  %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, (int(r["Start_Timestamp"]) - t0) / 1e6, r["Queue_Id"], r.get("Grid_Size"), r["Kernel_Name"][:90]))
# kernels of the OTHER queues (rebuild worker, insert side stream) longer than 0.25 ms, by name: count / max / total
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
mq = set(qid for qid, rs in byq.items() if any("pass_kernel" in r["Kernel_Name"] for r in rs))
first_pass = min(int(r["Start_Timestamp"]) for r in rows if "pass_kernel" in r["Kernel_Name"])
for r in rows:
    if r["Queue_Id"] in mq or int(r["Start_Timestamp"]) < first_pass: continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if d > ${OTHER_MS:-0.25}:
        a = agg[r["Kernel_Name"].split("(")[0][:70]]; a[0] += 1; a[1] = max(a[1], d); a[2] += d
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]): print("other-queue kernel > ${OTHER_MS:-0.25} ms: %-72s n %4d  max %7.3f ms  total %8.3f ms" % (k, v[0], v[1], v[2]))
# the main queue = the one holding pass_kernel: list its gaps > 1 ms
mainq = [qid for qid, rs in byq.items() if any("pass_kernel" in r["Kernel_Name"] for r in rs)]
print("main queues", mainq)
import json
out = []
for qid in mainq:
    rs = byq[qid]
    for a, b in zip(rs, rs[1:]):
        gap = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e6
        dur = (int(b["End_Timestamp"]) - int(b["Start_Timestamp"])) / 1e6
        if gap > ${EVENT_MS:-1.0} or dur > ${EVENT_MS:-1.0}:
            out.append((round((int(b["Start_Timestamp"]) - t0) / 1e6, 3), round(gap, 3), round(dur, 3), b["Kernel_Name"][:70]))
for o in out[:60]: print("main-queue event: at %.3f ms gap-before %.3f ms duration %.3f ms %s" % o)
PY
find $O -name "*kernel_trace.csv" -size +30M -delete
du -sh $O

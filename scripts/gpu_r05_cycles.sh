#!/bin/bash
# round 5: WHICH cycles of the configs[4] replay are slow beside a background rebuild?  Every cycle's time with the rebuild's
# state (lv_map_rebuild_status: 4 allocating, 5 allocated, 1 rebuilding / replaying, 2 ready, 0 idle), for the paced and a sliced form.
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_cycles
mkdir -p $O
F="LV_DEMO_FORCE_REBUILD=80,LV_DEMO_FORCE_REBUILD2=160"
LV_STREAM_ONLY_AB=1 LV_STREAM_AB="${SPEC:-paced120=$F,LV_DEMO_CYCLE_DUMP=$O/paced120.txt;s256=$F,LV_RELIN_PACED_WGS=0,LV_RELIN_SLICE_WGS=256,LV_DEMO_CYCLE_DUMP=$O/s256.txt}" timeout 1500 python scripts/stream_bench_cpp.py 2>$O/err.txt | tail -1 > $O/stream.json
python - <<PY
import glob, json
d = json.load(open("$O/stream.json"))
for k, v in d.items():
    if isinstance(v, dict): print(k, "cycle ms", v.get("cycle_ms"), "second", v["forced_rebuild"]["second_cycle_ms"])
for f in sorted(glob.glob("$O/*.txt")):
    if f.endswith("err.txt"): continue
    rows = [l.split() for l in open(f)]
    ms = sorted(float(r[1]) for r in rows[30:])
    med = ms[len(ms) // 2]
    print(f.split("/")[-1], "median", med, "cycles > 1.6 x median:")
    prev = None
    for r in rows[30:]:
        if float(r[1]) > 1.6 * med: print("   cycle", r[0], "ms", r[1], "state", r[2], "adopted", r[3], "journal", r[4])
    # state transitions
    for a, b in zip(rows, rows[1:]):
        if a[2] != b[2] or a[3] != b[3]: print("   transition at cycle", b[0], "state", a[2], "->", b[2], "adopted", b[3])
PY

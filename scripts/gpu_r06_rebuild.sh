#!/bin/bash
# round 6: the background rebuild with its second store allocated at set-up time (lv_map_reserve_rebuild) against the round-5 form
# (the first rebuild's worker allocates): the async tests, then REPS replays of configs[4] through the C++ host program with two
# forced background rebuilds each, both ways in one box (AB="name=ENV=v,...;name2=..." replaces the two variants).
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06_rebuild
mkdir -p $O
make -s -C limo-velo_amd/host 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_map_async.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
LV_STREAM_ONLY_AB=1 LV_STREAM_AB="${AB:-reserved=LV_DEMO_FORCE_REBUILD=80,LV_DEMO_FORCE_REBUILD2=160;worker_allocates=LV_DEMO_FORCE_REBUILD=80,LV_DEMO_FORCE_REBUILD2=160,LV_DEMO_NO_REBUILD_RESERVE=1}" LV_STREAM_REPS=${REPS:-8} timeout 2400 python scripts/stream_bench_cpp.py 2>$O/stream_cpp.err | tail -1 > $O/rebuild_replays.json
python - <<PY
import json
d = json.load(open("$O/rebuild_replays.json"))
for k, v in d.items():
    if isinstance(v, dict) and "cycle_ms" in v:
        c = v["cycle_ms"]; f = v.get("forced_rebuild") or {}; s2 = f.get("second_cycle_ms") or {}
        print("%-22s median %.3f p99 %.3f (%.2fx) max %.3f (%.2fx) | from the first forcing call: max %.3f | second rebuild: median %.3f p99 %.3f max %.3f | call %.3f ms | updates/s %.0f" % (
            k, c["median"], c["p99"], c["p99"]/c["median"], c["max"], c["max"]/c["median"], f.get("max_cycle_ms_from_there", 0), s2.get("median", 0), s2.get("p99", 0), s2.get("max", 0), f.get("call_ms", 0), v["updates_per_s"]))
PY
tail -3 $O/stream_cpp.err

#!/bin/bash
# round 5: background map rebuild under load — the tests, then BASELINE configs[4] through the C++ host program (stream_demo =
# the reference's loop over the shim; 10 M-point prior map) with a re-linearisation forced after update 100: in the background
# (lv_map_relinearise_async) and stop-the-world (lv_map_relinearise); cycle-time median / p99 / max in every record
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_stream
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_map_async.py -x -q -s 2>&1 | grep -v "^$" | tail -8
LV_STREAM_AB="forced_async=LV_DEMO_FORCE_REBUILD=80,LV_DEMO_FORCE_REBUILD2=160;forced_sync=LV_DEMO_FORCE_REBUILD=100:sync" timeout 1500 python scripts/stream_bench_cpp.py 2>$O/stream_cpp.err | tail -1 > $O/stream_cpp_cfg4_r05.json
python - <<PY
import json
d = json.load(open("$O/stream_cpp_cfg4_r05.json"))
for k, v in d.items():
    if isinstance(v, dict):
        print(k, v["updates_per_s"], "updates/s | cycle ms", v.get("cycle_ms"), "| forced", v.get("forced_rebuild"), "| rmse", round(v["rmse_vs_truth_m"], 5), "map", v["map_points"])
PY
tail -3 $O/stream_cpp.err

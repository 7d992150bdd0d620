#!/bin/bash
# round 6: BASELINE configs[4] at full scale on the single-replicated-level map — the Python harness with lockstep parity against
# the oracle every 6th update, and the C++ host program (the reference's loop over the shim) plain, with two forced background
# re-linearisations, and with a stop-the-world one.
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06_stream
mkdir -p $O
make -s -C limo-velo_amd/host 2>&1 | tail -2
LV_STREAM_LOCKSTEP=6 timeout 1500 python scripts/stream_bench.py 2>$O/stream_bench.err | tail -1 > $O/stream_bench_cfg4_r06.json
python -c "
import json; d=json.load(open('$O/stream_bench_cfg4_r06.json'))
print('python harness:', round(d['updates_per_s_end_to_end'],1), 'updates/s', {k: round(v,3) for k,v in d['stage_ms_per_update'].items()}, 'rmse', round(d['rmse_vs_truth_m'],5), 'lockstep', d['lockstep_vs_oracle'], 'bytes', d['map_stats']['bytes'], 'living', d['map_stats']['living'])"
LV_STREAM_AB="forced_async=LV_DEMO_FORCE_REBUILD=80,LV_DEMO_FORCE_REBUILD2=160;forced_sync=LV_DEMO_FORCE_REBUILD=100:sync" LV_STREAM_REPS=2 timeout 1500 python scripts/stream_bench_cpp.py 2>$O/stream_cpp.err | tail -1 > $O/stream_cpp_cfg4_r06.json
python - <<PY
import json
d = json.load(open("$O/stream_cpp_cfg4_r06.json"))
for k, v in d.items():
    if isinstance(v, dict):
        print(k, v["updates_per_s"], "updates/s | after 30:", v.get("updates_per_s_after_30"), "| cycle ms", v.get("cycle_ms"), "| forced", {kk: vv for kk, vv in (v.get("forced_rebuild") or {}).items() if kk != "note"}, "| rmse", round(v["rmse_vs_truth_m"], 5), "map", v["map_points"])
PY
tail -3 $O/stream_cpp.err

#!/bin/bash
# Round artefacts of BASELINE configs[4] at full scale (10 M-point map): the Python harness with lockstep parity against the
# oracle every 6th update (50 checks in 300 updates), and the C++ host program (the reference's loop over the shim).
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=${ROUND:-r05}
OUT=gpurun_out/streams_$R
mkdir -p $OUT
LV_STREAM_LOCKSTEP=6 timeout 1500 python scripts/stream_bench.py 2>$OUT/stream_bench.err | tail -1 > $OUT/stream_bench_cfg4_$R.json
python -c "
import json; d=json.load(open('$OUT/stream_bench_cfg4_$R.json'))
print('python harness:', round(d['updates_per_s_end_to_end'],1), 'updates/s', {k: round(v,3) for k,v in d['stage_ms_per_update'].items()}, 'rmse', round(d['rmse_vs_truth_m'],5), 'lockstep', d['lockstep_vs_oracle'], 'bytes', d['map_stats']['bytes'])"
LV_STREAM_REPS=2 timeout 1500 python scripts/stream_bench_cpp.py 2>$OUT/stream_cpp.err | tail -1 > $OUT/stream_cpp_cfg4_$R.json
python -c "
import json; d=json.load(open('$OUT/stream_cpp_cfg4_$R.json'))
for k,v in d.items():
    if isinstance(v,dict): print('C++ host', k, v['updates_per_s'], 'updates/s rmse', round(v['rmse_vs_truth_m'],5), 'map', v['map_points'])"

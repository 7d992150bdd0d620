#!/bin/bash
# which call of a cycle stalls while the background rebuild runs?  (LV_SLOW_CALL_MS diagnostic, stream_demo with a forced rebuild)
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_slow
mkdir -p $O
LV_STREAM_AB="forced_async=LV_DEMO_FORCE_REBUILD=80,LV_DEMO_FORCE_REBUILD2=160,LV_SLOW_CALL_MS=1.5" LV_STREAM_ONLY_AB=1 timeout 1500 python scripts/stream_bench_cpp.py 2>$O/stream_cpp.err | tail -1 > $O/stream.json
grep "slow call" $O/stream_cpp.err | sort | uniq -c | sort -rn | head -30
grep "slow call" $O/stream_cpp.err | head -60 > $O/slow_calls.txt
python -c "
import json; d=json.load(open('$O/stream.json'))
for k,v in d.items():
    if isinstance(v,dict): print(k, v['updates_per_s'], v.get('cycle_ms'), v.get('forced_rebuild'))"

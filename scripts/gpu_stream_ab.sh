#!/bin/bash
# A/B on ONE box: the mapping cycle at configs[4] scale with a previous build of the library (copy it to scripts/ab/liblimovelo_hip_old.so
# first: e.g. `git worktree add /tmp/old <commit> && make -C /tmp/old/limo-velo_amd/csrc`) and with the current one
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/stream_ab
cp limo-velo_amd/liblimovelo_hip.so /tmp/new.so
for v in old new old new; do
  if [ $v = old ]; then cp scripts/ab/liblimovelo_hip_old.so limo-velo_amd/liblimovelo_hip.so; else cp /tmp/new.so limo-velo_amd/liblimovelo_hip.so; fi
  LV_STREAM_MAP=${LV_STREAM_MAP:-10000000} LV_STREAM_UPDATES=200 timeout 900 python scripts/stream_bench.py 2>>gpurun_out/stream_ab/err.log > gpurun_out/stream_ab/$v.json
  python - <<P
import json
d=json.load(open("gpurun_out/stream_ab/$v.json"))
print("$v", round(d["updates_per_s_end_to_end"],1), {k: round(x,3) for k,x in d["stage_ms_per_update"].items()}, "build", round(d["map_build_ms"],1))
P
done
cp /tmp/new.so limo-velo_amd/liblimovelo_hip.so

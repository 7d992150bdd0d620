#!/bin/bash
# round 6, final tree: full suite + smoke + the round's profile artefacts (gpu_final.sh), the scan-size sweep behind DESIGN §5's
# prediction, the map memory / insert timings, configs[4] at scale (Python harness with lockstep parity; C++ host program).
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ROUND=r06 bash scripts/gpu_final.sh 2>&1 | tail -30
timeout 600 python scripts/shard_size_sweep.py > gpurun_out/shard_size_sweep_r06.txt 2>&1; tail -12 gpurun_out/shard_size_sweep_r06.txt
bash scripts/gpu_r06_map.sh > gpurun_out/r06_map_out.txt 2>&1; grep -E "^map|cycle" gpurun_out/r06_map_out.txt | head -6
bash scripts/gpu_r06_stream.sh > gpurun_out/r06_stream_out.txt 2>&1; head -3 gpurun_out/r06_stream_out.txt | cut -c1-400

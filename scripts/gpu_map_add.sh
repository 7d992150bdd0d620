#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
timeout 900 python -m pytest tests/test_gpu_map_add.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -15
timeout 500 python scripts/map_add_timing.py 2>gpurun_out/r2d/err.txt | tail -1 > gpurun_out/r2d/map_add_timing.json
python - <<PY
import json
d=json.load(open("gpurun_out/r2d/map_add_timing.json"))
for k in ("map_build_ms","map_add_scan_ms","map_sizes","map_add_host_points_ms","update_ms_on_incremental_map","relinearise_ms","update_ms_after_relinearise","evict_box_ms"): print(k, d[k])
print(d["stats_after_scan_adds"])
PY
tail -3 gpurun_out/r2d/err.txt
bash scripts/gpu_mapprof.sh > /dev/null 2>&1; python scripts/kstats.py gpurun_out/mapprof/stats/map_kernel_stats.csv 22

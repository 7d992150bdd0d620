#!/bin/bash
# per-kernel times of lv_map_build (1M points, 3 builds)
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/mapstats
rm -rf $OUT; mkdir -p $OUT
cat > /tmp/mb.py <<PY
import sys; sys.path.insert(0, "$GRAFT_REPO_ROOT")
import torch, lvamd; lvamd.load()
from limo_velo_amd import capi, synth
sc = synth.make_scene(1_048_576, 1024)
ctx = capi.Context(capi.default_params())
for _ in range(3):
    ctx.map_build(sc["map_xyz"]); ctx.synchronize()
PY
cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python /tmp/mb.py > $OUT/log.txt 2>&1
python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/s_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    print("%-70s calls %4s total %8.2f ms avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
print("sum of kernels per build: %.2f ms" % (tot / 3e6))
PY
rm -f $OUT/*kernel_trace.csv

#!/bin/bash
# round 5: how small must the background rebuild's launch slices be for the cycle's whole-CU workgroups not to notice?
# BASELINE configs[4] through the C++ host program with two forced background re-linearisations, LV_RELIN_SLICE_WGS swept.
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_slices
mkdir -p $O
F="LV_DEMO_FORCE_REBUILD=80,LV_DEMO_FORCE_REBUILD2=160"
LV_STREAM_AB="${SPEC:-s4096=$F;s1024=$F,LV_RELIN_SLICE_WGS=1024;s256=$F,LV_RELIN_SLICE_WGS=256;s64=$F,LV_RELIN_SLICE_WGS=64}" timeout 1500 python scripts/stream_bench_cpp.py 2>$O/stream_cpp.err | tail -1 > $O/stream_cpp_slices.json
python - <<PY
import json
d = json.load(open("$O/stream_cpp_slices.json"))
for k, v in d.items():
    if isinstance(v, dict):
        print(k, v["updates_per_s"], "updates/s | cycle ms", v.get("cycle_ms"), "| forced", v.get("forced_rebuild"), "| rmse", round(v["rmse_vs_truth_m"], 5), "map", v["map_points"])
PY
tail -3 $O/stream_cpp.err

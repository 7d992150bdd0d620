#!/bin/bash
# round 6: the single-replicated-level map against the library of the round's first commit (scripts/ab/base.so), in ONE box:
# headline bench with the whole cycle, map memory per point, 64k-point inserts (new space / same scan).
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r06_map
mkdir -p $OUT
for v in tree base; do
  if [ $v = tree ]; then L=$GRAFT_REPO_ROOT/limo-velo_amd/liblimovelo_hip.so; else L=$GRAFT_REPO_ROOT/scripts/ab/$v.so; fi
  [ -e $L ] || continue
  echo "==== $v"
  LV_LIB_PATH=$L timeout 600 python bench.py --no-cpu-baseline --steps 200 2>$OUT/bench_$v.err | tail -1 > $OUT/bench_$v.json
  python - <<P
import json
d=json.load(open("$OUT/bench_$v.json")); r=d["roofline"]
print("it/s", round(d["value"]), "pipelined", round(d.get("value_pipelined",0)), "kernel by launch", r.get("kernel_us_by_launch"), "parity ok", d["parity"]["ok"])
print("cycle", d.get("cycle_ms_64k"))
print("large_n", d["large_n"]["us_per_update"], "ext", d["ext"]["us_per_update"])
P
  LV_LIB_PATH=$L timeout 600 python scripts/map_mem.py 2>&1 | grep "^map" | tee $OUT/mem_$v.txt
  LV_LIB_PATH=$L timeout 600 python scripts/map_add_timing.py 2>/dev/null | tail -1 > $OUT/add_$v.json
  python - <<P
import json
d=json.load(open("$OUT/add_$v.json"))
print({k:d[k] for k in d if k!="stats_after_scan_adds"})
P
done

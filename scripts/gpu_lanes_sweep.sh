#!/bin/bash
# search_kernel (three-kernel pass) by lanes per scan point: what the VALU share of the selection networks costs at S = 2 / 4 / 8 / 16
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lanes
for s in 8 4 2 16 8; do
  LV_FUSED_PASS=0 timeout 300 python bench.py --no-cpu-baseline --no-parity --no-cycle --rotate 0 --steps 300 --lanes $s 2>/dev/null | tail -1 > gpurun_out/lanes/s$s.json
  python - <<P
import json
d=json.load(open("gpurun_out/lanes/s$s.json")); r=d["roofline"]
print("lanes", $s, "it/s", round(d["value"]), "search_kernel us", round(r["avg_kernel_us"],2), "per pass", r["last_update_match_us_per_pass"], "fit+solve", round(r["avg_fit_plus_solve_us"],2))
P
done

#!/bin/bash
# A/B of two builds of the library on the headline bench inside ONE box: the tree's build vs scripts/ab/liblimovelo_hip_var.so
# (built elsewhere, e.g. `make EXTRA=-DLV_PASS_DYNAMIC` in a copy of csrc/).  The variant first passes the pass-kernel parity tests.
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/lib_ab
cp limo-velo_amd/liblimovelo_hip.so /tmp/base.so
cp scripts/ab/liblimovelo_hip_var.so limo-velo_amd/liblimovelo_hip.so
timeout 900 python -m pytest tests/test_gpu_pass_kernel.py tests/test_gpu_parity.py -x -q > gpurun_out/lib_ab/pytest_var.log 2>&1
grep -E "passed|failed|error" gpurun_out/lib_ab/pytest_var.log | tail -2
for v in base var base var base var; do
  if [ $v = base ]; then cp /tmp/base.so limo-velo_amd/liblimovelo_hip.so; else cp scripts/ab/liblimovelo_hip_var.so limo-velo_amd/liblimovelo_hip.so; fi
  timeout 300 python bench.py --no-cpu-baseline --steps 400 2>/dev/null | tail -1 > gpurun_out/lib_ab/$v.json
  python - <<P
import json
d=json.load(open("gpurun_out/lib_ab/$v.json"))
r=d["roofline"]
ph=r.get("pass_kernel_phases",{}).get("per_launch_us",{})
print("$v", round(d["value"],0), "us/update", round(d["ms_per_step"]*1e3,2), "kernel", round(r["avg_kernel_us"],2), "search", ph.get("search"), "span", ph.get("span"))
P
done
cp /tmp/base.so limo-velo_amd/liblimovelo_hip.so

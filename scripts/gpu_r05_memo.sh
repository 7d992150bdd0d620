#!/bin/bash
# A/B of the voxel memo (pass_kernel: the level-0 probe's answer of a scan point is left for the next launch of the update)
# inside one box: the library of the commit before (scripts/ab/head_r05.so) / this tree with LV_VOXEL_MEMO=0 / =1, alternating on
# the headline bench, after the pass-kernel parity tests with the memo on.
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/memo_ab
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pass_kernel.py tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q > $O/pytest.log 2>&1
grep -E "passed|failed|error" $O/pytest.log | tail -3
cp limo-velo_amd/liblimovelo_hip.so /tmp/new.so
for v in head 0 1 head 0 1 head 0 1; do
  if [ $v = head ]; then cp scripts/ab/head_r05.so limo-velo_amd/liblimovelo_hip.so; m=0; else cp /tmp/new.so limo-velo_amd/liblimovelo_hip.so; m=$v; fi
  LV_VOXEL_MEMO=$m timeout 300 python bench.py --no-cpu-baseline --no-large --no-ext --steps 400 2>/dev/null | tail -1 > $O/memo_$v.json
  python - <<P
import json
d=json.load(open("$O/memo_$v.json"))
r=d["roofline"]
ph=r.get("pass_kernel_phases",{}).get("per_launch_us",{})
print("memo=$v", round(d["value"],0), "it/s; us/update", round(d["ms_per_step"]*1e3,2), "pipelined", round(d.get("value_pipelined",0)), "kernel", r.get("kernel_us_by_launch"), "search", ph.get("search"), "fits", ph.get("fits_partial", ph.get("fits")), "span", ph.get("span"))
P
done
cp /tmp/new.so limo-velo_amd/liblimovelo_hip.so
for v in 0 1; do
  LV_VOXEL_MEMO=$v SIZES=512,8192,65536,131072,262144 timeout 300 python scripts/shard_size_sweep.py 2>&1 | grep scan | sed "s/^/memo=$v /"
done

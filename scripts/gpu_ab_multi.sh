#!/bin/bash
# A/B/C... of several builds of the library on the headline bench inside ONE box (boxes differ by a few per cent):
# scripts/ab/*.so (built in this container: scripts/build_variant.sh) against the tree's build.  Usage on the GPU box:
#   bash scripts/gpu_ab_multi.sh [rounds] [extra bench args]
# Every variant first passes the pass-kernel parity tests (TESTS=0 skips), then the builds are benched in turn, `rounds` times.
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ROUNDS=${1:-2}
shift || true
OUT=gpurun_out/ab
mkdir -p $OUT
VARS="tree"
for f in scripts/ab/*.so; do [ -e "$f" ] && VARS="$VARS $(basename $f .so)"; done
libpath() { if [ $1 = tree ]; then echo $GRAFT_REPO_ROOT/limo-velo_amd/liblimovelo_hip.so; else echo $GRAFT_REPO_ROOT/scripts/ab/$1.so; fi; }
if [ "${TESTS:-1}" != 0 ]; then
  for v in $VARS; do
    LV_LIB_PATH=$(libpath $v) timeout 600 python -m pytest tests/test_gpu_pass_kernel.py tests/test_gpu_parity.py -x -q -k "${TESTK:-not cfg3 and not mailbox}" > $OUT/pytest_$v.log 2>&1
    echo "$v tests: $(grep -E 'passed|failed|error' $OUT/pytest_$v.log | tail -1)"
  done
fi
for r in $(seq 1 $ROUNDS); do
  for v in $VARS; do
    LV_LIB_PATH=$(libpath $v) timeout 300 python bench.py --no-cpu-baseline --no-parity --no-cycle --rotate 0 --steps 400 "$@" 2>$OUT/err_$v.log | tail -1 > $OUT/$v.$r.json
    python - <<P
import json
try:
    d=json.load(open("$OUT/$v.$r.json"))
    r=d["roofline"]; ph=(r.get("pass_kernel_phases") or {}).get("per_launch_us",{})
    print("%-22s" % "$v", "it/s", round(d["value"]), "us/update", round(d["ms_per_step"]*1e3,2), "kernel", round(r["avg_kernel_us"],2),
          "prologue", ph.get("prologue"), "search", ph.get("search"), "fit", ph.get("fit_and_partial"), "span", ph.get("span"))
except Exception as e:
    print("$v", "FAILED", e, open("$OUT/err_$v.log").read()[-400:])
P
  done
done
if [ "${CLOCKS:-0}" != 0 ]; then
  for v in $VARS; do
    echo "==== pass clocks: $v"
    LV_LIB_PATH=$(libpath $v) LV_PASS_CLK=1 timeout 300 python scripts/pass_clocks.py 20 2>&1 | tee $OUT/clocks_$v.txt | grep -v "p90" | head -60
  done
fi

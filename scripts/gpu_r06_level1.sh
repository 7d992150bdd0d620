#!/bin/bash
# round 6: where the headline update's first launch goes on the single-replicated-level map — which level decides the scan's
# points at the perturbed / converged pose (tree vs the round-5 library), then the builds under scripts/ab/ on the headline bench.
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06_level1
mkdir -p $O
for v in tree r05; do
  if [ $v = tree ]; then L=$GRAFT_REPO_ROOT/limo-velo_amd/liblimovelo_hip.so; else L=$GRAFT_REPO_ROOT/scripts/ab/$v.so; fi
  [ -e $L ] || continue
  echo "==== $v"
  LV_LIB_PATH=$L REPS=1 timeout 300 python scripts/r06_cycle_diag.py 2>&1 | tail -4 | tee $O/diag_$v.txt
done
TESTS=0 bash scripts/gpu_ab_multi.sh ${ROUNDS:-2} 2>&1 | tee $O/ab.txt

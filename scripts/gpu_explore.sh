#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for L in ${LANES:-1 2 4 8}; do
  echo "== lanes $L"; timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --lanes $L 2>&1 | python scripts/summ.py
done
for V in ${VOXELS:-0.4 0.7}; do
  echo "== voxel $V"; timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --voxel $V 2>&1 | python scripts/summ.py
done
if [ "${PROF:-1}" = "1" ]; then
cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
fi

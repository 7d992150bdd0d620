#!/bin/bash
# build a kernel variant on the GPU box and bench it:  VAR_FLAGS="..." LV_BLOCKS_PER_CU=8 bash scripts/gpu_variant.sh
set -u
cd $GRAFT_REPO_ROOT/limo-velo_amd/csrc
touch lv_match.hip
make -s EXTRA="${VAR_FLAGS:-}" 2>&1 | grep -E "error" -A5 | head
cd $GRAFT_REPO_ROOT
echo "== variant: ${VAR_FLAGS:-default} blocks/CU=${LV_BLOCKS_PER_CU:-4}"
for L in ${LANES:-8}; do
  timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --lanes $L 2>&1 | python scripts/summ.py
done

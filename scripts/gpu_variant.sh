#!/bin/bash
# build a kernel variant on the GPU box and trace it:  VAR_FLAGS="..." bash scripts/gpu_variant.sh
set -u
cd $GRAFT_REPO_ROOT/limo-velo_amd/csrc
touch lv_match.hip
make -s EXTRA="${VAR_FLAGS:-}" 2>&1 | grep -E "error" -A5 | head
cd $GRAFT_REPO_ROOT
echo "== variant: ${VAR_FLAGS:-default}"
bash scripts/gpu_trace.sh | grep -E "search|fit_red|it/s|match_red"

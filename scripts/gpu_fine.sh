#!/bin/bash
# experiment: fine-grained stamps inside the search kernel's stream+select phase
set -u
cd $GRAFT_REPO_ROOT/limo-velo_amd/csrc
touch lv_match.hip
make -s EXTRA="-DLV_FINE_STAMPS ${VAR_FLAGS:-}" 2>&1 | grep -E "error" -A5 | head
cd $GRAFT_REPO_ROOT
python scripts/phase_clocks.py ${LANES:-8} 2>&1 | grep -v "^solve"

#!/bin/bash
# end of a round: full GPU suite + smoke, then the round's profile artefacts
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/full
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/full/pytest.log 2>&1
grep -E "passed|failed|error" gpurun_out/full/pytest.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
ROUND=${ROUND:-r05} bash scripts/gpu_profile.sh 2>&1 | tail -25

#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
LV_PASS_CLK=1 timeout 300 python scripts/pass_clocks.py 3 0 2>&1 | tail -30

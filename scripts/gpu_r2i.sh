#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2i
LV_PASS_CLK=1 timeout 300 python scripts/pass_clocks.py > gpurun_out/r2i/clocks.txt 2>&1
tail -5 gpurun_out/r2i/clocks.txt

#!/bin/bash
# Host wall clock per stage of the C++ replay's cycle (stream_demo, LV_DEMO_TIMING=1), device-resident hand-overs; extra
# environment settings for an A/B go in front of the call:  LV_OVERLAP_INSERT=0 bash scripts/gpu_stream_timing.sh
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
if [ ! -e /tmp/stream_in.bin ]; then
python - <<'P'
import os, sys
ROOT = os.environ["GRAFT_REPO_ROOT"]
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")): sys.path.insert(0, p)
import lvamd; lvamd.load()
import test_gpu_shim as S
from limo_velo_amd import synth
M, N_AZ, N_REVS = int(os.environ.get("LV_STREAM_MAP", 2000000)), 2048, int(os.environ.get("LV_STREAM_REVS", 20))
stream = synth.make_stream(M, N_REVS, n_az=N_AZ)
pos0, _, vel0, _, q0 = synth.stream_truth(0.2)
x0 = synth.make_state(pos0 + [0.02, -0.015, 0.01], synth.quat_mul(q0, synth.quat_from_rotvec([0.002, -0.001, 0.003])), vel=vel0, grav=(0, 0, synth.STREAM_G))
S._write_stream_input("/tmp/stream_in.bin", 1, 0.01, stream, N_REVS, x0)
P
fi
for r in 1 2 3; do
  LV_DEMO_TIMING=1 $GRAFT_REPO_ROOT/limo-velo_amd/host/stream_demo /tmp/stream_in.bin /tmp/stream_out.bin 2>&1 | grep -E "host wall|updates_per_s" | cut -c1-260
done

#!/bin/bash
# Where the host time of the 100 Hz cycle goes: HIP API + kernel statistics of the C++ replay (stream_demo, device-resident hand-overs)
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/streamtrace
mkdir -p $OUT
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
cd /tmp
limo=$GRAFT_REPO_ROOT/limo-velo_amd/host/stream_demo
$limo /tmp/stream_in.bin /tmp/stream_out.bin | tail -1
timeout 300 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/t -o s -- $limo /tmp/stream_in.bin /tmp/stream_out.bin > $OUT/run.log 2>&1
tail -2 $OUT/run.log
for f in $(find $OUT/t -name "*hip_api_stats.csv" -o -name "*kernel_stats.csv" -o -name "*memory_copy_stats.csv"); do echo "== $f"; head -25 $f | cut -c1-160; cp $f $OUT/; done
find $OUT/t -name "*trace.csv" -delete
du -sh $OUT

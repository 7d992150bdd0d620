#!/bin/bash
# The launch sequence of one 100 Hz cycle of the C++ replay (stream_demo, device-resident): kernel + memory-copy trace in start
# order for a few cycles in the middle of the run: name, duration, gap to the previous end.
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/streamseq
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
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/t -o s -- $limo /tmp/stream_in.bin /tmp/stream_out.bin > $OUT/run.log 2>&1
python - <<'P'
import csv, glob, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/streamseq"
ev = []
for f in glob.glob(out + "/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:70]))
for f in glob.glob(out + "/t/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", r.get("Name", ""))))
ev.sort()
# cycles are delimited by the closing launch of the update
idx = [i for i, e in enumerate(ev) if "pass_kernel<false, true" in e[2]]
print("events", len(ev), "updates", len(idx))
a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 2]
lines = []
for i in range(a + 1, b + 1):
    s, e, n = ev[i]
    lines.append(f"{(s - ev[a][1]) / 1e3:9.1f} us  +{(s - ev[i - 1][1]) / 1e3:7.1f} gap  {(e - s) / 1e3:7.1f} us  {n}")
open(out + "/sequence.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
import numpy as np
cyc = np.array([(ev[idx[i + 1]][1] - ev[idx[i]][1]) / 1e3 for i in range(len(idx) - 1)])
print("cycle us: median", round(float(np.median(cyc)), 1), "mean", round(float(cyc.mean()), 1), "p90", round(float(np.percentile(cyc, 90)), 1), "max", round(float(cyc.max()), 1))
w = int(np.argmax(cyc[20:-5])) + 20
print("== the longest cycle (LiDAR message ingest):")
for i in range(idx[w] + 1, idx[w + 1] + 1):
    s_, e_, n_ = ev[i]
    if (e_ - s_) > 6000 or (s_ - ev[i - 1][1]) > 8000:
        print(f"{(s_ - ev[idx[w]][1]) / 1e3:9.1f} us  +{(s_ - ev[i - 1][1]) / 1e3:7.1f} gap  {(e_ - s_) / 1e3:7.1f} us  {n_}")
import collections
per = collections.Counter()
busy = 0
for i in range(idx[10] + 1, idx[-10] + 1):
    per[ev[i][2]] += 1; busy += ev[i][1] - ev[i][0]
nu = len(idx) - 20
print("per update:", {k: round(v / nu, 2) for k, v in per.most_common()})
print("submissions per update", round(sum(per.values()) / nu, 1), "device busy us per update", round(busy / nu / 1e3, 1), "wall us per update", round((ev[idx[-10]][1] - ev[idx[10]][1]) / nu / 1e3, 1))
P
find $OUT/t -name "*trace.csv" -delete

"""BASELINE configs[4] through the C++ host program: limo-velo_amd/host/stream_demo (main_loop.hpp = the reference's
src/main.cpp:52-128 over the shim's Accumulator / Compensator / Localizator / Mapper).  Same synthetic stream as
scripts/stream_bench.py (64 rings x n_az azimuth steps per 0.1 s sweep, delta = 0.01 s, prior map of LV_STREAM_MAP points,
mapping online), but the host side is compiled code with no per-stage synchronisation: what a ROS-free node would see.
Prints one JSON line: updates/s for the reference's by-value hand-overs and for the device-resident ones, RMSE vs truth."""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import lvamd  # noqa: E402

lvamd.load()
import test_gpu_shim as S  # noqa: E402
from limo_velo_amd import synth  # noqa: E402

M = int(os.environ.get("LV_STREAM_MAP", 10_000_000))
N_AZ = int(os.environ.get("LV_STREAM_AZ", 2048))
N_REVS = int(os.environ.get("LV_STREAM_REVS", 30))

host = os.path.join(ROOT, "limo-velo_amd", "host")
exe = os.path.join(host, "stream_demo")
if not os.path.exists(exe):
    subprocess.check_call(["make", "-s", "-C", host])
t0 = time.time()
stream = synth.make_stream(M, N_REVS, n_az=N_AZ)
gen_s = time.time() - t0
t_init = 0.30 - 0.1
pos0, _, vel0, _, q0 = synth.stream_truth(t_init)
x0 = synth.make_state(pos0 + [0.02, -0.015, 0.01], synth.quat_mul(q0, synth.quat_from_rotvec([0.002, -0.001, 0.003])), vel=vel0,
                      grav=(0, 0, synth.STREAM_G))
out = {"workload": f"{M}-pt prior map, 64 rings x {N_AZ} azimuth steps per 0.1 s sweep, delta = 0.01 s, {N_REVS} sweeps, mapping online, "
                   "1 GPU, C++ host (stream_demo over the shim)", "stream_generation_s": gen_s}
# LV_STREAM_AB="NAME=ENV1=v,ENV2=v;NAME2=..." : extra device-resident runs of the SAME stream with those environment
# settings (A/B of library knobs in one box), e.g. LV_STREAM_AB="separate_launches=LV_SMALL_WINDOW=0,LV_SMALL_INSERT=0"
variants = [] if os.environ.get("LV_STREAM_ONLY_AB") else [("device_resident", 1, {}), ("by_value", 0, {})]
for item in filter(None, os.environ.get("LV_STREAM_AB", "").split(";")):
    name, _, envs = item.partition("=")
    variants.append((name, 1, dict(e.split("=", 1) for e in envs.split(",") if e)))
with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
    for rep in range(int(os.environ.get("LV_STREAM_REPS", 1))):
        for name, on_device, env in variants:
            inp, res = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
            S._write_stream_input(inp, on_device, 0.01, stream, N_REVS, x0)
            cmd = [exe, inp, res]
            if os.environ.get("LV_STREAM_ROCPROF") and name != "by_value":   # kernel trace of this variant (scripts/gpu_r05_trace.sh)
                cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", os.path.join(os.environ["LV_STREAM_ROCPROF"], name), "-o", "t", "--"] + cmd
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=dict(os.environ, **env))
            if r.returncode != 0:
                raise SystemExit(r.stdout + r.stderr)
            if "slow call" in r.stderr:   # (LV_SLOW_CALL_MS diagnostic of the library: pass it on)
                sys.stderr.write("".join(ln + "\n" for ln in r.stderr.splitlines() if "slow call" in ln))
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
            rec = json.loads(line)
            t, x, npts = S._read_stream_output(res)
            truth = np.array([synth.stream_truth(tt)[0] for tt in t])
            rec["rmse_vs_truth_m"] = float(np.sqrt(np.mean(np.sum((x[:, :3] - truth) ** 2, axis=1))))
            rec["final_state_hash"] = float(np.abs(x[-1]).sum())
            out[name if rep == 0 else f"{name}#{rep + 1}"] = rec
print(json.dumps(out))

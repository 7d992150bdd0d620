"""Row f-2 / f-4 timing: PointCloud2 ingest, windowed de-skew + voxel grid of one sweep (GPU only)."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "oracle"); sys.path.insert(0, "tests")
import numpy as np
import torch  # noqa: F401
import lvamd; lvamd.load()
import lvoracle as lo
import cloud_messages as cm
from limo_velo_amd import capi

n = 131_072
raw, f, stamp = cm.make_message("velodyne", n, seed=2, wire=True, stamp_sec=100.35)
fmt = capi.CloudFormat(f["point_step"], f["off_x"], f["off_y"], f["off_z"], f["off_time"], f["time_type"], f["off_intensity"],
                       f["intensity_type"], f["off_range"], f["range_type"], f["relative_time"])
prm = capi.IngestParams(stamp, 0, 0, 0.1, 1, 4.0)
ctx = capi.Context()
reps = 20
ctx.cloud_ingest(raw, n, fmt, prm); ctx.cloud_clear(1e300); ctx.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    kept = ctx.cloud_ingest(raw, n, fmt, prm); ctx.cloud_clear(1e300)
ctx.synchronize()
t_ing = (time.perf_counter() - t0) / reps
kept = ctx.cloud_ingest(raw, n, fmt, prm)
pts = ctx.cloud_fetch(-1e300, 1e300)
t1, t2 = pts["time"][0], pts["time"][-1]
s = lo.motion_state(pos=(1.0, 2.0, 0.5), vel=(4.0, 0.5, 0.0), a=(0.3, -0.2, 9.9), w=(0.02, -0.01, 0.4), time=t1 - 0.004)
states = [s.copy()]
k = 0
while states[-1]["time"][0] < t2 + 0.004:
    k += 1
    s = lo.state_integrate(s, (0.3, -0.2, 9.9), (0.02, -0.01, 0.4), t1 - 0.004 + 0.01 * k)
    states.append(s.copy())
states = np.concatenate(states)
xt2 = states[-2:-1].copy()
ctx.scan_deskew_window(t1, t2, states, xt2, 0.5); ctx.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    nw = ctx.scan_deskew_window(t1, t2, states, xt2, 0.5)
ctx.synchronize()
t_desk = (time.perf_counter() - t0) / reps
print("ingest of a %d-point velodyne message (H2D + decode + time sort + append): %.3f ms, %d kept" % (n, t_ing * 1e3, kept))
print("windowed de-skew + 0.5 m voxel grid + Morton sort of %d buffered points: %.3f ms -> %d scan points" % (nw, t_desk * 1e3, ctx.scan_size()))

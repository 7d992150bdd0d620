"""BASELINE configs[4] at scale on ONE GPU: 64-ring stream (n_az azimuth steps per 0.1 s sweep), delta = 0.01 s
windows, rolling map of up to ~10M points, mapping online.  GPU pipeline only (the oracle comparison is
tests/test_gpu_stream.py at a size the CPU restatement finishes in minutes).  Prints one JSON line with the
per-stage times of the mapping cycle and the end-to-end update rate."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import lvamd  # noqa: E402

lvamd.load()
import lvoracle as oracle  # noqa: E402  (host plumbing only: lv_motion_state records of the surrounding states)
import test_gpu_stream as T  # noqa: E402
from limo_velo_amd import capi, synth  # noqa: E402

M = int(os.environ.get("LV_STREAM_MAP", 10_000_000))
N_AZ = int(os.environ.get("LV_STREAM_AZ", 2048))
N_UPD = int(os.environ.get("LV_STREAM_UPDATES", 300))
# LV_STREAM_LOCKSTEP = k: every k-th update is ALSO computed by the oracle from exactly what the device holds at that moment
# (propagated state and covariance, de-skewed scan, the incrementally maintained 10 M-point map) — "pose vs CPU reference" of
# BASELINE configs[4] at full scale.  The oracle searches the map points within 95 m of the sensor (its kd-tree over them; the scan
# reaches 80 m: every neighbour that can pass the MAX_DIST_PLANE gate lies inside; a full-map kd-tree per check would cost
# minutes of CPU); the checked updates are excluded from the timing.
LOCKSTEP = int(os.environ.get("LV_STREAM_LOCKSTEP", 0))

t0 = time.time()
stream = synth.make_stream(M, N_UPD // 10, n_az=N_AZ)
gen_s = time.time() - t0
Q = np.diag([1e-4] * 3 + [1e-2] * 3 + [1e-5] * 3 + [1e-4] * 3)
stage = {k: 0.0 for k in ("ingest", "predict", "deskew_window", "correct", "map_add_scan", "evict", "clear")}
with capi.Context() as ctx:
    t0 = time.perf_counter()
    ctx.map_build(stream["map_xyz"])
    build_ms = (time.perf_counter() - t0) * 1e3
    pipe = T.HipStream.__new__(T.HipStream)
    pipe.ctx, pipe.capi = ctx, capi
    pos0, _, vel0, _, q0 = synth.stream_truth(0.0)
    x = synth.make_state(pos0 + [0.02, -0.015, 0.01], synth.quat_mul(q0, synth.quat_from_rotvec([0.002, -0.001, 0.003])), vel=vel0,
                         grav=(0, 0, synth.STREAM_G))
    ctx.filter_set(x, synth.default_P0())
    traj, n_scan, n_upd = [], [], 0
    lock = {"checked": 0, "passes_equal": 0, "worst_dx": 0.0, "worst_dP_rel": 0.0, "map_points_near": 0}
    lock_s = 0.0
    msgs = [T.hesai_message(r) for r in stream["revs"]]
    wall0 = time.perf_counter()

    def timed(name, fn, *a):
        t = time.perf_counter()
        r = fn(*a)
        ctx.synchronize()
        stage[name] += time.perf_counter() - t
        return r

    for k in range(1, N_UPD + 1):
        t1, t2 = (k - 1) * T.DELTA, k * T.DELTA
        if (k - 1) % 10 == 0:
            raw, n, fmt, stamp = msgs[(k - 1) // 10]
            timed("ingest", pipe.ingest, raw, n, fmt, stamp)
        x_t1 = ctx.filter_get()[0]
        a1, w1 = synth.stream_imu(t1)
        states = [T._motion_from_filter(oracle, x_t1, t1, a1, w1)]
        a, w = synth.stream_imu(t2)
        timed("predict", pipe.predict, t2 - t1, Q, a, w)
        states.append(oracle.state_integrate(states[-1], a.astype(np.float32), w.astype(np.float32), t2))
        states = np.concatenate(states)
        n_ds = timed("deskew_window", pipe.window, t1, t2, states, states[-1:])
        if n_ds < T.MAX_POINTS2MATCH:
            continue
        if LOCKSTEP and k % LOCKSTEP == LOCKSTEP // 2:
            t_ls = time.perf_counter()
            x_b, P_b = ctx.filter_get()
            scan_b, map_b = ctx.scan_fetch(), ctx.map_fetch()
            passes_g = ctx.correct()
            x_g, P_g = ctx.filter_get()
            sensor = x_b[:3].astype(np.float32)
            near = map_b[np.sum((map_b - sensor) ** 2, axis=1) < np.float32(95.0 * 95.0)]
            tree = oracle.KdTree(near)
            x_o, P_o, passes_o, _, _ = oracle.update(x_b, P_b, near, scan_b, tree=tree, nthreads=int(os.environ.get("LV_ORACLE_THREADS", 16)))
            tree.close()
            lock["checked"] += 1
            lock["passes_equal"] += int(passes_g == passes_o)
            lock["worst_dx"] = max(lock["worst_dx"], float(np.abs(x_g - x_o).max()))
            lock["worst_dP_rel"] = max(lock["worst_dP_rel"], float(np.abs(P_g - P_o).max() / max(1.0, np.abs(P_o).max())))
            lock["map_points_near"] = int(len(near))
            lock_s += time.perf_counter() - t_ls
        else:
            timed("correct", pipe.correct)
        timed("map_add_scan", pipe.map_add)
        if k % 20 == 0:
            c = synth.stream_truth(t2)[0].astype(np.float32)
            timed("evict", pipe.evict, c - np.float32(150.0), c + np.float32(150.0))
        timed("clear", pipe.clear, t2 - T.EMPTY_LIDAR_TIME)
        traj.append(ctx.filter_get()[0])
        n_scan.append(n_ds)
        n_upd += 1
    wall = time.perf_counter() - wall0 - lock_s
    st = ctx.map_stats()
traj = np.array(traj)
truth = np.array([synth.stream_truth((i + 1) * T.DELTA)[0] for i in range(len(traj))])
out = {
    "workload": f"{M}-pt scene map, 64 rings x {N_AZ} azimuth steps per 0.1 s sweep, delta = 0.01 s, {N_UPD} windows, mapping online, 1 GPU",
    "updates": n_upd,
    "updates_per_s_end_to_end": n_upd / wall,
    "scan_points_per_update_mean": float(np.mean(n_scan)),
    "raw_points_per_sweep": int(np.mean([len(r["xyz"]) for r in stream["revs"]])),
    "stage_ms_per_update": {k: v / max(n_upd, 1) * 1e3 for k, v in stage.items()},
    "host_loop_ms_per_update": wall / max(n_upd, 1) * 1e3,
    "map_build_ms": build_ms,
    "map_stats": st,
    "rmse_vs_truth_m": float(np.sqrt(np.mean(np.sum((traj[:, :3] - truth) ** 2, axis=1)))),
    "stream_generation_s": gen_s,
    "lockstep_vs_oracle": lock if LOCKSTEP else None,
}
print(json.dumps(out))

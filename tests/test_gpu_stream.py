"""Row f-3 at the shape of BASELINE configs[4]: a 64-ring LiDAR + 100 Hz IMU stream along a trajectory, localised
at delta = 0.01 s (one update per 10 ms field of view, `src/main.cpp:48,61-71` with Initialization deltas = [0.01]),
mapping online into a rolling map — >= 300 consecutive updates.

Per update, exactly as the reference's loop (src/main.cpp:76-102, 118):
    loc.propagate_to(t2)                         -> lv_predict            (resident filter)
    comp.compensate(t1, t2) + downsample         -> lv_scan_deskew_window (points from the device LiDAR buffer, which
                                                                           lv_cloud_ingest filled from the driver's message)
    loc.correct(ds_compensated, t2)              -> lv_correct
    map.add(Xt2 * Xt2.I_Rt_L() * ds, t2, true)   -> lv_map_add_scan       (incremental, down-sampled, no host round trip)
    accum.clear_lidar(t2 - empty_lidar_time)     -> lv_cloud_clear
plus, every 20 updates, the rolling window (lv_map_evict_box around the vehicle).  The GPU pipeline and the oracle
pipeline (CPU restatement of every stage) consume the same messages; the trajectories must agree to rounding
(pose RMSE < 1e-6 m) and track the ground truth (< 3 cm).  The host plumbing between the stages (which IMU sample,
which states surround the window: Compensator::path / upsample / get_t2) is the same Python for both.
"""
import os

import numpy as np
import pytest

import cloud_messages as cm

pytestmark = pytest.mark.gpu

DELTA = 0.01
EMPTY_LIDAR_TIME = 1.0     # config/kitti.yaml:23
MAX_POINTS2MATCH = 10      # config/params.yaml:47
WINDOW = np.array([55.0, 55.0, 30.0], np.float32)


def _motion_from_filter(oracle, x, t, a, w):
    from limo_velo_amd import synth

    return oracle.motion_state(R=synth.quat_to_rot(x[3:7]), pos=x[0:3], vel=x[14:17], bw=x[17:20], ba=x[20:23], a=a, w=w,
                               time=t, RLI=synth.quat_to_rot(x[7:11]), tLI=x[11:14], g=(0, 0, -9.807))


def hesai_message(rev):
    """One sweep as a sensor_msgs/PointCloud2 payload in the hesai_ros::Point layout (absolute f64 stamps)."""
    dt = cm._fields("hesai", False)
    rec = np.zeros(len(rev["xyz"]), dt)
    rec["x"], rec["y"], rec["z"] = rev["xyz"].T
    rec["timestamp"] = rev["t"]
    rec["intensity"] = 100
    fmt = dict(point_step=dt.itemsize, off_x=0, off_y=4, off_z=8, off_time=dt.fields["timestamp"][1], time_type=1,
               off_intensity=dt.fields["intensity"][1], intensity_type=2, off_range=0, range_type=0, relative_time=0)
    return rec.tobytes(), len(rec), fmt, int(round(rev["stamp"] * 1e6))


def _fmt(mod, f):
    return mod.CloudFormat(f["point_step"], f["off_x"], f["off_y"], f["off_z"], f["off_time"], f["time_type"], f["off_intensity"],
                           f["intensity_type"], f["off_range"], f["range_type"], f["relative_time"])


class HipStream:
    def __init__(self, ctx, capi, map_xyz):
        self.ctx, self.capi = ctx, capi
        ctx.map_build(map_xyz)

    def filter_set(self, x, P):
        self.ctx.filter_set(x, P)

    def predict(self, dt, Q, a, w):
        self.ctx.predict(dt, Q, a, w)

    def ingest(self, raw, n, fmt, stamp):
        return self.ctx.cloud_ingest(raw, n, _fmt(self.capi, fmt), self.capi.IngestParams(stamp, 0, 0, 0.1, 4, 4.0))

    def window(self, t1, t2, states, xt2):
        self.ctx.scan_deskew_window(t1, t2, states, xt2, downsample_prec=0.5)
        return self.ctx.scan_size()

    def correct(self):
        return self.ctx.correct()

    def state(self):
        return self.ctx.filter_get()[0]

    def map_add(self):
        self.ctx.map_add_scan(downsample=True)

    def evict(self, lo, hi):
        return self.ctx.map_evict_box(lo, hi, keep_inside=True)

    def clear(self, t):
        self.ctx.cloud_clear(t)

    def map_size(self):
        return self.ctx.map_size()


class LockstepHipStream(HipStream):
    """Every `every`-th update is ALSO computed by the oracle from exactly what the device holds at that moment — the
    propagated state and covariance, the de-skewed scan, the incrementally maintained map — and must agree to 1e-8:
    per-update parity deep inside a long mapping run (hundreds of in-place inserts and evictions behind it)."""

    def __init__(self, ctx, capi, map_xyz, oracle, every, nthreads):
        super().__init__(ctx, capi, map_xyz)
        self.o, self.every, self.nt, self.k, self.checked, self.worst = oracle, every, nthreads, 0, 0, 0.0

    def correct(self):
        self.k += 1
        if self.k % self.every != self.every // 2:
            return self.ctx.correct()
        x, P = self.ctx.filter_get()
        scan, mp = self.ctx.scan_fetch(), self.ctx.map_fetch()
        passes = self.ctx.correct()
        xg, Pg = self.ctx.filter_get()
        xo, Po, po, _, _ = self.o.update(x, P, mp, scan, nthreads=self.nt)
        assert passes == po, (self.k, passes, po)
        d = float(np.abs(xg - xo).max())
        assert d < 1e-8 and np.abs(Pg - Po).max() < 1e-8 * max(1.0, np.abs(Po).max()), (self.k, d)
        self.worst = max(self.worst, d)
        self.checked += 1
        return passes


class OracleStream:
    def __init__(self, oracle, map_xyz, nthreads):
        self.o, self.map, self.nt = oracle, map_xyz.copy(), nthreads
        self.buf = np.zeros(0, oracle.POINT_DTYPE)

    def filter_set(self, x, P):
        self.x, self.P = x.copy(), P.copy()

    def predict(self, dt, Q, a, w):
        self.x, self.P = self.o.predict(self.x, self.P, dt, Q, a, w)

    def ingest(self, raw, n, fmt, stamp):
        pts = self.o.cloud_ingest(raw, n, _fmt(self.o, fmt), self.o.IngestParams(stamp, 0, 0, 0.1, 4, 4.0))
        self.buf = np.concatenate([self.buf, pts])     # Accumulator::push of every point (time order across sweeps)
        return len(pts)

    def window(self, t1, t2, states, xt2):
        sel = self.buf[(self.buf["time"] >= t1) & (self.buf["time"] <= t2)]   # Accumulator::get_points: closed interval
        if len(sel) == 0:
            self.scan = np.zeros((0, 3), np.float32)
            return 0
        xyz = np.stack([sel["x"], sel["y"], sel["z"]], axis=1)
        self.scan = self.o.voxelgrid(self.o.deskew(xyz, sel["time"], states, xt2), 0.5)
        return len(self.scan)

    def correct(self):
        self.x, self.P, passes, _, _ = self.o.update(self.x, self.P, self.map, self.scan, nthreads=self.nt)
        return passes

    def state(self):
        return self.x.copy()

    def map_add(self):
        self.map = self.o.map_add(self.map, self.o.transform_scan(self.x, self.scan), downsample=True)

    def evict(self, lo, hi):
        keep = np.all((self.map >= lo) & (self.map <= hi), axis=1)
        self.map = self.map[keep]
        return int((~keep).sum())

    def clear(self, t):
        self.buf = self.buf[self.buf["time"] > t]

    def map_size(self):
        return len(self.map)


def run_stream(pipe, oracle, stream, n_updates, log=None):
    from limo_velo_amd import synth

    Q = np.diag([1e-4] * 3 + [1e-2] * 3 + [1e-5] * 3 + [1e-4] * 3)   # Localizator.cpp:159-173 with config/params.yaml:39-42
    pos0, _, vel0, _, q0 = synth.stream_truth(0.0)
    x = synth.make_state(pos0 + [0.02, -0.015, 0.01], synth.quat_mul(q0, synth.quat_from_rotvec([0.002, -0.001, 0.003])), vel=vel0,
                         grav=(0, 0, synth.STREAM_G))
    pipe.filter_set(x, synth.default_P0())
    traj, times, sizes, skipped = [], [], [], 0
    per_rev = int(round(0.1 / DELTA))
    for k in range(1, n_updates + 1):
        t1, t2 = (k - 1) * DELTA, k * DELTA
        if (k - 1) % per_rev == 0:                      # the driver's message of the sweep these windows lie in
            raw, n, fmt, stamp = hesai_message(stream["revs"][(k - 1) // per_rev])
            pipe.ingest(raw, n, fmt, stamp)
        x_t1 = pipe.state()
        a1, w1 = synth.stream_imu(t1)
        states = [_motion_from_filter(oracle, x_t1, t1, a1, w1)]          # Compensator::path: the state before t1 ...
        a, w = synth.stream_imu(t2)                                        # ... and the IMU sample up to t2 (100 Hz)
        pipe.predict(t2 - t1, Q, a, w)                                     # Localizator::propagate_to(t2)
        states.append(oracle.state_integrate(states[-1], a.astype(np.float32), w.astype(np.float32), t2))
        states = np.concatenate(states)
        n_ds = pipe.window(t1, t2, states, states[-1:])                    # compensate(t1, t2) + downsample
        if n_ds < MAX_POINTS2MATCH:                                        # main.cpp:81
            skipped += 1
            continue
        passes = pipe.correct()                                            # loc.correct(ds_compensated, t2)
        xk = pipe.state()
        traj.append(xk)
        times.append(t2)
        pipe.map_add()                                                     # map.add(global_ds_compensated, t2, true)
        if k % 20 == 0:                                                    # rolling window around the vehicle
            c = synth.stream_truth(t2)[0].astype(np.float32)
            pipe.evict(c - WINDOW, c + WINDOW)
        pipe.clear(t2 - EMPTY_LIDAR_TIME)                                  # accum.clear_lidar (main.cpp:118)
        sizes.append((n_ds, passes, pipe.map_size()))
        if log is not None and k % 50 == 0:
            log(k, n_ds, passes, pipe.map_size())
    return np.array(traj), np.array(times), sizes, skipped


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stream_oracle.npz")


def test_cfg4_stream_300_updates(lv, oracle):
    """GPU pipeline vs the oracle pipeline over 300 updates.  The oracle's trajectory comes from the committed fixture
    tests/golden/stream_oracle.npz (made by tests/golden/make_golden_stream.py: minutes of CPU time), or is recomputed
    live with LV_STREAM_LIVE_ORACLE=1.  The two pipelines agree to rounding, not bitwise: a 1e-10 difference in the
    state flips the last bit of a few f32 world points, and with it now and then a box-rule decision of the map
    insert — so scan sizes are compared exactly, map sizes to 0.2 %, passes as explained below."""
    from limo_velo_amd import capi, synth

    n_updates = 300
    stream = synth.make_stream(1_048_576, n_updates // 10, n_az=512, map_radius=62.0)
    assert 150_000 < len(stream["map_xyz"]) < 700_000
    nthreads = max(8, min(64, (os.cpu_count() or 8)))
    with capi.Context() as ctx:
        pipe = LockstepHipStream(ctx, capi, stream["map_xyz"], oracle, 10, nthreads)
        tg, tt, sg, skg = run_stream(pipe, oracle, stream, n_updates)
        st = ctx.map_stats()
    assert pipe.checked == n_updates // 10
    if os.environ.get("LV_STREAM_LIVE_ORACLE") == "1" or not os.path.exists(GOLDEN):
        to, _, so, sko = run_stream(OracleStream(oracle, stream["map_xyz"], nthreads), oracle, stream, n_updates)
    else:
        g = np.load(GOLDEN)
        assert int(g["n_map0"]) == len(stream["map_xyz"]) and float(g["map_checksum"]) == float(stream["map_xyz"].astype(np.float64).sum())
        to, so, sko = g["traj"], [tuple(int(v) for v in r) for r in g["sizes"]], int(g["skipped"])
    assert skg == sko == 0 and len(tg) == len(to) == n_updates
    # scan sizes (voxel-grid leaves of the de-skewed window): the very same early on; free-running, a state that differs at
    # the 1e-5 m level (below) now and then moves one point across a leaf boundary
    nsg, nso = np.array([s[0] for s in sg]), np.array([s[0] for s in so])
    assert np.array_equal(nsg[:50], nso[:50]), np.flatnonzero(nsg != nso)
    assert np.abs(nsg - nso).max() <= 2 and np.count_nonzero(nsg != nso) <= n_updates // 20, (np.flatnonzero(nsg != nso), (nsg - nso)[nsg != nso])
    # Free-running, the two pipelines cannot stay bitwise together: map points are f32 (one ulp = 4e-6 m at 60 m), so a
    # 1e-10 difference in the state rounds a few inserted points differently, the next scans are matched against maps
    # that differ by micrometres, and the difference grows to the 1e-5 m level over a hundred mapping updates (the
    # per-update parity is what LockstepHipStream pins, to 1e-8).  On top of that the number of passes is a
    # discontinuous function of the state (dx vs LIMITS, src/main.cpp:145): an update that sits on the threshold gets one
    # pass more in one pipeline, i.e. a state that differs by less than LIMITS (1e-3), until the filter has pulled
    # both back together.  Hence: rounding-level agreement at the start, a bounded deviation throughout.
    flips = [i for i, (a, b) in enumerate(zip(sg, so)) if a[1] != b[1]]
    first = flips[0] if flips else n_updates
    dev = np.linalg.norm(tg[:, :3] - to[:, :3], axis=1)
    report = dict(first_pass_flip=first, n_flips=len(flips), max_dev_first_20=float(dev[:20].max()), max_dev_before_flip=float(dev[:first].max()),
                  max_dev=float(dev.max()), rmse_vs_oracle=float(np.sqrt(np.mean(dev ** 2))), map_sizes_end=(sg[-1][2], so[-1][2]),
                  lockstep_checked=pipe.checked, lockstep_worst=pipe.worst, stats=st)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        import json

        json.dump(report, open(os.path.join(out_dir, "stream_test_report.json"), "w"), indent=1)
    # (How fast the free-running difference grows depends on the realisation — on the summation order of the workgroup
    # partials, for instance, which differs between pass_kernel geometries: 2e-5 .. 1e-4 m before the first pass-count flip
    # have been seen; a realisation without any flip in 300 updates drifted to 3.03e-4 m by update 297.  The arithmetic itself
    # is pinned per update by LockstepHipStream, not here.)
    assert dev[:20].max() < 1e-7, report
    assert first >= 50 and dev[:min(first, 200)].max() < 3e-4, report
    assert len(flips) <= n_updates // 20, report
    assert dev.max() < 1e-3 and report["rmse_vs_oracle"] < 5e-4, report
    assert max(abs(a[2] - b[2]) / b[2] for a, b in zip(sg, so)) < 2e-3
    assert sg[:50] == so[:50]                                    # early on: the very same passes and map sizes
    assert st["incremental_adds"] >= n_updates - 2 and st["relinearisations"] <= 3, st   # the map was maintained in place
    truth = np.array([synth.stream_truth(t)[0] for t in tt])
    rmse_vs_truth = float(np.sqrt(np.mean(np.sum((tg[:, :3] - truth) ** 2, axis=1))))
    assert rmse_vs_truth < 0.03, rmse_vs_truth
    assert min(s[0] for s in sg) >= MAX_POINTS2MATCH and np.mean([s[1] for s in sg]) <= 4

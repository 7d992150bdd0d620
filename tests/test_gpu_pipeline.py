"""Row f-3 (driver): the per-scan loop of src/main.cpp:76-102 replayed end to end on a synthetic trajectory —
IMU propagate -> de-skew + voxel grid -> iterated correct -> map insert with down-sampling — once through the
C-ABI (everything on the GPU, filter state and map device-resident) and once through the oracle (CPU).  Both
consume identical inputs at every stage; the trajectories must agree to rounding and track the ground truth."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = 9.809


def _truth(t):
    """circular arc: speed 5 m/s, yaw rate 0.3 rad/s, level, 1.5 m above the ground"""
    v, om, yaw0 = 5.0, 0.3, 0.5
    yaw = yaw0 + om * t
    pos = np.array([3.0 + v / om * (math.sin(yaw) - math.sin(yaw0)), -2.0 - v / om * (math.cos(yaw) - math.cos(yaw0)), 1.5])
    R = np.array([[math.cos(yaw), -math.sin(yaw), 0], [math.sin(yaw), math.cos(yaw), 0], [0, 0, 1.0]])
    vel = np.array([v * math.cos(yaw), v * math.sin(yaw), 0.0])
    a_world = np.array([-v * om * math.sin(yaw), v * om * math.cos(yaw), 0.0])
    q = np.array([0, 0, math.sin(yaw / 2), math.cos(yaw / 2)])
    return pos, R, vel, a_world, q


def _imu(t):
    # the reference's convention (State.cpp:104 `R a - g`, Localizator.cpp:138 grav = -initial_gravity): a stationary
    # IMU reads (0, 0, -g)
    _, R, _, a_world, _ = _truth(t)
    return R.T @ (a_world - np.array([0, 0, G])), np.array([0.0, 0.0, 0.3])


def _motion_from_filter(oracle, x, t, a, w):
    from limo_velo_amd import synth

    return oracle.motion_state(R=synth.quat_to_rot(x[3:7]), pos=x[0:3], vel=x[14:17], bw=x[17:20], ba=x[20:23], a=a, w=w,
                               time=t, RLI=synth.quat_to_rot(x[7:11]), tLI=x[11:14], g=(0, 0, -9.807))


class OraclePipe:
    def __init__(self, oracle, map_xyz):
        self.o, self.map = oracle, map_xyz

    def filter_set(self, x, P):
        self.x, self.P = x.copy(), P.copy()

    def predict(self, dt, Q, a, w):
        self.x, self.P = self.o.predict(self.x, self.P, dt, Q, a, w)

    def deskew(self, xyz, times, states, xt2):
        self.scan = self.o.voxelgrid(self.o.deskew(xyz, times, states, xt2), 0.5)
        return self.scan

    def correct(self):
        self.x, self.P, passes, _, _ = self.o.update(self.x, self.P, self.map, self.scan)
        return passes

    def state(self):
        return self.x.copy()

    def map_add(self, pts):
        self.map = self.o.map_add(self.map, pts, downsample=True)

    def map_size(self):
        return len(self.map)


class HipPipe:
    def __init__(self, ctx, map_xyz):
        self.ctx = ctx
        ctx.map_build(map_xyz)

    def filter_set(self, x, P):
        self.ctx.filter_set(x, P)

    def predict(self, dt, Q, a, w):
        self.ctx.predict(dt, Q, a, w)

    def deskew(self, xyz, times, states, xt2):
        self.ctx.scan_deskew(xyz, times, states, xt2, downsample_prec=0.5)
        return self.ctx.scan_fetch()

    def correct(self):
        return self.ctx.correct()

    def state(self):
        return self.ctx.filter_get()[0]

    def map_add(self, pts):
        # the device holds both the scan (de-skew output) and the posterior state: no host round trip (src/main.cpp:92,102)
        self.ctx.map_add_scan(downsample=True)

    def map_size(self):
        return self.ctx.map_size()


def _run(pipe, oracle, sc, n_scans=6, raw_per_scan=12000):
    from limo_velo_amd import synth

    rng = np.random.default_rng(21)
    Q = np.diag([1e-4] * 3 + [1e-2] * 3 + [1e-5] * 3 + [1e-4] * 3)
    pos0, _, vel0, _, q0 = _truth(0.0)
    x = synth.make_state(pos0 + [0.03, -0.02, 0.01], synth.quat_mul(q0, synth.quat_from_rotvec([0.002, -0.001, 0.004])),
                         vel=vel0, grav=(0, 0, G))
    pipe.filter_set(x, synth.default_P0())
    traj, sizes = [], []
    for k in range(n_scans):
        t1, t2 = 0.1 * k, 0.1 * (k + 1)
        x_t1 = pipe.state()
        # IMU samples at 100 Hz inside (t1, t2]
        imu_t = [t1 + 0.01 * (j + 1) for j in range(10)]
        a1, w1 = _imu(t1)
        states = [_motion_from_filter(oracle, x_t1, t1, a1, w1)]   # Compensator::path / upsample (host plumbing)
        last = t1
        for tj in imu_t:
            a, w = _imu(tj)
            pipe.predict(tj - last, Q, a, w)                         # Localizator::propagate_to
            states.append(oracle.state_integrate(states[-1], a.astype(np.float32), w.astype(np.float32), tj))
            last = tj
        states = np.concatenate(states)
        xt2 = states[-1:]                                            # Compensator::get_t2 (t2 is a state stamp here)
        # raw scan: surface points seen from the moving sensor, each expressed in the LiDAR frame at its own time
        near = sc["map_xyz"][rng.integers(0, len(sc["map_xyz"]), raw_per_scan)].astype(np.float64)
        near += rng.uniform(-0.01, 0.01, near.shape)
        times = np.sort(rng.uniform(t1, t2, raw_per_scan))
        raw = np.empty((raw_per_scan, 3), np.float32)
        for i, (pw, tp) in enumerate(zip(near, times)):
            p, R, _, _, _ = _truth(tp)
            raw[i] = (R.T @ (pw - p)).astype(np.float32)
        keep = (np.linalg.norm(raw, axis=1) > 4.0) & (np.linalg.norm(raw, axis=1) < 60.0)
        ds = pipe.deskew(raw[keep], times[keep], states, xt2)
        passes = pipe.correct()                                      # Localizator::correct
        xk = pipe.state()
        traj.append(xk)
        sizes.append((len(ds), passes))
        pipe.map_add(oracle.transform_scan(xk, ds))                  # map.add(Xt2 * Xt2.I_Rt_L() * ds, t2, true)
    return np.array(traj), sizes, pipe.map_size()


def test_replay_gpu_equals_oracle_and_tracks_truth(lv, oracle):
    from limo_velo_amd import capi, synth

    sc = synth.make_scene(200_000, 16)
    with capi.Context() as ctx:
        tg, sg, mg = _run(HipPipe(ctx, sc["map_xyz"]), oracle, sc)
    to, so, mo = _run(OraclePipe(oracle, sc["map_xyz"]), oracle, sc)
    assert sg == so and mg == mo and mg < 200_000 + sum(s[0] for s in sg)
    rmse_pos = float(np.sqrt(np.mean(np.sum((tg[:, :3] - to[:, :3]) ** 2, axis=1))))
    assert rmse_pos < 1e-6, rmse_pos
    assert np.abs(tg - to).max() < 1e-5
    truth = np.array([_truth(0.1 * (k + 1))[0] for k in range(len(tg))])
    assert np.sqrt(np.mean(np.sum((tg[:, :3] - truth) ** 2, axis=1))) < 0.03

"""Deterministic synthetic scene for the LIMO-Velo KF-update hot path.

The reference ships no data, so tests/ and bench.py use this "street canyon" of planar
patches: ground + 4 walls + random axis-aligned boxes, sampled area-uniformly so that
`R3Math::is_plane` (reference src/Utils/Utils.cpp:59-66) accepts most matches.  The scan is drawn
from the same surfaces with a fresh seed, restricted to 4..80 m range (mirrors `min_dist`,
config/params.yaml:34) and expressed in the LiDAR frame through a ground-truth pose; the filter
starts from a perturbed pose so several IKFoM passes are needed (LIMITS 1e-3, main.cpp:145).

Random numbers: numpy PCG64 *raw* 64-bit outputs only, uniform = (u >> 11) * 2^-53 — bit-portable
across platforms and numpy versions (no distribution code involved).  Noise is uniform +-sigma*sqrt(3).
"""
from __future__ import annotations

import math

import numpy as np

SEED_MAP = 0x4C494D4F  # "LIMO"
SEED_SCAN = 0x56454C4F  # "VELO"

# field order of state_ikfom as 26 doubles (quaternions x,y,z,w) — see include/limovelo_hip.h lv_state
STATE_LEN = 26


class _Rng:
    def __init__(self, seed: int):
        self._bg = np.random.PCG64(seed)

    def uniform(self, n: int) -> np.ndarray:
        raw = self._bg.random_raw(n).astype(np.uint64)
        return (raw >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def quat_from_rpy(roll: float, pitch: float, yaw: float) -> np.ndarray:
    """ZYX euler -> quaternion (x, y, z, w)."""
    cr, sr = math.cos(roll / 2), math.sin(roll / 2)
    cp, sp = math.cos(pitch / 2), math.sin(pitch / 2)
    cy, sy = math.cos(yaw / 2), math.sin(yaw / 2)
    return np.array([
        sr * cp * cy - cr * sp * sy,
        cr * sp * cy + sr * cp * sy,
        cr * cp * sy - sr * sp * cy,
        cr * cp * cy + sr * sp * sy,
    ])


def quat_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by + ay * bw + az * bx - ax * bz,
        aw * bz + az * bw + ax * by - ay * bx,
        aw * bw - ax * bx - ay * by - az * bz,
    ])


def quat_from_rotvec(v) -> np.ndarray:
    v = np.asarray(v, dtype=np.float64)
    th = float(np.linalg.norm(v))
    if th < 1e-12:
        return np.array([0.5 * v[0], 0.5 * v[1], 0.5 * v[2], 1.0])
    s = math.sin(th / 2) / th
    return np.array([v[0] * s, v[1] * s, v[2] * s, math.cos(th / 2)])


def quat_to_rot(q) -> np.ndarray:
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ])


def make_state(pos, rot, offR=(0, 0, 0, 1), offT=(0, 0, 0), vel=(0, 0, 0), bg=(0, 0, 0), ba=(0, 0, 0),
               grav=(0, 0, -9.809)) -> np.ndarray:
    """Pack a state_ikfom as 26 float64: pos3 rot4 offset_R_L_I4 offset_T_L_I3 vel3 bg3 ba3 grav3."""
    s = np.concatenate([np.asarray(x, dtype=np.float64).ravel() for x in (pos, rot, offR, offT, vel, bg, ba, grav)])
    assert s.shape == (STATE_LEN,)
    return s


def default_P0() -> np.ndarray:
    """Initial covariance of Localizator::init_IKFoM_state (reference Localizator.cpp:144-150)."""
    P = np.eye(23)
    for i in (6, 7, 8, 9, 10, 11):
        P[i, i] = 0.00001
    for i in (15, 16, 17):
        P[i, i] = 0.0001
    for i in (18, 19, 20):
        P[i, i] = 0.001
    for i in (21, 22):
        P[i, i] = 0.00001
    return P


def _surfaces(rng: _Rng, m_points: int, density: float = 25.0, wall_h: float = 8.0, max_boxes: int | None = 64):
    """Rectangles (origin, edge u, edge v) of the scene; total area ~ m_points / density.  max_boxes = None: one box per
    ~15 600 map points at any size (a 10M-point scene then keeps the box density of the 1M one — with the default cap
    of 64 a large scene is mostly bare ground, x / y / yaw hardly observable)."""
    n_boxes = int(max(4, m_points // 15625))
    if max_boxes is not None:
        n_boxes = min(n_boxes, max_boxes)
    u = rng.uniform(n_boxes * 5)
    side_x = 2.0 + 4.0 * u[0::5]
    side_y = 2.0 + 4.0 * u[1::5]
    height = 2.0 + 4.0 * u[2::5]
    box_area = float(np.sum(side_x * side_y + 2 * height * (side_x + side_y)))
    target = m_points / density
    # (2L)^2 + 4 * (2L) * wall_h + box_area = target
    rem = max(target - box_area, 0.25 * target)
    L = (-8 * wall_h + math.sqrt((8 * wall_h) ** 2 + 16 * rem)) / 8.0
    rects = []
    rects.append(((-L, -L, 0.0), (2 * L, 0, 0), (0, 2 * L, 0)))  # ground
    rects.append(((-L, -L, 0.0), (2 * L, 0, 0), (0, 0, wall_h)))  # y = -L
    rects.append(((-L, L, 0.0), (2 * L, 0, 0), (0, 0, wall_h)))  # y = +L
    rects.append(((-L, -L, 0.0), (0, 2 * L, 0), (0, 0, wall_h)))  # x = -L
    rects.append(((L, -L, 0.0), (0, 2 * L, 0), (0, 0, wall_h)))  # x = +L
    cx = (2 * u[3::5] - 1) * (L - 8.0)
    cy = (2 * u[4::5] - 1) * (L - 8.0)
    for b in range(n_boxes):
        x0, y0 = cx[b] - side_x[b] / 2, cy[b] - side_y[b] / 2
        sx, sy, h = side_x[b], side_y[b], height[b]
        rects.append(((x0, y0, h), (sx, 0, 0), (0, sy, 0)))  # top
        rects.append(((x0, y0, 0), (sx, 0, 0), (0, 0, h)))
        rects.append(((x0, y0 + sy, 0), (sx, 0, 0), (0, 0, h)))
        rects.append(((x0, y0, 0), (0, sy, 0), (0, 0, h)))
        rects.append(((x0 + sx, y0, 0), (0, sy, 0), (0, 0, h)))
    o = np.array([r[0] for r in rects], dtype=np.float64)
    eu = np.array([r[1] for r in rects], dtype=np.float64)
    ev = np.array([r[2] for r in rects], dtype=np.float64)
    area = np.linalg.norm(np.cross(eu, ev), axis=1)
    return o, eu, ev, area, L


def _sample(rng: _Rng, surf, n: int, sigma: float) -> np.ndarray:
    o, eu, ev, area, _ = surf
    cdf = np.cumsum(area) / np.sum(area)
    u = rng.uniform(n * 6).reshape(n, 6)
    which = np.minimum(np.searchsorted(cdf, u[:, 0], side="right"), len(area) - 1)
    p = o[which] + u[:, 1:2] * eu[which] + u[:, 2:3] * ev[which]
    p += (2.0 * u[:, 3:6] - 1.0) * (sigma * math.sqrt(3.0))
    return p


XALOC_EXTRINSICS = dict(  # config/xaloc.yaml:16-21 (LiDAR -> IMU)
    t=(-0.17, 0.0, -0.04),
)


def make_scene(m_points: int, n_points: int, *, sigma: float = 0.01, extrinsics: str = "identity",
               seed_map: int = SEED_MAP, seed_scan: int = SEED_SCAN, rmin: float = 4.0, rmax: float = 80.0):
    """Returns dict(map_xyz [M,3] f32 world frame, scan_xyz [N,3] f32 LiDAR frame, x_true, x_init
    (26 f64 each), P0 [23,23] f64, L)."""
    rng_m = _Rng(seed_map)
    surf = _surfaces(rng_m, m_points)
    map_xyz = _sample(rng_m, surf, m_points, sigma).astype(np.float32)

    pos_true = np.array([3.0, -2.0, 1.5])
    q_true = quat_from_rpy(math.radians(2.0), math.radians(-1.0), math.radians(30.0))
    if extrinsics == "identity":
        offR, offT = np.array([0.0, 0.0, 0.0, 1.0]), np.zeros(3)
    elif extrinsics == "xaloc":
        offR = quat_from_rpy(0.0, 0.0, math.radians(1.5))
        offT = np.array(XALOC_EXTRINSICS["t"])
    else:
        raise ValueError(extrinsics)
    R = quat_to_rot(q_true)
    RLI = quat_to_rot(offR)
    sensor = R @ offT + pos_true

    rng_s = _Rng(seed_scan)
    chunks, have = [], 0
    while have < n_points:
        c = _sample(rng_s, surf, max(4096, 2 * (n_points - have)), sigma)
        r = np.linalg.norm(c - sensor, axis=1)
        c = c[(r >= rmin) & (r <= rmax)]
        chunks.append(c)
        have += len(c)
    pw = np.concatenate(chunks)[:n_points]
    # p_w = R (RLI p_l + tLI) + pos   =>   p_l = RLI^T (R^T (p_w - pos) - tLI)
    pl = ((pw - pos_true) @ R - offT) @ RLI
    scan_xyz = pl.astype(np.float32)

    x_true = make_state(pos_true, q_true, offR, offT)
    dq = quat_from_rotvec(np.radians([0.5, -0.4, 0.8]))
    x_init = make_state(pos_true + np.array([0.10, -0.07, 0.05]), quat_mul(q_true, dq), offR, offT)
    return dict(map_xyz=map_xyz, scan_xyz=scan_xyz, x_true=x_true, x_init=x_init, P0=default_P0(), L=surf[4])


def make_ring_scene(m_points: int, n_rings: int, n_az: int, *, fov_deg=(-15.0, 15.0), sigma: float = 0.01,
                    extrinsics: str = "identity", seed_map: int = SEED_MAP, seed_scan: int = SEED_SCAN, rmin: float = 4.0,
                    rmax: float = 80.0):
    """Same map and poses as make_scene, but the scan is what a spinning LiDAR sees: n_rings
    elevation rings between fov_deg, n_az azimuth steps each, every ray cast from the true sensor pose onto the
    scene's rectangles (nearest hit within [rmin, rmax]); rays without a hit are dropped, so the scan has at most
    n_rings * n_az points.  Returns the same dict as make_scene."""
    rng_m = _Rng(seed_map)
    surf = _surfaces(rng_m, m_points)
    map_xyz = _sample(rng_m, surf, m_points, sigma).astype(np.float32)
    o, eu, ev, _, L = surf

    pos_true = np.array([3.0, -2.0, 1.5])
    q_true = quat_from_rpy(math.radians(2.0), math.radians(-1.0), math.radians(30.0))
    if extrinsics == "identity":
        offR, offT = np.array([0.0, 0.0, 0.0, 1.0]), np.zeros(3)
    elif extrinsics == "xaloc":
        offR = quat_from_rpy(0.0, 0.0, math.radians(1.5))
        offT = np.array(XALOC_EXTRINSICS["t"])
    else:
        raise ValueError(extrinsics)
    R = quat_to_rot(q_true)
    RLI = quat_to_rot(offR)
    sensor = R @ offT + pos_true

    el = np.radians(np.linspace(fov_deg[0], fov_deg[1], n_rings))
    az = np.linspace(0.0, 2.0 * math.pi, n_az, endpoint=False)
    ce, se = np.cos(el)[:, None], np.sin(el)[:, None]
    dl = np.stack([ce * np.cos(az)[None, :], ce * np.sin(az)[None, :], se * np.ones_like(az)[None, :]], axis=-1).reshape(-1, 3)
    dw = dl @ (R @ RLI).T   # ray directions in the world

    nrm = np.cross(eu, ev)
    nrm /= np.linalg.norm(nrm, axis=1)[:, None]
    uu, vv = np.sum(eu * eu, axis=1), np.sum(ev * ev, axis=1)
    best = np.full(len(dw), np.inf)
    for c0 in range(0, len(dw), 32768):   # rays x rectangles in chunks
        d = dw[c0:c0 + 32768]
        den = d @ nrm.T                                         # [rays, rects]
        num = np.sum((o - sensor) * nrm, axis=1)[None, :]
        with np.errstate(divide="ignore", invalid="ignore"):
            t = num / den
        t[~np.isfinite(t) | (t < rmin) | (t > rmax)] = np.inf
        with np.errstate(invalid="ignore"):
            hit = sensor[None, None, :] + t[:, :, None] * d[:, None, :]
            rel = hit - o[None, :, :]
            a = np.sum(rel * eu[None, :, :], axis=2) / uu[None, :]
            b = np.sum(rel * ev[None, :, :], axis=2) / vv[None, :]
        t[(a < 0) | (a > 1) | (b < 0) | (b > 1) | ~np.isfinite(a) | ~np.isfinite(b)] = np.inf
        best[c0:c0 + 32768] = np.min(t, axis=1)
    ok = np.isfinite(best)
    pw = sensor[None, :] + best[ok, None] * dw[ok]
    rng_s = _Rng(seed_scan)
    pw = pw + (2.0 * rng_s.uniform(len(pw) * 3).reshape(-1, 3) - 1.0) * (sigma * math.sqrt(3.0))
    pl = ((pw - pos_true) @ R - offT) @ RLI
    scan_xyz = pl.astype(np.float32)

    x_true = make_state(pos_true, q_true, offR, offT)
    dq = quat_from_rotvec(np.radians([0.5, -0.4, 0.8]))
    x_init = make_state(pos_true + np.array([0.10, -0.07, 0.05]), quat_mul(q_true, dq), offR, offT)
    return dict(map_xyz=map_xyz, scan_xyz=scan_xyz, x_true=x_true, x_init=x_init, P0=default_P0(), L=L)


def make_extra_scan(m_points: int, n_points: int, k: int, *, sigma: float = 0.01, seed_map: int = SEED_MAP,
                    rmin: float = 4.0, rmax: float = 80.0):
    """k-th additional scan of the SAME scene as make_scene(m_points, ...) taken from another ground-truth pose
    (poses spread on a circle of radius 0.45 L, heading along the tangent, small roll / pitch), with its own
    perturbed start state.  Used by bench.py --rotate to cycle scans whose neighbourhood buckets do not all fit
    in the Infinity Cache together.  Returns dict(scan_xyz, x_true, x_init)."""
    rng_m = _Rng(seed_map)
    surf = _surfaces(rng_m, m_points)
    L = surf[4]
    ang = 2.0 * math.pi * (k * 0.381966011)   # golden-angle spacing: any number of poses stays spread out
    rad = 0.45 * L
    pos_true = np.array([rad * math.cos(ang), rad * math.sin(ang), 1.5 + 0.1 * (k % 3)])
    q_true = quat_from_rpy(math.radians(1.0 + (k % 2)), math.radians(-1.0), ang + math.pi / 2)
    R = quat_to_rot(q_true)
    rng_s = _Rng(SEED_SCAN + 7919 * (k + 1))
    chunks, have = [], 0
    while have < n_points:
        c = _sample(rng_s, surf, max(4096, 2 * (n_points - have)), sigma)
        r = np.linalg.norm(c - pos_true, axis=1)
        c = c[(r >= rmin) & (r <= rmax)]
        chunks.append(c)
        have += len(c)
    pw = np.concatenate(chunks)[:n_points]
    scan_xyz = ((pw - pos_true) @ R).astype(np.float32)
    x_true = make_state(pos_true, q_true)
    dq = quat_from_rotvec(np.radians([0.5, -0.4, 0.8]))
    x_init = make_state(pos_true + np.array([0.10, -0.07, 0.05]), quat_mul(q_true, dq))
    return dict(scan_xyz=scan_xyz, x_true=x_true, x_init=x_init)


# ---- BASELINE configs[4]: a LiDAR + IMU stream along a trajectory through the scene ----------------------------------
STREAM_G = 9.809


def stream_truth(t: float, v: float = 9.0, om: float = 0.3, yaw0: float = 0.5, p0=(-20.0, -25.0, 1.5)):
    """Ground truth of the stream: a level arc (speed v, yaw rate om).  Returns pos, R (body -> world), vel, world
    acceleration, quaternion (x, y, z, w)."""
    yaw = yaw0 + om * t
    pos = np.array([p0[0] + v / om * (math.sin(yaw) - math.sin(yaw0)), p0[1] - v / om * (math.cos(yaw) - math.cos(yaw0)), p0[2]])
    R = np.array([[math.cos(yaw), -math.sin(yaw), 0], [math.sin(yaw), math.cos(yaw), 0], [0, 0, 1.0]])
    vel = np.array([v * math.cos(yaw), v * math.sin(yaw), 0.0])
    a_world = np.array([-v * om * math.sin(yaw), v * om * math.cos(yaw), 0.0])
    q = np.array([0, 0, math.sin(yaw / 2), math.cos(yaw / 2)])
    return pos, R, vel, a_world, q


def stream_imu(t: float, om: float = 0.3, **kw):
    """IMU reading in the reference's convention (State.cpp:104 `R a - g`, Localizator.cpp:138): a stationary IMU reads
    (0, 0, -g)."""
    _, R, _, a_world, _ = stream_truth(t, om=om, **kw)
    return R.T @ (a_world - np.array([0, 0, STREAM_G])), np.array([0.0, 0.0, om])


def make_stream(m_points: int, n_revs: int, *, n_rings: int = 64, n_az: int = 1024, rev_time: float = 0.1, fov_deg=(-24.8, 2.0),
                sigma: float = 0.01, seed_map: int = SEED_MAP, seed_scan: int = SEED_SCAN, rmin: float = 4.0, rmax: float = 80.0,
                map_radius: float | None = None, t0: float = 0.0):
    """A spinning 64-ring LiDAR (HDL-64E-like vertical field of view) carried along stream_truth() through the scene of
    make_scene(m_points, ...): every azimuth step fires all rings at its own time, from the pose the sensor has THEN
    (so the raw points are skewed exactly as a moving sensor's are).  Returns dict(
      map_xyz  [M', 3] f32  the scene's map points (only those within map_radius of the start, if given),
      revs     list of n_revs dicts(xyz [n, 3] f32 LiDAR frame at firing time, t [n] f64 absolute stamps, stamp = end of the sweep),
      L)."""
    rng_m = _Rng(seed_map)
    surf = _surfaces(rng_m, m_points, max_boxes=64 if m_points <= 1_048_576 else None)
    map_xyz = _sample(rng_m, surf, m_points, sigma).astype(np.float32)
    o_all, eu_all, ev_all, _, L = surf
    if map_radius is not None:
        p_start = stream_truth(t0)[0]
        map_xyz = map_xyz[np.linalg.norm(map_xyz[:, :2] - p_start[:2].astype(np.float32), axis=1) < map_radius]
    el = np.radians(np.linspace(fov_deg[0], fov_deg[1], n_rings))
    ce, se = np.cos(el), np.sin(el)
    rng_s = _Rng(seed_scan)
    revs = []
    ctr_all = o_all + 0.5 * (eu_all + ev_all)
    half_all = 0.5 * (np.linalg.norm(eu_all, axis=1) + np.linalg.norm(ev_all, axis=1))
    for r in range(n_revs):
        # only the rectangles a ray of this sweep can reach (a 10M-point scene has thousands)
        p_mid = stream_truth(t0 + (r + 0.5) * rev_time)[0]
        near = np.linalg.norm(ctr_all - p_mid, axis=1) < rmax + half_all + 5.0
        o, eu, ev = o_all[near], eu_all[near], ev_all[near]
        nrm = np.cross(eu, ev)
        nrm /= np.linalg.norm(nrm, axis=1)[:, None]
        uu, vv = np.sum(eu * eu, axis=1), np.sum(ev * ev, axis=1)
        t_az = t0 + (r + (np.arange(n_az) + 1.0) / n_az) * rev_time            # firing time of every azimuth step
        az = 2.0 * math.pi * np.arange(n_az) / n_az
        pos = np.empty((n_az, 3))
        Rw = np.empty((n_az, 3, 3))
        for i, t in enumerate(t_az):
            p, R, _, _, _ = stream_truth(float(t))
            pos[i], Rw[i] = p, R
        dl = np.stack([ce[None, :] * np.cos(az)[:, None], ce[None, :] * np.sin(az)[:, None], se[None, :] * np.ones(n_az)[:, None]], axis=-1)
        dw = np.einsum("aij,arj->ari", Rw, dl).reshape(-1, 3)                   # [n_az * n_rings, 3]
        org = np.repeat(pos, n_rings, axis=0)
        tt = np.repeat(t_az, n_rings)
        best = np.full(len(dw), np.inf)
        for c0 in range(0, len(dw), 16384):
            d, og = dw[c0:c0 + 16384], org[c0:c0 + 16384]
            den = d @ nrm.T
            num = np.einsum("rk,rk->r", o, nrm)[None, :] - og @ nrm.T
            with np.errstate(divide="ignore", invalid="ignore"):
                t = num / den
            t[~np.isfinite(t) | (t < rmin) | (t > rmax)] = np.inf
            with np.errstate(invalid="ignore"):
                hit = og[:, None, :] + t[:, :, None] * d[:, None, :]
                rel = hit - o[None, :, :]
                a = np.sum(rel * eu[None, :, :], axis=2) / uu[None, :]
                b = np.sum(rel * ev[None, :, :], axis=2) / vv[None, :]
            t[(a < 0) | (a > 1) | (b < 0) | (b > 1) | ~np.isfinite(a) | ~np.isfinite(b)] = np.inf
            best[c0:c0 + 16384] = np.min(t, axis=1)
        ok = np.isfinite(best)
        pw = org[ok] + best[ok, None] * dw[ok]
        pw = pw + (2.0 * rng_s.uniform(len(pw) * 3).reshape(-1, 3) - 1.0) * (sigma * math.sqrt(3.0))
        # into the LiDAR frame of the firing instant (extrinsics = identity)
        idx = np.nonzero(ok)[0] // n_rings
        pl = np.einsum("nji,nj->ni", Rw[idx], pw - pos[idx])
        revs.append(dict(xyz=pl.astype(np.float32), t=tt[ok].copy(), stamp=float(t0 + (r + 1) * rev_time)))
    return dict(map_xyz=map_xyz, revs=revs, L=L)

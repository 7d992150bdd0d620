"""ctypes binding of the CPU oracle (oracle/liblvoracle.so).

TEST INFRASTRUCTURE ONLY — see oracle/lv_oracle.h.  Importable from tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg; never from the product package.  Parity: the in-tree half of the path is
pinned to the reference's own compiled sources (oracle/_ref, tests/test_oracle_ref.py); the absent
dependencies' half (kNN order, QR internals, esekf algebra) stays unpinned — see lv_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liblvoracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "lv_oracle.cpp")
    hdr = os.path.join(_HERE, "lv_oracle.h")
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.exists(f) and os.path.getmtime(f) > os.path.getmtime(_LIB_PATH) for f in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "-s"])
    return _LIB_PATH


class Params(C.Structure):
    _fields_ = [
        ("max_num_iters", C.c_int),
        ("num_match_points", C.c_int),
        ("max_dist_plane", C.c_double),
        ("planes_threshold", C.c_float),
        ("estimate_extrinsics", C.c_int),
        ("lidar_noise", C.c_double),
        ("limits", C.c_double * 23),
        ("degeneracy_mode", C.c_int),
        ("degeneracy_threshold", C.c_double),
    ]


class IterOut(C.Structure):
    _fields_ = [
        ("HTH", C.c_double * 144),
        ("HTh", C.c_double * 12),
        ("sum_h2", C.c_double),
        ("n_valid", C.c_int64),
    ]

    def as_dict(self):
        return dict(HTH=np.array(self.HTH).reshape(12, 12), HTh=np.array(self.HTh), sum_h2=float(self.sum_h2),
                    n_valid=int(self.n_valid))


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.lvo_kdtree_build.restype = C.c_void_p
        _lib.lvo_kdtree_size.restype = C.c_size_t
        _lib.lvo_plane_fit.restype = C.c_int
        _lib.lvo_update.restype = C.c_int
        _lib.lvo_kf_step.restype = C.c_int
        _lib.lvo_map_add.restype = C.c_size_t
        _lib.lvo_voxelgrid.restype = C.c_size_t
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def default_params(**kw) -> Params:
    p = Params()
    lib().lvo_default_params(C.byref(p))
    for k, v in kw.items():
        if k == "limits":
            for i in range(23):
                p.limits[i] = float(v[i])
        else:
            setattr(p, k, v)
    return p


def state_to_pose(state) -> np.ndarray:
    s = _f64(state)
    out = np.zeros(24, dtype=np.float32)
    lib().lvo_state_to_pose(_p(s, C.c_double), _p(out, C.c_float))
    return out


def transform_scan(state, scan_xyz) -> np.ndarray:
    pose = state_to_pose(state)
    scan = _f32(scan_xyz)
    out = np.empty_like(scan)
    lib().lvo_transform_scan(_p(pose, C.c_float), _p(scan, C.c_float), C.c_size_t(len(scan)), _p(out, C.c_float))
    return out


def knn_brute(map_xyz, q_xyz, k=5, nthreads=8):
    m, q = _f32(map_xyz), _f32(q_xyz)
    n = len(q)
    idx = np.empty((n, k), dtype=np.uint32)
    d2 = np.empty((n, k), dtype=np.float32)
    found = np.empty(n, dtype=np.int32)
    ties = C.c_int64(0)
    lib().lvo_knn_brute(_p(m, C.c_float), C.c_size_t(len(m)), _p(q, C.c_float), C.c_size_t(n), k,
                        _p(idx, C.c_uint32), _p(d2, C.c_float), _p(found, C.c_int32), C.byref(ties), nthreads)
    return idx, d2, found, int(ties.value)


class KdTree:
    def __init__(self, map_xyz):
        self.map_xyz = _f32(map_xyz)
        self.handle = C.c_void_p(lib().lvo_kdtree_build(_p(self.map_xyz, C.c_float), C.c_size_t(len(self.map_xyz))))

    def knn(self, q_xyz, k=5, nthreads=8):
        q = _f32(q_xyz)
        n = len(q)
        idx = np.empty((n, k), dtype=np.uint32)
        d2 = np.empty((n, k), dtype=np.float32)
        found = np.empty(n, dtype=np.int32)
        lib().lvo_kdtree_knn(self.handle, _p(q, C.c_float), C.c_size_t(n), k, _p(idx, C.c_uint32), _p(d2, C.c_float),
                             _p(found, C.c_int32), nthreads)
        return idx, d2, found

    def close(self):
        if self.handle:
            lib().lvo_kdtree_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def plane_fit(near_xyz, sq_dists, params=None):
    prm = params or default_params()
    near = _f32(near_xyz)
    sq = _f32(sq_dists)
    abcd = np.zeros(4, dtype=np.float32)
    ok = lib().lvo_plane_fit(_p(near, C.c_float), _p(sq, C.c_float), len(sq), C.byref(prm), _p(abcd, C.c_float))
    return bool(ok), abcd


def plane_fit_f64(near_xyz):
    near = _f32(near_xyz)
    abcd = np.zeros(4, dtype=np.float64)
    lib().lvo_plane_fit_f64(_p(near, C.c_float), len(near), _p(abcd, C.c_double))
    return abcd


def calculate_H_row(state, p_w, abcd, dist, estimate_extrinsics=False):
    s = _f64(state)
    pw, ab = _f32(p_w), _f32(abcd)
    row = np.zeros(12)
    h = C.c_double(0)
    lib().lvo_calculate_H_row(_p(s, C.c_double), _p(pw, C.c_float), _p(ab, C.c_float), C.c_float(dist),
                              int(estimate_extrinsics), _p(row, C.c_double), C.byref(h))
    return row, float(h.value)


def iterate(state, map_xyz, scan_xyz, params=None, tree: KdTree | None = None, nthreads=8, details=True):
    prm = params or default_params()
    s, m, q = _f64(state), _f32(map_xyz), _f32(scan_xyz)
    n, k = len(q), prm.num_match_points
    out = IterOut()
    d = {}
    if details:
        d = dict(knn_idx=np.empty((n, k), np.uint32), knn_d2=np.empty((n, k), np.float32), valid=np.empty(n, np.uint8),
                 abcd=np.empty((n, 4), np.float32), dist=np.empty(n, np.float32), Hrows=np.empty((n, 12), np.float64),
                 h=np.empty(n, np.float64))
    lib().lvo_iterate(_p(s, C.c_double), C.byref(prm), tree.handle if tree else None, _p(m, C.c_float),
                      C.c_size_t(len(m)), _p(q, C.c_float), C.c_size_t(n), C.byref(out),
                      _p(d.get("knn_idx"), C.c_uint32), _p(d.get("knn_d2"), C.c_float), _p(d.get("valid"), C.c_uint8),
                      _p(d.get("abcd"), C.c_float), _p(d.get("dist"), C.c_float), _p(d.get("Hrows"), C.c_double),
                      _p(d.get("h"), C.c_double), nthreads)
    res = out.as_dict()
    res.update(d)
    return res


def set_qr_backsub_columns(on: bool) -> None:
    """Experiment switch (NOT the default, never used as the reference of a parity test): the plane fit's back substitution in
    Eigen 3.3's column-oriented order instead of the row-oriented one the oracle, the stand-in Eigen and the device share."""
    lib().lvo_set_qr_backsub_columns(1 if on else 0)


def update(state, P, map_xyz, scan_xyz, params=None, tree: KdTree | None = None, nthreads=8):
    """Full iterated update.  Returns (x_post[26], P_post[23,23], passes, trace[passes,49], per-pass sums)."""
    prm = params or default_params()
    x = _f64(state).copy()
    Pm = _f64(P).copy().reshape(23, 23)
    m, q = _f32(map_xyz), _f32(scan_xyz)
    npass = prm.max_num_iters + 1
    trace = np.zeros((npass, 49))
    sums = (IterOut * npass)()
    passes = lib().lvo_update(_p(x, C.c_double), _p(Pm, C.c_double), C.byref(prm), tree.handle if tree else None,
                              _p(m, C.c_float), C.c_size_t(len(m)), _p(q, C.c_float), C.c_size_t(len(q)),
                              _p(trace, C.c_double), sums, nthreads)
    return x, Pm, passes, trace[:passes], [sums[i].as_dict() for i in range(passes)]


def kf_step(x, x_prop, P_prop, sums: dict, params=None, finalize=True):
    prm = params or default_params()
    xs = _f64(x).copy()
    xp = _f64(x_prop)
    Pp = _f64(P_prop).reshape(23, 23)
    io = IterOut()
    HTH = _f64(sums["HTH"]).ravel()
    for i in range(144):
        io.HTH[i] = HTH[i]
    for i in range(12):
        io.HTh[i] = float(sums["HTh"][i])
    io.sum_h2 = float(sums.get("sum_h2", 0.0))
    io.n_valid = int(sums.get("n_valid", 1))
    dx = np.zeros(23)
    Pout = np.zeros((23, 23))
    conv = lib().lvo_kf_step(_p(xs, C.c_double), _p(xp, C.c_double), _p(Pp, C.c_double), C.byref(prm), C.byref(io),
                             _p(dx, C.c_double), int(finalize), _p(Pout, C.c_double))
    return xs, dx, bool(conv), Pout


def degeneracy(sums: dict, params=None):
    """The degeneracy stage alone: returns (eigenvalues[6] of the pose block, possibly modified sums)."""
    prm = params or default_params()
    io = IterOut()
    HTH = _f64(sums["HTH"]).ravel()
    for i in range(144):
        io.HTH[i] = HTH[i]
    for i in range(12):
        io.HTh[i] = float(sums["HTh"][i])
    io.sum_h2 = float(sums.get("sum_h2", 0.0))
    io.n_valid = int(sums.get("n_valid", 1))
    eig = np.zeros(6)
    lib().lvo_degeneracy(C.byref(io), C.byref(prm), _p(eig, C.c_double))
    return eig, io.as_dict()


def boxplus(x, dx):
    xs = _f64(x).copy()
    d = _f64(dx)
    lib().lvo_boxplus(_p(xs, C.c_double), _p(d, C.c_double))
    return xs


def boxminus(x, other):
    a, b = _f64(x), _f64(other)
    d = np.zeros(23)
    lib().lvo_boxminus(_p(a, C.c_double), _p(b, C.c_double), _p(d, C.c_double))
    return d


def predict(x, P, dt, Q, acc, gyro):
    xs = _f64(x).copy()
    Pm = _f64(P).copy().reshape(23, 23)
    Qm = _f64(Q).reshape(12, 12)
    a, g = _f64(acc), _f64(gyro)
    lib().lvo_predict(_p(xs, C.c_double), _p(Pm, C.c_double), C.c_double(dt), _p(Qm, C.c_double), _p(a, C.c_double),
                      _p(g, C.c_double))
    return xs, Pm


def map_add(map_xyz, new_xyz, downsample=True, box_length=0.2):
    m, k = _f32(map_xyz).reshape(-1, 3), _f32(new_xyz).reshape(-1, 3)
    out = np.empty((len(m) + len(k), 3), np.float32)
    n = lib().lvo_map_add(_p(m, C.c_float), C.c_size_t(len(m)), _p(k, C.c_float), C.c_size_t(len(k)), int(downsample),
                          C.c_float(box_length), _p(out, C.c_float))
    return out[:n].copy()


MOTION_DTYPE = np.dtype([("R", "f4", 9), ("pos", "f4", 3), ("vel", "f4", 3), ("bw", "f4", 3), ("ba", "f4", 3), ("g", "f4", 3),
                         ("RLI", "f4", 9), ("tLI", "f4", 3), ("a", "f4", 3), ("w", "f4", 3), ("pad_", "f4", 2), ("time", "f8")])
assert MOTION_DTYPE.itemsize == 184


def motion_state(R=None, pos=(0, 0, 0), vel=(0, 0, 0), a=(0, 0, 9.807), w=(0, 0, 0), time=0.0, RLI=None, tLI=(0, 0, 0),
                 g=(0, 0, -9.807), bw=(0, 0, 0), ba=(0, 0, 0)):
    s = np.zeros(1, MOTION_DTYPE)
    s["R"] = np.eye(3, dtype=np.float32).ravel() if R is None else np.asarray(R, np.float32).ravel()
    s["RLI"] = np.eye(3, dtype=np.float32).ravel() if RLI is None else np.asarray(RLI, np.float32).ravel()
    for k, v in (("pos", pos), ("vel", vel), ("a", a), ("w", w), ("tLI", tLI), ("g", g), ("bw", bw), ("ba", ba)):
        s[k] = np.asarray(v, np.float32)
    s["time"] = time
    return s


def state_integrate(state, a, w, t):
    s = state.copy()
    av, wv = _f32(a), _f32(w)
    lib().lvo_state_integrate(s.ctypes.data_as(C.c_void_p), _p(av, C.c_float), _p(wv, C.c_float), C.c_double(t))
    return s


def sincos_f32(x):
    v = _f32(x).ravel()
    sn, cs = np.empty_like(v), np.empty_like(v)
    lib().lvo_sincos_f32(_p(v, C.c_float), C.c_size_t(len(v)), _p(sn, C.c_float), _p(cs, C.c_float))
    return sn, cs


def sincos_vs_libm(x):
    """(arguments whose sin differs, whose cos differs, largest difference in ulps) between the pinned polynomial of row
    f-2 and this platform's sinf / cosf."""
    v = _f32(x).ravel()
    ns, nc, mu = C.c_int64(0), C.c_int64(0), C.c_int32(0)
    lib().lvo_sincos_vs_libm(_p(v, C.c_float), C.c_size_t(len(v)), C.byref(ns), C.byref(nc), C.byref(mu))
    return int(ns.value), int(nc.value), int(mu.value)


def deskew(xyz, times, states, Xt2):
    p = _f32(xyz).reshape(-1, 3)
    t = _f64(times)
    st = np.ascontiguousarray(states)
    out = np.empty_like(p)
    lib().lvo_deskew(_p(p, C.c_float), _p(t, C.c_double), C.c_size_t(len(p)), st.ctypes.data_as(C.c_void_p), C.c_size_t(len(st)),
                     np.ascontiguousarray(Xt2).ctypes.data_as(C.c_void_p), _p(out, C.c_float))
    return out


def voxelgrid(xyz, leaf):
    p = _f32(xyz).reshape(-1, 3)
    out = np.empty_like(p)
    n = lib().lvo_voxelgrid(_p(p, C.c_float), C.c_size_t(len(p)), C.c_float(leaf), _p(out, C.c_float))
    return out[:n].copy()


# ---- row f-4: LiDAR wire formats ---------------------------------------------------------------------------
class CloudFormat(C.Structure):
    _fields_ = [("point_step", C.c_uint32), ("off_x", C.c_uint32), ("off_y", C.c_uint32), ("off_z", C.c_uint32),
                ("off_time", C.c_uint32), ("time_type", C.c_int), ("off_intensity", C.c_uint32), ("intensity_type", C.c_int),
                ("off_range", C.c_uint32), ("range_type", C.c_int), ("relative_time", C.c_int)]


class IngestParams(C.Structure):
    _fields_ = [("header_stamp_usec", C.c_uint64), ("stamp_beginning", C.c_int), ("offset_beginning", C.c_int),
                ("full_rotation_time", C.c_double), ("downsample_rate", C.c_int), ("min_dist", C.c_float)]


POINT_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("z", "f4"), ("pad_", "f4"), ("time", "f8"), ("intensity", "f4"), ("range", "f4")])
assert POINT_DTYPE.itemsize == 32


def cloud_ingest(raw: bytes, n: int, fmt: CloudFormat, prm: IngestParams) -> np.ndarray:
    """Accumulator::process restated: returns the kept, time-sorted reference Points (POINT_DTYPE)."""
    out = np.zeros(max(n, 1), POINT_DTYPE)
    buf = (C.c_char * len(raw)).from_buffer_copy(raw)
    lib().lvo_cloud_ingest.restype = C.c_size_t
    k = lib().lvo_cloud_ingest(buf, C.c_size_t(n), C.byref(fmt), C.byref(prm), out.ctypes.data_as(C.c_void_p))
    return out[:k].copy()

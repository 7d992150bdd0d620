"""ctypes binding of oracle/_ref/liblvref.so — the REFERENCE's own in-tree sources (/root/reference/src) compiled in place against
stand-in headers (oracle/ref_build/).  TEST INFRASTRUCTURE ONLY: used by tests/test_oracle_ref.py to pin the oracle
(oracle/lv_oracle.cpp) to what the reference's code computes, function by function.  The reference mount exists only in the
build container; on the GPU box the prebuilt library (it travels with gpurun, it is not in git) is used as is, and without
either the tests skip."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

import lvoracle as lo

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_ref", "liblvref.so")
REFERENCE = os.environ.get("LV_REFERENCE_DIR", "/root/reference")


def build() -> str | None:
    """make -C oracle/ref_build when the reference mount is there; the path of the library, or None when it cannot exist."""
    if os.path.isdir(os.path.join(REFERENCE, "src")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "ref_build"), f"REF={REFERENCE}"])
    return _LIB_PATH if os.path.exists(_LIB_PATH) else None


class Config(C.Structure):
    _fields_ = [("estimate_extrinsics", C.c_int), ("max_num_iters", C.c_int), ("num_match_points", C.c_int), ("max_points2match", C.c_int),
                ("max_dist_plane", C.c_double), ("planes_threshold", C.c_float), ("lidar_noise", C.c_double), ("degeneracy_threshold", C.c_double),
                ("limits", C.c_double * 23), ("initial_gravity", C.c_float * 3), ("I_Rotation_L", C.c_float * 9), ("I_Translation_L", C.c_float * 3),
                ("cov_acc", C.c_double), ("cov_gyro", C.c_double), ("cov_bias_acc", C.c_double), ("cov_bias_gyro", C.c_double),
                ("full_rotation_time", C.c_double), ("imu_rate", C.c_double), ("real_time_delay", C.c_double), ("min_dist", C.c_double),
                ("offset_beginning", C.c_int), ("stamp_beginning", C.c_int), ("downsample_rate", C.c_int), ("lidar_type", C.c_int),
                ("downsample_prec", C.c_float)]


LIDAR = {"velodyne": 0, "hesai": 1, "ouster": 2, "custom": 3}
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = build()
        if path is None:
            raise RuntimeError("oracle/_ref/liblvref.so is not built and the reference mount is absent")
        _lib = C.CDLL(path)
        for f in ("lvr_map_size", "lvr_match", "lvr_deskew", "lvr_path", "lvr_cloud_ingest", "lvr_buffer_window"):
            getattr(_lib, f).restype = C.c_size_t
        _lib.lvr_plane.restype = C.c_int
        _lib.lvr_update.restype = C.c_int
    return _lib


def default_config(**kw) -> Config:
    """config/params.yaml defaults of the keys the compiled sources read (params.yaml:17-53)."""
    c = Config()
    c.estimate_extrinsics, c.max_num_iters, c.num_match_points, c.max_points2match = 0, 3, 5, 10
    c.max_dist_plane, c.planes_threshold, c.lidar_noise, c.degeneracy_threshold = 2.0, 0.05, 1e-3, 5.0
    for i in range(23):
        c.limits[i] = 1e-3
    c.initial_gravity[:] = [0.0, 0.0, -9.807]
    c.I_Rotation_L[:] = [1, 0, 0, 0, 1, 0, 0, 0, 1]
    c.I_Translation_L[:] = [0, 0, 0]
    c.cov_acc, c.cov_gyro, c.cov_bias_acc, c.cov_bias_gyro = 1e-2, 1e-4, 1e-4, 1e-5
    c.full_rotation_time, c.imu_rate, c.real_time_delay, c.min_dist = 0.1, 400.0, 1.0, 3.0
    c.offset_beginning, c.stamp_beginning, c.downsample_rate, c.lidar_type, c.downsample_prec = 0, 0, 1, 0, 0.2
    for k, v in kw.items():
        if k in ("limits", "initial_gravity", "I_Rotation_L", "I_Translation_L"):
            getattr(c, k)[:] = list(v)
        elif k == "lidar_type" and isinstance(v, str):
            c.lidar_type = LIDAR[v]
        else:
            setattr(c, k, v)
    return c


def set_config(c: Config | None = None, **kw):
    c = c or default_config(**kw)
    lib().lvr_set_config(C.byref(c))
    return c


def reset():
    lib().lvr_reset()


def state_to_pose(state) -> np.ndarray:
    out = np.zeros(24, np.float32)
    lib().lvr_state_to_pose(lo._p(lo._f64(state), C.c_double), lo._p(out, C.c_float))
    return out


def transform(state, scan_xyz) -> np.ndarray:
    p = lo._f32(scan_xyz).reshape(-1, 3)
    out = np.empty_like(p)
    lib().lvr_transform(lo._p(lo._f64(state), C.c_double), lo._p(p, C.c_float), C.c_size_t(len(p)), lo._p(out, C.c_float))
    return out


def map_add(xyz, time=0.0, downsample=False):
    p = lo._f32(xyz).reshape(-1, 3)
    lib().lvr_map_add(lo._p(p, C.c_float), C.c_size_t(len(p)), C.c_double(time), C.c_int(int(downsample)))


def map_size() -> int:
    return int(lib().lvr_map_size())


def map_fetch() -> np.ndarray:
    out = np.zeros((map_size(), 3), np.float32)
    lib().lvr_map_fetch(lo._p(out, C.c_float))
    return out


def match(state, scan_xyz) -> dict:
    p = lo._f32(scan_xyz).reshape(-1, 3)
    n = len(p)
    src, pw, abcd, dist = np.zeros(n, np.uint32), np.zeros((n, 3), np.float32), np.zeros((n, 4), np.float32), np.zeros(n, np.float32)
    k = lib().lvr_match(lo._p(lo._f64(state), C.c_double), lo._p(p, C.c_float), C.c_size_t(n), lo._p(src, C.c_uint32), lo._p(pw, C.c_float),
                        lo._p(abcd, C.c_float), lo._p(dist, C.c_float))
    return dict(src=src[:k].copy(), p_world=pw[:k].copy(), abcd=abcd[:k].copy(), dist=dist[:k].copy())


def plane(near_xyz, sq_dists):
    p = lo._f32(near_xyz).reshape(-1, 3)
    sq = lo._f32(sq_dists)
    abcd = np.zeros(4, np.float32)
    ok = lib().lvr_plane(lo._p(p, C.c_float), lo._p(sq, C.c_float), C.c_int(len(p)), lo._p(abcd, C.c_float))
    return bool(ok), abcd


def estimate_plane(near_xyz) -> np.ndarray:
    p = lo._f32(near_xyz).reshape(-1, 3)
    abcd = np.zeros(4, np.float32)
    lib().lvr_estimate_plane(lo._p(p, C.c_float), C.c_int(len(p)), lo._p(abcd, C.c_float))
    return abcd


def calculate_H(state, p_world, abcd):
    pw, ab = lo._f32(p_world).reshape(-1, 3), lo._f32(abcd).reshape(-1, 4)
    n = len(pw)
    H, h, dist = np.zeros((n, 12)), np.zeros(n), np.zeros(n, np.float32)
    lib().lvr_calculate_H(lo._p(lo._f64(state), C.c_double), C.c_size_t(n), lo._p(pw, C.c_float), lo._p(ab, C.c_float), lo._p(H, C.c_double),
                          lo._p(h, C.c_double), lo._p(dist, C.c_float))
    return H, h, dist


def update(state, P, scan_xyz):
    x = lo._f64(state).copy()
    Pm = lo._f64(P).copy()
    p = lo._f32(scan_xyz).reshape(-1, 3)
    sums = (lo.IterOut * 16)()
    tr = np.zeros((16, 26))
    n = lib().lvr_update(lo._p(x, C.c_double), lo._p(Pm, C.c_double), lo._p(p, C.c_float), C.c_size_t(len(p)), sums, lo._p(tr, C.c_double))
    return x, Pm.reshape(23, 23), int(n), tr[:n].copy(), [sums[i].as_dict() for i in range(n)]


def initialize(a, w, q_xyzw, t):
    x, P = np.zeros(26), np.zeros((23, 23))
    lib().lvr_initialize(lo._p(lo._f32(a), C.c_float), lo._p(lo._f32(w), C.c_float), lo._p(lo._f32(q_xyzw), C.c_float), C.c_double(t),
                         lo._p(x, C.c_double), lo._p(P, C.c_double))
    return x, P


def propagate(state, P, last_time_integrated, imu_a, imu_w, imu_t, t):
    x, Pm = lo._f64(state).copy(), lo._f64(P).copy()
    a, w, ts = lo._f32(imu_a).reshape(-1, 3), lo._f32(imu_w).reshape(-1, 3), lo._f64(imu_t)
    lib().lvr_propagate(lo._p(x, C.c_double), lo._p(Pm, C.c_double), C.c_double(last_time_integrated), lo._p(a, C.c_float), lo._p(w, C.c_float),
                        lo._p(ts, C.c_double), C.c_size_t(len(ts)), C.c_double(t))
    return x, Pm.reshape(23, 23)


def state_integrate(state, a, w, t):
    s = state.copy()
    lib().lvr_state_integrate(s.ctypes.data_as(C.c_void_p), lo._p(lo._f32(a), C.c_float), lo._p(lo._f32(w), C.c_float), C.c_double(t))
    return s


def deskew(xyz, times, states, Xt2):
    p, t = lo._f32(xyz).reshape(-1, 3), lo._f64(times)
    st = np.ascontiguousarray(states)
    out = np.full_like(p, np.nan)
    k = lib().lvr_deskew(lo._p(p, C.c_float), lo._p(t, C.c_double), C.c_size_t(len(p)), st.ctypes.data_as(C.c_void_p), C.c_size_t(len(st)),
                         np.ascontiguousarray(Xt2).ctypes.data_as(C.c_void_p), lo._p(out, C.c_float))
    return out, int(k)


def path(states, imu_a, imu_w, imu_t, t1, t2, cap=4096):
    st = np.ascontiguousarray(states)
    a, w, ts = lo._f32(imu_a).reshape(-1, 3), lo._f32(imu_w).reshape(-1, 3), lo._f64(imu_t)
    out = np.zeros(cap, lo.MOTION_DTYPE)
    k = lib().lvr_path(st.ctypes.data_as(C.c_void_p), C.c_size_t(len(st)), lo._p(a, C.c_float), lo._p(w, C.c_float), lo._p(ts, C.c_double),
                       C.c_size_t(len(ts)), C.c_double(t1), C.c_double(t2), out.ctypes.data_as(C.c_void_p), C.c_size_t(cap))
    return out[:k].copy()


_DT = {"i1": 1, "u1": 2, "i2": 3, "u2": 4, "i4": 5, "u4": 6, "f4": 7, "f8": 8}


def cloud_ingest(raw: bytes, n: int, dtype: np.dtype, stamp_usec: int) -> np.ndarray:
    """PointCloudProcessor::msg2points -> downsample -> sort_points on a message whose fields are those of the numpy record
    dtype (names, offsets, types) — Config.LiDAR_type etc. as set by set_config."""
    names = list(dtype.names)
    arr = (C.c_char_p * len(names))(*[s.encode() for s in names])
    offs = np.array([dtype.fields[k][1] for k in names], np.uint32)
    dts = np.array([_DT[dtype.fields[k][0].str.lstrip("<|=")] for k in names], np.uint8)
    out = np.zeros(max(n, 1), lo.POINT_DTYPE)
    buf = (C.c_uint8 * len(raw)).from_buffer_copy(raw)
    k = lib().lvr_cloud_ingest(buf, C.c_size_t(n), C.c_uint32(dtype.itemsize), C.c_int(len(names)), arr, lo._p(offs, C.c_uint32), lo._p(dts, C.c_uint8),
                               C.c_uint64(stamp_usec), out.ctypes.data_as(C.c_void_p))
    return out[:k].copy()


def buffer_window(times, t1, t2, clear_t=None) -> np.ndarray:
    t = lo._f64(times)
    out = np.zeros(len(t))
    k = lib().lvr_buffer_window(lo._p(t, C.c_double), C.c_size_t(len(t)), C.c_double(t1), C.c_double(t2),
                                C.c_double(-1e301 if clear_t is None else clear_t), lo._p(out, C.c_double))
    return out[:k].copy()

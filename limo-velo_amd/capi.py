"""ctypes binding of the C-ABI (include/limovelo_hip.h) for tests, smoke() and bench.py.

This is plumbing, not the product: the product is liblimovelo_hip.so (HIP, gfx950) and the C++
Mapper / Localizator shim in limo-velo_amd/host/.  There is NO CPU fallback: loading fails loudly if
the HIP library has not been built, and Context() fails if no GPU is present.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (LV_LIB_PATH: another build of the same library, for A/B measurements of two builds in one box — scripts/gpu_ab_multi.sh)
LIB_PATH = os.environ.get("LV_LIB_PATH") or os.path.join(_HERE, "liblimovelo_hip.so")

LV_OK = 0
PEER_HANDLE_BYTES = 128   # LV_PEER_HANDLE_BYTES: two HIP IPC handles (gather buffers, flag word)
SUMS_LEN = 96
NS = 23

# every symbol include/limovelo_hip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "lv_default_params", "lv_last_error", "lv_version", "lv_create", "lv_destroy", "lv_set_stream", "lv_get_stream",
    "lv_synchronize", "lv_map_build", "lv_map_add", "lv_map_add_scan", "lv_map_evict_box", "lv_map_evict_oldest", "lv_map_relinearise", "lv_map_get_stats",
    "lv_map_size", "lv_map_fetch", "lv_scan_set", "lv_scan_deskew", "lv_scan_downsample", "lv_scan_size", "lv_scan_fetch", "lv_iterate", "lv_pseudo_measurement",
    "lv_map_relinearise_async", "lv_map_reserve_rebuild", "lv_map_rebuild_status", "lv_update", "lv_filter_set", "lv_filter_get", "lv_predict", "lv_correct", "lv_get_degeneracy_values", "lv_update_begin", "lv_pass_reduce", "lv_sums_device_ptr", "lv_set_sums_buffer", "lv_pass_solve", "lv_update_end",
    "lv_set_capture", "lv_fetch_knn", "lv_fetch_matches", "lv_fetch_neighbors", "lv_set_record_dump", "lv_last_update_fused", "lv_last_passes", "lv_set_fused_pass", "lv_set_option", "lv_get_pass_clocks", "lv_pass_geometry", "lv_fetch_rows", "lv_calculate_H", "lv_get_timing", "lv_set_profiling", "lv_get_phase_clocks", "lv_get_solve_clocks", "lv_get_level_histogram",
    "lv_comm_unique_id", "lv_comm_init", "lv_comm_destroy", "lv_comm_world", "lv_comm_set_shard_max", "lv_set_comm_fused", "lv_comm_set_host_gather", "lv_comm_peer_export", "lv_comm_peer_init",
    "lv_cloud_format_preset", "lv_cloud_ingest", "lv_cloud_size", "lv_cloud_fetch", "lv_cloud_clear", "lv_cloud_reserve", "lv_reserve_stream", "lv_scan_deskew_window",
]


def pseudo_measurement(sums: dict, estimate_extrinsics: bool):
    """lv_pseudo_measurement: (h_x [rows, 12], h [rows]) with h_x^T h_x = H^T H and h_x^T h = H^T h — the Eigen-free half of
    IKFoM::h_share_model for an unmodified esekf loop (host arithmetic: no context, no device)."""
    rec = Sums()
    HTH = np.ascontiguousarray(sums["HTH"], np.float64).ravel()
    for i in range(144):
        rec.HTH[i] = HTH[i]
    for i in range(12):
        rec.HTh[i] = float(sums["HTh"][i])
    rec.sum_h2 = float(sums.get("sum_h2", 0.0))
    rec.n_valid = int(sums.get("n_valid", 1))
    hx = np.zeros((12, 12))
    h = np.zeros(12)
    rows = C.c_int(0)
    lib = load_library()
    lib.lv_pseudo_measurement.argtypes = [C.POINTER(Sums), C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    if lib.lv_pseudo_measurement(C.byref(rec), int(bool(estimate_extrinsics)), hx.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p),
                                 C.byref(rows)) != LV_OK:
        raise RuntimeError(lib.lv_last_error().decode())
    return hx[: rows.value].copy(), h[: rows.value].copy()


def pass_geometry(n_scan: int, n_cus: int = 256):
    """(searching workgroups, steps per round, rounds, dedicated bookkeeper) of pass_kernel — host logic only."""
    out = (C.c_int * 4)()
    lib = load_library()
    lib.lv_pass_geometry.argtypes = [C.c_size_t, C.c_int, C.POINTER(C.c_int)]
    if lib.lv_pass_geometry(n_scan, n_cus, out) != LV_OK:
        raise RuntimeError(lib.lv_last_error().decode())
    return tuple(int(v) for v in out)


class CloudFormat(C.Structure):  # lv_cloud_format
    _fields_ = [("point_step", C.c_uint32), ("off_x", C.c_uint32), ("off_y", C.c_uint32), ("off_z", C.c_uint32),
                ("off_time", C.c_uint32), ("time_type", C.c_int), ("off_intensity", C.c_uint32), ("intensity_type", C.c_int),
                ("off_range", C.c_uint32), ("range_type", C.c_int), ("relative_time", C.c_int)]


class IngestParams(C.Structure):  # lv_ingest_params
    _fields_ = [("header_stamp_usec", C.c_uint64), ("stamp_beginning", C.c_int), ("offset_beginning", C.c_int),
                ("full_rotation_time", C.c_double), ("downsample_rate", C.c_int), ("min_dist", C.c_float)]


POINT_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("z", "f4"), ("pad_", "f4"), ("time", "f8"), ("intensity", "f4"), ("range", "f4")])
LIDAR_VELODYNE, LIDAR_HESAI, LIDAR_OUSTER, LIDAR_CUSTOM = 0, 1, 2, 3


class Params(C.Structure):
    _fields_ = [
        ("MAX_NUM_ITERS", C.c_int),
        ("NUM_MATCH_POINTS", C.c_int),
        ("MAX_DIST_PLANE", C.c_double),
        ("PLANES_THRESHOLD", C.c_float),
        ("estimate_extrinsics", C.c_int),
        ("LiDAR_noise", C.c_double),
        ("LIMITS", C.c_double * NS),
        ("degeneracy_threshold", C.c_double),
        ("voxel_size", C.c_float),
        ("lanes_per_query", C.c_int),
        ("degeneracy_mode", C.c_int),
        ("print_degeneracy_values", C.c_int),
    ]


class Sums(C.Structure):
    _fields_ = [("HTH", C.c_double * 144), ("HTh", C.c_double * 12), ("sum_h2", C.c_double), ("n_valid", C.c_int64)]

    def as_dict(self):
        return dict(HTH=np.array(self.HTH).reshape(12, 12), HTh=np.array(self.HTh), sum_h2=float(self.sum_h2),
                    n_valid=int(self.n_valid))


class Timing(C.Structure):
    _fields_ = [("last_update_ms", C.c_float), ("last_reduce_ms", C.c_float), ("last_solve_ms", C.c_float),
                ("last_passes", C.c_int), ("fallback_queries", C.c_int), ("pass_match_ms", C.c_float * 8),
                ("pass_solve_ms", C.c_float * 8), ("mailbox_resyncs", C.c_int), ("pass_collective_ms", C.c_float * 8)]


# lv_motion_state (include/limovelo_hip.h): the f32 State members State::propagate_f reads, 184 bytes
MOTION_DTYPE = np.dtype([("R", "f4", 9), ("pos", "f4", 3), ("vel", "f4", 3), ("bw", "f4", 3), ("ba", "f4", 3), ("g", "f4", 3),
                         ("RLI", "f4", 9), ("tLI", "f4", 3), ("a", "f4", 3), ("w", "f4", 3), ("pad_", "f4", 2), ("time", "f8")])
assert MOTION_DTYPE.itemsize == 184


def motion_state(time=0.0, R=None, pos=(0, 0, 0), vel=(0, 0, 0), a=(0, 0, 9.807), w=(0, 0, 0), RLI=None, tLI=(0, 0, 0),
                 g=(0, 0, -9.807), bw=(0, 0, 0), ba=(0, 0, 0)) -> np.ndarray:
    """One lv_motion_state record (a numpy array of length 1)."""
    s = np.zeros(1, MOTION_DTYPE)
    s["R"] = np.eye(3, dtype=np.float32).ravel() if R is None else np.asarray(R, np.float32).ravel()
    s["RLI"] = np.eye(3, dtype=np.float32).ravel() if RLI is None else np.asarray(RLI, np.float32).ravel()
    for k, v in (("pos", pos), ("vel", vel), ("a", a), ("w", w), ("tLI", tLI), ("g", g), ("bw", bw), ("ba", ba)):
        s[k] = np.asarray(v, np.float32)
    s["time"] = time
    return s


class MapStats(C.Structure):  # lv_map_stats
    _fields_ = [("living", C.c_uint64), ("ids", C.c_uint64), ("capacity", C.c_uint64), ("pool_used", C.c_uint64 * 4),
                ("pool_cap", C.c_uint64 * 4), ("slots_used", C.c_uint64 * 4), ("slots_cap", C.c_uint64 * 4),
                ("tombstones", C.c_uint64), ("dropped", C.c_uint64), ("relinearisations", C.c_uint64),
                ("incremental_adds", C.c_uint64), ("bytes", C.c_uint64)]


class LvError(RuntimeError):
    pass


_lib = None


def load_library() -> C.CDLL:
    """dlopen liblimovelo_hip.so.  Raises (never falls back) if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LvError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          f"(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        lib.lv_last_error.restype = C.c_char_p
        lib.lv_version.restype = C.c_char_p
        lib.lv_scan_size.restype = C.c_size_t
        lib.lv_scan_size.argtypes = [C.c_void_p]
        lib.lv_map_size.restype = C.c_size_t
        lib.lv_cloud_size.restype = C.c_size_t
        lib.lv_cloud_size.argtypes = [C.c_void_p]
        lib.lv_map_size.argtypes = [C.c_void_p]
        lib.lv_get_stream.restype = C.c_void_p
        lib.lv_get_stream.argtypes = [C.c_void_p]
        lib.lv_sums_device_ptr.restype = C.c_void_p
        lib.lv_sums_device_ptr.argtypes = [C.c_void_p]
        lib.lv_destroy.restype = None
        lib.lv_destroy.argtypes = [C.c_void_p]
        lib.lv_default_params.restype = None
        _lib = lib
    return _lib


def default_params(**kw) -> Params:
    p = Params()
    load_library().lv_default_params(C.byref(p))
    for k, v in kw.items():
        if k == "LIMITS":
            for i in range(NS):
                p.LIMITS[i] = float(v[i])
        else:
            setattr(p, k, v)
    return p


def _points(a):
    """Accepts [N,3] float32 (stride 12) or a structured/2-D array with a custom byte stride."""
    a = np.asarray(a)
    if a.dtype != np.float32 or a.ndim != 2 or a.shape[1] < 3 or not a.flags["C_CONTIGUOUS"]:
        a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.strides[0], a.shape[0]


class Context:
    def __init__(self, params: Params | None = None, device: int = 0):
        self.lib = load_library()
        self.params = params or default_params()
        self.h = C.c_void_p()
        self._check(self.lib.lv_create(C.byref(self.params), int(device), C.byref(self.h)))
        # pre-bound buffers for the hot call (keeps the Python overhead of update() at a few microseconds)
        self._xb = np.zeros(26)
        self._Pb = np.zeros((NS, NS))
        self._passes = C.c_int(0)
        self._xp = self._xb.ctypes.data_as(C.c_void_p)
        self._Pp = self._Pb.ctypes.data_as(C.c_void_p)
        self._passes_ref = C.byref(self._passes)

    def _check(self, rc):
        if rc != LV_OK:
            raise LvError(f"limovelo_hip error {rc}: {self.lib.lv_last_error().decode()}")

    def close(self):
        if self.h:
            self.lib.lv_destroy(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- Mapper side
    def map_build(self, pts):
        a, stride, n = _points(pts)
        self._check(self.lib.lv_map_build(self.h, a.ctypes.data_as(C.c_void_p), C.c_size_t(stride), C.c_size_t(n)))

    def map_add(self, pts, downsample=False):
        a, stride, n = _points(pts)
        self._check(self.lib.lv_map_add(self.h, a.ctypes.data_as(C.c_void_p), C.c_size_t(stride), C.c_size_t(n),
                                        int(bool(downsample))))

    def map_add_scan(self, downsample=True):
        """The mapping step on the device: current scan -> world with the device-held state -> insert."""
        self._check(self.lib.lv_map_add_scan(self.h, int(bool(downsample))))

    def map_evict_box(self, lo, hi, keep_inside=True) -> int:
        lo = np.ascontiguousarray(lo, np.float32)
        hi = np.ascontiguousarray(hi, np.float32)
        n = C.c_size_t(0)
        self._check(self.lib.lv_map_evict_box(self.h, lo.ctypes.data_as(C.c_void_p), hi.ctypes.data_as(C.c_void_p), int(bool(keep_inside)),
                                              C.byref(n)))
        return int(n.value)

    def map_evict_oldest(self, n_oldest: int) -> int:
        n = C.c_size_t(0)
        self._check(self.lib.lv_map_evict_oldest(self.h, C.c_size_t(n_oldest), C.byref(n)))
        return int(n.value)

    def map_relinearise_async(self):
        self._check(self.lib.lv_map_relinearise_async(self.h))

    def map_reserve_rebuild(self):
        """lv_map_reserve_rebuild: the second store of the background rebuild allocated (and touched) now, at set-up time."""
        self._check(self.lib.lv_map_reserve_rebuild(self.h))

    def map_rebuild_status(self, wait=False) -> dict:
        out = (C.c_uint64 * 4)()
        self._check(self.lib.lv_map_rebuild_status(self.h, C.c_int(int(wait)), out))
        return dict(state=int(out[0]), started=int(out[1]), adopted=int(out[2]), journal=int(out[3]))

    def map_relinearise(self):
        self._check(self.lib.lv_map_relinearise(self.h))

    def map_stats(self) -> dict:
        st = MapStats()
        self._check(self.lib.lv_map_get_stats(self.h, C.byref(st)))
        return {k: (list(getattr(st, k)) if k in ("pool_used", "pool_cap", "slots_used", "slots_cap") else int(getattr(st, k)))
                for k, _ in MapStats._fields_}

    def map_size(self) -> int:
        return int(self.lib.lv_map_size(self.h))

    def map_fetch(self) -> np.ndarray:
        n = self.map_size()
        out = np.empty((n, 3), np.float32)
        self._check(self.lib.lv_map_fetch(self.h, out.ctypes.data_as(C.c_void_p), C.c_size_t(n)))
        return out

    # --- Localizator side
    def scan_set(self, pts):
        a, stride, n = _points(pts)
        self._n = n
        self._check(self.lib.lv_scan_set(self.h, a.ctypes.data_as(C.c_void_p), C.c_size_t(stride), C.c_size_t(n)))

    def scan_deskew(self, xyz, times, states, Xt2, downsample_prec=0.5):
        """xyz [N,3] f32, times [N] f64, states / Xt2: numpy records with the lv_motion_state layout (184 B)."""
        n = len(xyz)
        rec = np.zeros(n, dtype=[("x", "f4"), ("y", "f4"), ("z", "f4"), ("pad", "f4"), ("time", "f8"), ("intensity", "f4"), ("range", "f4")])
        xyz = np.asarray(xyz, np.float32)
        rec["x"], rec["y"], rec["z"], rec["time"] = xyz[:, 0], xyz[:, 1], xyz[:, 2], np.asarray(times, np.float64)
        st = np.ascontiguousarray(states)
        x2 = np.ascontiguousarray(Xt2)
        assert st.dtype.itemsize == 184 and x2.dtype.itemsize == 184
        self._check(self.lib.lv_scan_deskew(self.h, rec.ctypes.data_as(C.c_void_p), C.c_size_t(32), C.c_size_t(16), C.c_size_t(n),
                                            st.ctypes.data_as(C.c_void_p), C.c_size_t(len(st)), x2.ctypes.data_as(C.c_void_p),
                                            C.c_float(downsample_prec)))
        self._n = self.scan_size()

    def scan_downsample(self, xyz, downsample_prec=0.5):
        a, stride, n = _points(xyz)
        self._check(self.lib.lv_scan_downsample(self.h, a.ctypes.data_as(C.c_void_p), C.c_size_t(stride), C.c_size_t(n), C.c_float(downsample_prec)))
        self._n = self.scan_size()

    # --- row f-4: LiDAR wire formats
    def cloud_format_preset(self, lidar_type: int) -> CloudFormat:
        f = CloudFormat()
        self._check(self.lib.lv_cloud_format_preset(int(lidar_type), C.byref(f)))
        return f

    def cloud_ingest(self, raw: bytes, n: int, fmt: CloudFormat, prm: IngestParams) -> int:
        buf = (C.c_char * len(raw)).from_buffer_copy(raw)
        kept = C.c_size_t(0)
        self._check(self.lib.lv_cloud_ingest(self.h, buf, C.c_size_t(n), C.byref(fmt), C.byref(prm), C.byref(kept)))
        return int(kept.value)

    def cloud_size(self) -> int:
        return int(self.lib.lv_cloud_size(self.h))

    def cloud_fetch(self, t1: float, t2: float) -> np.ndarray:
        cap = max(self.cloud_size(), 1)
        out = np.zeros(cap, POINT_DTYPE)
        n = C.c_size_t(0)
        self._check(self.lib.lv_cloud_fetch(self.h, C.c_double(t1), C.c_double(t2), out.ctypes.data_as(C.c_void_p), C.c_size_t(cap), C.byref(n)))
        return out[: n.value].copy()

    def cloud_reserve(self, max_points_per_message: int, point_step: int, buffer_points: int):
        self._check(self.lib.lv_cloud_reserve(self.h, C.c_size_t(max_points_per_message), C.c_size_t(point_step), C.c_size_t(buffer_points)))

    def reserve_stream(self, max_window_points: int, max_scan_points: int):
        self._check(self.lib.lv_reserve_stream(self.h, C.c_size_t(max_window_points), C.c_size_t(max_scan_points)))

    def cloud_clear(self, t: float):
        self._check(self.lib.lv_cloud_clear(self.h, C.c_double(t)))

    def scan_deskew_window(self, t1, t2, states, Xt2, downsample_prec=0.5) -> int:
        st = np.ascontiguousarray(states)
        x2 = np.ascontiguousarray(Xt2)
        assert st.dtype.itemsize == 184 and x2.dtype.itemsize == 184
        nw = C.c_size_t(0)
        self._check(self.lib.lv_scan_deskew_window(self.h, C.c_double(t1), C.c_double(t2), st.ctypes.data_as(C.c_void_p), C.c_size_t(len(st)),
                                                   x2.ctypes.data_as(C.c_void_p), C.c_float(downsample_prec), C.byref(nw)))
        self._n = self.scan_size()
        return int(nw.value)

    def scan_size(self) -> int:
        return int(self.lib.lv_scan_size(self.h))

    def scan_fetch(self) -> np.ndarray:
        n = self.scan_size()
        out = np.empty((n, 3), np.float32)
        self._check(self.lib.lv_scan_fetch(self.h, out.ctypes.data_as(C.c_void_p), C.c_size_t(n)))
        return out

    def iterate(self, state) -> dict:
        s = np.ascontiguousarray(state, np.float64)
        out = Sums()
        self._check(self.lib.lv_iterate(self.h, s.ctypes.data_as(C.c_void_p), C.byref(out)))
        return out.as_dict()

    def update(self, state, P, want_trace=True):
        if not want_trace:
            self._xb[:] = state
            self._Pb[:] = np.asarray(P).reshape(NS, NS)
            rc = self.lib.lv_update(self.h, self._xp, self._Pp, self._passes_ref, None, None)
            if rc != LV_OK:
                self._check(rc)
            return self._xb.copy(), self._Pb.copy(), self._passes.value, None, []
        x = np.ascontiguousarray(state, np.float64).copy()
        Pm = np.ascontiguousarray(P, np.float64).copy().reshape(NS, NS)
        npass = self.params.MAX_NUM_ITERS + 1
        passes = C.c_int(0)
        sums = (Sums * npass)()
        trace = np.zeros((npass, 49))
        self._check(self.lib.lv_update(self.h, x.ctypes.data_as(C.c_void_p), Pm.ctypes.data_as(C.c_void_p), C.byref(passes),
                                       sums if want_trace else None,
                                       trace.ctypes.data_as(C.c_void_p) if want_trace else None))
        n = passes.value
        return x, Pm, n, trace[:n], [sums[i].as_dict() for i in range(n)] if want_trace else []

    # --- resident filter (row f-3)
    def filter_set(self, state, P):
        x = np.ascontiguousarray(state, np.float64)
        Pm = np.ascontiguousarray(P, np.float64)
        self._check(self.lib.lv_filter_set(self.h, x.ctypes.data_as(C.c_void_p), Pm.ctypes.data_as(C.c_void_p)))

    def filter_get(self):
        x = np.zeros(26)
        Pm = np.zeros((NS, NS))
        self._check(self.lib.lv_filter_get(self.h, x.ctypes.data_as(C.c_void_p), Pm.ctypes.data_as(C.c_void_p)))
        return x, Pm

    def predict(self, dt, Q, acc, gyro):
        Qm = np.ascontiguousarray(Q, np.float64)
        a = np.ascontiguousarray(acc, np.float64)
        g = np.ascontiguousarray(gyro, np.float64)
        self._check(self.lib.lv_predict(self.h, C.c_double(dt), Qm.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p),
                                        g.ctypes.data_as(C.c_void_p)))

    def correct(self, want_passes=True) -> int:
        p = C.c_int(0)
        self._check(self.lib.lv_correct(self.h, C.byref(p) if want_passes else None))
        return p.value

    def degeneracy_values(self) -> np.ndarray:
        """[passes, 6] eigenvalues of the pose block of H^T H of the last update (degeneracy_mode >= 1)."""
        eig = np.zeros((16, 6))
        n = C.c_int(0)
        self._check(self.lib.lv_get_degeneracy_values(self.h, eig.ctypes.data_as(C.c_void_p), 16, C.byref(n)))
        return eig[: n.value].copy()

    def update_begin(self, state, P):
        x = np.ascontiguousarray(state, np.float64)
        Pm = np.ascontiguousarray(P, np.float64)
        self._check(self.lib.lv_update_begin(self.h, x.ctypes.data_as(C.c_void_p), Pm.ctypes.data_as(C.c_void_p)))

    def pass_reduce(self):
        self._check(self.lib.lv_pass_reduce(self.h))

    def pass_solve(self):
        self._check(self.lib.lv_pass_solve(self.h))

    def sums_device_ptr(self) -> int:
        return int(self.lib.lv_sums_device_ptr(self.h))

    # --- collective inside the library (row e)
    def comm_unique_id(self, rccl_library: str | None = None) -> bytes:
        buf = C.create_string_buffer(128)
        lib = rccl_library.encode() if rccl_library else None
        self._check(self.lib.lv_comm_unique_id(lib, buf))
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world: int, rccl_library: str | None = None):
        if len(unique_id) != 128:
            raise ValueError("unique id must be 128 bytes")
        lib = rccl_library.encode() if rccl_library else None
        self._check(self.lib.lv_comm_init(self.h, lib, C.c_char_p(unique_id), int(rank), int(world)))

    def comm_destroy(self):
        self._check(self.lib.lv_comm_destroy(self.h))

    def comm_world(self) -> int:
        return int(self.lib.lv_comm_world(self.h))

    def set_sums_buffer(self, device_ptr: int | None):
        self._check(self.lib.lv_set_sums_buffer(self.h, C.c_void_p(device_ptr or 0)))

    def update_end(self):
        x = np.zeros(26)
        Pm = np.zeros((NS, NS))
        passes = C.c_int(0)
        self._check(self.lib.lv_update_end(self.h, x.ctypes.data_as(C.c_void_p), Pm.ctypes.data_as(C.c_void_p), C.byref(passes)))
        return x, Pm, passes.value

    def set_stream(self, stream_handle: int | None):
        self._check(self.lib.lv_set_stream(self.h, C.c_void_p(stream_handle or 0)))

    def get_stream(self) -> int:
        return int(self.lib.lv_get_stream(self.h) or 0)

    def synchronize(self):
        self._check(self.lib.lv_synchronize(self.h))

    def set_capture(self, on: bool):
        self._check(self.lib.lv_set_capture(self.h, int(on)))

    def set_profiling(self, mode):
        self._check(self.lib.lv_set_profiling(self.h, int(mode)))

    def level_histogram(self) -> list:
        out = (C.c_int * 8)()
        self._check(self.lib.lv_get_level_histogram(self.h, out))
        return list(out)

    def solve_clocks(self) -> np.ndarray:
        out = np.zeros((16, 16), np.int64)
        self._check(self.lib.lv_get_solve_clocks(self.h, out.ctypes.data_as(C.c_void_p), 256))
        return out

    def phase_clocks(self, capacity=4096) -> np.ndarray:
        out = np.zeros((capacity, 8), np.int64)
        nb = C.c_int(0)
        self._check(self.lib.lv_get_phase_clocks(self.h, out.ctypes.data_as(C.c_void_p), capacity, C.byref(nb)))
        return out[:nb.value]

    def timing(self) -> dict:
        t = Timing()
        self._check(self.lib.lv_get_timing(self.h, C.byref(t)))
        out = {k: getattr(t, k) for k, _ in Timing._fields_}
        out["pass_match_ms"] = list(t.pass_match_ms)
        out["pass_solve_ms"] = list(t.pass_solve_ms)
        out["pass_collective_ms"] = list(t.pass_collective_ms)
        return out

    # --- fetches
    def fetch_knn(self):
        n = self._n
        k = self.params.NUM_MATCH_POINTS
        idx = np.empty((n, k), np.uint32)
        d2 = np.empty((n, k), np.float32)
        self._check(self.lib.lv_fetch_knn(self.h, idx.ctypes.data_as(C.c_void_p), d2.ctypes.data_as(C.c_void_p)))
        return idx, d2

    def set_record_dump(self, on=True):
        """pass_kernel (one launch per pass) also stores its hand-over records to memory (for fetch_neighbors)."""
        self._check(self.lib.lv_set_record_dump(self.h, int(on)))

    def comm_set_shard_max(self, n_max: int):
        """The largest shard of the current scan over the ranks (same value on every rank, after every scan_set)."""
        self.lib.lv_comm_set_shard_max.argtypes = [C.c_void_p, C.c_size_t]
        self._check(self.lib.lv_comm_set_shard_max(self.h, int(n_max)))

    def set_comm_fused(self, on=True):
        self._check(self.lib.lv_set_comm_fused(self.h, int(on)))

    GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int)

    def comm_set_host_gather(self, rank: int, world: int, fn):
        """The one-launch-per-pass multi-rank form with the caller's transport (lv_comm_set_host_gather).
        fn(slots: float64 array of world * n, n, rank, world) fills every other rank's slots[r * n:(r + 1) * n] in place
        (this rank's partials are already at [rank * n:(rank + 1) * n]); None removes the transport."""
        self.lib.lv_comm_set_host_gather.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        if fn is None:
            self._check(self.lib.lv_comm_set_host_gather(self.h, 0, 1, None, None))
            self._gather_cb = None
            return

        def tramp(_user, slots, bytes_per_rank, r, w):
            try:
                n = bytes_per_rank // 8
                arr = np.ctypeslib.as_array((C.c_double * (n * w)).from_address(slots))
                fn(arr, n, r, w)
                return 0
            except Exception as e:  # noqa: BLE001 — an exception must not unwind through the C frames
                print(f"[limo_velo_amd] host gather callback failed: {e!r}", flush=True)
                return 1

        cb = self.GATHER_FN(tramp)
        self._check(self.lib.lv_comm_set_host_gather(self.h, int(rank), int(world), C.cast(cb, C.c_void_p), None))
        self._gather_cb = cb   # (keeps the trampoline alive as long as the library may call it)

    def comm_peer_export(self) -> bytes:
        buf = (C.c_ubyte * PEER_HANDLE_BYTES)()
        self._check(self.lib.lv_comm_peer_export(self.h, buf))
        return bytes(buf)

    def comm_peer_init(self, rank: int, world: int, handles):
        blob = b"".join(handles)
        assert len(blob) == PEER_HANDLE_BYTES * world
        self._check(self.lib.lv_comm_peer_init(self.h, int(rank), int(world), blob))

    def set_fused_pass(self, on=True):
        self._check(self.lib.lv_set_fused_pass(self.h, int(on)))

    def set_option(self, name: str, value: int):
        self._check(self.lib.lv_set_option(self.h, name.encode(), int(value)))

    def pass_clocks(self, slots=None):
        """[launch][workgroup slot][32] stamps of the last update's pass_kernel launches, and the number of search
        workgroups n (slot n - 1 of a launch = its bookkeeping workgroup, stamp 10 = books done).  The slot count
        (the library's stride: its workgroup limit + 1) is asked from the library."""
        if slots is None:
            q = C.c_int(0)
            self._check(self.lib.lv_get_pass_clocks(self.h, None, 0, C.byref(q)))
            slots = q.value
        nl = self.params.MAX_NUM_ITERS + 2
        buf = np.zeros((nl, slots, 32), np.int64)
        n = C.c_int(0)
        self._check(self.lib.lv_get_pass_clocks(self.h, buf.ctypes.data_as(C.c_void_p), slots, C.byref(n)))
        return buf, n.value

    def last_update_fused(self) -> bool:
        return bool(self.lib.lv_last_update_fused(self.h))

    def last_passes(self) -> int:
        """Passes of the last update whose results have been fetched (lv_update, or lv_correct + lv_filter_get)."""
        return int(self.lib.lv_last_passes(self.h))

    def fetch_neighbors(self):
        """Neighbour coordinates / squared distances / world points / found counts out of the hand-over records of
        the most recent pass (works for the non-capturing, timed kernels)."""
        n = self._n
        k = self.params.NUM_MATCH_POINTS
        nbr = np.empty((n, k, 3), np.float32)
        d2 = np.empty((n, k), np.float32)
        pw = np.empty((n, 3), np.float32)
        found = np.empty(n, np.int32)
        self._check(self.lib.lv_fetch_neighbors(self.h, nbr.ctypes.data_as(C.c_void_p), d2.ctypes.data_as(C.c_void_p),
                                                pw.ctypes.data_as(C.c_void_p), found.ctypes.data_as(C.c_void_p)))
        return nbr, d2, pw, found

    def fetch_matches(self):
        n = self._n
        valid = np.empty(n, np.uint8)
        pw = np.empty((n, 3), np.float32)
        abcd = np.empty((n, 4), np.float32)
        dist = np.empty(n, np.float32)
        self._check(self.lib.lv_fetch_matches(self.h, valid.ctypes.data_as(C.c_void_p), pw.ctypes.data_as(C.c_void_p),
                                              abcd.ctypes.data_as(C.c_void_p), dist.ctypes.data_as(C.c_void_p)))
        return valid, pw, abcd, dist

    def calculate_H(self, state, p_world, abcd, dist):
        s = np.ascontiguousarray(state, np.float64)
        pw = np.ascontiguousarray(p_world, np.float32)
        ab = np.ascontiguousarray(abcd, np.float32)
        di = np.ascontiguousarray(dist, np.float32)
        n = len(di)
        H = np.zeros((n, 12))
        h = np.zeros(n)
        self._check(self.lib.lv_calculate_H(self.h, s.ctypes.data_as(C.c_void_p), pw.ctypes.data_as(C.c_void_p),
                                            ab.ctypes.data_as(C.c_void_p), di.ctypes.data_as(C.c_void_p), C.c_size_t(n),
                                            H.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p)))
        return H, h

    def fetch_rows(self):
        n = self._n
        H = np.empty((n, 12), np.float64)
        h = np.empty(n, np.float64)
        self._check(self.lib.lv_fetch_rows(self.h, H.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p)))
        return H, h

"""Synthetic sensor_msgs/PointCloud2 payloads in the point layouts of the LiDAR drivers the reference supports
(include/Headers/Common.hpp:109-221) — shared by the CPU and GPU tests of row f-4."""
import numpy as np


def _fields(kind, wire):
    """numpy structured dtype (explicit offsets) of one point record; wire=True: a packed on-the-wire layout with
    unaligned fields as real drivers publish it, wire=False: the PCL in-memory layout (lv_cloud_format_preset)."""
    if kind == "velodyne":
        if wire:   # x y z intensity ring(u16) time(f32): 22 bytes
            return np.dtype({"names": ["x", "y", "z", "intensity", "ring", "time"], "formats": ["<f4", "<f4", "<f4", "<f4", "<u2", "<f4"],
                             "offsets": [0, 4, 8, 12, 16, 18], "itemsize": 22})
        return np.dtype({"names": ["x", "y", "z", "intensity", "time", "ring"], "formats": ["<f4", "<f4", "<f4", "<f4", "<f4", "<u2"],
                         "offsets": [0, 4, 8, 16, 20, 24], "itemsize": 32})
    if kind == "hesai":
        if wire:   # x y z intensity(u8) timestamp(f64) ring(u16): 23 bytes
            return np.dtype({"names": ["x", "y", "z", "intensity", "timestamp", "ring"], "formats": ["<f4", "<f4", "<f4", "u1", "<f8", "<u2"],
                             "offsets": [0, 4, 8, 12, 13, 21], "itemsize": 23})
        return np.dtype({"names": ["x", "y", "z", "intensity", "timestamp", "ring"], "formats": ["<f4", "<f4", "<f4", "u1", "<f8", "<u2"],
                         "offsets": [0, 4, 8, 16, 24, 32], "itemsize": 48})
    if kind == "ouster":
        return np.dtype({"names": ["x", "y", "z", "intensity", "t", "reflectivity", "ring", "range"],
                         "formats": ["<f4", "<f4", "<f4", "<f4", "<u4", "<u2", "u1", "<u4"],
                         "offsets": [0, 4, 8, 16, 20, 24, 26, 28], "itemsize": 32})
    if kind == "custom":
        return np.dtype({"names": ["x", "y", "z", "intensity", "range", "timestamp", "ring"],
                         "formats": ["<f4", "<f4", "<f4", "<f4", "<f4", "<f8", "<u2"], "offsets": [0, 4, 8, 20, 24, 32, 40], "itemsize": 48})
    raise ValueError(kind)


def make_message(kind, n, seed=1, wire=False, stamp_sec=1_700_000_000.25, sweep=0.1):
    """Returns (raw bytes, format dict, header_stamp_usec).  Points on a ring pattern 1..60 m from the sensor, stamps
    increasing over one sweep with a sprinkling of duplicates and small inversions (so the time sort has work)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    dt = _fields(kind, wire)
    rec = np.zeros(n, dt)
    r = rng.uniform(1.0, 60.0, n)
    az = np.linspace(0, 2 * np.pi, n, endpoint=False) + rng.uniform(-0.01, 0.01, n)
    el = rng.uniform(-0.26, 0.26, n)
    rec["x"], rec["y"], rec["z"] = (r * np.cos(el) * np.cos(az)).astype(np.float32), (r * np.cos(el) * np.sin(az)).astype(np.float32), (r * np.sin(el)).astype(np.float32)
    frac = np.arange(n) / max(n - 1, 1)
    frac = np.round(frac * 4096) / 4096          # duplicates
    jitter = rng.integers(0, 3, n) * (1.0 / 8192)  # local inversions
    rel = (frac + jitter) * sweep
    fmt = dict(point_step=dt.itemsize, off_x=0, off_y=4, off_z=8, off_range=0, range_type=0)
    if kind == "velodyne":
        rec["time"] = (rel - sweep).astype(np.float32)   # relative to the end of the sweep (params.yaml:31 "usual")
        rec["intensity"] = rng.uniform(0, 255, n).astype(np.float32)
        fmt.update(off_time=dt.fields["time"][1], time_type=0, off_intensity=dt.fields["intensity"][1], intensity_type=1, relative_time=1)
    elif kind == "hesai":
        rec["timestamp"] = stamp_sec + rel
        rec["intensity"] = rng.integers(0, 256, n).astype(np.uint8)
        fmt.update(off_time=dt.fields["timestamp"][1], time_type=1, off_intensity=dt.fields["intensity"][1], intensity_type=2, relative_time=0)
    elif kind == "ouster":
        rec["t"] = np.round(rel * 1e9).astype(np.uint32)
        rec["reflectivity"] = rng.integers(0, 65536, n).astype(np.uint16)
        rec["range"] = np.round(r * 1000).astype(np.uint32)
        rec["intensity"] = rng.uniform(0, 1000, n).astype(np.float32)
        fmt.update(off_time=20, time_type=2, off_intensity=24, intensity_type=3, off_range=28, range_type=4, relative_time=1)
    else:
        rec["timestamp"] = stamp_sec + rel
        rec["intensity"] = rng.uniform(0, 255, n).astype(np.float32)
        fmt.update(off_time=32, time_type=1, off_intensity=20, intensity_type=1, relative_time=0)
    return rec.tobytes(), fmt, int(round(stamp_sec * 1e6))

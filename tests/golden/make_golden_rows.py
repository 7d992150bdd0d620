"""Generates tests/golden/rows_kat.npz — known-answer vectors of the "next" rows of SURVEY §8 from the CPU oracle:
f-1 map insert with ikd-Tree down-sampling, f-2 de-skew + voxel grid, f-3 one IMU prediction, f-4 PointCloud2 ingest.
Inputs are regenerated deterministically by the tests (tests/cloud_messages.py, limo_velo_amd.synth), only digests and
small outputs are stored.  PARITY UNPINNED (see make_golden.py): these pin the oracle, they are not reference outputs.

    python tests/golden/make_golden_rows.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import lvamd  # noqa: E402

lvamd.load()
from limo_velo_amd import synth  # noqa: E402

import cloud_messages as cm  # noqa: E402
import lvoracle as lo  # noqa: E402


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def rows_inputs():
    """Everything the GPU test needs to rebuild the same inputs."""
    sc = synth.make_scene(50_000, 2_000)
    raw, f, stamp = cm.make_message("velodyne", 20_000, seed=7, wire=True, stamp_sec=100.35)
    fmt = (f["point_step"], f["off_x"], f["off_y"], f["off_z"], f["off_time"], f["time_type"], f["off_intensity"], f["intensity_type"],
           f["off_range"], f["range_type"], f["relative_time"])
    prm = (stamp, 0, 0, 0.1, 2, 4.0)
    return sc, raw, fmt, prm


def deskew_path(lo_mod, t1):
    s = lo_mod.motion_state(pos=(1.0, 2.0, 0.5), vel=(4.0, 0.5, 0.0), a=(0.3, -0.2, 9.9), w=(0.02, -0.01, 0.4), time=t1 - 0.004)
    states = [s.copy()]
    for k in range(1, 14):
        s = lo_mod.state_integrate(s, (0.3, -0.2 + 0.01 * k, 9.9), (0.02, -0.01, 0.4 - 0.01 * k), t1 - 0.004 + 0.01 * k)
        states.append(s.copy())
    return np.concatenate(states)


def main():
    sc, raw, fmt, prm = rows_inputs()
    # f-4
    pts = lo.cloud_ingest(raw, 20_000, lo.CloudFormat(*fmt), lo.IngestParams(*prm))
    # f-2 on the ingested sweep
    t1, t2 = pts["time"][0] + 0.02, pts["time"][-1] - 0.01
    sel = pts[(pts["time"] >= t1) & (pts["time"] <= t2)]
    states = deskew_path(lo, t1)
    xyz = np.stack([sel["x"], sel["y"], sel["z"]], axis=1)
    desk = lo.deskew(xyz, sel["time"], states, states[-2:-1])
    ds = lo.voxelgrid(desk, 0.5)
    # f-1
    new_pts = (sc["map_xyz"][:3000] + np.float32(0.013)).astype(np.float32)
    merged = lo.map_add(sc["map_xyz"], new_pts, downsample=True)
    out = dict(
        ingest_sha256=np.array(digest(pts)), ingest_n=np.array(len(pts)), ingest_first=pts[:16].copy().view(np.uint8), ingest_last=pts[-16:].copy().view(np.uint8),
        t1=np.array(t1), t2=np.array(t2), window_n=np.array(len(sel)),
        deskew_sha256=np.array(digest(desk)), voxelgrid_sha256=np.array(digest(ds)), voxelgrid_n=np.array(len(ds)), voxelgrid_first=ds[:32],
        map_add_sha256=np.array(digest(merged)), map_add_n=np.array(len(merged)), map_add_tail=merged[-32:],
    )
    path = os.path.join(HERE, "rows_kat.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

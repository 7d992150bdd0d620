"""The reference-style C++ API (limo-velo_amd/host: Mapper / Localizator over the C-ABI) driven by a
C++ program the way src/main.cpp drives the reference, checked against the oracle."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_shim_demo_matches_oracle(oracle, scene_small, tmp_path):
    exe = os.path.join(ROOT, "limo-velo_amd", "host", "shim_demo")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(exe)])
    sc = scene_small
    scan = sc["scan_xyz"][:1500]
    sc["map_xyz"].tofile(tmp_path / "map.f32")
    scan.tofile(tmp_path / "scan.f32")
    sc["x_init"].tofile(tmp_path / "x.f64")
    sc["P0"].tofile(tmp_path / "P.f64")
    out = tmp_path / "out.bin"
    r = subprocess.run([exe, str(tmp_path / "map.f32"), str(tmp_path / "scan.f32"), str(tmp_path / "x.f64"),
                        str(tmp_path / "P.f64"), str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = np.fromfile(out, dtype=np.uint8)
    hdr = raw[:32].view(np.float64)
    n_map, n_match, passes, t = int(hdr[0]), int(hdr[1]), int(hdr[2]), hdr[3]
    off = 32
    x = raw[off:off + 26 * 8].view(np.float64); off += 26 * 8
    P = raw[off:off + 529 * 8].view(np.float64).reshape(23, 23); off += 529 * 8
    H = raw[off:off + n_match * 12 * 8].view(np.float64).reshape(n_match, 12); off += n_match * 12 * 8
    h = raw[off:off + n_match * 8].view(np.float64); off += n_match * 8
    rec = raw[off:off + n_match * 32].view(np.float32).reshape(n_match, 8)

    tree = oracle.KdTree(sc["map_xyz"])
    o = oracle.iterate(sc["x_init"], sc["map_xyz"], scan, tree=tree)
    sel = o["valid"].astype(bool)
    assert n_map == 50_000 and n_match == int(sel.sum()) and t == 0.1
    pw = oracle.transform_scan(sc["x_init"], scan)
    assert np.array_equal(rec[:, :3].view(np.uint32), pw[sel].view(np.uint32))          # Match::point
    assert np.array_equal(rec[:, 3:7].view(np.uint32), o["abcd"][sel].view(np.uint32))  # Match::plane.n
    assert np.array_equal(rec[:, 7].view(np.uint32), o["dist"][sel].view(np.uint32))    # Match::distance
    assert np.array_equal(H, o["Hrows"][sel]) and np.array_equal(h, o["h"][sel])        # calculate_H
    xo, Po, po, _, _ = oracle.update(sc["x_init"], sc["P0"], sc["map_xyz"], scan, tree=tree)
    assert passes == po
    assert np.abs(x - xo).max() < 1e-9 and np.abs(P - Po).max() < 1e-10


def _write_stream_input(path, on_device, delta, stream, n_revs, x0):
    """The binary input of limo-velo_amd/host/stream_demo.cpp."""
    import struct

    from limo_velo_amd import synth

    import test_gpu_stream as T

    with open(path, "wb") as f:
        f.write(struct.pack("<IId", 0x5453564C, int(on_device), delta))
        f.write(struct.pack("<I", len(stream["map_xyz"])))
        f.write(np.ascontiguousarray(stream["map_xyz"], np.float32).tobytes())
        t_imu = np.arange(0, int(round(n_revs * 0.1 / 0.01)) + 12) * 0.01
        f.write(struct.pack("<I", len(t_imu)))
        for t in t_imu:
            a, w = synth.stream_imu(float(t))
            q = synth.stream_truth(float(t))[4]
            f.write(struct.pack("<d", float(t)) + np.asarray(a, np.float32).tobytes() + np.asarray(w, np.float32).tobytes()
                    + np.asarray(q, np.float32).tobytes())
        f.write(struct.pack("<I", n_revs))
        for r in range(n_revs):
            raw, n, fmt, stamp = T.hesai_message(stream["revs"][r])
            assert fmt["point_step"] == 48
            f.write(struct.pack("<dQI", stream["revs"][r]["stamp"], stamp, n))
            f.write(raw)
    np.ascontiguousarray(x0, np.float64).tofile(str(path) + ".x0")


def _read_stream_output(path):
    raw = np.fromfile(path, dtype=np.uint8)
    n = int(raw[:4].view(np.uint32)[0])
    rec = raw[4:4 + n * (8 + 208 + 4)].reshape(n, 220)
    t = rec[:, :8].copy().view(np.float64)[:, 0]
    x = rec[:, 8:216].copy().view(np.float64).reshape(n, 26)
    npts = rec[:, 216:220].copy().view(np.uint32)[:, 0]
    return t, x, npts


def test_reference_main_loop_through_the_shim(lv, tmp_path):
    """src/main.cpp:52-128 (main_loop.hpp: the reference's own lines minus the ROS publishers) run by a C++ program over
    the shim's Accumulator / Compensator / Localizator / Mapper on a 100 Hz stream: once with the reference's by-value
    hand-overs (compensate -> downsample -> correct -> Xt2 * Xt2.I_Rt_L() * ds -> map.add), once with the three calls
    that keep the scan on the device.  Both track the ground truth and agree with each other."""
    from limo_velo_amd import synth

    host = os.path.join(ROOT, "limo-velo_amd", "host")
    exe = os.path.join(host, "stream_demo")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", host])
    n_revs, delta = 9, 0.01
    stream = synth.make_stream(1_048_576, n_revs, n_az=512, map_radius=62.0)
    t_init = 0.30 - 0.1        # Accumulator::ready at the 31st IMU sample; initial_time = its stamp - real_time_delay
    pos0, _, vel0, _, q0 = synth.stream_truth(t_init)
    x0 = synth.make_state(pos0 + [0.02, -0.015, 0.01], synth.quat_mul(q0, synth.quat_from_rotvec([0.002, -0.001, 0.003])), vel=vel0,
                          grav=(0, 0, synth.STREAM_G))
    res = {}
    for on_device in (0, 1):
        inp, out = tmp_path / f"in{on_device}.bin", tmp_path / f"out{on_device}.bin"
        _write_stream_input(inp, on_device, delta, stream, n_revs, x0)
        r = subprocess.run([exe, str(inp), str(out)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        if os.environ.get("LV_DEMO_VERBOSE"):
            print(r.stderr)
        res[on_device] = _read_stream_output(out)
    t0, xa, na = res[0]
    t1, xb, nb = res[1]
    assert len(t0) >= 50 and np.array_equal(t0, t1) and np.array_equal(na, nb)
    assert np.allclose(np.diff(t0), delta, atol=1e-9)                      # one localisation per 10 ms field of view
    truth = np.array([synth.stream_truth(t)[0] for t in t0])
    for x in (xa, xb):
        err = np.linalg.norm(x[:, :3] - truth, axis=1)
        assert np.sqrt(np.mean(err ** 2)) < 0.03, err
    # the device-resident hand-overs change nothing beyond rounding: the two free-running runs see world points that differ
    # in the last f32 bit now and then (one ulp = 4e-6 m at 60 m), which 70 mapping updates amplify to the 1e-5 m level in
    # position (the first 20 updates agree to 1e-6); an update that sits on the LIMITS threshold (src/main.cpp:145) may take one pass more in one run, which moves
    # the weakly observable states (biases, gravity) by up to LIMITS = 1e-3 until the filter has pulled them back together
    # (which realisation one gets depends on the summation order of the workgroup partials, for instance)
    assert np.abs(xa[:20] - xb[:20]).max() < 1e-6 and np.abs(xa[:, :3] - xb[:, :3]).max() < 1e-3 and np.abs(xa - xb).max() < 3e-3


def test_reference_main_loop_mapping_offline(lv, tmp_path):
    """The other mapping branch of the reference's loop (src/main.cpp:105-116, `mapping_online: false`): no insert per
    localisation; once per FULL_ROTATION_TIME the whole sweep [t2 - full_rotation_time, t2] is de-skewed to t2, taken to the
    world frame, down-sampled there and added (Mapper::hasToMap).  Both hand-over modes of the localisation step track the
    truth, agree with each other, and the map grows by whole sweeps."""
    import re

    from limo_velo_amd import synth

    host = os.path.join(ROOT, "limo-velo_amd", "host")
    exe = os.path.join(host, "stream_demo")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", host])
    n_revs, delta = 9, 0.01
    stream = synth.make_stream(1_048_576, n_revs, n_az=512, map_radius=62.0)
    t_init = 0.30 - 0.1
    pos0, _, vel0, _, q0 = synth.stream_truth(t_init)
    x0 = synth.make_state(pos0 + [0.02, -0.015, 0.01], synth.quat_mul(q0, synth.quat_from_rotvec([0.002, -0.001, 0.003])), vel=vel0,
                          grav=(0, 0, synth.STREAM_G))
    res, sizes = {}, {}
    env = dict(os.environ, LV_DEMO_MAPPING_OFFLINE="1")
    for on_device in (0, 1):
        inp, out = tmp_path / f"in{on_device}.bin", tmp_path / f"out{on_device}.bin"
        _write_stream_input(inp, on_device, delta, stream, n_revs, x0)
        r = subprocess.run([exe, str(inp), str(out)], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        res[on_device] = _read_stream_output(out)
        sizes[on_device] = int(re.search(r"map (\d+) points", r.stdout).group(1))
    t0, xa, na = res[0]
    t1, xb, nb = res[1]
    assert len(t0) >= 50 and np.array_equal(t0, t1) and np.array_equal(na, nb)
    truth = np.array([synth.stream_truth(t)[0] for t in t0])
    for x in (xa, xb):
        err = np.linalg.norm(x[:, :3] - truth, axis=1)
        assert np.sqrt(np.mean(err ** 2)) < 0.03, err
    # (free-running, and in this mode the map only changes once per sweep: a last-bit difference of a world point lives on
    # until the next insert instead of being averaged out by the next window's: 4.5e-5 measured)
    assert np.abs(xa - xb).max() < 1.5e-4
    n_prior = len(stream["map_xyz"])
    assert sizes[0] > n_prior and sizes[1] > n_prior       # whole sweeps were added ...
    assert abs(sizes[0] - sizes[1]) <= 0.001 * n_prior     # ... the same ones (up to a last-bit voxel flip) in both modes

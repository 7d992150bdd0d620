"""Row f-4 on the GPU: PointCloud2 payloads of the four driver layouts -> lv_cloud_ingest -> the device LiDAR
buffer, compared record for record (bit-exact: xyz, f64 time, intensity, range, order) with the oracle's
restatement of Accumulator::process; window fetch / clear (Accumulator::get_points / clear_lidar); and the de-skew of a
buffered window against lv_scan_deskew on the same points."""
import numpy as np
import pytest

import cloud_messages as cm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi(lv):
    from limo_velo_amd import capi as c

    return c


def _formats(capi, oracle, f):
    args = (f["point_step"], f["off_x"], f["off_y"], f["off_z"], f["off_time"], f["time_type"], f["off_intensity"], f["intensity_type"],
            f["off_range"], f["range_type"], f["relative_time"])
    return capi.CloudFormat(*args), oracle.CloudFormat(*args)


@pytest.mark.parametrize("kind,wire", [("velodyne", False), ("velodyne", True), ("hesai", False), ("hesai", True), ("ouster", False), ("custom", False)])
@pytest.mark.parametrize("rate,stamp_beginning,offset_beginning", [(4, 0, 0), (1, 1, 1), (3, 0, 1)])
def test_ingest_matches_oracle(capi, oracle, lv, kind, wire, rate, stamp_beginning, offset_beginning):
    n = 60_000
    raw, f, stamp = cm.make_message(kind, n, seed=11, wire=wire)
    gf, of = _formats(capi, oracle, f)
    want = oracle.cloud_ingest(raw, n, of, oracle.IngestParams(stamp, stamp_beginning, offset_beginning, 0.1, rate, 4.0))
    with capi.Context() as ctx:
        kept = ctx.cloud_ingest(raw, n, gf, capi.IngestParams(stamp, stamp_beginning, offset_beginning, 0.1, rate, 4.0))
        assert kept == len(want) == ctx.cloud_size()
        got = ctx.cloud_fetch(-1e300, 1e300)
    assert got.tobytes() == want.tobytes()


def test_presets_are_the_pcl_layouts(capi, lv):
    with capi.Context() as ctx:
        for kind, lidar in (("velodyne", capi.LIDAR_VELODYNE), ("hesai", capi.LIDAR_HESAI), ("ouster", capi.LIDAR_OUSTER), ("custom", capi.LIDAR_CUSTOM)):
            _, f, _ = cm.make_message(kind, 8, wire=False)
            p = ctx.cloud_format_preset(lidar)
            for k, v in f.items():
                assert getattr(p, k) == v, (kind, k)


def test_buffer_window_and_clear(capi, oracle, lv):
    """Three consecutive sweeps are pushed; get_points(t1, t2) is the closed time interval, oldest first;
    clear_lidar(t) drops everything with time <= t."""
    msgs, allpts = [], []
    with capi.Context() as ctx:
        for k in range(3):
            raw, f, stamp = cm.make_message("hesai", 20_000, seed=20 + k, stamp_sec=1_700_000_000.0 + 0.1 * k + 0.05, sweep=0.09)
            gf, of = _formats(capi, oracle, f)
            prm = (stamp, 0, 0, 0.1, 2, 4.0)
            ctx.cloud_ingest(raw, 20_000, gf, capi.IngestParams(*prm))
            allpts.append(oracle.cloud_ingest(raw, 20_000, of, oracle.IngestParams(*prm)))
        ref = np.concatenate(allpts)
        assert np.all(np.diff(ref["time"]) >= 0)   # the generator's sweeps do not overlap
        assert ctx.cloud_size() == len(ref)
        t1, t2 = ref["time"][len(ref) // 3], ref["time"][2 * len(ref) // 3]
        win = ctx.cloud_fetch(t1, t2)
        sel = ref[(ref["time"] >= t1) & (ref["time"] <= t2)]
        assert win.tobytes() == sel.tobytes()
        assert len(ctx.cloud_fetch(t2 + 10, t2 + 20)) == 0
        ctx.cloud_clear(t1)
        rest = ref[ref["time"] > t1]
        assert ctx.cloud_size() == len(rest)
        assert ctx.cloud_fetch(-1e300, 1e300).tobytes() == rest.tobytes()
        ctx.cloud_clear(1e300)
        assert ctx.cloud_size() == 0


def test_deskew_window_equals_deskew_of_the_same_points(capi, oracle, lv):
    raw, f, stamp = cm.make_message("velodyne", 40_000, seed=5, wire=True, stamp_sec=100.35)
    gf, of = _formats(capi, oracle, f)
    prm = (stamp, 0, 0, 0.1, 2, 4.0)
    pts = oracle.cloud_ingest(raw, 40_000, of, oracle.IngestParams(*prm))
    t0, t3 = pts["time"][0], pts["time"][-1]
    t1, t2 = t0 + 0.02, t3 - 0.01
    sel = pts[(pts["time"] >= t1) & (pts["time"] <= t2)]
    # a short IMU-upsampled path surrounding the window
    states = []
    s = oracle.motion_state(pos=(1.0, 2.0, 0.5), vel=(4.0, 0.5, 0.0), a=(0.3, -0.2, 9.9), w=(0.02, -0.01, 0.4), time=t1 - 0.004)
    states.append(s.copy())
    for k in range(1, 14):
        s = oracle.state_integrate(s, (0.3, -0.2 + 0.01 * k, 9.9), (0.02, -0.01, 0.4 - 0.01 * k), t1 - 0.004 + 0.01 * k)
        states.append(s.copy())
    states = np.concatenate(states)
    assert states["time"][0] <= sel["time"][0] and sel["time"][-1] <= states["time"][-1]
    xt2 = states[-2:-1].copy()
    xyz = np.stack([sel["x"], sel["y"], sel["z"]], axis=1)
    with capi.Context() as a, capi.Context() as b:
        a.scan_deskew(xyz, sel["time"], states, xt2, downsample_prec=0.5)
        want = a.scan_fetch()
        b.cloud_ingest(raw, 40_000, gf, capi.IngestParams(*prm))
        nw = b.scan_deskew_window(t1, t2, states, xt2, downsample_prec=0.5)
        got = b.scan_fetch()
    assert nw == len(sel)
    assert got.shape == want.shape and got.tobytes() == want.tobytes()
    ds = oracle.voxelgrid(oracle.deskew(xyz, sel["time"], states, xt2), 0.5)
    assert got.tobytes() == ds.tobytes()


@pytest.mark.parametrize("n_msg,leaf,lo_frac,hi_frac", [(6000, 0.5, 0.05, 0.95), (20000, 1.0, 0.1, 0.9), (40000, 2.0, 0.2, 0.9),
                                                        (40000, 0.5, 0.2, 0.9), (40000, 1.0, 0.45, 0.55), (3000, 0.5, 0.0, 1.0)])
def test_large_window_chain_equals_the_general_chain(capi, oracle, lv, scene_small, n_msg, leaf, lo_frac, hi_frac):
    """Windows beyond 2048 raw points take four launches + the library sort (deskew_bounds_kernel straight from the LiDAR
    buffer, leaf keys, sort, window_tail_kernel: heads, scan, centroids, Morton order, tile order by one workgroup, the counts
    back as a note) instead of the thirteen-launch chain; beyond 4096 points OUT the tail declines and the general chain runs.
    Same points in the same order, the same Morton / tile order behind them: the update that consumes the scan is bit-identical."""
    sc = scene_small
    raw, f, stamp = cm.make_message("velodyne", n_msg, seed=11, wire=True, stamp_sec=50.2)
    gf, of = _formats(capi, oracle, f)
    prm = (stamp, 0, 0, 0.1, 2, 4.0)
    pts = oracle.cloud_ingest(raw, n_msg, of, oracle.IngestParams(*prm))
    t0, t3 = pts["time"][0], pts["time"][-1]
    t1, t2 = t0 + lo_frac * (t3 - t0), t0 + hi_frac * (t3 - t0)
    sel = pts[(pts["time"] >= t1) & (pts["time"] <= t2)]
    s = oracle.motion_state(pos=(0.5, -0.3, 0.2), vel=(3.0, 0.4, 0.0), a=(0.2, -0.1, 9.8), w=(0.01, -0.02, 0.3), time=t1 - 0.003)
    states = [s.copy()]
    k = 0
    while states[-1]["time"][0] < t2 + 0.002:
        k += 1
        s = oracle.state_integrate(s, (0.2, -0.1 + 0.01 * k, 9.8), (0.01, -0.02, 0.3 - 0.01 * k), t1 - 0.003 + 0.01 * k)
        states.append(s.copy())
    states = np.concatenate(states)
    xt2 = states[-2:-1].copy()
    res = {}
    with capi.Context() as ctx:
        ctx.map_build(sc["map_xyz"])
        ctx.cloud_ingest(raw, n_msg, gf, capi.IngestParams(*prm))
        for on in (1, 0):
            ctx.set_option("large_window", on)
            nw = ctx.scan_deskew_window(t1, t2, states, xt2, downsample_prec=leaf)
            got = ctx.scan_fetch()
            x, P, passes, tr, sums = ctx.update(sc["x_init"], sc["P0"])
            res[on] = (nw, got, x, P, passes, [s_["n_valid"] for s_ in sums])
    assert res[1][0] == res[0][0] == len(sel)
    assert res[1][1].shape == res[0][1].shape and res[1][1].tobytes() == res[0][1].tobytes()
    assert res[1][4] == res[0][4] and res[1][5] == res[0][5]
    assert np.array_equal(res[1][2], res[0][2]) and np.array_equal(res[1][3], res[0][3])
    xyz = np.stack([sel["x"], sel["y"], sel["z"]], axis=1)
    ds = oracle.voxelgrid(oracle.deskew(xyz, sel["time"], states, xt2), leaf)
    assert res[1][1].tobytes() == ds.tobytes()

"""Row f-2: Compensator::compensate (de-skew) + voxel-grid down-sampling on the device (lv_scan_deskew) against
the oracle's restatement of src/Modules/Compensator.cpp:123-163 / State::propagate_f / pcl::VoxelGrid."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _motion(oracle, n_pts, seed=3):
    rng = np.random.default_rng(seed)
    s0 = oracle.motion_state(pos=(3, -2, 1.5), vel=(12.0, 1.0, -0.2), a=(0.8, -0.3, -9.6), w=(0.05, -0.1, 1.4), time=10.0,
                             tLI=(-0.17, 0.0, -0.04), bw=(0.01, 0.0, -0.02), ba=(0.05, -0.02, 0.01))
    states = [s0]
    for k in range(1, 6):  # IMU-upsampled states every 20 ms (Compensator::upsample)
        prev = states[-1]
        a = prev["a"][0] + rng.normal(scale=0.3, size=3).astype(np.float32)
        w = prev["w"][0] + rng.normal(scale=0.05, size=3).astype(np.float32)
        states.append(oracle.state_integrate(prev, a, w, 10.0 + 0.02 * k))
    states = np.concatenate(states)
    times = np.sort(rng.uniform(10.0, 10.1, n_pts))
    times[:3] = [10.0, 10.02, 10.04]  # exactly on state stamps: first containing interval wins
    times = np.sort(times)
    xyz = (rng.uniform(-1, 1, (n_pts, 3)) * [60, 60, 6]).astype(np.float32)
    return xyz, times, states


def test_deskew_bit_exact(lv, oracle):
    from limo_velo_amd import capi

    xyz, times, states = _motion(oracle, 20000)
    xt2 = states[-1:]
    ref = oracle.deskew(xyz, times, states, xt2)
    assert np.isfinite(ref).all() and np.abs(ref - xyz).max() > 0.3  # the motion really moves the points
    with capi.Context() as ctx:
        ctx.scan_deskew(xyz, times, states, xt2, downsample_prec=0.0)
        got = ctx.scan_fetch()
    assert np.array_equal(_bits(got), _bits(ref))


def test_deskew_voxelgrid_and_scan_feed(lv, oracle, scene_small):
    from limo_velo_amd import capi

    sc = scene_small
    xyz, times, states = _motion(oracle, 30000)
    xt2 = oracle.state_integrate(states[3:4], states[3]["a"], states[3]["w"], 10.07)  # Compensator::get_t2
    ref = oracle.voxelgrid(oracle.deskew(xyz, times, states, xt2), 0.5)
    assert 1000 < len(ref) < 30000
    with capi.Context() as ctx:
        ctx.map_build(sc["map_xyz"])
        ctx.scan_deskew(xyz, times, states, xt2, downsample_prec=0.5)
        got = ctx.scan_fetch()
        assert got.shape == ref.shape
        assert np.array_equal(_bits(got), _bits(ref))
        a = ctx.iterate(sc["x_init"])        # the de-skewed scan is the current scan ...
        ia, da = ctx.fetch_knn()
        ctx.scan_set(ref)                     # ... exactly as if it had been uploaded
        b = ctx.iterate(sc["x_init"])
        ib, db = ctx.fetch_knn()
        assert np.array_equal(ia, ib) and np.array_equal(_bits(da), _bits(db))
        assert a["n_valid"] == b["n_valid"] and np.array_equal(a["HTH"], b["HTH"])


def test_points_outside_the_state_window_are_dropped(lv, oracle):
    from limo_velo_amd import capi

    xyz, times, states = _motion(oracle, 2000)
    times[-50:] += 1.0  # beyond the last state
    xt2 = states[-1:]
    ref = oracle.voxelgrid(oracle.deskew(xyz[:-50], times[:-50], states, xt2), 0.5)
    with capi.Context() as ctx:
        ctx.scan_deskew(xyz, times, states, xt2, downsample_prec=0.5)
        got = ctx.scan_fetch()
    assert np.array_equal(_bits(got), _bits(ref))


def test_downsample_alone_is_the_voxel_grid(lv, oracle, scene_small):
    """lv_scan_downsample = Compensator::downsample (Compensator.cpp:104-107,148-163) on already compensated points."""
    from limo_velo_amd import capi

    pts = scene_small["scan_xyz"]
    with capi.Context() as ctx:
        for leaf in (0.5, 0.2):
            ctx.scan_downsample(pts, leaf)
            got = ctx.scan_fetch()
            want = oracle.voxelgrid(pts, leaf)
            assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32)), leaf
        ctx.scan_downsample(pts, 0.0)      # no voxel grid: the points become the scan as they are
        assert np.array_equal(ctx.scan_fetch().view(np.uint32), pts.view(np.uint32))


@pytest.mark.parametrize("n_pts,leaf", [(3, 0.5), (63, 0.5), (700, 0.5), (2048, 0.5), (2048, 0.2), (1500, 0.0), (2049, 0.5)])
def test_small_window_chain_equals_the_general_path(lv, oracle, scene_small, n_pts, leaf):
    """Windows of up to 2048 points run de-skew -> bounds -> leaf keys -> stable sort -> heads -> scan -> centroids -> Morton
    sort -> tile order in ONE launch (window_small_kernel); larger ones (and the knob off) the twelve-launch chain.  Same
    points in the same order, the same Morton order and tile order behind them: the iterated update that consumes the scan is
    bit-identical."""
    from limo_velo_amd import capi

    sc = scene_small
    xyz, times, states = _motion(oracle, n_pts)
    xyz = (xyz * np.float32([0.3, 0.3, 0.5])).astype(np.float32)     # inside the test map
    xt2 = states[-1:]
    res = {}
    with capi.Context() as ctx:
        ctx.map_build(sc["map_xyz"])
        for on in (1, 0):
            ctx.set_option("small_window", on)
            ctx.scan_deskew(xyz, times, states, xt2, downsample_prec=leaf)
            pts = ctx.scan_fetch()
            x, P, passes, tr, sums = ctx.update(sc["x_init"], sc["P0"])
            res[on] = (pts, x, P, passes, [s["n_valid"] for s in sums])
    ref = oracle.deskew(xyz, times, states, xt2)
    if leaf > 0:
        ref = oracle.voxelgrid(ref, leaf)
    assert np.array_equal(_bits(res[1][0]), _bits(ref))
    assert np.array_equal(_bits(res[1][0]), _bits(res[0][0]))
    assert res[1][3] == res[0][3] and res[1][4] == res[0][4]
    assert np.array_equal(res[1][1], res[0][1]) and np.array_equal(res[1][2], res[0][2])

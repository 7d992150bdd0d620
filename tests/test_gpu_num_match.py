"""NUM_MATCH_POINTS other than 5 (the reference reads it at run time: src/main.cpp:146, Mapper.cpp:85-86, Utils.cpp:33-40,
Plane.cpp:37): the general-K build of the three-kernel pass against the oracle — neighbours, planes, rows bit for bit, the
iterated update to the tolerances of test_gpu_parity.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from test_gpu_parity import TOL_STATE, _bits, _compare_pass  # noqa: E402

KS = [3, 4, 6, 7, 8]


@pytest.fixture(scope="module")
def capi(lv):
    from limo_velo_amd import capi as c

    return c


@pytest.mark.parametrize("k", KS)
def test_single_pass_and_update(capi, oracle, scene_small, k):
    sc = scene_small
    tree = oracle.KdTree(sc["map_xyz"])
    prm_o = oracle.default_params(num_match_points=k)
    # strays far from every surface: the MAX_DIST_PLANE gate on the k-th distance and the bounded stop see both sides
    rng = np.random.default_rng(k)
    stray = sc["scan_xyz"][:150] + rng.uniform(-3.0, 3.0, (150, 3)).astype(np.float32)
    scan = np.concatenate([sc["scan_xyz"][:1850], stray])
    with capi.Context(capi.default_params(NUM_MATCH_POINTS=k)) as ctx:
        ctx.map_build(sc["map_xyz"])
        ctx.scan_set(scan)
        g, o = _compare_pass(ctx, oracle, sc["x_init"], sc["map_xyz"], scan, tree, prm_o)
        assert o["knn_idx"].shape == (len(scan), k) and g["n_valid"] > 300
        xo, Po, po, tro, so = oracle.update(sc["x_init"], sc["P0"], sc["map_xyz"], scan, params=prm_o, tree=tree)
        x, P, passes, tr, sums = ctx.update(sc["x_init"], sc["P0"])
        assert not ctx.last_update_fused()      # one launch per pass is built for k = 5
        # the non-capturing kernels' hand-over records (neighbour coordinates, no indices) of the last pass
        nbr, d2, pw, found = ctx.fetch_neighbors()
    assert passes == po
    for i in range(passes):
        assert sums[i]["n_valid"] == so[i]["n_valid"], f"pass {i}"
        assert np.abs(tr[i] - tro[i]).max() < TOL_STATE, f"pass {i}"
    assert np.abs(x - xo).max() < TOL_STATE and np.abs(P - Po).max() < 1e-10
    last = tr[passes - 2][23:49].copy() if passes > 1 else sc["x_init"]     # the state the last pass was linearised at
    ol = oracle.iterate(last, sc["map_xyz"], scan, tree=tree, params=prm_o)
    # the timed launches are BOUNDED: a point whose k-th neighbour is not closer than MAX_DIST_PLANE (dropped at Plane.cpp:40-43)
    # may be reported without neighbours; every other record is the exact answer
    max_d2 = float(capi.default_params().MAX_DIST_PLANE) ** 2
    near = ol["knn_d2"][:, k - 1].astype(np.float64) < max_d2
    assert (~near).sum() > 10 and near.sum() > 1500
    rejected = (found < k) | ~(d2[:, k - 1].astype(np.float64) < max_d2)
    assert rejected[~near].all()
    assert np.array_equal(_bits(d2[near]), _bits(ol["knn_d2"][near]))
    assert np.array_equal(_bits(nbr[near]), _bits(sc["map_xyz"][ol["knn_idx"][near]]))
    assert np.array_equal(found[near], np.full(near.sum(), k))
    assert np.array_equal(_bits(pw), _bits(oracle.transform_scan(last, scan)))


@pytest.mark.parametrize("k", [3, 8])
def test_estimate_extrinsics_and_small_maps(capi, oracle, lv, k):
    from limo_velo_amd import synth

    sc = synth.make_scene(50_000, 1500, extrinsics="xaloc")
    prm_o = oracle.default_params(estimate_extrinsics=1, num_match_points=k)
    with capi.Context(capi.default_params(estimate_extrinsics=1, NUM_MATCH_POINTS=k)) as ctx:
        ctx.map_build(sc["map_xyz"])
        ctx.scan_set(sc["scan_xyz"])
        g, o = _compare_pass(ctx, oracle, sc["x_init"], sc["map_xyz"], sc["scan_xyz"], None, prm_o)
        assert np.abs(o["HTH"][6:, 6:]).max() > 0
        scan = sc["scan_xyz"][:300]
        ctx.scan_set(scan)
        for m in (1, k - 1, k, k + 2):     # fewer than / exactly / just over k map points
            ctx.map_build(sc["map_xyz"][:m])
            gi = ctx.iterate(sc["x_init"])
            idx, d2 = ctx.fetch_knn()
            oi, od, found, _ = oracle.knn_brute(sc["map_xyz"][:m], oracle.transform_scan(sc["x_init"], scan), k=k)
            assert np.array_equal(idx, oi) and np.array_equal(_bits(d2), _bits(od))
            if m < k:
                assert gi["n_valid"] == 0


def test_values_outside_the_range_are_refused(capi):
    for k in (0, 2, 9):
        with pytest.raises(capi.LvError):
            capi.Context(capi.default_params(NUM_MATCH_POINTS=k))
    with pytest.raises(capi.LvError):      # the general build runs eight lanes per point
        capi.Context(capi.default_params(NUM_MATCH_POINTS=4, lanes_per_query=4))


def test_resident_filter_and_map_insert_with_k4(capi, oracle, scene_small):
    """The calls the reference's main loop makes (lv_filter_set / lv_predict / lv_correct / lv_filter_get, lv_map_add_scan) with
    NUM_MATCH_POINTS = 4: the resident route equals the update by value bit for bit, and the posterior matches the oracle's."""
    sc = scene_small
    prm_o = oracle.default_params(num_match_points=4)
    Q = np.diag([1e-4] * 3 + [1e-2] * 3 + [1e-5] * 3 + [1e-4] * 3)
    with capi.Context(capi.default_params(NUM_MATCH_POINTS=4)) as ctx:
        ctx.map_build(sc["map_xyz"])
        ctx.scan_set(sc["scan_xyz"])
        ctx.filter_set(sc["x_init"], sc["P0"])
        ctx.predict(0.005, Q, [0.1, -0.05, 9.81], [0.01, 0.02, -0.01])
        xp, Pp = ctx.filter_get()
        passes = ctx.correct()
        xr, Pr = ctx.filter_get()
        xv, Pv, pv, _, _ = ctx.update(xp, Pp)
        assert passes == pv and np.array_equal(xr, xv) and np.array_equal(Pr, Pv)
        n0 = ctx.map_size()
        ctx.map_add_scan(True)          # the scan, transformed by the posterior, joins the map (Mapper::add, main.cpp:102)
        assert ctx.map_size() > n0
        g = ctx.iterate(xr)             # and the next search sees it
        idx, d2 = ctx.fetch_knn()
        assert idx.shape == (len(sc["scan_xyz"]), 4) and (d2[:, 0] == 0).mean() > 0.02     # (a survivor finds itself)
    xo, Po, po, _, _ = oracle.update(xp, Pp, sc["map_xyz"], sc["scan_xyz"], params=prm_o)
    assert po == passes and np.abs(xr - xo).max() < TOL_STATE and np.abs(Pr - Po).max() < 1e-10

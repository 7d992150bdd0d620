"""GPU path against the committed known-answer vectors (tests/golden/cfg0_kat.npz): no oracle run, no
reference tree needed on the GPU box."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_cfg0_against_golden(lv, scene_small):
    from limo_velo_amd import capi

    g = np.load("tests/golden/cfg0_kat.npz")
    sc = scene_small
    with capi.Context() as ctx:
        ctx.map_build(sc["map_xyz"])
        ctx.scan_set(sc["scan_xyz"])
        s = ctx.iterate(g["x_init"])
        idx, d2 = ctx.fetch_knn()
        valid, pw, abcd, dist = ctx.fetch_matches()
        H, h = ctx.fetch_rows()
        assert np.array_equal(idx, g["knn_idx"])
        assert np.array_equal(d2.view(np.uint32), g["knn_d2"].view(np.uint32))
        assert np.array_equal(valid, g["valid"])
        assert np.array_equal(abcd.view(np.uint32), g["abcd"].view(np.uint32))
        assert np.array_equal(dist.view(np.uint32), g["dist"].view(np.uint32))
        assert np.array_equal(H[:64], g["Hrows_first64"]) and np.array_equal(h[:64], g["h_first64"])
        assert s["n_valid"] == int(g["n_valid"])
        assert np.abs(s["HTH"] - g["HTH"]).max() <= 1e-10 * np.abs(g["HTH"]).max()
        x, P, passes, trace, sums = ctx.update(g["x_init"], g["P0"])
        assert passes == int(g["passes"])
        assert [q["n_valid"] for q in sums] == list(g["n_valid_per_pass"])
        assert np.abs(trace - g["trace"]).max() < 1e-9
        assert np.abs(x - g["x_post"]).max() < 1e-9 and np.abs(P - g["P_post"]).max() < 1e-10

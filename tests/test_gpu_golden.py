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


def test_next_rows_against_golden(lv):
    """Rows f-4 (ingest), f-2 (de-skew + voxel grid) and f-1 (down-sampled insert) against the committed
    known-answer digests of tests/golden/rows_kat.npz (the inputs are rebuilt deterministically — the IMU path with the
    oracle's state integrator —, the outputs are compared with the committed digests, not with a live oracle run)."""
    import hashlib
    import sys

    sys.path.insert(0, "tests/golden")
    import make_golden_rows as mg
    from limo_velo_amd import capi

    def digest(a):
        return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()

    g = np.load("tests/golden/rows_kat.npz")
    sc, raw, fmt, prm = mg.rows_inputs()
    with capi.Context() as ctx:
        kept = ctx.cloud_ingest(raw, 20_000, capi.CloudFormat(*fmt), capi.IngestParams(*prm))
        pts = ctx.cloud_fetch(-1e300, 1e300)
        assert kept == int(g["ingest_n"]) and digest(pts) == str(g["ingest_sha256"])
        assert pts[:16].copy().view(np.uint8).tobytes() == g["ingest_first"].tobytes()
        t1, t2 = float(g["t1"]), float(g["t2"])
        states = mg.deskew_path(mg.lo, t1)
        nw = ctx.scan_deskew_window(t1, t2, states, states[-2:-1], downsample_prec=0.5)
        ds = ctx.scan_fetch()
        assert nw == int(g["window_n"]) and len(ds) == int(g["voxelgrid_n"])
        assert digest(ds) == str(g["voxelgrid_sha256"]) and np.array_equal(ds[:32], g["voxelgrid_first"])
        ctx.scan_deskew_window(t1, t2, states, states[-2:-1], downsample_prec=0.0)   # de-skew only
        # the scan order after lv_scan_deskew* is the input order of the points (scan_fetch returns d_raw)
        assert digest(ctx.scan_fetch()) == str(g["deskew_sha256"])
        ctx.map_build(sc["map_xyz"])
        ctx.map_add((sc["map_xyz"][:3000] + np.float32(0.013)).astype(np.float32), downsample=True)
        merged = ctx.map_fetch()
        assert len(merged) == int(g["map_add_n"]) and digest(merged) == str(g["map_add_sha256"])
        assert np.array_equal(merged[-32:], g["map_add_tail"])


@pytest.mark.parametrize("tag,extrinsics,est", [("id", "identity", 0), ("ext", "xaloc", 1)])
def test_hip_path_against_the_reference_generated_fixture(lv, tag, extrinsics, est):
    """The HIP path against tests/golden/ref_cfg0.npz — arrays written by the reference's own compiled sources
    (tests/golden/make_golden_ref.py) — with neither the oracle nor the reference present: world points of every scan point,
    the chosen set, plane coefficients, residuals and Jacobian rows bit for bit; the iterated update to 1e-9."""
    from limo_velo_amd import capi, synth

    g = np.load("tests/golden/ref_cfg0.npz")
    sc = synth.make_scene(50_000, 2_000, extrinsics=extrinsics)
    assert float(g[tag + "_map_checksum"]) == float(sc["map_xyz"].astype(np.float64).sum())
    b = lambda a: np.ascontiguousarray(a).view(np.uint32 if a.dtype == np.float32 else np.uint64)
    with capi.Context(capi.default_params(estimate_extrinsics=est)) as ctx:
        ctx.map_build(sc["map_xyz"])
        ctx.scan_set(sc["scan_xyz"])
        for st, x in (("init", sc["x_init"]), ("true", sc["x_true"])):
            k = f"{tag}_{st}_"
            ctx.iterate(x)
            valid, pw, abcd, dist = ctx.fetch_matches()
            H, h = ctx.fetch_rows()
            v = valid.astype(bool)
            assert np.array_equal(b(pw), b(g[k + "p_world_all"]))
            assert np.array_equal(np.nonzero(v)[0], g[k + "src"])
            assert np.array_equal(b(abcd[v]), b(g[k + "abcd"])) and np.array_equal(b(dist[v]), b(g[k + "dist"]))
            assert np.array_equal(b(H[v]), b(g[k + "H"])) and np.array_equal(b(h[v]), b(g[k + "h"]))
        x, P, passes, tr, sums = ctx.update(sc["x_init"], sc["P0"])
        assert passes == int(g[tag + "_update_passes"]) and [q["n_valid"] for q in sums] == list(g[tag + "_update_n_valid"])
        tol = 1e-9 if not est else 2e-6
        assert np.abs(x - g[tag + "_update_x"]).max() < tol

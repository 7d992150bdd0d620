"""pass_kernel (one launch per measurement pass: the solve of the previous pass in every workgroup + search + plane fits,
limo-velo_amd/csrc/lv_pass_dev.hpp) against the three-kernel pass (search / fit / solve) and the oracle: the two routes run the
same device functions and differ only in the (fixed) order the workgroup partials are summed in, so states agree to ~1e-13;
every geometry of pass_grid_size is crossed (one / two search steps, dedicated / searching bookkeeping workgroup), every way
an update can end (all passes, early convergence, a pass without matches), both entries (lv_update, lv_correct)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_X, TOL_P_REL = 1e-12, 1e-9


@pytest.fixture(scope="module")
def capi(lv):
    from limo_velo_amd import capi as c

    return c


def _both(ctx, sc, x0, P0, scan=None):
    if scan is not None:
        ctx.scan_set(scan)
    ctx.set_fused_pass(True)
    xf, Pf, pf, trf, sf = ctx.update(x0, P0)
    fused = ctx.last_update_fused()
    ctx.set_fused_pass(False)
    xt, Pt, pt, trt, st = ctx.update(x0, P0)
    assert not ctx.last_update_fused()
    ctx.set_fused_pass(True)
    return (xf, Pf, pf, trf, sf, fused), (xt, Pt, pt, trt, st)


def _agree(a, b, tol_x=TOL_X, tol_p_rel=TOL_P_REL):
    xf, Pf, pf, trf, sf, _ = a
    xt, Pt, pt, trt, st = b
    assert pf == pt
    np.testing.assert_allclose(xf, xt, rtol=0, atol=tol_x)
    np.testing.assert_allclose(Pf, Pt, rtol=tol_p_rel, atol=1e-13)   # (small entries are differences of larger ones)
    assert [s["n_valid"] for s in sf] == [s["n_valid"] for s in st]
    for u, v in zip(sf, st):
        scale = max(np.abs(v["HTH"]).max(), 1.0)
        assert np.abs(u["HTH"] - v["HTH"]).max() <= 1e-12 * scale
    np.testing.assert_allclose(np.asarray(trf), np.asarray(trt), rtol=0, atol=10 * tol_x)   # dx_ and the state after every pass


@pytest.mark.parametrize("n", [1, 31, 33, 1000, 4096, 8192, 32_640, 32_768, 32_769, 65_280, 65_281, 65_536, 65_537, 100_000, 131_072])
def test_geometries_match_the_three_kernel_pass(capi, lv, n):
    """n crosses pass_grid_size's cases on a 256-CU part: <= 1024 tiles of 32 points: one search step (32 768 points fill
    every CU: the bookkeeper searches too; below that a dedicated one); up to 2040 tiles: two steps + dedicated bookkeeper;
    2041..2048 tiles: every CU searches, the books follow the bookkeeper's fits; up to 4096 tiles: two rounds per workgroup
    (the records and the candidate stage are reused, the fit wavefronts accumulate across the rounds)."""
    from limo_velo_amd import synth

    sc = synth.make_scene(300_000, 131_072)
    with capi.Context() as ctx:
        ctx.map_build(sc["map_xyz"])
        a, b = _both(ctx, sc, sc["x_init"], sc["P0"], sc["scan_xyz"][:n])
        assert a[5]                      # the one-launch route really ran
        _agree(a, b)


def test_multi_round_scans_keep_the_one_launch_pass(capi, lv):
    """Beyond two steps a workgroup takes rounds; from the second round on the separate multi-round instantiation runs a
    round's plane fits on four wavefronts beside the next round's search (task and queue counters alternate by round parity,
    the fit wavefronts release the records before the searchers overwrite them).  Three, four and six rounds against the
    three-kernel pass; estimate_extrinsics keeps round 3's barrier form and its limit of three rounds."""
    from limo_velo_amd import synth

    sc = synth.make_scene(300_000, 330_000)
    with capi.Context() as ctx:
        ctx.map_build(sc["map_xyz"])
        for n in (190_000, 200_000, 262_144, 330_000):
            a, b = _both(ctx, sc, sc["x_init"], sc["P0"], sc["scan_xyz"][:n])
            assert a[5], n                   # one launch per pass
            _agree(a, b)
        ctx.set_option("fused_multi_round", 0)   # round 3's rule: four rounds go to the three-kernel pass
        ctx.scan_set(sc["scan_xyz"][:200_000])
        ctx.update(sc["x_init"], sc["P0"])
        assert not ctx.last_update_fused()
    # estimate_extrinsics (config/xaloc.yaml:13): round 5 gave it the overlapped multi-round form too — a fit wavefront's 64 rows
    # of 14 doubles staged in two halves inside its own 6 KB of candidate stage — so it keeps one launch per pass beyond three
    # rounds; its sums must equal, BITWISE, those of round 3's barrier form (same rows, same contraction order)
    scx = synth.make_scene(300_000, 330_000, extrinsics="xaloc")
    with capi.Context(capi.default_params(estimate_extrinsics=True)) as ctx:
        ctx.map_build(scx["map_xyz"])
        for n in (131_072, 190_000, 200_000, 330_000):
            a, b = _both(ctx, scx, scx["x_init"], scx["P0"], scx["scan_xyz"][:n])
            assert a[5], n
            # (the 12-column solve at these sizes has a condition number of ~3e6 — tests/test_gpu_configs.py — so the 1e-16
            # difference of the two forms' summation orders in pass 0 is a 1e-9 .. 1e-8 difference of the state from pass 1 on:
            # pass 0's sums are held to 1e-12, the passes' match counts to equality, the posterior to the EXT tolerance)
            assert a[2] == b[2] and [v["n_valid"] for v in a[4]] == [v["n_valid"] for v in b[4]], n
            assert np.abs(a[4][0]["HTH"] - b[4][0]["HTH"]).max() <= 1e-12 * np.abs(b[4][0]["HTH"]).max(), n
            assert np.abs(a[0] - b[0]).max() < 2e-6 and np.abs(a[1] - b[1]).max() < 1e-5 * max(1.0, np.abs(b[1]).max()), n
        ctx.set_option("fused_multi_round", 0)   # round 3's rule
        ctx.scan_set(scx["scan_xyz"][:200_000])
        ctx.update(scx["x_init"], scx["P0"])
        assert not ctx.last_update_fused()


def test_multi_round_scans_with_open_and_stray_points(capi, lv):
    """The multi-round instantiation with every path of a round in play: a sparse map (many points need level 1, some the coarse
    levels' queue) and stray points far from every surface (bounded stops), four and five rounds per workgroup, against the
    three-kernel pass."""
    from limo_velo_amd import synth

    sc = synth.make_scene(60_000, 300_000)
    rng = np.random.default_rng(3)
    scan = sc["scan_xyz"].copy()
    pick = rng.choice(len(scan), 3000, replace=False)
    scan[pick] += rng.uniform(-4.0, 4.0, (3000, 3)).astype(np.float32)
    with capi.Context() as ctx:
        ctx.map_build(sc["map_xyz"])
        for n in (230_000, 300_000):
            a, b = _both(ctx, sc, sc["x_init"], sc["P0"], scan[:n])
            assert a[5]
            _agree(a, b)
            assert ctx.timing()["fallback_queries"] >= 0


def test_multi_round_sums_are_bitwise_stable_across_repetitions(capi, lv):
    """in the multi-round instantiation a fit wavefront's staged rows used to overlap the candidate stage of
    the fit wavefront before it, which refills that stage as soon as it rejoins the next round's search — a timing-dependent
    corruption of H^T H / H^T h or of neighbour coordinates.  The rows now live inside the fit wavefront's own stage area.  This
    is the stress test the finding asks for: many repetitions of multi-round updates (3 / 5 / 9 rounds per workgroup; a sparse map
    so that search tasks differ wildly in length and fit wavefronts rejoin the search at different times), every per-pass sums
    record, state and covariance compared BITWISE with the first repetition, and the first repetition with the three-kernel pass."""
    from limo_velo_amd import synth

    sc = synth.make_scene(80_000, 600_000)
    rng = np.random.default_rng(11)
    scan = sc["scan_xyz"].copy()
    pick = rng.choice(len(scan), 6000, replace=False)
    scan[pick] += rng.uniform(-3.0, 3.0, (6000, 3)).astype(np.float32)
    with capi.Context() as ctx:
        ctx.map_build(sc["map_xyz"])
        for n, reps in ((150_000, 60), (300_000, 40), (600_000, 20)):
            a, b = _both(ctx, sc, sc["x_init"], sc["P0"], scan[:n])
            assert a[5], n
            _agree(a, b)
            x0, P0, p0, tr0, s0 = a[:5]
            for r in range(reps):
                x, P, p, tr, sm = ctx.update(sc["x_init"], sc["P0"])
                assert ctx.last_update_fused()
                assert p == p0 and np.array_equal(x, x0) and np.array_equal(P, P0), (n, r)
                for u, v in zip(sm, s0):
                    assert u["n_valid"] == v["n_valid"] and np.array_equal(u["HTH"], v["HTH"]) and np.array_equal(u["HTh"], v["HTh"]), (n, r)


@pytest.mark.parametrize("iters", [0, 1, 2, 3])
def test_pass_counts_and_oracle(capi, oracle, scene_small, iters):
    sc = scene_small
    prm = capi.default_params(MAX_NUM_ITERS=iters)
    with capi.Context(prm) as ctx:
        ctx.map_build(sc["map_xyz"])
        a, b = _both(ctx, sc, sc["x_init"], sc["P0"], sc["scan_xyz"])
        _agree(a, b)
    xo, Po, po, tro, so = oracle.update(sc["x_init"], sc["P0"], sc["map_xyz"], sc["scan_xyz"], params=oracle.default_params(max_num_iters=iters))
    assert a[2] == po == iters + 1
    assert np.abs(a[0] - xo).max() < 1e-9 and np.abs(a[1] - Po).max() < 1e-9
    assert [s["n_valid"] for s in a[4]] == [s["n_valid"] for s in so]


def test_early_convergence_ends_the_update_inside_a_prologue(capi, oracle, scene_small):
    """From the true pose dx falls under LIMITS twice in a row before MAX_NUM_ITERS + 1 passes: the launch whose prologue
    finds that out searches nothing, its bookkeeping workgroup writes the terminal results, the launches behind it exit."""
    sc = scene_small
    with capi.Context() as ctx:
        ctx.map_build(sc["map_xyz"])
        a, b = _both(ctx, sc, sc["x_true"], sc["P0"], sc["scan_xyz"])
        _agree(a, b)
        assert a[5] and a[2] < ctx.params.MAX_NUM_ITERS + 1
        # the next update on the same context starts clean
        a2, b2 = _both(ctx, sc, sc["x_init"], sc["P0"])
        _agree(a2, b2)
        assert a2[2] == ctx.params.MAX_NUM_ITERS + 1
    xo, Po, po, _, _ = oracle.update(sc["x_true"], sc["P0"], sc["map_xyz"], sc["scan_xyz"])
    assert a[2] == po and np.abs(a[0] - xo).max() < 1e-9


def test_passes_without_matches(capi, scene_small):
    """A scan 500 m from the map: every pass has n_valid = 0 (h_share_model: valid = false -> continue): the state does not
    move, the covariance returned is the propagated one, all MAX_NUM_ITERS + 1 passes are counted — on both routes."""
    sc = scene_small
    far = sc["scan_xyz"][:3000] + np.float32([500.0, 0.0, 0.0])
    with capi.Context() as ctx:
        ctx.map_build(sc["map_xyz"])
        a, b = _both(ctx, sc, sc["x_init"], sc["P0"], far)
        assert a[5]
    for x, P, p, tr, sums in (a[:5], b):
        assert p == 4 and [s["n_valid"] for s in sums] == [0] * 4
        assert np.array_equal(x, sc["x_init"]) and np.array_equal(P, sc["P0"])


def test_correlated_covariance_and_resident_filter(capi, scene_small):
    sc = scene_small
    rng = np.random.default_rng(3)
    A = rng.normal(0, 1, (23, 23))
    P0 = sc["P0"] + 1e-3 * (A @ A.T)
    with capi.Context() as ctx:
        ctx.map_build(sc["map_xyz"])
        a, b = _both(ctx, sc, sc["x_init"], P0, sc["scan_xyz"])
        _agree(a, b)
        # lv_correct (state resident on the device: a begin kernel installs it, the first launch is "mode 2")
        res = {}
        for fused in (True, False):
            ctx.set_fused_pass(fused)
            ctx.filter_set(sc["x_init"], P0)
            p = ctx.correct()
            assert ctx.last_update_fused() == fused
            res[fused] = (ctx.filter_get(), p)
        ctx.set_fused_pass(True)
    (xf, Pf), pf = res[True]
    (xt, Pt), pt = res[False]
    assert pf == pt == a[2]
    np.testing.assert_allclose(xf, xt, rtol=0, atol=TOL_X)
    np.testing.assert_allclose(Pf, Pt, rtol=TOL_P_REL, atol=1e-13)   # (small entries are differences of larger ones)
    np.testing.assert_allclose(xf, a[0], rtol=0, atol=TOL_X)   # lv_correct == lv_update from the same state


def test_many_updates_are_bit_reproducible(capi, scene_small):
    """No atomics on the data path (the queue's slot counter only decides which wavefront serves a point; the bookkeeping
    workgroup is chosen by last launch's timings but computes what any other would): identical bits run to run."""
    sc = scene_small
    with capi.Context() as ctx:
        ctx.map_build(sc["map_xyz"])
        ctx.scan_set(sc["scan_xyz"])
        x0, P0, p0, _, _ = ctx.update(sc["x_init"], sc["P0"], want_trace=False)
        for _ in range(200):
            x, P, p, _, _ = ctx.update(sc["x_init"], sc["P0"], want_trace=False)
            assert p == p0 and np.array_equal(x, x0) and np.array_equal(P, P0)


def test_extrinsics_variant(capi, scene_small, monkeypatch):
    """estimate_extrinsics = true (12-column rows, 92 live sums, 12 x 12 gain blocks): the one-launch form (the default since
    round 3: 22.1 k vs 21.4 k it/s at the headline size) must agree with the three-kernel pass."""
    sc = scene_small
    with capi.Context(capi.default_params(estimate_extrinsics=1)) as ctx:
        ctx.map_build(sc["map_xyz"])
        a, b = _both(ctx, sc, sc["x_init"], sc["P0"], sc["scan_xyz"])
        assert a[5]
        # (the extrinsics are weakly observable from one scan: the 12 x 12 gain blocks amplify the 1e-16 difference of the
        # summation orders more than the 6 x 6 ones do)
        _agree(a, b, tol_x=1e-10, tol_p_rel=1e-6)


@pytest.mark.parametrize("case", ["all_passes", "early_convergence", "no_matches", "three_kernel"])
def test_filter_get_from_the_mailbox_equals_the_copy(capi, scene_small, case):
    """After lv_correct the posterior is read from the host-mapped mailbox its finishing pass writes (a poll) instead of a
    device-to-host copy + stream synchronise: the same bits, however the update ended, and only while the mailbox still IS the
    resident filter (a predict / filter_set / other update in between goes back to the copy)."""
    sc = scene_small
    scan = sc["scan_xyz"]
    x0 = sc["x_init"]
    if case == "early_convergence":
        x0 = sc["x_true"]
    if case == "no_matches":
        scan = scan[:3000] + np.float32([500.0, 0.0, 0.0])
    res = {}
    with capi.Context() as ctx:
        ctx.map_build(sc["map_xyz"])
        ctx.scan_set(scan)
        ctx.set_fused_pass(case != "three_kernel")
        for mail in (1, 0, 1):
            ctx.set_option("mail_filter", mail)
            ctx.filter_set(x0, sc["P0"])
            ctx.correct(want_passes=False)   # (no synchronisation inside lv_correct)
            x, P = ctx.filter_get()
            res.setdefault(mail, []).append((x.copy(), P.copy(), ctx.last_passes()))
        # a predict after the correct: the mailbox no longer is the filter
        ctx.set_option("mail_filter", 1)
        ctx.filter_set(x0, sc["P0"])
        ctx.correct(want_passes=False)
        Q = np.eye(12) * 1e-4
        ctx.predict(0.01, Q, np.array([0.1, 0.0, 9.8]), np.array([0.0, 0.0, 0.1]))
        xa, Pa = ctx.filter_get()
        ctx.set_option("mail_filter", 0)
        xb, Pb = ctx.filter_get()
        assert np.array_equal(xa, xb) and np.array_equal(Pa, Pb)
        assert not np.array_equal(xa, res[1][0][0])
    for a in res[1]:
        assert np.array_equal(a[0], res[0][0][0]) and np.array_equal(a[1], res[0][0][1]) and a[2] == res[0][0][2]

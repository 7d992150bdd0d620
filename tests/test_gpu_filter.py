"""Row f-3: device-resident filter — esekf::predict on the GPU (lv_predict) and correction of the resident state
(lv_correct) against the oracle's restatement of IKFoM predict + iterated update, over several
propagate -> correct cycles as src/main.cpp:76-85 runs them."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _Q():  # Localizator::propagate (Localizator.cpp:164-168) with config/params.yaml:39-42
    return np.diag([1e-4] * 3 + [1e-2] * 3 + [1e-5] * 3 + [1e-4] * 3)


def test_predict_matches_oracle(lv, oracle, scene_small):
    from limo_velo_amd import capi

    sc = scene_small
    rng = np.random.default_rng(4)
    x, P = sc["x_init"].copy(), sc["P0"].copy()
    x[14:17] = [1.5, -0.3, 0.05]      # velocity
    x[17:20] = [0.01, -0.02, 0.005]   # gyro bias
    x[20:23] = [0.05, 0.02, -0.03]    # accel bias
    with capi.Context() as ctx:
        ctx.filter_set(x, P)
        for i in range(40):
            acc = np.array([0.3, -0.1, 9.81]) + rng.normal(scale=0.2, size=3)
            gyro = np.array([0.02, -0.01, 0.3]) + rng.normal(scale=0.05, size=3)
            dt = 0.005 if i % 7 else 0.0123
            ctx.predict(dt, _Q(), acc, gyro)
            x, P = oracle.predict(x, P, dt, _Q(), acc, gyro)
        xg, Pg = ctx.filter_get()
    assert np.abs(xg - x).max() < 1e-12
    assert np.abs(Pg - P).max() < 1e-12 * max(1.0, np.abs(P).max())
    assert abs(np.linalg.norm(xg[23:26]) - 9.809) < 1e-9


def test_propagate_correct_cycles(lv, oracle, scene_small):
    from limo_velo_amd import capi

    sc = scene_small
    tree = oracle.KdTree(sc["map_xyz"])
    x, P = sc["x_init"].copy(), sc["P0"].copy()
    acc, gyro = np.array([0.0, 0.0, 9.809]), np.array([0.0, 0.0, 0.0])  # static sensor: the map stays valid
    scans = [sc["scan_xyz"][:1200], sc["scan_xyz"][600:1900], sc["scan_xyz"][100:2000]]
    with capi.Context() as ctx:
        ctx.map_build(sc["map_xyz"])
        ctx.filter_set(x, P)
        for scan in scans:
            for _ in range(5):
                ctx.predict(0.002, _Q(), acc, gyro)
                x, P = oracle.predict(x, P, 0.002, _Q(), acc, gyro)
            ctx.scan_set(scan)
            passes = ctx.correct()
            x, P, po, _, _ = oracle.update(x, P, sc["map_xyz"], scan, tree=tree)
            assert passes == po
            xg, Pg = ctx.filter_get()
            assert np.abs(xg - x).max() < 1e-8, np.abs(xg - x).max()
            assert np.abs(Pg - P).max() < 1e-8 * max(1.0, np.abs(P).max())
            x, P = xg, Pg  # continue from the device state so rounding does not accumulate in the comparison
    assert np.linalg.norm(x[:3] - sc["x_true"][:3]) < 5e-3


def test_batched_predictions_and_lazy_filter_copies_change_no_bit(lv, scene_small):
    """lv_predict calls are queued and launched up to eight steps at a time; after lv_correct the posterior stays in the update's
    working copy until something needs it in the filter's own buffer (the next prediction reads it from there, the next
    correct starts from there).  A propagate -> correct -> propagate sequence gives the same bits with one launch per
    prediction, with the copies made eagerly (mail_filter = 0 reads the filter buffer), and with a changing Q in the queue."""
    from limo_velo_amd import capi

    sc = scene_small
    rng = np.random.default_rng(9)
    steps = [(0.005 if i % 5 else 0.011, np.array([0.2, -0.1, 9.8]) + rng.normal(scale=0.2, size=3),
              np.array([0.01, 0.02, 0.2]) + rng.normal(scale=0.05, size=3)) for i in range(23)]
    Q2 = _Q() * 1.5
    res = {}
    for batch, mail in ((1, 1), (0, 1), (1, 0), (0, 0)):
        with capi.Context() as ctx:
            ctx.set_option("batch_predict", batch)
            ctx.set_option("mail_filter", mail)
            ctx.map_build(sc["map_xyz"])
            ctx.scan_set(sc["scan_xyz"][:5000])
            ctx.filter_set(sc["x_init"], sc["P0"])
            out = []
            for i, (dt, a, g) in enumerate(steps):
                ctx.predict(dt, Q2 if i in (11, 12) else _Q(), a, g)      # (a different Q in the middle of a queue)
                if i in (2, 12, 13, 22):
                    ctx.correct(want_passes=False)
                    if i == 13:
                        ctx.correct(want_passes=False)                      # two updates without a prediction in between
                    if i != 12:
                        out.append(ctx.filter_get())
            # a by-value update in between must not disturb the resident filter
            ctx.predict(0.004, _Q(), steps[0][1], steps[0][2])
            ctx.correct(want_passes=False)
            ctx.update(sc["x_init"], sc["P0"], want_trace=False)
            ctx.predict(0.004, _Q(), steps[1][1], steps[1][2])
            out.append(ctx.filter_get())
            res[(batch, mail)] = out
    ref = res[(0, 0)]
    for key, out in res.items():
        assert len(out) == len(ref)
        for (xa, Pa), (xb, Pb) in zip(out, ref):
            assert np.array_equal(xa, xb) and np.array_equal(Pa, Pb), key


def test_filter_set_hands_over_lazily_and_is_never_lost(lv, oracle, scene_small):
    """lv_filter_set (round 4) neither uploads nor waits: the filter stays in pinned host memory until something needs it on
    the device.  Whatever comes next must see exactly the filter that was set: lv_filter_get (no device involved), lv_correct
    (the prior rides in its first launch's arguments: same result as the update by value), lv_predict (uploaded first), an
    lv_update / lv_iterate by value in between (they use the device's working copy but leave the resident filter alone), a second
    lv_filter_set (replaces the first)."""
    from limo_velo_amd import capi

    sc = scene_small
    x0, P0 = sc["x_init"].copy(), sc["P0"].copy()
    x1 = oracle.boxplus(x0, np.r_[0.02, -0.01, 0.015, 0.003, -0.002, 0.004, np.zeros(17)])
    acc, gyro = np.array([0.1, -0.05, 9.81]), np.array([0.01, 0.02, -0.01])
    with capi.Context() as ctx:
        ctx.map_build(sc["map_xyz"])
        ctx.scan_set(sc["scan_xyz"])
        xu, Pu, pu, _, _ = ctx.update(x0, P0)                    # the reference result: update by value from (x0, P0)
        # set -> get
        ctx.filter_set(x1, P0)
        xg, Pg = ctx.filter_get()
        assert np.array_equal(xg, x1) and np.array_equal(Pg, P0)
        # set (replaces) -> correct: the by-value update, bit for bit
        ctx.filter_set(x0, P0)
        assert ctx.correct() == pu
        xg, Pg = ctx.filter_get()
        assert np.array_equal(xg, xu) and np.array_equal(Pg, Pu)
        # set -> an update by value and a capturing pass in between -> correct still starts from the filter that was set
        ctx.filter_set(x0, P0)
        ctx.update(x1, P0)
        ctx.iterate(x1)
        assert ctx.correct() == pu
        xg, Pg = ctx.filter_get()
        assert np.array_equal(xg, xu) and np.array_equal(Pg, Pu)
        # set -> predict (needs the filter on the device) -> get
        ctx.filter_set(x0, P0)
        ctx.predict(0.01, _Q(), acc, gyro)
        xg, Pg = ctx.filter_get()
        xo, Po = oracle.predict(x0, P0, 0.01, _Q(), acc, gyro)
        assert np.abs(xg - xo).max() < 1e-12 and np.abs(Pg - Po).max() < 1e-12 * max(1.0, np.abs(Po).max())
        # many set + correct pairs enqueued without waiting (bench.py's timed step), one wait at the end
        for _ in range(20):
            ctx.filter_set(x0, P0)
            ctx.correct(want_passes=False)
        xg, Pg = ctx.filter_get()
        assert np.array_equal(xg, xu) and np.array_equal(Pg, Pu) and ctx.last_passes() == pu

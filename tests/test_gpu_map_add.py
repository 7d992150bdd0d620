"""Row f-1: Mapper::add -> KD_TREE::Add_Points(points, downsample) on the device (lv_map_add) against the
oracle's sequential restatement of ikd-Tree's box rule.  The resulting map must be identical point for point
and in the same order (the order is the index space of the kNN results)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi(lv):
    from limo_velo_amd import capi as c

    return c


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_add_without_downsample_appends(capi, oracle, scene_small):
    sc = scene_small
    a, b = sc["map_xyz"][:30000], sc["map_xyz"][30000:]
    with capi.Context() as ctx:
        ctx.map_build(a)
        ctx.map_add(b, downsample=False)
        assert ctx.map_size() == 50_000
        assert np.array_equal(_bits(ctx.map_fetch()), _bits(sc["map_xyz"]))
        ctx.scan_set(sc["scan_xyz"][:500])
        ctx.iterate(sc["x_init"])
        idx, d2 = ctx.fetch_knn()
        oi, od, _, _ = oracle.knn_brute(sc["map_xyz"], oracle.transform_scan(sc["x_init"], sc["scan_xyz"][:500]))
        assert np.array_equal(idx, oi) and np.array_equal(_bits(d2), _bits(od))


def test_downsampled_add_matches_sequential_rule(capi, oracle, scene_small):
    sc = scene_small
    rng = np.random.default_rng(11)
    base = sc["map_xyz"][:20000]                      # Build(): no down-sampling, boxes may hold several points
    scan1 = sc["map_xyz"][20000:35000]
    scan2 = np.concatenate([sc["map_xyz"][35000:], base[rng.integers(0, len(base), 2000)] + np.float32(0.01)])
    dup = np.concatenate([scan2[:100], scan2[:100]])  # repeated points inside one batch
    ref = base
    with capi.Context() as ctx:
        ctx.map_build(base)
        for batch in (scan1, scan2, dup, scan1[:1], scan1[:0]):
            ctx.map_add(batch, downsample=True)
            ref = oracle.map_add(ref, batch, downsample=True)
            got = ctx.map_fetch()
            assert got.shape == ref.shape, (got.shape, ref.shape)
            assert np.array_equal(_bits(got), _bits(ref))
        assert len(ref) < 20000 + 15000 + len(scan2) + 200
        # the search structure was rebuilt over the new map: kNN must match the oracle on it
        ctx.scan_set(sc["scan_xyz"][:800])
        g = ctx.iterate(sc["x_init"])
        idx, d2 = ctx.fetch_knn()
        oi, od, _, _ = oracle.knn_brute(ref, oracle.transform_scan(sc["x_init"], sc["scan_xyz"][:800]))
        assert np.array_equal(idx, oi) and np.array_equal(_bits(d2), _bits(od))
        o = oracle.iterate(sc["x_init"], ref, sc["scan_xyz"][:800])
        assert g["n_valid"] == o["n_valid"]


def test_first_add_on_empty_map_and_lattice_ties(capi, oracle):
    # points exactly on box boundaries / equidistant from box centres exercise the tie rule
    g = (np.stack(np.meshgrid(np.arange(-10, 10), np.arange(-10, 10), [0, 1]), -1).reshape(-1, 3) * 0.1).astype(np.float32)
    rng = np.random.default_rng(2)
    new = g[rng.permutation(len(g))][:300] + np.float32(0.05)
    with capi.Context() as ctx:
        ctx.map_add(g, downsample=True)  # Mapper::add on an empty map builds (Mapper.cpp:26); the shim calls build
        ref = oracle.map_add(np.zeros((0, 3), np.float32), g, downsample=True)
        assert np.array_equal(_bits(ctx.map_fetch()), _bits(ref))
        ctx.map_add(new, downsample=True)
        ref = oracle.map_add(ref, new, downsample=True)
        assert np.array_equal(_bits(ctx.map_fetch()), _bits(ref))


def _knn_matches(ctx, oracle, ref, state, scan):
    ctx.scan_set(scan)
    g = ctx.iterate(state)
    idx, d2 = ctx.fetch_knn()
    oi, od, _, _ = oracle.knn_brute(ref, oracle.transform_scan(state, scan))
    assert np.array_equal(idx, oi), f"kNN index mismatches at {(idx != oi).any(axis=1).sum()} points"
    assert np.array_equal(_bits(d2), _bits(od))
    o = oracle.iterate(state, ref, scan)
    assert g["n_valid"] == o["n_valid"]
    return g


def test_fifty_incremental_adds_keep_map_and_search_exact(capi, oracle, lv):
    """The mapping cycle of src/main.cpp:102 fifty times over: a 200k-point map takes fifty down-sampled scans (revisited
    and new space), every one inserted IN PLACE (no rebuild: incremental_adds counts them).  After each add the map
    equals the oracle's sequential lvo_map_add point for point and in order; the exact 5-NN (indices in the shifted
    index space, distance bits) is re-checked along the way and at the end, through the capturing kernels and through
    the timed ones."""
    from limo_velo_amd import synth

    sc = synth.make_scene(200_000, 4000)
    rng = np.random.default_rng(5)
    ref = sc["map_xyz"]
    L = float(sc["L"])
    with capi.Context() as ctx:
        ctx.map_build(ref)
        for step in range(50):
            # world points of a scan: noisy copies of map points in a drifting window (revisits: the box rule keeps or
            # replaces occupants) plus fresh points beyond the mapped walls (new buckets, new voxels)
            c = np.array([-0.6 * L + 0.02 * L * step, 0.3 * L - 0.01 * L * step, 0.0], np.float32)
            near = ref[np.linalg.norm(ref - c, axis=1) < 25.0]
            pick = near[rng.integers(0, len(near), 1500)] + rng.normal(0, 0.03, (1500, 3)).astype(np.float32)
            fresh = (rng.uniform(-1, 1, (300, 3)) * [6, 6, 0.02] + [L + 4.0 + 0.1 * step, c[1], 0.5]).astype(np.float32)
            batch = np.concatenate([pick, fresh]).astype(np.float32)
            ctx.map_add(batch, downsample=True)
            ref = oracle.map_add(ref, batch, downsample=True)
            assert ctx.map_size() == len(ref), step
            if step % 10 == 9 or step == 0:
                assert np.array_equal(_bits(ctx.map_fetch()), _bits(ref)), f"map differs after add {step}"
                _knn_matches(ctx, oracle, ref, sc["x_init"], sc["scan_xyz"][:1500])
        st = ctx.map_stats()
        assert st["incremental_adds"] == 50 and st["living"] == len(ref)
        assert st["relinearisations"] <= 2, st          # in place, not rebuilt
        assert st["tombstones"] > 0 or st["relinearisations"] > 0
        # queries in the freshly mapped strip and in the revisited window
        probe = np.concatenate([fresh[:200], pick[:300]])
        ident = sc["x_true"].copy()
        ident[:3] = 0
        ident[3:7] = [0, 0, 0, 1]
        _knn_matches(ctx, oracle, ref, ident, probe)
        # the timed (non-capturing) kernels on the incrementally maintained structure: full update vs oracle on `ref`
        ctx.scan_set(sc["scan_xyz"])
        ctx.set_record_dump(True)   # pass_kernel keeps its hand-over records in LDS; the same kernel also stores them for the fetch
        x, P, passes, tr, sums = ctx.update(sc["x_init"], sc["P0"])
        nbr, d2, pw, found = ctx.fetch_neighbors()
        ctx.set_record_dump(False)
        xo, Po, po, tro, so = oracle.update(sc["x_init"], sc["P0"], ref, sc["scan_xyz"])
        assert passes == po and [s["n_valid"] for s in sums] == [s["n_valid"] for s in so]
        assert np.abs(x - xo).max() < 1e-9
        # a re-linearisation must not change anything observable
        ctx.map_relinearise()
        assert np.array_equal(_bits(ctx.map_fetch()), _bits(ref))
        x2, P2, p2, _, _ = ctx.update(sc["x_init"], sc["P0"])
        assert p2 == passes and np.abs(x2 - x).max() < 1e-12
        _knn_matches(ctx, oracle, ref, sc["x_init"], sc["scan_xyz"][:1000])


def test_evictions_and_rolling_window(capi, oracle, scene_small):
    sc = scene_small
    ref = sc["map_xyz"]
    rng = np.random.default_rng(8)
    with capi.Context() as ctx:
        ctx.map_build(ref)
        lo, hi = np.array([-12.0, -15.0, -1.0], np.float32), np.array([15.0, 11.0, 9.0], np.float32)
        inside = np.all((ref >= lo) & (ref <= hi), axis=1)
        assert ctx.map_evict_box(lo, hi, keep_inside=True) == int((~inside).sum())
        ref = ref[inside]
        assert ctx.map_size() == len(ref)
        assert np.array_equal(_bits(ctx.map_fetch()), _bits(ref))
        _knn_matches(ctx, oracle, ref, sc["x_init"], sc["scan_xyz"][:800])   # indices are ranks among the living
        for step in range(3):
            batch = (ref[rng.integers(0, len(ref), 3000)] + rng.normal(0, 0.05, (3000, 3))).astype(np.float32)
            ctx.map_add(batch, downsample=True)
            ref = oracle.map_add(ref, batch, downsample=True)
        assert np.array_equal(_bits(ctx.map_fetch()), _bits(ref))
        k = len(ref) // 4
        assert ctx.map_evict_oldest(k) == k
        ref = ref[k:]
        assert np.array_equal(_bits(ctx.map_fetch()), _bits(ref))
        _knn_matches(ctx, oracle, ref, sc["x_init"], sc["scan_xyz"][:800])
        hole = np.all((ref >= -3.0) & (ref <= 3.0), axis=1)
        assert ctx.map_evict_box([-3.0] * 3, [3.0] * 3, keep_inside=False) == int(hole.sum())
        ref = ref[~hole]
        _knn_matches(ctx, oracle, ref, sc["x_init"], sc["scan_xyz"][:800])
        # the whole update on the thinned map
        ctx.scan_set(sc["scan_xyz"])
        x, P, passes, _, sums = ctx.update(sc["x_init"], sc["P0"])
        xo, Po, po, _, so = oracle.update(sc["x_init"], sc["P0"], ref, sc["scan_xyz"])
        assert passes == po and [s["n_valid"] for s in sums] == [s["n_valid"] for s in so]
        assert np.abs(x - xo).max() < 1e-9
        # everything goes: no map (Localizator::correct returns, Localizator.cpp:24); the next add builds again
        assert ctx.map_evict_box([-1e4] * 3, [1e4] * 3, keep_inside=False) == len(ref)
        assert ctx.map_size() == 0
        assert ctx.update(sc["x_init"], sc["P0"])[2] == 0
        ctx.map_add(sc["map_xyz"][:5000], downsample=True)   # Add_Points(points, true) into an empty map: the rule among the new points
        ref = oracle.map_add(np.zeros((0, 3), np.float32), sc["map_xyz"][:5000], downsample=True)
        assert np.array_equal(_bits(ctx.map_fetch()), _bits(ref))
        _knn_matches(ctx, oracle, ref, sc["x_init"], sc["scan_xyz"][:300])


def test_map_add_scan_stays_on_the_device(capi, oracle, scene_small):
    """lv_map_add_scan = `map.add(Xt2 * Xt2.I_Rt_L() * ds_compensated, t2, true)` (src/main.cpp:92,102) with the scan and
    the state the device already holds: same map as transforming on the host with the oracle's f32 arithmetic."""
    sc = scene_small
    ref = sc["map_xyz"][:30000]
    with capi.Context() as ctx:
        ctx.map_build(ref)
        scan = sc["scan_xyz"].copy()
        scan[7] = [np.nan, 0, 0]                      # a no-return point is skipped
        ctx.scan_set(scan)
        x, P, passes, _, _ = ctx.update(sc["x_init"], sc["P0"])
        ctx.map_add_scan(downsample=True)             # state: the last update's result
        world = oracle.transform_scan(x, np.delete(scan, 7, axis=0))
        ref = oracle.map_add(ref, world, downsample=True)
        assert np.array_equal(_bits(ctx.map_fetch()), _bits(ref))
        # resident filter as the state source
        ctx.filter_set(sc["x_true"], sc["P0"])
        ctx.scan_set(sc["scan_xyz"][:700])
        ctx.map_add_scan(downsample=False)
        ref = oracle.map_add(ref, oracle.transform_scan(sc["x_true"], sc["scan_xyz"][:700]), downsample=False)
        assert np.array_equal(_bits(ctx.map_fetch()), _bits(ref))
        assert ctx.map_stats()["dropped"] == 1


def test_map_add_scan_builds_an_empty_map(capi, oracle, scene_small):
    """Mapper::add on an empty map BUILDS it from the cloud as it is (Mapper.cpp:22-27: no down-sampling, whatever the
    flag says) — lv_map_add_scan follows Mapper::add; lv_map_add follows KD_TREE::Add_Points (the box rule applies)."""
    sc = scene_small
    scan = sc["scan_xyz"][:1500].copy()
    scan[3] = [np.inf, 0, 0]
    with capi.Context() as ctx:
        ctx.filter_set(sc["x_true"], sc["P0"])
        ctx.scan_set(scan)
        assert ctx.map_size() == 0
        ctx.map_add_scan(downsample=True)
        want = oracle.transform_scan(sc["x_true"], np.delete(scan, 3, axis=0))
        assert ctx.map_size() == len(want)
        assert np.array_equal(_bits(ctx.map_fetch()), _bits(want))
        _knn_matches(ctx, oracle, want, sc["x_true"], sc["scan_xyz"][:200])
        # the next one is an insert with the box rule
        ctx.scan_set(sc["scan_xyz"][1500:])
        ctx.map_add_scan(downsample=True)
        want = oracle.map_add(want, oracle.transform_scan(sc["x_true"], sc["scan_xyz"][1500:]), downsample=True)
        assert np.array_equal(_bits(ctx.map_fetch()), _bits(want))


def test_small_batch_front_kernel_equals_the_separate_launches(lv, scene_small):
    """Insert batches of up to 2048 points run box keys -> sort -> box rule -> scan -> ids -> voxel groups in ONE workgroup
    launch; the knob off (and larger batches) the seven separate launches: identical map contents, order and search results
    after every one of 12 consecutive small inserts."""
    from limo_velo_amd import capi

    sc = scene_small
    rng = np.random.default_rng(7)
    batches = [(sc["map_xyz"][rng.integers(0, len(sc["map_xyz"]), n)] + rng.normal(0, 0.05, (n, 3))).astype(np.float32)
               for n in (1, 17, 640, 2048, 800, 1200, 33, 2047, 5, 900, 1999, 64)]
    maps = {}
    for on in (1, 0):
        with capi.Context() as ctx:
            ctx.set_option("small_insert", on)
            ctx.map_build(sc["map_xyz"][:20000])
            sizes = []
            for b in batches:
                ctx.map_add(b, downsample=True)
                sizes.append(ctx.map_size())
            ctx.scan_set(sc["scan_xyz"])
            g = ctx.iterate(sc["x_init"])
            idx, d2 = ctx.fetch_knn()
            maps[on] = (ctx.map_fetch(), sizes, idx, d2, g["n_valid"], g["HTH"])
    assert maps[1][1] == maps[0][1]
    assert np.array_equal(maps[1][0].view(np.uint32), maps[0][0].view(np.uint32))
    assert np.array_equal(maps[1][2], maps[0][2]) and np.array_equal(maps[1][3].view(np.uint32), maps[0][3].view(np.uint32))
    assert maps[1][4] == maps[0][4] and np.array_equal(maps[1][5], maps[0][5])


def test_eviction_by_runs_equals_eviction_by_points(lv, scene_small):
    """lv_map_evict_box applies the box test to the RUNS (bucket / list bounding boxes: untouched, dropped whole, or walked entry
    by entry) instead of searching every evicted point's 81 runs: the same living points, the same search results and the
    same behaviour of later inserts, for both senses of the box, a box that cuts nothing, and one that takes everything."""
    from limo_velo_amd import capi

    sc = scene_small
    rng = np.random.default_rng(21)
    batches = [(sc["map_xyz"][rng.integers(0, len(sc["map_xyz"]), n)] + rng.normal(0, 0.05, (n, 3))).astype(np.float32) for n in (3000, 900)]
    res = {}
    for sweep in (1, 0):
        with capi.Context() as ctx:
            ctx.set_option("sweep_evict", sweep)
            ctx.map_build(sc["map_xyz"])
            out = []
            out.append(ctx.map_evict_box([-14.0, -17.5, -1.0], [16.25, 12.0, 9.0], keep_inside=True))
            ctx.map_add(batches[0], downsample=True)
            out.append(ctx.map_evict_box([-2.5, -3.0, -5.0], [3.5, 2.0, 5.0], keep_inside=False))
            out.append(ctx.map_evict_box([-1e4] * 3, [1e4] * 3, keep_inside=True))      # cuts nothing
            ctx.map_add(batches[1], downsample=True)
            out.append(ctx.map_evict_box([100.0] * 3, [101.0] * 3, keep_inside=False))  # nothing inside
            out.append(ctx.map_size())
            ctx.scan_set(sc["scan_xyz"])
            g = ctx.iterate(sc["x_init"])
            idx, d2 = ctx.fetch_knn()
            x, P, passes, _, sums = ctx.update(sc["x_init"], sc["P0"])
            living = ctx.map_fetch()
            out.append(ctx.map_evict_box([-1e4] * 3, [1e4] * 3, keep_inside=False))     # takes everything
            out.append(ctx.map_size())
            res[sweep] = (out, living, idx, d2, g["n_valid"], x, P, passes, [s["n_valid"] for s in sums])
    a, b = res[1], res[0]
    assert a[0] == b[0] and a[0][0] > 0 and a[0][1] > 0 and a[0][2] == 0 and a[0][3] == 0 and a[0][-1] == 0
    assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3].view(np.uint32), b[3].view(np.uint32))
    assert a[4] == b[4] and a[7] == b[7] and a[8] == b[8]
    assert np.array_equal(a[5], b[5]) and np.array_equal(a[6], b[6])


def test_rebuild_gives_memory_back_after_the_map_shrank(lv):
    """A rolling window that shrank far below its initial extent (configs[4]: a 10 M-point map cut to the 2.4 M points around the
    sensor) must not keep pools and tables sized for the old map for ever: the (re)build that a re-linearisation runs re-allocates
    pools 3x / tables 8x larger than the map now wants.  The search on the shrunk structure = the search on a fresh build of the
    same points, bit for bit."""
    from limo_velo_amd import capi, synth

    sc = synth.make_scene(4_000_000, 8_192)
    with capi.Context() as ctx:
        ctx.map_build(sc["map_xyz"])
        st0 = ctx.map_stats()
        c = sc["x_init"][:3].astype(np.float32)
        gone = ctx.map_evict_box(c - np.float32([18, 18, 10]), c + np.float32([18, 18, 10]), keep_inside=True)
        living = ctx.map_size()
        assert gone > 0 and living * 5 < 4_000_000, (gone, living)
        ctx.map_relinearise()
        st1 = ctx.map_stats()
        assert st1["living"] == living and st1["ids"] == living
        # ([0]: the pool of the replicated level-0 buckets — the one that holds the memory; [1]: the voxel lists, which only grow)
        assert st1["pool_cap"][0] * 2 < st0["pool_cap"][0], (st0["pool_cap"], st1["pool_cap"])
        assert st1["bytes"] < 0.6 * st0["bytes"], (st0["bytes"], st1["bytes"])
        pts = ctx.map_fetch()
        ctx.scan_set(sc["scan_xyz"])
        xa, Pa, pa, tra, sa = ctx.update(sc["x_init"], sc["P0"])
        # ... and it still takes inserts (the pools' free part was laid out for the new size)
        ctx.map_add_scan(downsample=True)
        assert ctx.map_size() > living
        ctx.update(sc["x_init"], sc["P0"])
    with capi.Context() as ctx:
        ctx.map_build(pts)
        ctx.scan_set(sc["scan_xyz"])
        xb, Pb, pb, trb, sb = ctx.update(sc["x_init"], sc["P0"])
    assert pa == pb and np.array_equal(xa, xb) and np.array_equal(Pa, Pb)
    assert [s["n_valid"] for s in sa] == [s["n_valid"] for s in sb]


def test_reobserved_ground_is_compacted_in_place_and_stays_exact(capi, oracle, scene_small):
    """The same ground scanned again and again (a sensor at rest; bench.py's cycle): every scan's points compete with the occupants of
    their 0.2 m boxes, the losers stay behind as tombstones, buckets outgrow their room by their DEAD entries and are compacted
    where they lie (lv_mapinc.hpp inc_compact_* — the device takes the ballot form of the gather, the host emulation of
    tests/test_mapinc_emulation.py the loop form).  After every scan: the map equals the oracle's point for point, in order; the
    5-NN of a probe scan — level 0, and level 1 over the tile groups' regions from a perturbed pose — equal brute force; and the
    bucket pool has not grown (no run moved, no group was laid out again, no re-linearisation)."""
    sc = scene_small
    rng = np.random.default_rng(5)
    ref = sc["map_xyz"].copy()
    scan = (sc["map_xyz"][rng.integers(0, len(ref), 12_000)] + rng.normal(0, 0.02, (12_000, 3))).astype(np.float32)
    probe = sc["scan_xyz"][:800]
    with capi.Context() as ctx:
        ctx.map_build(ref)
        used0 = None
        for step in range(12):
            batch = (scan + rng.normal(0, 0.004, scan.shape)).astype(np.float32)
            ctx.map_add(batch, downsample=True)
            ref = oracle.map_add(ref, batch, downsample=True)
            assert ctx.map_size() == len(ref)
            if step % 3 == 2 or step == 11:
                assert np.array_equal(_bits(ctx.map_fetch()), _bits(ref)), f"step {step}"
                ctx.scan_set(probe)
                for state in (sc["x_true"], sc["x_init"]):
                    ctx.iterate(state)
                    idx, d2 = ctx.fetch_knn()
                    oi, od, _, _ = oracle.knn_brute(ref, oracle.transform_scan(state, probe))
                    assert np.array_equal(idx, oi) and np.array_equal(_bits(d2), _bits(od)), f"step {step}"
            st = ctx.map_stats()
            if step == 3:
                used0 = st["pool_used"][0]
        assert st["relinearisations"] == 0
        assert st["tombstones"] > 0
        assert st["pool_used"][0] <= used0 * 1.02, (used0, st["pool_used"][0])   # (the churn of the later scans did not cost pool space)

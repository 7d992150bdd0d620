"""Host emulation of the incremental-map kernels (limo-velo_amd/csrc/lv_mapinc.hpp) against the oracle.

The kernels that maintain the map in place (ikd-Tree Add_Points with down-sampling, tombstones, relocation of
buckets, evictions — SURVEY §8 row f-1) are one-thread-per-item code with plain atomics.  tests/emu compiles that
very source for the host and runs it as loops (forward, reverse and shuffled "thread" orders), so the bookkeeping is
checked here, without a GPU: after every operation the structure's invariants must hold (tests/emu/mapinc_emu.cpp:
emu_check) and the living points, order included, must equal the oracle's lvo_map_add.  The GPU tests
(tests/test_gpu_map_add.py) then check the real launches and the searches on top.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")
LIB = os.path.join(EMU_DIR, "_build", "libmapinc_emu.so")


@pytest.fixture(scope="module")
def emu():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    srcs = [os.path.join(EMU_DIR, "mapinc_emu.cpp"), os.path.join(EMU_DIR, "hip", "hip_runtime.h"),
            os.path.join(ROOT, "limo-velo_amd", "csrc", "lv_mapinc.hpp"), os.path.join(ROOT, "limo-velo_amd", "csrc", "lv_device.hpp")]
    if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs):
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-I" + EMU_DIR, "-o", LIB, srcs[0]])
    lib = C.CDLL(LIB)
    lib.emu_create.restype = C.c_void_p
    lib.emu_create.argtypes = [C.c_float, C.c_uint32]
    lib.emu_destroy.argtypes = [C.c_void_p]
    lib.emu_set_order.argtypes = [C.c_int, C.c_uint64]
    lib.emu_add.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
    lib.emu_evict_box.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.emu_evict_box.restype = C.c_uint32
    lib.emu_evict_oldest.argtypes = [C.c_void_p, C.c_uint32]
    lib.emu_evict_oldest.restype = C.c_uint32
    lib.emu_relinearise.argtypes = [C.c_void_p]
    lib.emu_size.argtypes = [C.c_void_p]
    lib.emu_size.restype = C.c_uint32
    lib.emu_ids.argtypes = [C.c_void_p]
    lib.emu_ids.restype = C.c_uint32
    lib.emu_relinearisations.argtypes = [C.c_void_p]
    lib.emu_relinearisations.restype = C.c_uint64
    lib.emu_relocations.argtypes = [C.c_void_p]
    lib.emu_relocations.restype = C.c_uint64
    lib.emu_regrouped.argtypes = [C.c_void_p]
    lib.emu_regrouped.restype = C.c_uint64
    lib.emu_compacted.argtypes = [C.c_void_p]
    lib.emu_compacted.restype = C.c_uint64
    lib.emu_tombstones.argtypes = [C.c_void_p]
    lib.emu_tombstones.restype = C.c_uint32
    lib.emu_fetch.argtypes = [C.c_void_p, C.c_void_p]
    lib.emu_fetch.restype = C.c_uint32
    lib.emu_check.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    return lib


class Map:
    def __init__(self, lib, cell=0.5, pool_reserve=4096):
        self.lib, self.h = lib, lib.emu_create(cell, pool_reserve)

    def close(self):
        self.lib.emu_destroy(self.h)

    def add(self, pts, downsample):
        p = np.ascontiguousarray(pts, np.float32)
        self.lib.emu_add(self.h, p.ctypes.data_as(C.c_void_p), len(p), int(downsample))

    def fetch(self):
        out = np.empty((self.lib.emu_size(self.h), 3), np.float32)
        n = self.lib.emu_fetch(self.h, out.ctypes.data_as(C.c_void_p))
        assert n == len(out)
        return out

    def check(self):
        buf = C.create_string_buffer(512)
        ok = self.lib.emu_check(self.h, buf, 512)
        assert ok, buf.value.decode()

    def evict_box(self, lo, hi, keep_inside):
        lo, hi = np.asarray(lo, np.float32), np.asarray(hi, np.float32)
        return self.lib.emu_evict_box(self.h, lo.ctypes.data_as(C.c_void_p), hi.ctypes.data_as(C.c_void_p), int(keep_inside))

    def evict_oldest(self, n):
        return self.lib.emu_evict_oldest(self.h, n)


def _surface_points(rng, n, lo=-6.0, hi=6.0):
    """points near a few planes (ground + two walls), like a scan's world points"""
    p = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    which = rng.integers(0, 3, n)
    p[which == 0, 2] = rng.normal(0.0, 0.01, (which == 0).sum())
    p[which == 1, 0] = hi + rng.normal(0.0, 0.01, (which == 1).sum())
    p[which == 2, 1] = lo + rng.normal(0.0, 0.01, (which == 2).sum())
    return p.astype(np.float32)


@pytest.mark.parametrize("order", [0, 1, 2])
def test_adds_with_downsampling_equal_the_oracle(emu, oracle, order):
    emu.emu_set_order(order, 12345)
    rng = np.random.default_rng(7 + order)
    base = _surface_points(rng, 6000)
    mp = Map(emu)
    ref = base.copy()
    mp.add(base, False)            # Mapper::add on an empty map builds it
    mp.check()
    assert np.array_equal(mp.fetch(), ref)
    for step in range(12):
        lo = -6.0 + 0.7 * step     # the window drifts: revisited space and new space in every batch
        new = _surface_points(rng, 1500, lo, lo + 12.0)
        if step % 4 == 3:
            new[::50] = new[1::50][: len(new[::50])]          # exact duplicates inside the batch
            new[5] = ref[17]                                    # and of a map point
        ds = step % 5 != 4
        mp.add(new, ds)
        ref = oracle.map_add(ref, new, downsample=ds)
        mp.check()
        got = mp.fetch()
        assert got.shape == ref.shape, (step, got.shape, ref.shape)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), f"step {step}"
    assert emu.emu_tombstones(mp.h) > 0 or emu.emu_relinearisations(mp.h) > 0   # deletions did happen
    mp.close()


def test_relocation_and_overflow_paths(emu, oracle):
    """A tiny pool reserve forces buckets to move and, soon, the pool to run full (-> re-linearisation); dense
    batches into few voxels force long tails (ordering of a batch's own entries)."""
    emu.emu_set_order(2, 99)
    rng = np.random.default_rng(3)
    base = _surface_points(rng, 800, -2.0, 2.0)
    mp = Map(emu, pool_reserve=64)
    mp.add(base, False)
    ref = base.copy()
    for step in range(10):
        new = (rng.uniform(-1.0, 1.0, (400, 3)) * [1.0, 1.0, 0.02]).astype(np.float32)   # everything in a handful of voxels
        mp.add(new, step % 2 == 0)
        ref = oracle.map_add(ref, new, downsample=step % 2 == 0)
        mp.check()
        assert np.array_equal(mp.fetch().view(np.uint32), ref.view(np.uint32)), f"step {step}"
    assert emu.emu_relinearisations(mp.h) >= 1
    assert emu.emu_relocations(mp.h) >= 1
    assert emu.emu_regrouped(mp.h) >= 1   # ... and the tile groups those moves broke up were laid out again (emu_check holds them to it)
    mp.close()


def test_evictions(emu, oracle):
    emu.emu_set_order(2, 5)
    rng = np.random.default_rng(11)
    base = _surface_points(rng, 5000)
    mp = Map(emu)
    mp.add(base, False)
    ref = base.copy()
    # rolling window: keep a box
    lo, hi = np.array([-3.0, -4.0, -1.0], np.float32), np.array([6.5, 4.0, 6.5], np.float32)
    inside = np.all((ref >= lo) & (ref <= hi), axis=1)
    n = mp.evict_box(lo, hi, True)
    ref = ref[inside]
    assert n == (~inside).sum()
    mp.check()
    assert np.array_equal(mp.fetch().view(np.uint32), ref.view(np.uint32))
    # adds after an eviction (ids and ranks now differ), with down-sampling against the survivors
    for step in range(4):
        new = _surface_points(rng, 1200, -5.0, 7.0)
        mp.add(new, True)
        ref = oracle.map_add(ref, new, downsample=True)
        mp.check()
        assert np.array_equal(mp.fetch().view(np.uint32), ref.view(np.uint32)), f"step {step}"
    # the oldest third goes
    k = len(ref) // 3
    assert mp.evict_oldest(k) == k
    ref = ref[k:]
    mp.check()
    assert np.array_equal(mp.fetch().view(np.uint32), ref.view(np.uint32))
    # carve a hole, then fill it again
    n = mp.evict_box([-1.0, -1.0, -1.0], [1.0, 1.0, 1.0], False)
    hole = np.all((ref >= -1.0) & (ref <= 1.0), axis=1)
    assert n == hole.sum()
    ref = ref[~hole]
    mp.check()
    new = (rng.uniform(-1.2, 1.2, (900, 3)) * [1.0, 1.0, 0.01]).astype(np.float32)
    mp.add(new, True)
    ref = oracle.map_add(ref, new, downsample=True)
    mp.check()
    assert np.array_equal(mp.fetch().view(np.uint32), ref.view(np.uint32))
    # everything goes; Add_Points(points, true) into the empty map applies the box rule among the new points
    assert mp.evict_box([-100, -100, -100], [100, 100, 100], False) == len(ref)
    mp.check()
    dense = (rng.uniform(-1.0, 1.0, (1500, 3)) * [1.0, 1.0, 0.05]).astype(np.float32)
    mp.add(dense, True)
    mp.check()
    ref = oracle.map_add(np.zeros((0, 3), np.float32), dense, downsample=True)
    assert len(ref) < len(dense)
    assert np.array_equal(mp.fetch().view(np.uint32), ref.view(np.uint32))
    mp.close()


def test_non_finite_points_are_dropped(emu, oracle):
    emu.emu_set_order(0, 1)
    rng = np.random.default_rng(2)
    base = _surface_points(rng, 2000)
    mp = Map(emu)
    mp.add(base, False)
    new = _surface_points(rng, 500)
    bad = new.copy()
    bad[3, 0] = np.nan
    bad[10, 1] = np.inf
    bad[11, 2] = -np.inf
    mp.add(bad, True)
    good = np.delete(new, [3, 10, 11], axis=0)
    ref = oracle.map_add(base, good, downsample=True)
    mp.check()
    assert np.array_equal(mp.fetch().view(np.uint32), ref.view(np.uint32))
    mp.close()


def test_work_order_helpers_of_the_insert(emu):
    """inc_block_of: the XCD-aware workgroup -> item-block map must be a bijection of [0, n) for every grid size (a hole would
    skip work items, a collision would run them twice) that keeps the workgroups of one XCD (b % 8) on one contiguous stretch;
    inc_box_key: the Morton key of a 0.2 m box must be injective on the boxes (it is the key of the box table and of the
    grouping sort) and order boxes along the Z-curve."""
    import ctypes as C

    emu.emu_block_of.argtypes = [C.c_uint32, C.c_uint32]
    emu.emu_block_of.restype = C.c_uint32
    emu.emu_box_key.argtypes = [C.c_float, C.c_float, C.c_float, C.c_float]
    emu.emu_box_key.restype = C.c_uint64
    for n in list(range(1, 70)) + [255, 256, 257, 1000, 4099]:
        img = [emu.emu_block_of(b, n) for b in range(n)]
        assert sorted(img) == list(range(n)), n
        for x in range(min(8, n)):
            mine = sorted(img[b] for b in range(x, n, 8))
            assert mine == list(range(mine[0], mine[0] + len(mine))), (n, x)      # contiguous per XCD
    rng = np.random.default_rng(3)
    pts = rng.uniform(-300.0, 300.0, (20_000, 3)).astype(np.float32)
    box = np.floor(pts / np.float32(0.2)).astype(np.int64)
    keys = np.array([emu.emu_box_key(float(p[0]), float(p[1]), float(p[2]), 0.2) for p in pts], np.uint64)
    # same box <=> same key
    _, inv_b = np.unique(box, axis=0, return_inverse=True)
    _, inv_k = np.unique(keys, return_inverse=True)
    assert len(np.unique(inv_b)) == len(np.unique(inv_k))
    assert len(np.unique(np.stack([inv_b.ravel(), inv_k.ravel()], 1), axis=0)) == len(np.unique(inv_k))
    # Z-curve: de-interleaving the key gives back the (offset) box coordinates
    def compact(k, shift):
        v = np.zeros(len(k), np.int64)
        for bit in range(21):
            v |= ((k >> np.uint64(3 * bit + shift)) & np.uint64(1)).astype(np.int64) << bit
        return v
    off = 1 << 20
    assert np.array_equal(compact(keys, 0) - off, box[:, 0]) and np.array_equal(compact(keys, 1) - off, box[:, 1])
    assert np.array_equal(compact(keys, 2) - off, box[:, 2])


@pytest.mark.parametrize("order", [0, 2])
def test_reobserved_ground_compacts_runs_in_place(emu, oracle, order):
    """The same surface scanned again and again (what a LiDAR at rest does, and bench.py's cycle): every scan's points compete with
    the occupants of their 0.2 m boxes, the losers stay behind as tombstones, and a bucket outgrows its room by its DEAD entries.
    Such a run is compacted where it lies (inc_compact_gather / _scatter) instead of moving with its whole tile group: ascending
    ids, back-positions and the tile groups' regions are held by emu_check after every batch, the living points by the oracle."""
    emu.emu_set_order(order, 4242)
    rng = np.random.default_rng(11)
    base = _surface_points(rng, 5000, -4.0, 4.0)
    mp = Map(emu)
    mp.add(base, False)
    ref = base.copy()
    scan = _surface_points(rng, 2500, -4.0, 4.0)
    for step in range(14):
        jit = (scan + rng.normal(0, 0.004, scan.shape)).astype(np.float32)   # the same ground, sensor noise
        mp.add(jit, True)
        ref = oracle.map_add(ref, jit, downsample=True)
        mp.check()
        assert np.array_equal(mp.fetch().view(np.uint32), ref.view(np.uint32)), f"step {step}"
    assert emu.emu_compacted(mp.h) >= 1
    mp.close()

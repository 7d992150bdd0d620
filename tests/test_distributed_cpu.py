"""The N>1 path on CPU: world_size-2 `gloo` run of limo_velo_amd.distributed.ShardedUpdater.

The per-rank engine used here is built on the CPU oracle (test infrastructure) so that the sharding,
the per-pass all-reduce of the 96-double record and the pass loop are exercised without a GPU; on a
GPU node the same ShardedUpdater drives HipEngine over RCCL (bench.py --gpus N)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def pack_record(s):
    """lv_sums -> the 96-double device record layout of include/limovelo_hip.h."""
    rec = np.zeros(96)
    k = 0
    for i in range(12):
        for j in range(i, 12):
            rec[k] = s["HTH"][i, j]
            k += 1
    rec[78:90] = s["HTh"]
    rec[90] = s["n_valid"]
    rec[91] = s["sum_h2"]
    return rec


def unpack_record(rec):
    HTH = np.zeros((12, 12))
    k = 0
    for i in range(12):
        for j in range(i, 12):
            HTH[i, j] = HTH[j, i] = rec[k]
            k += 1
    return dict(HTH=HTH, HTh=rec[78:90].copy(), n_valid=int(round(rec[90])), sum_h2=float(rec[91]))


class OracleEngine:
    """CPU stand-in for HipEngine with the same begin/reduce/solve/end protocol and the device-side pass
    bookkeeping of lv_solve.hip (t, iter from -1, done)."""

    def __init__(self, lo, torch, map_xyz, max_iters=3):
        self.lo, self.torch, self.map_xyz = lo, torch, map_xyz
        self.tree = lo.KdTree(map_xyz)
        self.max_passes = max_iters + 1
        self.maximum_iter = max_iters

    def scan_set(self, pts):
        self.scan = pts

    def update_fused(self, x, P):
        xo, Po, passes, _, _ = self.lo.update(x, P, self.map_xyz, self.scan, tree=self.tree)
        return xo, Po, passes

    def begin(self, x, P):
        self.x, self.x_prop, self.P_prop, self.P_post = x.copy(), x.copy(), P.copy(), P.copy()
        self.t, self.iter, self.done, self.passes = 0, -1, False, 0

    def reduce(self):
        if self.done:
            self.rec = self.torch.zeros(96, dtype=self.torch.float64)
        elif len(self.scan) == 0:
            self.rec = self.torch.zeros(96, dtype=self.torch.float64)
        else:
            s = self.lo.iterate(self.x, self.map_xyz, self.scan, tree=self.tree, details=False)
            self.rec = self.torch.from_numpy(pack_record(s))
        return self.rec

    def solve(self):
        if self.done:
            return
        s = unpack_record(self.rec.numpy())
        self.passes += 1
        if s["n_valid"] == 0:
            self.iter += 1
            self.done = self.iter >= self.maximum_iter
            return
        xn, dx, conv, Pn = self.lo.kf_step(self.x, self.x_prop, self.P_prop, s)
        self.x = xn
        self.t += int(conv)
        last = self.t > 1 or self.iter == self.maximum_iter - 1
        self.iter += 1
        if last:
            self.P_post, self.done = Pn, True

    def end(self):
        return self.x, self.P_post, self.passes


def _worker(rank, world, port, out_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    import lvamd

    lvamd.load()
    from limo_velo_amd import synth
    from limo_velo_amd.distributed import ShardedUpdater

    import lvoracle as lo

    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = synth.make_scene(50_000, 2_001)  # odd size: uneven shards
    upd = ShardedUpdater(OracleEngine(lo, torch, sc["map_xyz"]), rank, world, dist, torch)
    upd.scan_set(sc["scan_xyz"])
    x, P, passes = upd.update(sc["x_init"], sc["P0"])
    out_q.put((rank, upd.n_local, x, P, passes))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds(lv):
    from limo_velo_amd.distributed import shard_bounds

    for n in (0, 1, 7, 2001, 65536):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1


def test_two_rank_gloo_update_equals_single_process(oracle, lv):
    import torch.multiprocessing as mp

    from limo_velo_amd import synth

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sc = synth.make_scene(50_000, 2_001)
    tree = oracle.KdTree(sc["map_xyz"])
    xo, Po, po, _, _ = oracle.update(sc["x_init"], sc["P0"], sc["map_xyz"], sc["scan_xyz"], tree=tree)
    assert res[0][1] + res[1][1] == 2001 and abs(res[0][1] - res[1][1]) <= 1
    for rank, n_local, x, P, passes in res:
        assert passes == po
        assert np.abs(x - xo).max() < 1e-10 and np.abs(P - Po).max() < 1e-12
    assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][3], res[1][3])  # ranks agree bitwise


class _GatherEngine(OracleEngine):
    """CPU stand-in for the one-launch-per-pass multi-rank form (lv_comm_set_host_gather / lv_comm_set_shard_max): every rank
    cuts its shard into the SAME number of workgroup slices (pass_kernel's geometry for the largest shard, the C-ABI's own
    lv_pass_geometry), leaves one compact 32-double partial per slice in its slot of a gather buffer, the ranks' slots are
    exchanged by the gather function limo_velo_amd.distributed.init_host_gather installs, and every rank folds world x slices
    partials in the same fixed order before its solve."""

    def comm_set_host_gather(self, rank, world, fn):   # what init_host_gather calls on a capi.Context
        self.rank, self.world, self.gather_fn = rank, world, fn

    def set_shard_max(self, n_max, capi):
        self.nwg = capi.pass_geometry(max(int(n_max), 1), 256)[0]

    @staticmethod
    def compact(s):   # the 29 live sums of the 6-column case in a 32-double record
        rec = np.zeros(32)
        k = 0
        for i in range(6):
            for j in range(i, 6):
                rec[k] = s["HTH"][i, j]
                k += 1
        rec[21:27] = s["HTh"][:6]
        rec[27] = s["n_valid"]
        rec[28] = s["sum_h2"]
        return rec

    def reduce(self):
        n = 32 * self.nwg
        slots = np.zeros(n * self.world)
        if not self.done and len(self.scan):
            cuts = np.linspace(0, len(self.scan), self.nwg + 1).astype(int)
            for g in range(self.nwg):
                if cuts[g + 1] > cuts[g]:
                    s = self.lo.iterate(self.x, self.map_xyz, self.scan[cuts[g]:cuts[g + 1]], tree=self.tree, details=False)
                    slots[self.rank * n + 32 * g:self.rank * n + 32 * (g + 1)] = self.compact(s)
        self.gather_fn(slots, n, self.rank, self.world)
        tot = np.zeros(32)
        for r in range(self.world * self.nwg):   # the fold: every record, rank-major, in order
            tot += slots[32 * r:32 * (r + 1)]
        HTH = np.zeros((12, 12))
        k = 0
        for i in range(6):
            for j in range(i, 6):
                HTH[i, j] = HTH[j, i] = tot[k]
                k += 1
        HTh = np.zeros(12)
        HTh[:6] = tot[21:27]
        self.rec = self.torch.from_numpy(pack_record(dict(HTH=HTH, HTh=HTh, n_valid=int(round(tot[27])), sum_h2=float(tot[28]))))
        return self.rec


def _gather_worker(rank, world, port, n_scan, out_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    import lvamd

    lvamd.load()
    from limo_velo_amd import capi, synth
    from limo_velo_amd.distributed import init_host_gather, shard_bounds

    import lvoracle as lo

    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = synth.make_scene(50_000, max(n_scan, 8))
    scan = sc["scan_xyz"][:n_scan]
    eng = _GatherEngine(lo, torch, sc["map_xyz"])
    init_host_gather(eng, dist, torch, rank, world)           # the product's gather function, on a stand-in context
    lo_, hi_ = shard_bounds(n_scan, rank, world)
    eng.scan_set(scan[lo_:hi_])
    l0, h0 = shard_bounds(n_scan, 0, world)
    eng.set_shard_max(h0 - l0, capi)                          # every rank: the largest shard (rank 0's)
    eng.begin(sc["x_init"], sc["P0"])
    for _ in range(eng.max_passes):
        eng.reduce()
        eng.solve()
    x, P, passes = eng.end()
    out_q.put((rank, hi_ - lo_, eng.nwg, x, P, passes))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_scan", [2001, 1])
def test_two_rank_gloo_one_launch_form(oracle, lv, n_scan):
    """The protocol of the one-launch-per-pass multi-rank form on CPU (world size 2, gloo): same slice count on both ranks
    from the largest shard, partial slots exchanged by the gather function the product installs (init_host_gather), fixed-order
    fold — ranks bitwise equal, the single-process update within 1e-10; uneven shards and an empty one."""
    import torch.multiprocessing as mp

    from limo_velo_amd import synth

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, n_scan, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sc = synth.make_scene(50_000, max(n_scan, 8))
    tree = oracle.KdTree(sc["map_xyz"])
    xo, Po, po, _, _ = oracle.update(sc["x_init"], sc["P0"], sc["map_xyz"], sc["scan_xyz"][:n_scan], tree=tree)
    assert res[0][1] + res[1][1] == n_scan and res[0][2] == res[1][2] >= 1
    for rank, n_local, nwg, x, P, passes in res:
        assert passes == po
        assert np.abs(x - xo).max() < 1e-10 and np.abs(P - Po).max() < 1e-12
    assert np.array_equal(res[0][3], res[1][3]) and np.array_equal(res[0][4], res[1][4])  # ranks agree bitwise


def _run_world(target, world, args, timeout=300):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + tuple(args) + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=timeout) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def _allreduce_worker_n(rank, world, port, n_scan, out_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), OMP_NUM_THREADS="1")
    import torch
    import torch.distributed as dist

    import lvamd

    lvamd.load()
    from limo_velo_amd import synth
    from limo_velo_amd.distributed import ShardedUpdater

    import lvoracle as lo

    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = synth.make_scene(50_000, max(n_scan, 8))
    upd = ShardedUpdater(OracleEngine(lo, torch, sc["map_xyz"]), rank, world, dist, torch)
    upd.scan_set(sc["scan_xyz"][:n_scan])
    x, P, passes = upd.update(sc["x_init"], sc["P0"])
    out_q.put((rank, upd.n_local, x, P, passes))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_scan", [2001, 5])
def test_eight_rank_gloo_both_forms(oracle, lv, n_scan):
    """World size 8 (the node the north star names) on CPU over gloo, both multi-rank forms' protocols — the per-pass all-reduce
    of the 96-double record (ShardedUpdater, what bench.py falls back to) and the one-launch form's gather of workgroup partials
    (init_host_gather, the product's gather function; the same slots and fixed-order fold as the RCCL all-gather and the
    peer-mapped pull) — with uneven shards (2001 = 251 x 1 + 250 x 7) and with EMPTY shards (5 points: ranks 5-7 hold none and
    still take part in every exchange).  All eight ranks bitwise equal; the single-process update within 1e-10."""
    from limo_velo_amd import synth
    from limo_velo_amd.distributed import shard_bounds

    sc = synth.make_scene(50_000, max(n_scan, 8))
    tree = oracle.KdTree(sc["map_xyz"])
    xo, Po, po, _, _ = oracle.update(sc["x_init"], sc["P0"], sc["map_xyz"], sc["scan_xyz"][:n_scan], tree=tree)
    tol_P = 1e-12 if n_scan > 100 else 1e-9     # (five matches: a barely determined solve, conditioning 1e3 worse)
    want = [shard_bounds(n_scan, r, 8)[1] - shard_bounds(n_scan, r, 8)[0] for r in range(8)]
    res = _run_world(_allreduce_worker_n, 8, (n_scan,))
    assert [r[1] for r in res] == want and sum(want) == n_scan
    for rank, n_local, x, P, passes in res:
        assert passes == po and np.abs(x - xo).max() < 1e-10 and np.abs(P - Po).max() < tol_P
        assert np.array_equal(x, res[0][2]) and np.array_equal(P, res[0][3])
    res = _run_world(_gather_worker, 8, (n_scan,))
    assert [r[1] for r in res] == want and len({r[2] for r in res}) == 1
    for rank, n_local, nwg, x, P, passes in res:
        assert passes == po and np.abs(x - xo).max() < 1e-10 and np.abs(P - Po).max() < tol_P
        assert np.array_equal(x, res[0][3]) and np.array_equal(P, res[0][4])

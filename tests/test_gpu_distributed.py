"""The multi-GPU engine (HipEngine, split-form C-ABI, RCCL all-reduce on the torch tensor that backs the sums
record) exercised on ONE GPU: a world-size-1 `nccl` group drives exactly the code path bench.py uses for
--gpus N, and must reproduce the single-GPU update (same arithmetic, different but fixed summation order of the
block partials: group records + final record here, one direct fold inside solve_kernel there)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_split_path_with_rccl_world1(lv, scene_small):
    import torch
    import torch.distributed as dist

    from limo_velo_amd import capi
    from limo_velo_amd.distributed import HipEngine, ShardedUpdater

    sc = scene_small
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        with capi.Context() as ref:
            ref.map_build(sc["map_xyz"])
            ref.scan_set(sc["scan_xyz"])
            x1, P1, p1, _, _ = ref.update(sc["x_init"], sc["P0"])
        with capi.Context() as ctx:
            ctx.map_build(sc["map_xyz"])
            eng = HipEngine(ctx, torch, multi=True)  # torch tensor as sums record, torch's stream
            upd = ShardedUpdater(eng, 0, 2, dist, torch)  # world=2 logic: takes the all-reduce branch ...
            upd.world = 1  # ... but shard as one rank
            upd.scan_set(sc["scan_xyz"])
            upd.world = 2
            for _ in range(3):
                x2, P2, p2 = upd.update(sc["x_init"], sc["P0"])
            torch.cuda.synchronize()
            rec = eng.sums.cpu().numpy()
        assert p1 == p2
        # summation-order difference only: ~1e-16 relative on H^T H -> far below 1e-12 on the state / covariance
        np.testing.assert_allclose(x2, x1, rtol=0, atol=1e-12)
        np.testing.assert_allclose(P2, P1, rtol=1e-9, atol=1e-15)
        assert rec[90] > 1500  # n_valid of the last pass sits in the all-reduced record
    finally:
        dist.destroy_process_group()


def test_library_communicator_world1(lv, scene_small):
    """lv_comm_init (RCCL bound and driven by the library itself) with a one-rank communicator: the update takes
    the multi-GPU route (rank record -> ncclAllReduce on the context stream -> solve from the record) and must
    reproduce the plain single-GPU update bit for bit (same fold order, the all-reduce of one rank is a copy)."""
    import torch

    from limo_velo_amd import capi
    from limo_velo_amd.distributed import torch_rccl_path

    sc = scene_small
    torch.cuda.set_device(0)
    path = torch_rccl_path(torch)
    with capi.Context() as ref:
        ref.map_build(sc["map_xyz"])
        ref.scan_set(sc["scan_xyz"])
        x0, P0, p0, _, _ = ref.update(sc["x_init"], sc["P0"])   # one launch per pass (the single-GPU default)
        assert ref.last_update_fused()
        ref.set_fused_pass(False)                               # the three-kernel pass: what the communicator route runs
        x1, P1, p1, _, _ = ref.update(sc["x_init"], sc["P0"])
        assert not ref.last_update_fused()
    with capi.Context() as ctx:
        ctx.map_build(sc["map_xyz"])
        ctx.scan_set(sc["scan_xyz"])
        assert ctx.comm_world() == 1
        uid = ctx.comm_unique_id(path)
        assert len(uid) == 128 and any(uid)
        ctx.comm_init(uid, 0, 1, path)
        for _ in range(3):
            x2, P2, p2, _, _ = ctx.update(sc["x_init"], sc["P0"], want_trace=False)
        # resident-filter route through the same pass loop
        ctx.filter_set(sc["x_init"], sc["P0"])
        p3 = ctx.correct()
        x3, P3 = ctx.filter_get()
        # the one-launch-per-pass form with a communicator: told the largest shard, the update all-gathers the workgroup
        # partials instead (one rank: in place, a copy) — the very kernel, buffers and fold of the multi-GPU route
        ctx.scan_set(sc["scan_xyz"])
        ctx.comm_set_shard_max(len(sc["scan_xyz"]))
        for _ in range(2):
            x5, P5, p5, _, s5 = ctx.update(sc["x_init"], sc["P0"])
        assert ctx.last_update_fused()
        ctx.filter_set(sc["x_init"], sc["P0"])
        p6 = ctx.correct()
        assert ctx.last_update_fused()
        x6, P6 = ctx.filter_get()
        # a rank whose shard is smaller than the largest one launches the same grid: workgroups without a tile contribute
        # zero partials (same sums, folded in a different grouping) ...
        ctx.scan_set(sc["scan_xyz"])
        ctx.comm_set_shard_max(3 * len(sc["scan_xyz"]) + 17)
        x7, P7, p7, _, s7 = ctx.update(sc["x_init"], sc["P0"])
        assert ctx.last_update_fused()
        # ... and a rank without any point still runs every launch and every collective
        ctx.scan_set(sc["scan_xyz"][:0])
        ctx.comm_set_shard_max(len(sc["scan_xyz"]))
        x8, P8, p8, _, s8 = ctx.update(sc["x_init"], sc["P0"])
        assert ctx.last_update_fused() and p8 == 4 and [s["n_valid"] for s in s8] == [0] * 4 and np.array_equal(x8, sc["x_init"])
        ctx.scan_set(sc["scan_xyz"])
        ctx.comm_set_shard_max(len(sc["scan_xyz"]))
        ctx.set_comm_fused(False)
        ctx.update(sc["x_init"], sc["P0"], want_trace=False)
        assert not ctx.last_update_fused()
        ctx.set_comm_fused(True)
        ctx.scan_set(sc["scan_xyz"])                      # a new scan forgets the shard size: three-kernel form until told again
        ctx.update(sc["x_init"], sc["P0"], want_trace=False)
        assert not ctx.last_update_fused()
        ctx.comm_destroy()
        x4, P4, p4, _, _ = ctx.update(sc["x_init"], sc["P0"], want_trace=False)
    assert p0 == p1 == p2 == p3 == p4
    assert np.array_equal(x1, x2) and np.array_equal(P1, P2)
    assert np.array_equal(x1, x3) and np.array_equal(P1, P3)
    # without a communicator the update is one launch per pass again: same arithmetic, the workgroup partials are summed
    # in a different (fixed) order -> ~1e-16 relative on H^T H
    assert np.array_equal(x0, x4) and np.array_equal(P0, P4)
    # gathered partials of one rank = its own partials in the same order: bit for bit the single-GPU one-launch update
    assert p5 == p6 == p0 and np.array_equal(x0, x5) and np.array_equal(P0, P5)
    assert np.array_equal(x0, x6) and np.array_equal(P0, P6)
    assert p7 == p0 and [s["n_valid"] for s in s7] == [s["n_valid"] for s in s5]
    np.testing.assert_allclose(x7, x0, rtol=0, atol=1e-12)
    np.testing.assert_allclose(P7, P0, rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(x0, x1, rtol=0, atol=1e-12)
    np.testing.assert_allclose(P0, P1, rtol=1e-9, atol=1e-15)


def _world2_worker(rank, world, port, n_scan, out_q):
    """One rank of a world-size-2 run with BOTH ranks on GPU 0: HIP engine, split-form C-ABI, the 96-double record
    all-reduced through a gloo group (staged through host memory — RCCL refuses two ranks on one device)."""
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    torch.cuda.init()
    torch.cuda.set_device(0)
    import lvamd

    lvamd.load()
    from limo_velo_amd import capi, synth
    from limo_velo_amd.distributed import HipEngine, ShardedUpdater

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = synth.make_scene(50_000, max(n_scan, 8))
        scan = sc["scan_xyz"][:n_scan]
        with capi.Context() as ctx:
            ctx.map_build(sc["map_xyz"])
            eng = HipEngine(ctx, torch, multi=True, host_staged=True)
            upd = ShardedUpdater(eng, rank, world, dist, torch)
            upd.scan_set(scan)
            for _ in range(2):
                x, P, passes = upd.update(sc["x_init"], sc["P0"])
            torch.cuda.synchronize()
        out_q.put((rank, upd.n_local, x, P, passes))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_scan", [2001, 1])
def test_world2_hip_engine_on_one_gpu(lv, n_scan):
    """The sharded HIP path with world = 2 (uneven shards: 1001 + 1000 points; one EMPTY shard: 1 + 0): both ranks end
    bitwise equal, and within 1e-10 of the single-process update of the whole scan (only the summation order of
    the record differs)."""
    import torch.multiprocessing as mp

    from limo_velo_amd import capi, synth

    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_world2_worker, args=(r, 2, port, n_scan, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    sc = synth.make_scene(50_000, max(n_scan, 8))
    with capi.Context() as ref:
        ref.map_build(sc["map_xyz"])
        ref.scan_set(sc["scan_xyz"][:n_scan])
        x1, P1, p1, _, _ = ref.update(sc["x_init"], sc["P0"])
    assert res[0][1] + res[1][1] == n_scan and abs(res[0][1] - res[1][1]) <= 1
    assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][3], res[1][3]) and res[0][4] == res[1][4]
    for _, _, x, P, passes in res:
        assert passes == p1
        assert np.abs(x - x1).max() < 1e-10
        assert np.abs(P - P1).max() < 1e-10 * max(1.0, np.abs(P1).max())


def _world2_gather_worker(rank, world, port, n_scan, out_q, ext=0):
    """One rank of a world-size-2 run with BOTH ranks on GPU 0, one launch per pass: the workgroup partials of the two
    ranks are all-gathered through a gloo group (lv_comm_set_host_gather) — the kernels, gather buffers, geometry and fold
    are the ones the RCCL route (lv_comm_init + lv_comm_set_shard_max) runs on N GPUs."""
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    torch.cuda.init()
    torch.cuda.set_device(0)
    import lvamd

    lvamd.load()
    from limo_velo_amd import capi, synth
    from limo_velo_amd.distributed import HipEngine, ShardedUpdater, init_host_gather

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = synth.make_scene(50_000, max(n_scan, 8), extrinsics="xaloc" if ext else "identity")
        scan = sc["scan_xyz"][:n_scan]
        with capi.Context(capi.default_params(estimate_extrinsics=ext)) as ctx:
            if ext:
                ctx.set_option("fused_ext", 1)   # (12-column rows: 96-double partials in the gather slots)
            ctx.map_build(sc["map_xyz"])
            init_host_gather(ctx, dist, torch, rank, world)
            upd = ShardedUpdater(HipEngine(ctx, torch, multi=False, library_comm=True), rank, world, dist, torch)
            upd.scan_set(scan)
            for _ in range(2):
                x, P, passes = upd.update(sc["x_init"], sc["P0"])
            fused = ctx.last_update_fused()
            # resident-filter route through the same launches
            ctx.filter_set(sc["x_init"], sc["P0"])
            p2 = ctx.correct()
            x2, P2 = ctx.filter_get()
            # without the largest shard there is no transport for the three-kernel form: a clean error, no hang
            ctx.scan_set(scan[:0] if upd.n_local == 0 else scan[:upd.n_local])
            try:
                ctx.update(sc["x_init"], sc["P0"], want_trace=False)
                refused = False
            except Exception:  # noqa: BLE001
                refused = True
        out_q.put((rank, upd.n_local, x, P, passes, fused, np.array_equal(x, x2) and np.array_equal(P, P2) and p2 == passes, refused))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_scan,ext", [(2001, 0), (1, 0), (20_000, 0), (2001, 1), (20_000, 1)])
def test_world2_one_launch_form_on_one_gpu(lv, n_scan, ext):
    """The one-launch-per-pass multi-rank form with world = 2 (uneven shards 1001 + 1000; one EMPTY shard 1 + 0; 10 000 +
    10 000: several workgroups per rank): both ranks end bitwise equal and within 1e-10 of the single-process update of
    the whole scan (the partials of the two shards are folded in a different grouping)."""
    import torch.multiprocessing as mp

    from limo_velo_amd import capi, synth

    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_world2_gather_worker, args=(r, 2, port, n_scan, q, ext)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    sc = synth.make_scene(50_000, max(n_scan, 8), extrinsics="xaloc" if ext else "identity")
    with capi.Context(capi.default_params(estimate_extrinsics=ext)) as ref:
        if ext:
            ref.set_option("fused_ext", 1)
        ref.map_build(sc["map_xyz"])
        ref.scan_set(sc["scan_xyz"][:n_scan])
        x1, P1, p1, _, _ = ref.update(sc["x_init"], sc["P0"])
    assert res[0][1] + res[1][1] == n_scan and abs(res[0][1] - res[1][1]) <= 1
    assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][3], res[1][3]) and res[0][4] == res[1][4]
    # (12-column solve: the different grouping of the partial sums is amplified by the conditioning of the extrinsic columns,
    # as between the two single-GPU routes in test_gpu_pass_kernel.py::test_extrinsics_variant)
    tol_x, tol_P = (1e-10, 1e-10) if not ext else (1e-9, 1e-6)
    for _, _, x, P, passes, fused, same_filter, refused in res:
        assert fused and same_filter and refused
        assert passes == p1
        assert np.abs(x - x1).max() < tol_x
        assert np.abs(P - P1).max() < tol_P * max(1.0, np.abs(P1).max())


def _world2_peer_worker(rank, world, port, n_scan, out_q, ext=0, absent_peer=False):
    """One rank of a world-size-2 run with BOTH ranks on GPU 0 over PEER-MAPPED memory (lv_comm_peer_export / _init: HIP IPC):
    after every pass one small kernel publishes this rank's partials and pulls the other rank's slot straight out of its
    buffers.  absent_peer: rank 1 maps the buffers but never runs an update — rank 0's wait must end with an error, not hang."""
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if absent_peer:
        os.environ["LV_PEER_TIMEOUT_MS"] = "200"
    import torch
    import torch.distributed as dist

    torch.cuda.init()
    torch.cuda.set_device(0)
    import lvamd

    lvamd.load()
    from limo_velo_amd import capi, synth
    from limo_velo_amd.distributed import HipEngine, ShardedUpdater, init_peer_gather

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = synth.make_scene(50_000, max(n_scan, 8), extrinsics="xaloc" if ext else "identity")
        scan = sc["scan_xyz"][:n_scan]
        with capi.Context(capi.default_params(estimate_extrinsics=ext)) as ctx:
            if ext:
                ctx.set_option("fused_ext", 1)
            ctx.map_build(sc["map_xyz"])
            init_peer_gather(ctx, dist, rank, world)
            ctx_handles = bytes(capi.PEER_HANDLE_BYTES)
            upd = ShardedUpdater(HipEngine(ctx, torch, multi=False, library_comm=True), rank, world, dist, torch)
            upd.scan_set(scan)
            if absent_peer:
                err = None
                if rank == 0:
                    try:
                        ctx.comm_peer_init(rank, world, [ctx_handles] * world)   # a second init on a live exchange: refused
                        err = "second lv_comm_peer_init was accepted"
                    except Exception as e:  # noqa: BLE001
                        assert "already set up" in str(e), str(e)
                    import ctypes as C

                    x0, P0, np_ = np.ascontiguousarray(sc["x_init"]).copy(), np.ascontiguousarray(sc["P0"]).copy(), C.c_int(0)
                    rc = ctx.lib.lv_update(ctx.h, x0.ctypes.data_as(C.c_void_p), P0.ctypes.data_as(C.c_void_p), C.byref(np_), None, None)
                    if rc != 0:
                        err = err or ctx.lib.lv_last_error().decode()
                    assert np.array_equal(x0, sc["x_init"]) and np.array_equal(P0, sc["P0"])   # nothing of the failed update came back
                    try:                      # sticky: the context refuses further work until the exchange is torn down
                        ctx.filter_set(sc["x_init"], sc["P0"])
                        ctx.correct()
                        err = "lv_correct after a failed exchange was accepted"
                    except Exception as e:  # noqa: BLE001
                        assert "peer-mapped gather" in str(e), str(e)
                dist.barrier()
                out_q.put((rank, err))
                return
            for _ in range(3):     # (several updates: the flag words only grow, the buffers are reused)
                x, P, passes = upd.update(sc["x_init"], sc["P0"])
            fused = ctx.last_update_fused()
            ctx.filter_set(sc["x_init"], sc["P0"])
            p2 = ctx.correct()
            x2, P2 = ctx.filter_get()
        out_q.put((rank, upd.n_local, x, P, passes, fused, np.array_equal(x, x2) and np.array_equal(P, P2) and p2 == passes))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_scan,ext", [(2001, 0), (1, 0), (20_000, 0), (2001, 1)])
def test_world2_peer_mapped_gather_on_one_gpu(lv, n_scan, ext):
    """The one-launch-per-pass multi-rank form with the partials exchanged through peer-mapped memory (HIP IPC; across GPUs the
    same mapping goes over xGMI): both ranks end bitwise equal and within 1e-10 of the single-process update of the whole
    scan; lv_correct takes the same launches."""
    import torch.multiprocessing as mp

    from limo_velo_amd import capi, synth

    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_world2_peer_worker, args=(r, 2, port, n_scan, q, ext)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    sc = synth.make_scene(50_000, max(n_scan, 8), extrinsics="xaloc" if ext else "identity")
    with capi.Context(capi.default_params(estimate_extrinsics=ext)) as ref:
        if ext:
            ref.set_option("fused_ext", 1)
        ref.map_build(sc["map_xyz"])
        ref.scan_set(sc["scan_xyz"][:n_scan])
        x1, P1, p1, _, _ = ref.update(sc["x_init"], sc["P0"])
    assert res[0][1] + res[1][1] == n_scan and abs(res[0][1] - res[1][1]) <= 1
    assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][3], res[1][3]) and res[0][4] == res[1][4]
    tol_x, tol_P = (1e-10, 1e-10) if not ext else (1e-9, 1e-6)
    for _, _, x, P, passes, fused, same_filter in res:
        assert fused and same_filter
        assert passes == p1
        assert np.abs(x - x1).max() < tol_x
        assert np.abs(P - P1).max() < tol_P * max(1.0, np.abs(P1).max())


def test_peer_mapped_gather_times_out_instead_of_hanging(lv):
    """A rank that never publishes its partials (here: it simply does not run the update) must not hang the others: the pull
    kernel gives up after LV_PEER_TIMEOUT_MS (default 2 s; 200 ms here) and lv_update reports LV_ESTATE with the caller's state untouched; a
    second lv_comm_peer_init on a live exchange is refused."""
    import torch.multiprocessing as mp

    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_world2_peer_worker, args=(r, 2, port, 2001, q, 0, True)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0][1] is not None and "did not publish" in res[0][1]


@pytest.mark.parametrize("n_scan", [2001, 3])
def test_world4_peer_mapped_gather_on_one_gpu(lv, n_scan):
    """Four ranks on ONE GPU over the peer-mapped exchange: every rank publishes once and pulls THREE
    peers' slots per pass; with 3 points one rank holds none and still publishes / pulls.  All four ranks bitwise equal, within
    1e-10 of the single-process update.  (Same device: the ranks share an L2 — the protocol and its bookkeeping are what this
    exercises; no run across GPUs exists yet.)"""
    import torch.multiprocessing as mp

    from limo_velo_amd import capi, synth

    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_world2_peer_worker, args=(r, 4, port, n_scan, q, 0)) for r in range(4)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    sc = synth.make_scene(50_000, max(n_scan, 8))
    with capi.Context() as ref:
        ref.map_build(sc["map_xyz"])
        ref.scan_set(sc["scan_xyz"][:n_scan])
        x1, P1, p1, _, _ = ref.update(sc["x_init"], sc["P0"])
    assert sum(r[1] for r in res) == n_scan and max(r[1] for r in res) - min(r[1] for r in res) <= 1
    for _, _, x, P, passes, fused, same_filter in res:
        assert fused and same_filter and passes == p1
        assert np.array_equal(x, res[0][2]) and np.array_equal(P, res[0][3])
        assert np.abs(x - x1).max() < 1e-10 and np.abs(P - P1).max() < 1e-10 * max(1.0, np.abs(P1).max())

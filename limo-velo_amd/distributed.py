"""Multi-GPU form of the iterated update: scan points sharded across ranks, map replicated, one
all-reduce(sum) of the 96-double record per IKFoM pass, everything else local.

The sharding / collective / loop logic is engine-agnostic so that it is exercised on CPU with the
`gloo` backend (tests/test_distributed_cpu.py plugs a CPU engine built on the oracle — test
infrastructure); the product engine is `HipEngine` over the C-ABI, and it is the only engine this
package ships.  One process per GPU; `dist` is torch.distributed (backend "nccl" == RCCL on ROCm).
"""
from __future__ import annotations

import numpy as np

SUMS_LEN = 96


def shard_bounds(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced split of n scan points: ranks < n % world get one extra point."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def torch_rccl_path(torch) -> str:
    """The librccl torch itself loaded (<torch>/lib/librccl.so): the library must bind the same copy."""
    import os

    return os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")


def init_library_comm(ctx, dist, torch, rank: int, world: int):
    """Creates the library-side RCCL communicator (lv_comm_init): rank 0 makes the id, torch.distributed
    carries the 128 bytes to the other ranks, then every rank joins.  After this, plain ctx.update() on every
    rank IS the multi-GPU update (all-reduce issued by the library on its own stream, no host round trips)."""
    path = torch_rccl_path(torch)
    try:  # every rank probes the binding first, so a rank that cannot load librccl does not leave the others
        uid, ok = ctx.comm_unique_id(path), 1  # waiting inside a collective
    except Exception:  # noqa: BLE001
        uid, ok = None, 0
    if world > 1:
        t = torch.tensor([ok], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = int(t.item())
    if not ok:
        raise RuntimeError("librccl could not be bound on every rank")
    box = [uid if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    ctx.comm_init(box[0], rank, world, path)


def init_host_gather(ctx, dist, torch, rank: int, world: int):
    """The one-launch-per-pass multi-rank form without librccl: the workgroup partials of the ranks are all-gathered
    through `dist` on CPU tensors (lv_comm_set_host_gather).  For process groups RCCL cannot serve — two ranks on one
    GPU in the tests, a `gloo` group — with exactly the kernels, buffers and fold of the RCCL route."""

    def gather(slots, n, r, w):
        mine = torch.from_numpy(slots[r * n:(r + 1) * n].copy())
        outs = [torch.empty(n, dtype=torch.float64) for _ in range(w)]
        dist.all_gather(outs, mine)
        for q in range(w):
            if q != r:
                slots[q * n:(q + 1) * n] = outs[q].numpy()

    ctx.comm_set_host_gather(rank, world, gather)


def init_peer_gather(ctx, dist, rank: int, world: int):
    """The one-launch-per-pass multi-rank form over peer-mapped memory (lv_comm_peer_export / lv_comm_peer_init): every rank
    exports the HIP IPC handle of its gather buffers, `dist` (any backend: the handles are 128 plain bytes) carries them to
    all ranks, every rank maps the others'.  After this, plain ctx.update() on every rank is the multi-GPU update with a
    one-shot peer read per pass instead of a collective."""
    mine = ctx.comm_peer_export()
    handles = [None] * world
    dist.all_gather_object(handles, mine)
    ctx.comm_peer_init(rank, world, handles)


class HipEngine:
    """Per-rank engine over the C-ABI split form (lv_update_begin / lv_pass_reduce / lv_pass_solve /
    lv_update_end).  The sums record lives in a torch tensor so RCCL can reduce it in place."""

    def __init__(self, ctx, torch, multi: bool, library_comm: bool = False, host_staged: bool = False):
        # library_comm: the context owns an RCCL communicator (init_library_comm) — ctx.update() is already the
        # multi-GPU update; multi: the collective is torch.distributed's, driven pass by pass from Python
        self.ctx, self.torch, self.multi = ctx, torch, multi and not library_comm
        self.library_comm = library_comm
        # host_staged: the record crosses the process boundary through host memory (a CPU process group such as
        # gloo): used to run several ranks on ONE GPU, where RCCL refuses duplicate devices
        self.host_staged = host_staged
        multi = self.multi
        self.max_passes = ctx.params.MAX_NUM_ITERS + 1
        self.sums = None
        self.stream = None
        if multi:
            # One explicit (non-default) torch stream carries BOTH the library's kernels and the point RCCL
            # orders its collective against: lv_set_stream(NULL) would mean "the context's own stream", which
            # torch knows nothing about.
            self.stream = torch.cuda.Stream()
            with torch.cuda.stream(self.stream):
                self.sums = torch.zeros(SUMS_LEN, dtype=torch.float64, device=f"cuda:{torch.cuda.current_device()}")
            self.stream.synchronize()
            ctx.set_sums_buffer(self.sums.data_ptr())
            ctx.set_stream(self.stream.cuda_stream)

    def stream_ctx(self):
        import contextlib

        return self.torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def scan_set(self, pts):
        self.ctx.scan_set(pts)

    def update_fused(self, x, P):
        xo, Po, passes, _, _ = self.ctx.update(x, P, want_trace=False)
        return xo, Po, passes

    def begin(self, x, P):
        self.ctx.update_begin(x, P)

    def reduce(self):
        self.ctx.pass_reduce()
        return self.sums

    def allreduce(self, rec, dist):
        if not self.host_staged:
            dist.all_reduce(rec, op=dist.ReduceOp.SUM)   # RCCL, ordered on the engine's stream
            return
        host = rec.cpu()                                  # synchronises the engine's stream
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        rec.copy_(host)                                   # enqueued on the engine's stream, before the solve

    def solve(self):
        self.ctx.pass_solve()

    def end(self):
        return self.ctx.update_end()


class ShardedUpdater:
    def __init__(self, ctx_or_engine, rank: int, world: int, dist, torch):
        self.rank, self.world, self.dist = rank, world, dist
        if hasattr(ctx_or_engine, "begin"):
            self.engine = ctx_or_engine
        else:
            self.engine = HipEngine(ctx_or_engine, torch, multi=world > 1)
        self.n_local = 0

    def scan_set(self, scan_xyz):
        scan_xyz = np.ascontiguousarray(scan_xyz, dtype=np.float32)
        lo, hi = shard_bounds(len(scan_xyz), self.rank, self.world)
        self.n_local = hi - lo
        self.engine.scan_set(scan_xyz[lo:hi])
        if getattr(self.engine, "library_comm", False):
            # the library's own communicator: tell it the largest shard (rank 0's), so that every rank sizes the
            # one-launch-per-pass grid alike (lv_comm_set_shard_max)
            lo0, hi0 = shard_bounds(len(scan_xyz), 0, self.world)
            self.engine.ctx.comm_set_shard_max(hi0 - lo0)

    def update(self, x, P):
        """Returns (x_post, P_post, passes).  Every rank computes the identical posterior: the solve
        consumes only the all-reduced record, which is bitwise identical on all ranks."""
        if self.world == 1 or getattr(self.engine, "library_comm", False):
            return self.engine.update_fused(x, P)
        ctx = self.engine.stream_ctx() if hasattr(self.engine, "stream_ctx") else None
        if ctx is None:
            import contextlib

            ctx = contextlib.nullcontext()
        with ctx:
            self.engine.begin(x, P)
            for _ in range(self.engine.max_passes):
                rec = self.engine.reduce()
                if hasattr(self.engine, "allreduce"):
                    self.engine.allreduce(rec, self.dist)
                else:
                    self.dist.all_reduce(rec, op=self.dist.ReduceOp.SUM)
                self.engine.solve()
            return self.engine.end()

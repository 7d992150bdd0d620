"""Per-kernel profile target for the map insert (run under rocprofv3 --kernel-trace --stats): MODE=same — the headline cycle's
case (bench.py cycle_64k: the same 65 536-point scan inserted again and again: nearly every point loses against the occupant of
its 0.2 m box) — or MODE=new — twelve different scans (about a third of each survives).  Prints the synchronised wall time per
insert as well."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lvamd  # noqa: E402

lvamd.load()
from limo_velo_amd import capi, synth  # noqa: E402

M, N = 1_048_576, 65_536
mode = os.environ.get("MODE", "same")
sc = synth.make_scene(M, N)
extra = [synth.make_extra_scan(M, N, k) for k in range(12)] if mode == "new" else None
with capi.Context() as ctx:
    ctx.map_build(sc["map_xyz"])
    ts, dead = [], []
    for k in range(12):
        e = extra[k] if extra else dict(scan_xyz=sc["scan_xyz"], x_init=sc["x_init"])
        ctx.scan_set(e["scan_xyz"])
        ctx.update(e["x_init"], sc["P0"], want_trace=False)
        ctx.synchronize()
        tomb0 = ctx.map_stats()["tombstones"]
        t0 = time.perf_counter()
        ctx.map_add_scan(downsample=True)
        ctx.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        dead.append((ctx.map_stats()["tombstones"] - tomb0) // 28)   # (28 entries per deleted point: its 27 level-0 runs + its voxel list)
    print(mode, "insert ms (synchronised):", [round(t, 3) for t in ts], "map", ctx.map_size(), "deleted per insert", dead)

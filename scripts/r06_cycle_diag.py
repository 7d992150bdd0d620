"""Round 6 diagnostic: which level decides the scan's points before / after the scan was inserted into the map, and what an update costs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lvamd; lvamd.load()
from limo_velo_amd import capi, synth
sc = synth.make_scene(1_048_576, 65_536)
with capi.Context() as ctx:
    ctx.map_build(sc["map_xyz"]); ctx.scan_set(sc["scan_xyz"])
    def upd():
        for _ in range(3): ctx.update(sc["x_init"], sc["P0"], want_trace=False)
        ctx.synchronize(); a = time.perf_counter()
        for _ in range(20): ctx.update(sc["x_init"], sc["P0"], want_trace=False)
        ctx.synchronize(); return (time.perf_counter() - a) / 20 * 1e3
    ctx.iterate(sc["x_init"]); h1 = ctx.level_histogram(); ctx.iterate(sc["x_true"]); h2 = ctx.level_histogram()
    print("fresh build: perturbed", h1[:6], "converged", h2[:6], "update ms", round(upd(), 4), ctx.map_stats()["slots_used"][:2])
    for rep in range(int(os.environ.get('REPS', '3'))):
        ctx.update(sc["x_init"], sc["P0"], want_trace=False)
        ctx.synchronize(); t0 = time.perf_counter(); ctx.map_add_scan(True); ctx.synchronize(); t_add = (time.perf_counter() - t0) * 1e3
        ctx.iterate(sc["x_init"]); h1 = ctx.level_histogram(); ctx.iterate(sc["x_true"]); h2 = ctx.level_histogram()
        print("after insert", rep, ": perturbed", h1[:6], "converged", h2[:6], "update ms", round(upd(), 4), "size", ctx.map_size(), "add ms", round(t_add, 3), {k: ctx.map_stats()[k] for k in ("pool_used", "pool_cap", "relinearisations", "tombstones")})

"""Host-side cost of enqueuing one update (13 kernel launches) vs its device time: is the loop launch-bound?"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import lvamd; lvamd.load()
from limo_velo_amd import capi, synth
sc = synth.make_scene(1_048_576, 65_536)
ctx = capi.Context(); ctx.map_build(sc["map_xyz"]); ctx.scan_set(sc["scan_xyz"])
for _ in range(20):
    ctx.update(sc["x_init"], sc["P0"], want_trace=False)
n = 200
enq = tot = 0.0
for _ in range(n):
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.update_begin(sc["x_init"], sc["P0"])
    for _ in range(4):
        ctx.pass_reduce(); ctx.pass_solve()
    t1 = time.perf_counter()
    ctx.update_end()
    t2 = time.perf_counter()
    enq += t1 - t0; tot += t2 - t0
print("enqueue of one update (split API from Python, 1 + 4 x 4 launches): %.1f us; until results: %.1f us" % (enq / n * 1e6, tot / n * 1e6))

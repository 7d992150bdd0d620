"""Device memory of the map structure per million points (lv_map_get_stats) + build / insert timing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import lvamd; lvamd.load()
from limo_velo_amd import capi, synth
for m in (1_048_576, 4_000_000):
    sc = synth.make_scene(m, 65_536)
    with capi.Context() as ctx:
        free0 = torch.cuda.mem_get_info()[0]
        t0 = time.perf_counter(); ctx.map_build(sc["map_xyz"]); ctx.synchronize(); t1 = time.perf_counter()
        ctx.map_build(sc["map_xyz"]); ctx.synchronize(); t2 = time.perf_counter()
        st = ctx.map_stats()
        free1 = torch.cuda.mem_get_info()[0]
        ctx.scan_set(sc["scan_xyz"])
        ctx.update(sc["x_init"], sc["P0"], want_trace=False)
        ts = []
        for i in range(6):
            ctx.synchronize(); a = time.perf_counter(); ctx.map_add_scan(True); ctx.synchronize(); ts.append(time.perf_counter() - a)
        print(f"map {m}: stats bytes {st['bytes'] / 1e9:.3f} GB ({st['bytes'] / m * 1e6 / 1e9:.2f} GB per million), device memory taken {(free0 - free1) / 1e9:.3f} GB, "
              f"build {1e3 * (t1 - t0):.1f} ms first / {1e3 * (t2 - t1):.1f} ms warm, map_add_scan(64k) ms {[round(1e3 * t, 2) for t in ts]}, size {ctx.map_size()}")

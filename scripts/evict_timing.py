"""lv_map_evict_box on the headline map (1 M points) and on a larger one: by runs (inc_evict_sweep_kernel) vs by points
(per-point search of the 81 runs), a rolling-window step (a box that moves by a few metres: a thin shell goes) and a big cut."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lvamd; lvamd.load()
from limo_velo_amd import capi, synth

M = int(os.environ.get("LV_EVICT_MAP", 4_000_000))
sc = synth.make_scene(M, 4096)
xyz = sc["map_xyz"]
lo0, hi0 = xyz.min(axis=0), xyz.max(axis=0)
out = {"map_points": int(len(xyz))}
for sweep in (1, 0):
    with capi.Context() as ctx:
        ctx.set_option("sweep_evict", sweep)
        ctx.map_build(xyz)
        ctx.synchronize()
        rec = {}
        # rolling window: the box shrinks by 2 % of the extent per step on one side (a shell of the map goes each time)
        ext = hi0 - lo0
        lo, hi = lo0 + 0.1 * ext, hi0 - 0.1 * ext      # (a window inside the map, as a sensor-centred rolling window is)
        t0 = time.perf_counter()
        rec["first_cut_evicted"] = ctx.map_evict_box(lo, hi, keep_inside=True)
        rec["first_cut_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
        ts, ns = [], []
        for step in range(6):
            lo[0] += 0.02 * (hi0[0] - lo0[0])
            t0 = time.perf_counter()
            n = ctx.map_evict_box(lo, hi, keep_inside=True)
            ts.append((time.perf_counter() - t0) * 1e3); ns.append(n)
        rec["shell_steps_ms"] = [round(t, 3) for t in ts]; rec["shell_steps_evicted"] = ns
        c = 0.5 * (lo0 + hi0); h = 0.25 * (hi0 - lo0)
        t0 = time.perf_counter()
        n = ctx.map_evict_box(c - h, c + h, keep_inside=True)
        rec["big_cut_ms"] = round((time.perf_counter() - t0) * 1e3, 3); rec["big_cut_evicted"] = n
        rec["living_after"] = ctx.map_size()
        out["by_runs" if sweep else "by_points"] = rec
print(json.dumps(out))

"""Timing of the mapping step (row f-1): lv_map_add / lv_map_add_scan of a 64k-point scan into a 1M-point map."""
import json
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lvamd  # noqa: E402

lvamd.load()
from limo_velo_amd import capi, synth  # noqa: E402

M, N = 1_048_576, 65_536
sc = synth.make_scene(M, N)
extra = [synth.make_extra_scan(M, N, k) for k in range(12)]
out = {}
with capi.Context() as ctx:
    t0 = time.perf_counter()
    ctx.map_build(sc["map_xyz"])
    out["map_build_ms"] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    ctx.map_build(sc["map_xyz"])
    out["map_build_again_ms"] = (time.perf_counter() - t0) * 1e3
    # device-resident mapping step: scan + posterior on the device
    times, sizes = [], []
    for e in extra:
        ctx.scan_set(e["scan_xyz"])
        x, P, passes, _, _ = ctx.update(e["x_init"], sc["P0"], want_trace=False)
        ctx.synchronize()
        t0 = time.perf_counter()
        ctx.map_add_scan(downsample=True)
        times.append((time.perf_counter() - t0) * 1e3)
        sizes.append(ctx.map_size())
    out["map_add_scan_ms"] = [round(t, 3) for t in times]
    out["map_sizes"] = sizes
    st = ctx.map_stats()
    out["stats_after_scan_adds"] = st
    # host-provided points (validated + staged through pinned memory)
    world = (sc["map_xyz"][np.random.default_rng(1).integers(0, M, N)] + np.float32(0.03)).astype(np.float32)
    t0 = time.perf_counter()
    ctx.map_add(world, downsample=True)
    out["map_add_host_points_ms"] = (time.perf_counter() - t0) * 1e3
    # the update on the incrementally grown map vs a fresh build of the same points
    ctx.scan_set(sc["scan_xyz"])
    for _ in range(20):
        ctx.update(sc["x_init"], sc["P0"], want_trace=False)
    t0 = time.perf_counter()
    for _ in range(200):
        ctx.update(sc["x_init"], sc["P0"], want_trace=False)
    out["update_ms_on_incremental_map"] = (time.perf_counter() - t0) / 200 * 1e3
    t0 = time.perf_counter()
    ctx.map_relinearise()
    out["relinearise_ms"] = (time.perf_counter() - t0) * 1e3
    for _ in range(20):
        ctx.update(sc["x_init"], sc["P0"], want_trace=False)
    t0 = time.perf_counter()
    for _ in range(200):
        ctx.update(sc["x_init"], sc["P0"], want_trace=False)
    out["update_ms_after_relinearise"] = (time.perf_counter() - t0) / 200 * 1e3
    t0 = time.perf_counter()
    n = ctx.map_evict_box([-30, -30, -5], [30, 30, 20], keep_inside=True)
    out["evict_box_ms"] = (time.perf_counter() - t0) * 1e3
    out["evicted"] = n
    out["stats_end"] = ctx.map_stats()
print(json.dumps(out))

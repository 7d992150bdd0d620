"""PCIe-inclusive timing: lv_scan_set (host scan in: repack + H2D + Morton sort) + lv_update (state out)."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (torch runtime first)
import lvamd; lvamd.load()
from limo_velo_amd import capi, synth
sc = synth.make_scene(1_048_576, 65_536)
ctx = capi.Context(); ctx.map_build(sc["map_xyz"])
for _ in range(5):
    ctx.scan_set(sc["scan_xyz"]); ctx.update(sc["x_init"], sc["P0"], want_trace=False)
ctx.synchronize()
n = 100
t0 = time.perf_counter()
for _ in range(n):
    ctx.scan_set(sc["scan_xyz"]); ctx.synchronize()
t1 = time.perf_counter()
for _ in range(n):
    ctx.update(sc["x_init"], sc["P0"], want_trace=False)
t2 = time.perf_counter()
for _ in range(n):
    ctx.scan_set(sc["scan_xyz"]); ctx.update(sc["x_init"], sc["P0"], want_trace=False)
t3 = time.perf_counter()
print("scan_set ms %.3f | update ms %.3f | scan_set+update ms %.3f -> %.0f iters/s PCIe-inclusive" % (
    (t1 - t0) / n * 1e3, (t2 - t1) / n * 1e3, (t3 - t2) / n * 1e3, 4 * n / (t3 - t2)))
t0 = time.perf_counter(); ctx.map_build(sc["map_xyz"]); print("map_build (1M pts) ms %.1f" % ((time.perf_counter() - t0) * 1e3))
# row f-1: one Mapper::add of a 64k-point scan (world frame) into the 1M-point map, with ikd-Tree down-sampling
import numpy as np
new_pts = (sc["map_xyz"][:65_536] + np.float32(0.013)).astype(np.float32)
ctx.synchronize(); t0 = time.perf_counter(); ctx.map_add(new_pts, downsample=True); ctx.synchronize()
print("map_add (64k pts into 1M, downsample) ms %.1f -> map size %d" % ((time.perf_counter() - t0) * 1e3, ctx.map_size()))

import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import lvamd; lvamd.load()
from limo_velo_amd import synth
import lvoracle as lo
sc = synth.make_scene(1_048_576, 65_536)
t0 = time.perf_counter(); tree = lo.KdTree(sc["map_xyz"]); print("kd build s", time.perf_counter() - t0)
for th in (1, 3, 8, 16, 32, 64, 128, 256):
    if th > (os.cpu_count() or 1): break
    lo.update(sc["x_init"], sc["P0"], sc["map_xyz"], sc["scan_xyz"], tree=tree, nthreads=th)
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 3.0:
        _, _, p, _, _ = lo.update(sc["x_init"], sc["P0"], sc["map_xyz"], sc["scan_xyz"], tree=tree, nthreads=th); n += p
    print("threads", th, "passes/s", n / (time.perf_counter() - t0))

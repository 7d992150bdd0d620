"""What one rank of an N-GPU run sees: the 64k-point scan cut to 64k / N points against the full 1M-point map
(no collective: an upper bound for the strong-scaling run), plus larger scans."""
import sys, time
import ctypes as C
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import lvamd; lvamd.load()
from limo_velo_amd import capi, synth
SIZES = [int(v) for v in os.environ.get("SIZES", "512,2048,8192,16384,32768,65536,131072,196608,262144").split(",")]
sc = synth.make_scene(1_048_576, max(262_144, max(SIZES)))
ctx = capi.Context(); ctx.map_build(sc["map_xyz"])
if len(sys.argv) > 1:
    ctx.set_fused_pass(int(sys.argv[1]) != 0)   # 0: the three-kernel pass, 1: one launch per pass (the default), 2: ... whatever the rounds
    if int(sys.argv[1]) == 2:
        ctx.set_option("fused_multi_round", 1)
for n in SIZES:
    ctx.scan_set(sc["scan_xyz"][:n])
    for _ in range(10):
        ctx.update(sc["x_init"], sc["P0"], want_trace=False)
    reps = 100 if n <= 262144 else 30
    ctx.synchronize(); t0 = time.perf_counter(); p = 0
    for _ in range(reps):
        p += ctx.update(sc["x_init"], sc["P0"], want_trace=False)[2]
    dt = time.perf_counter() - t0
    # the same update on the device-resident filter, enqueued back to back (one synchronisation per region: bench.py's step)
    x0 = np.ascontiguousarray(sc["x_init"], np.float64); P0 = np.ascontiguousarray(sc["P0"], np.float64)
    x0p, P0p = x0.ctypes.data_as(C.c_void_p), P0.ctypes.data_as(C.c_void_p)
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        if ctx.lib.lv_filter_set(ctx.h, x0p, P0p) or ctx.lib.lv_correct(ctx.h, None):
            raise RuntimeError(ctx.lib.lv_last_error().decode())
    ctx.synchronize(); dr = time.perf_counter() - t0
    print("fused" if ctx.last_update_fused() else "3-kernel", "scan %7d points: %.1f us per update, %.0f iterations/s, %.2f Gpoint-passes/s; resident, back to back: %.1f us per update" % (n, dt / reps * 1e6, p / dt, p * n / reps / (dt / reps) / 1e9 / (p / reps) * (p / reps), dr / reps * 1e6))

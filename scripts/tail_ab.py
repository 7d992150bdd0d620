"""A/B of lv_set_option "fused_tail": synchronised / pipelined step rates at the headline size, alternating in one process."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import lvamd
lvamd.load()
from limo_velo_amd import capi, synth

sc = synth.make_scene(1_048_576, 65_536)
x0 = np.ascontiguousarray(sc["x_init"]); P0 = np.ascontiguousarray(sc["P0"])
xp, Pp = x0.ctypes.data_as(C.c_void_p), P0.ctypes.data_as(C.c_void_p)
xg, Pg = np.zeros(26), np.zeros(529)
xgp, Pgp = xg.ctypes.data_as(C.c_void_p), Pg.ctypes.data_as(C.c_void_p)
with capi.Context() as ctx:
    ctx.map_build(sc["map_xyz"]); ctx.scan_set(sc["scan_xyz"])
    lib, h = ctx.lib, ctx.h
    for tail in (0, 1, 0, 1, 0, 1):
        ctx.set_option("fused_tail", tail)
        for _ in range(50):
            lib.lv_filter_set(h, xp, Pp); lib.lv_correct(h, None); lib.lv_filter_get(h, xgp, Pgp)
        res = []
        for mode in ("sync", "pipelined"):
            best = []
            for _ in range(5):
                ctx.synchronize(); t0 = time.perf_counter()
                for _ in range(200):
                    lib.lv_filter_set(h, xp, Pp); lib.lv_correct(h, None)
                    if mode == "sync": lib.lv_filter_get(h, xgp, Pgp)
                ctx.synchronize()
                best.append((time.perf_counter() - t0) / 200)
            res.append(sorted(best)[2])
        print(f"fused_tail={tail}: synchronised {res[0] * 1e6:7.2f} us per update ({4 / res[0]:8.0f} it/s)   pipelined {res[1] * 1e6:7.2f} us ({4 / res[1]:8.0f} it/s)")

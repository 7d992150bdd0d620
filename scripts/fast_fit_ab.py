"""A/B of the opt-in approximate plane fit: synchronised step rate and in-kernel phase stamps (fits + partial) at the headline
size, exact vs fast_fit, alternating in one process.  LV_PASS_CLK=1 must be set (phase stamps)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (runtime order: see tests/conftest.py)
import lvamd
lvamd.load()
from limo_velo_amd import capi, synth

sc = synth.make_scene(1_048_576, 65_536)
with capi.Context() as ctx:
    ctx.map_build(sc["map_xyz"]); ctx.scan_set(sc["scan_xyz"])
    for fast in (0, 1, 0, 1):
        ctx.set_option("fast_fit", fast)
        for _ in range(20):
            ctx.update(sc["x_init"], sc["P0"], want_trace=False)
        acc, t0 = [], time.perf_counter()
        for i in range(200):
            ctx.filter_set(sc["x_init"], sc["P0"]); ctx.correct(want_passes=False); ctx.filter_get()
        dt = (time.perf_counter() - t0) / 200
        for i in range(20):
            ctx.update(sc["x_init"], sc["P0"], want_trace=False)
            clk, nwg = ctx.pass_clocks()
            w = clk[:, :nwg, 16:].astype(np.float64) / 100.0
            acc.append([[np.median(w[li, :, 3] - w[li, :, 0]), np.median(w[li, :, 6] - w[li, :, 3]), np.median(w[li, :, 9] - w[li, :, 6]),
                         w[li, :, 9].max() - w[li, :, 0].min()] for li in range(clk.shape[0] - 1)])
        m = np.median(np.array(acc), axis=0)
        print(f"fast_fit={fast}: {4 / dt:9.0f} it/s ({dt * 1e6:6.1f} us per update, synchronised, with phase stamps on) | per launch: prologue "
              f"{np.round(m[:, 0], 2)} search {np.round(m[:, 1], 2)} fits+partial {np.round(m[:, 2], 2)} span {np.round(m[:, 3], 2)}")

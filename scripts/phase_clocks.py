"""Per-phase shader-clock breakdown of search_kernel / fit_reduce_kernel / solve_kernel (instrumentation, GPU only)."""
import sys

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lvamd

lvamd.load()
from limo_velo_amd import capi, synth

lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sc = synth.make_scene(1_048_576, 65_536)
ctx = capi.Context(capi.default_params(lanes_per_query=lanes))
ctx.map_build(sc["map_xyz"])
ctx.scan_set(sc["scan_xyz"])
for _ in range(3):
    ctx.update(sc["x_init"], sc["P0"], want_trace=False)
ctx.set_profiling(2)
names = ["search: scan+pose", "search: probe", "search: stream+select", "search: winners+store", "(kernel boundary)", "fit: record+fit", "fit: contraction"]
for state, label in ((sc["x_true"], "converged pose"), (sc["x_init"], "perturbed pose")):
    ctx.update_begin(state, sc["P0"])
    ctx.pass_reduce()
    clk2 = ctx.phase_clocks()
    ctx.update_end()
    nb = len(clk2) // 2
    clk, wall = clk2[:nb], clk2[nb:]
    w0, w1 = wall[:, 0], wall[:, 1]
    print(label, "WALL(100MHz ticks): first start 0, last start %d, first end %d, last end %d  (=%.1f us span)" % (
        w0.max() - w0.min(), w1.min() - w0.min(), w1.max() - w0.min(), (w1.max() - w0.min()) / 100.0))
    d = np.diff(clk, axis=1).astype(np.float64)
    tot = clk[:, 7] - clk[:, 0]
    span = clk[:, 7].max() - clk[:, 0].min()
    print(label, "blocks", len(clk), "| per-block total cycles mean %.0f max %.0f | first start -> last end %.0f" % (tot.mean(), tot.max(), span))
    print("   start spread (cycles):", clk[:, 0].max() - clk[:, 0].min())
    for i, nm in enumerate(names):
        print("   %-24s mean %8.0f  p50 %8.0f  max %8.0f" % (nm, d[:, i].mean(), np.median(d[:, i]), d[:, i].max()))

ctx.set_profiling(0)
ctx.update(sc["x_init"], sc["P0"], want_trace=False)
sclk = ctx.solve_clocks()
stamps = {0: "loads issued", 1: "fold + HTH", 4: "Pr, W = A1 + HTH", 5: "gauss-jordan", 6: "X, K, dx", 7: "boxplus", 8: "store + pose consts", 9: "terminal P"}
for p in range(4):
    row, prev = [], None
    for i in sorted(stamps):
        v = int(sclk[p, i])
        if v == 0:
            continue
        if prev is not None:
            row.append("%s %d" % (stamps[i], v - prev))
        prev = v
    print("solve pass", p, "|", " | ".join(row))

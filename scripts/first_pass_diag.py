"""First pass of the benchmark (pose off by 0.1 m / 0.8 deg): level histogram of a capturing pass and the time of a one-pass update."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import lvamd; lvamd.load()
from limo_velo_amd import capi, synth
sc = synth.make_scene(1_048_576, 65_536)
with capi.Context(capi.default_params(MAX_NUM_ITERS=0)) as ctx:
    ctx.map_build(sc["map_xyz"]); ctx.scan_set(sc["scan_xyz"])
    ctx.iterate(sc["x_init"])
    print("level histogram (capturing pass):", ctx.level_histogram(), "fallback", ctx.timing()["fallback_queries"])
    for _ in range(20): ctx.update(sc["x_init"], sc["P0"], want_trace=False)
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(200): ctx.update(sc["x_init"], sc["P0"], want_trace=False)
    ctx.synchronize(); print("one-pass update us:", (time.perf_counter() - t0) / 200 * 1e6, "fallback per update", ctx.timing()["fallback_queries"])
    ctx.set_profiling(True); ctx.update(sc["x_init"], sc["P0"], want_trace=False); print("kernel us", ctx.timing()["pass_match_ms"][0] * 1e3); ctx.set_profiling(False)
    # the same from the converged pose (no coarse points at all)
    for _ in range(20): ctx.update(sc["x_true"], sc["P0"], want_trace=False)
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(200): ctx.update(sc["x_true"], sc["P0"], want_trace=False)
    ctx.synchronize(); print("one-pass update from the true pose us:", (time.perf_counter() - t0) / 200 * 1e6)

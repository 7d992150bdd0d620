"""Diagnostic (library built with -DLV_MEMO_STATS, LV_LIB_PATH=scripts/ab/memo_stats.so): how many level-0 probes of the headline
update the voxel memo answers."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import lvamd; lvamd.load()
from limo_velo_amd import capi, synth
sc = synth.make_scene(1_048_576, 65_536)
ctx = capi.Context(); ctx.map_build(sc["map_xyz"]); ctx.scan_set(sc["scan_xyz"])
h0 = ctx.level_histogram()
x, P, p, tr, sm = ctx.update(sc["x_init"], sc["P0"])
h1 = ctx.level_histogram()
d = [b - a for a, b in zip(h0, h1)]
print("passes", p, "fused", ctx.last_update_fused(), "per launch (hits, misses):", [(d[2 * i], d[2 * i + 1]) for i in range(1, 4)])
tr = np.asarray(tr)
print("state change per pass (position, m):", [float(np.abs(tr[i][:3]).max()) for i in range(len(tr))] if tr.ndim == 2 else tr.shape)
print("cell", ctx.map_stats())

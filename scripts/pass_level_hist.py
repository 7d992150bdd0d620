"""Which level decides the scan's points in each pass of the headline update (capturing launches at the states the update went through)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lvamd; lvamd.load()
from limo_velo_amd import capi, synth
sc = synth.make_scene(1_048_576, 65_536)
with capi.Context() as ctx:
    ctx.map_build(sc["map_xyz"]); ctx.scan_set(sc["scan_xyz"])
    x, P, passes, tr, sums = ctx.update(sc["x_init"], sc["P0"])
    states = [sc["x_init"]] + [tr[i][23:49].copy() for i in range(passes - 1)]
    for i, st in enumerate(states):
        ctx.iterate(st)
        print("pass", i, "decided at level 0 / 1 / lists (3 x 3 x 3) / lists (larger) / every id / none:", ctx.level_histogram()[:6])

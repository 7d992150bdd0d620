import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lvamd; lvamd.load()
from limo_velo_amd import capi, synth
sc = synth.make_scene(1_048_576, 65_536)
for vox in ([float(v) for v in sys.argv[1:]] or [0.5, 0.6, 0.7]):
    ctx = capi.Context(capi.default_params(voxel_size=vox))
    ctx.map_build(sc["map_xyz"]); ctx.scan_set(sc["scan_xyz"])
    ctx.iterate(sc["x_init"]); h1 = ctx.level_histogram()
    ctx.iterate(sc["x_true"]); h2 = ctx.level_histogram()
    print("voxel", vox, "perturbed pose:", h1[:5], " converged pose:", h2[:5])
    ctx.close()

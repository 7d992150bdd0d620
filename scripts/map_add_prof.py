"""Per-kernel profile target: a few device-resident mapping steps (run under rocprofv3 --kernel-trace --stats)."""
import sys

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lvamd  # noqa: E402

lvamd.load()
from limo_velo_amd import capi, synth  # noqa: E402

M, N = 1_048_576, 65_536
sc = synth.make_scene(M, N)
extra = [synth.make_extra_scan(M, N, k) for k in range(6)]
with capi.Context() as ctx:
    ctx.map_build(sc["map_xyz"])
    for e in extra:
        ctx.scan_set(e["scan_xyz"])
        ctx.update(e["x_init"], sc["P0"], want_trace=False)
        ctx.map_add_scan(downsample=True)
    print(ctx.map_stats())

"""Wall-clock phase stamps of the CLOSING launch of an update (one workgroup: fold + solve + terminal books) by scan size:
LV_PASS_CLK=1 python scripts/closing_clocks.py [sizes...]"""
import os, sys
os.environ.setdefault("LV_PASS_CLK", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lvamd
lvamd.load()
from limo_velo_amd import capi, synth

sizes = [int(v) for v in sys.argv[1:]] or [2048, 16384, 65536]
sc = synth.make_scene(1_048_576, max(sizes))
W = 16
with capi.Context() as ctx:
    ctx.map_build(sc["map_xyz"])
    for n in sizes:
        ctx.scan_set(sc["scan_xyz"][:n])
        rows = []
        for i in range(25):
            ctx.update(sc["x_init"], sc["P0"])
            clk, nwg = ctx.pass_clocks()
            if i >= 5:
                k = clk[-1, 0]
                rows.append([(k[W + j] - k[W + 0]) / 100 for j in (1, 2, 10)])
        m = np.median(np.array(rows), axis=0)
        print(f"{n:7d} points, {nwg:3d} partials: fold done at {m[0]:.2f} us | W + gauss-jordan at {m[1]:.2f} | gain, [+], terminal books, mailbox: end at {m[2]:.2f} us "
              f"(from the workgroup's first instruction)")

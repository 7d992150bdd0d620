"""Phase stamps of pass_kernel (one launch per pass): LV_PASS_CLK=1 python scripts/pass_clocks.py [MAX_NUM_ITERS ...]
Prints, for the LAST searching launch of an update, the median / p90 / max duration of every phase over the search
workgroups, in shader cycles and in wall-clock microseconds, plus the launch-wide picture (first start, last end)."""
import os, sys
os.environ.setdefault("LV_PASS_CLK", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lvamd
lvamd.load()
from limo_velo_amd import capi, synth

sc = synth.make_scene(1_048_576, 65_536)
names = ["prologue", "search step 0", "search step 1", "barrier", "fit rows", "contraction", "tail"]
for npass in (int(a) for a in (sys.argv[1:] or ["3"])):
    prm = capi.default_params(MAX_NUM_ITERS=npass)
    with capi.Context(prm) as c2:
        c2.map_build(sc["map_xyz"])
        c2.scan_set(sc["scan_xyz"])
        c2.set_fused_pass(True)
        for _ in range(5):
            c2.update(sc["x_init"], sc["P0"])
        assert c2.last_update_fused()
        clk = c2.pass_clocks()
    sh, wl = clk[:, :8], clk[:, 8:]
    print(f"== MAX_NUM_ITERS={npass}: last searching launch, {len(clk)} workgroups")
    t0 = wl[:, 0].min()
    print(f"   launch: first start 0, last start {(wl[:,0].max()-t0)/100:.2f} us, first end {(wl[:,7].min()-t0)/100:.2f} us, last end {(wl[:,7].max()-t0)/100:.2f} us")
    for i, nm in enumerate(names):
        d = (sh[:, i + 1] - sh[:, i]).astype(np.float64)
        w = (wl[:, i + 1] - wl[:, i]).astype(np.float64) / 100.0
        print(f"   {nm:14s} cycles med {np.median(d):8.0f} p90 {np.percentile(d,90):8.0f} max {d.max():8.0f} | us med {np.median(w):6.2f} p90 {np.percentile(w,90):6.2f} max {w.max():6.2f}")

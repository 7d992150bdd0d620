"""Phase stamps of pass_kernel (one launch per pass): LV_PASS_CLK=1 python scripts/pass_clocks.py
Prints the launch-level timeline of one update (wall clock: first workgroup start, last workgroup end, gap to the
previous launch) and, per launch, the median / p90 / max duration of every phase over the workgroups."""
import os, sys
os.environ.setdefault("LV_PASS_CLK", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lvamd
lvamd.load()
from limo_velo_amd import capi, synth

sc = synth.make_scene(1_048_576, 65_536)
names = ["fold", "W + gauss-jordan", "gain, [+], consts", "search step 0", "search step 1", "barrier", "fit rows", "contraction", "tail"]
with capi.Context() as ctx:
    ctx.map_build(sc["map_xyz"])
    ctx.scan_set(sc["scan_xyz"])
    ctx.set_fused_pass(True)
    for _ in range(5):
        ctx.update(sc["x_init"], sc["P0"])
    assert ctx.last_update_fused()
    clk, n = ctx.pass_clocks()
W = 16
t0 = clk[0, :n, W].min()
prev_end = None
for li in range(clk.shape[0]):
    closing = li == clk.shape[0] - 1
    nw = 1 if closing else n
    sl = clk[li, :nw]
    start = sl[:, W]
    s0 = start.min()
    keeper_end = sl[nw - 1, W + 10]
    e1 = max(keeper_end, 0 if closing else sl[:, W + 9].max())
    gap = "" if prev_end is None else f"gap to previous launch's last end {(s0 - prev_end) / 100:.2f} us"
    print(f"launch {li}: first start {(s0 - t0) / 100:8.2f} us, last end {(e1 - t0) / 100:8.2f} us, span {(e1 - s0) / 100:6.2f} us  {gap}")
    prev_end = e1
    if closing:
        continue
    se = sl[:, W + 9]
    print(f"      workgroups: start 0 .. {(start.max() - s0) / 100:.2f} us; search+fit end {(se.min() - s0) / 100:.2f} .. {(se.max() - s0) / 100:.2f}; "
          f"bookkeeper: search+fit end {(sl[nw - 1, W + 9] - s0) / 100:.2f}, books done {(keeper_end - s0) / 100:.2f}")
    sh = sl[:, :W]
    first = 3 if li == 0 else 0
    for i in range(first, 9):
        d = (sh[:, i + 1] - sh[:, i]).astype(np.float64)
        print(f"      {names[i]:18s} cycles med {np.median(d):8.0f} p90 {np.percentile(d, 90):8.0f} max {d.max():8.0f}")

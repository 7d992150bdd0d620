"""Phase stamps of pass_kernel (one launch per pass): LV_PASS_CLK=1 python scripts/pass_clocks.py [n_updates]
Medians over n_updates updates (headline workload) of: the launch-level timeline (wall clock: span from the first workgroup
start to the last end, gap to the previous launch) and, per launch, the phase durations (median / p90 / max over the
workgroups, in shader cycles) and the bookkeeping workgroup's extra time."""
import os, sys
os.environ.setdefault("LV_PASS_CLK", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lvamd
lvamd.load()
from limo_velo_amd import capi, synth

NU = int(sys.argv[1]) if len(sys.argv) > 1 else 30
sc = synth.make_scene(1_048_576, 65_536)
names = ["fold", "W + gauss-jordan", "gain, [+], consts", "search: wavefront 0, first task(s)", "search: wavefront 0, later tasks", "barrier", "fit rows", "contraction", "tail"]
W = 16
runs = []
with capi.Context() as ctx:
    ctx.map_build(sc["map_xyz"])
    ctx.scan_set(sc["scan_xyz"])
    ctx.set_fused_pass(True)
    for _ in range(5):
        ctx.update(sc["x_init"], sc["P0"])
    for _ in range(NU):
        ctx.update(sc["x_init"], sc["P0"])
        assert ctx.last_update_fused()
        clk, n = ctx.pass_clocks()
        runs.append(clk.copy())
nl = runs[0].shape[0]
med = lambda v: float(np.median(v))
tot = []
for li in range(nl):
    closing = li == nl - 1
    nw = 1 if closing else n
    spans, gaps, books, ends_min, ends_max, keep_end = [], [], [], [], [], []
    phases = [[] for _ in range(9)]
    bk = []
    fits = []
    for clk in runs:
        sl = clk[li, :nw]
        s0 = sl[:, W].min()
        ki = int(np.argmax(sl[:, W + 10]))     # the launch's bookkeeping workgroup (the only one with stamp 10)
        kend = sl[ki, W + 10]
        e1 = max(kend, 0 if closing else sl[:, W + 9].max())
        spans.append((e1 - s0) / 100)
        if li:
            pl = clk[li - 1, :(n if li - 1 < nl - 1 else 1)]
            pe = max(pl[:, W + 10].max(), pl[:, W + 9].max())
            gaps.append((s0 - pe) / 100)
        if not closing:
            se = sl[:, W + 9]
            ends_min.append((se.min() - s0) / 100); ends_max.append((se.max() - s0) / 100)
            keep_end.append((sl[ki, W + 9] - s0) / 100)
            kk = sl[ki]
            books.append((kend - kk[W + 9]) / 100)     # the books follow the bookkeeper's own search and fits
            fits.append((kk[W + 8] - kk[W + 6]) / 100)
            if kk[W + 11] > 0:
                bk.append([(kk[W + 11] - kk[W + 9]) / 100, (kk[W + 12] - kk[W + 11]) / 100, (kk[W + 13] - kk[W + 12]) / 100, (kk[W + 10] - kk[W + 13]) / 100])
            sh = sl[:, :W]
            for i in range(3 if li == 0 else 0, 9):
                d = (sh[:, i + 1] - sh[:, i]).astype(np.float64)
                phases[i].append((np.median(d), np.percentile(d, 90), d.max()))
    tot.append(med(spans) + (med(gaps) if gaps else 0))
    g = f"gap {med(gaps):5.2f}" if gaps else "         "
    if closing:
        print(f"launch {li} (closing): span {med(spans):6.2f} us  {g}")
        continue
    print(f"launch {li}: span {med(spans):6.2f} us (min {min(spans):6.2f} max {max(spans):6.2f})  {g}  workgroups' search+fit end {med(ends_min):6.2f} .. {med(ends_max):6.2f}; "
          f"bookkeeper: end {med(keep_end):6.2f} + books {med(books):5.2f} us")
    if bk:
        b = np.median(np.array(bk), axis=0)
        print(f"      books: copy + stores + identity {b[0]:.2f} us, manifold blocks {b[1]:.2f}, dx_new + congruence + stores {b[2]:.2f}, gauss-jordan + store {b[3]:.2f}")
    for i in range(3 if li == 0 else 0, 9):
        a = np.array(phases[i])
        print(f"      {names[i]:18s} cycles med {np.median(a[:,0]):8.0f} p90 {np.median(a[:,1]):8.0f} max {np.median(a[:,2]):8.0f}")
print(f"sum of spans + gaps: {sum(tot):.1f} us")

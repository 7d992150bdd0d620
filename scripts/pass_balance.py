"""How evenly pass_kernel's workgroups finish: LV_PASS_CLK=1 python scripts/pass_balance.py [n_updates]
Per launch: the distribution of the workgroups' search + fit end (us from the launch's first start), its correlation with the
workgroup index / XCD (bid % 8), and how stable a workgroup's end is from update to update (stable = the tiles it was dealt,
not the machine)."""
import os, sys
os.environ.setdefault("LV_PASS_CLK", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lvamd
lvamd.load()
from limo_velo_amd import capi, synth

NU = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sc = synth.make_scene(1_048_576, 65_536)
W = 16
ends, starts, searches = [], [], []
with capi.Context() as ctx:
    ctx.map_build(sc["map_xyz"])
    ctx.scan_set(sc["scan_xyz"])
    for _ in range(5):
        ctx.update(sc["x_init"], sc["P0"])
    for _ in range(NU):
        ctx.update(sc["x_init"], sc["P0"])
        clk, n = ctx.pass_clocks()
        c = clk[:-1, :n].astype(np.float64)
        s0 = c[:, :, W].min(axis=1, keepdims=True)
        ends.append((c[:, :, W + 9] - s0) / 100)
        starts.append((c[:, :, W] - s0) / 100)
        searches.append((c[:, :, W + 6] - c[:, :, W + 3]) / 100)   # prologue end .. barrier after the search
ends, starts, searches = np.array(ends), np.array(starts), np.array(searches)   # [update, launch, wg]
for li in range(ends.shape[1]):
    e, st, se = ends[:, li], starts[:, li], searches[:, li]
    m = np.median(e, axis=0)                    # a workgroup's typical end
    pct = np.percentile(m, [0, 10, 50, 90, 100])
    bid = np.arange(m.size)
    xcd = [float(np.median(m[bid % 8 == x])) for x in range(8)]
    resid = e - m                               # update-to-update jitter of one workgroup
    print(f"launch {li}: end of search+fit per workgroup  min {pct[0]:.2f} p10 {pct[1]:.2f} med {pct[2]:.2f} p90 {pct[3]:.2f} max {pct[4]:.2f} us;"
          f" mean {m.mean():.2f}; start spread {np.median(st.max(axis=1)):.2f}")
    print(f"      search phase alone: min {np.median(se, axis=0).min():.2f} med {np.median(se):.2f} max {np.median(se, axis=0).max():.2f}")
    print(f"      corr(end, bid) {np.corrcoef(m, bid)[0, 1]:+.2f}; by XCD {[round(x, 2) for x in xcd]}; jitter of one workgroup (std) {resid.std():.2f} us,"
          f" spread of the typical ends (std) {m.std():.2f} us")
    stm = np.median(st, axis=0)                 # a workgroup's typical start after the launch's first
    print("      typical START by XCD (bid % 8):", [round(float(np.median(stm[bid % 8 == x])), 2) for x in range(8)],
          " by bid octile:", [round(float(stm[i].mean()), 2) for i in np.array_split(np.argsort(bid), 8)], f" max {stm.max():.2f}")
    q = np.array_split(np.argsort(bid), 8)
    print("      typical end by bid octile:", [round(float(m[i].mean()), 2) for i in q])
    slow = np.argsort(-m)[:8]
    print("      slowest workgroups:", [(int(b), round(float(m[b]), 2)) for b in slow])

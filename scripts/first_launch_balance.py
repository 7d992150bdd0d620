"""Round 6: the first launch of the headline update, per workgroup (LV_PASS_CLK=1): when its search reaches the barrier, when it ends —
which workgroups decide the launch's span and how much of their time lies behind the barrier (list levels + fits)."""
import os, sys
os.environ.setdefault("LV_PASS_CLK", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lvamd; lvamd.load()
from limo_velo_amd import capi, synth
sc = synth.make_scene(1_048_576, 65_536)
W=16
with capi.Context() as ctx:
    ctx.map_build(sc["map_xyz"]); ctx.scan_set(sc["scan_xyz"])
    for _ in range(5): ctx.update(sc["x_init"], sc["P0"])
    E=[];S=[];B=[]
    for _ in range(15):
        ctx.update(sc["x_init"], sc["P0"])
        clk,n = ctx.pass_clocks()
        c = clk[0,:n].astype(np.float64)
        s0 = c[:,W].min()
        E.append((c[:,W+9]-s0)/100); S.append((c[:,W+6]-c[:,W+3])/100); B.append((c[:,W+5]-c[:,W+3])/100)
    e=np.median(np.array(E),axis=0); s=np.median(np.array(S),axis=0); b=np.median(np.array(B),axis=0)
    print("launch 0 end per WG: pct", np.percentile(e,[0,10,25,50,75,90,95,99,100]).round(1))
    print("search phase (to barrier): pct", np.percentile(s,[0,10,50,90,99,100]).round(1))
    print("wave0 left task loop: pct", np.percentile(b,[0,10,50,90,99,100]).round(1))
    order=np.argsort(-e)[:12]
    print("slowest WGs:", [(int(i), round(float(e[i]),1), round(float(s[i]),1)) for i in order])
    print("corr end vs index", np.corrcoef(np.arange(n), e)[0,1].round(3))
    # level-1 count per WG from the static dealing: recompute which points each WG gets and whether they are level-1 (CPU estimate by kd-tree)

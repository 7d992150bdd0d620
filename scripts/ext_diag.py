"""estimate_extrinsics at the headline size: both routes of the update against the oracle, pass by pass (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "oracle")
import numpy as np
import torch  # noqa: F401
import lvamd; lvamd.load()
import lvoracle as lo
from limo_velo_amd import capi, synth

m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1_048_576, 65_536)
sc = synth.make_scene(m, n, extrinsics="xaloc")
tree = lo.KdTree(sc["map_xyz"])
prm_o = lo.default_params(estimate_extrinsics=1)
xo, Po, po, tro, so = lo.update(sc["x_init"], sc["P0"], sc["map_xyz"], sc["scan_xyz"], params=prm_o, tree=tree)
for route in ("one-launch", "three-kernel"):
    with capi.Context(capi.default_params(estimate_extrinsics=1)) as ctx:
        ctx.set_option("fused_ext", int(route == "one-launch"))
        ctx.map_build(sc["map_xyz"]); ctx.scan_set(sc["scan_xyz"])
        x, P, p, tr, sums = ctx.update(sc["x_init"], sc["P0"])
    print(route, "passes", p, po, "max|dx|", np.abs(x - xo).max(), "max|dP|", np.abs(P - Po).max())
    states = [sc["x_init"]] + [tr[i][23:49].copy() for i in range(p - 1)]
    for i in range(p):
        o = lo.iterate(states[i], sc["map_xyz"], sc["scan_xyz"], params=prm_o, tree=tree, details=False)
        rel = np.abs(sums[i]["HTH"] - o["HTH"]).max() / np.abs(o["HTH"]).max()
        # the oracle's solve from the DEVICE's sums at the DEVICE's state: isolates the solve
        r = lo.kf_step(states[i], sc["x_init"], sc["P0"], sums[i], params=prm_o, finalize=False)
        print(f"  pass {i}: trace diff vs oracle run {np.abs(tr[i] - tro[i]).max():.3e} (dx part {np.abs(tr[i][:23] - tro[i][:23]).max():.3e});"
              f" sums rel diff at device state {rel:.2e}; n_valid {sums[i]['n_valid']} {o['n_valid']};"
              f" device dx vs oracle kf_step(device sums, device state): {np.abs(np.asarray(r[1])[:23] - tr[i][:23]).max() if len(r) > 1 else -1:.3e}")

"""Generates tests/golden/cfg0_kat.npz — known-answer vectors of BASELINE configs[0]
(2k-pt synthetic scan vs 50k-pt map, 3 IKFoM iterations) from the CPU oracle.

PARITY UNPINNED: the reference ships no golden vectors and cannot be built here,
so these vectors pin the ORACLE (regression) and give the GPU box a fixture that does not depend on
re-running the oracle; they are not reference outputs.

    python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import lvamd  # noqa: E402

lvamd.load()
from limo_velo_amd import synth  # noqa: E402

import lvoracle as lo  # noqa: E402


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    sc = synth.make_scene(50_000, 2_000)
    tree = lo.KdTree(sc["map_xyz"])
    it = lo.iterate(sc["x_init"], sc["map_xyz"], sc["scan_xyz"], tree=tree)
    x, P, passes, trace, sums = lo.update(sc["x_init"], sc["P0"], sc["map_xyz"], sc["scan_xyz"], tree=tree)
    out = dict(
        map_sha256=np.array(digest(sc["map_xyz"])), scan_sha256=np.array(digest(sc["scan_xyz"])),
        x_init=sc["x_init"], P0=sc["P0"],
        knn_idx=it["knn_idx"], knn_d2=it["knn_d2"], valid=it["valid"], abcd=it["abcd"], dist=it["dist"],
        HTH=it["HTH"], HTh=it["HTh"], n_valid=np.array(it["n_valid"]), sum_h2=np.array(it["sum_h2"]),
        Hrows_first64=it["Hrows"][:64], h_first64=it["h"][:64],
        x_post=x, P_post=P, passes=np.array(passes), trace=trace,
        n_valid_per_pass=np.array([s["n_valid"] for s in sums]),
    )
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cfg0_kat.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

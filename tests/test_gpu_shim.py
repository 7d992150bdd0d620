"""The reference-style C++ API (limo-velo_amd/host: Mapper / Localizator over the C-ABI) driven by a
C++ program the way src/main.cpp drives the reference, checked against the oracle."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_shim_demo_matches_oracle(oracle, scene_small, tmp_path):
    exe = os.path.join(ROOT, "limo-velo_amd", "host", "shim_demo")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(exe)])
    sc = scene_small
    scan = sc["scan_xyz"][:1500]
    sc["map_xyz"].tofile(tmp_path / "map.f32")
    scan.tofile(tmp_path / "scan.f32")
    sc["x_init"].tofile(tmp_path / "x.f64")
    sc["P0"].tofile(tmp_path / "P.f64")
    out = tmp_path / "out.bin"
    r = subprocess.run([exe, str(tmp_path / "map.f32"), str(tmp_path / "scan.f32"), str(tmp_path / "x.f64"),
                        str(tmp_path / "P.f64"), str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = np.fromfile(out, dtype=np.uint8)
    hdr = raw[:32].view(np.float64)
    n_map, n_match, passes, t = int(hdr[0]), int(hdr[1]), int(hdr[2]), hdr[3]
    off = 32
    x = raw[off:off + 26 * 8].view(np.float64); off += 26 * 8
    P = raw[off:off + 529 * 8].view(np.float64).reshape(23, 23); off += 529 * 8
    H = raw[off:off + n_match * 12 * 8].view(np.float64).reshape(n_match, 12); off += n_match * 12 * 8
    h = raw[off:off + n_match * 8].view(np.float64); off += n_match * 8
    rec = raw[off:off + n_match * 32].view(np.float32).reshape(n_match, 8)

    tree = oracle.KdTree(sc["map_xyz"])
    o = oracle.iterate(sc["x_init"], sc["map_xyz"], scan, tree=tree)
    sel = o["valid"].astype(bool)
    assert n_map == 50_000 and n_match == int(sel.sum()) and t == 0.1
    pw = oracle.transform_scan(sc["x_init"], scan)
    assert np.array_equal(rec[:, :3].view(np.uint32), pw[sel].view(np.uint32))          # Match::point
    assert np.array_equal(rec[:, 3:7].view(np.uint32), o["abcd"][sel].view(np.uint32))  # Match::plane.n
    assert np.array_equal(rec[:, 7].view(np.uint32), o["dist"][sel].view(np.uint32))    # Match::distance
    assert np.array_equal(H, o["Hrows"][sel]) and np.array_equal(h, o["h"][sel])        # calculate_H
    xo, Po, po, _, _ = oracle.update(sc["x_init"], sc["P0"], sc["map_xyz"], scan, tree=tree)
    assert passes == po
    assert np.abs(x - xo).max() < 1e-9 and np.abs(P - Po).max() < 1e-10

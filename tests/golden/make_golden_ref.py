"""Generates tests/golden/ref_cfg0.npz FROM THE REFERENCE'S OWN CODE: oracle/_ref/liblvref.so (/root/reference/src compiled in place,
oracle/ref_build) run in this container on BASELINE configs[0] (2k-pt scan vs 50k-pt map, identity extrinsics and xaloc's with
estimate_extrinsics).  The fixture travels where the reference cannot: tests/test_oracle.py / tests/test_gpu_golden.py compare the
oracle and the HIP path with these arrays even on a box where neither /root/reference nor the prebuilt library exists.
What "the reference's own code" covers — the library compiles the reference's callers, its un-vendored dependencies
are stand-ins (oracle/ref_build: ikd-Tree's and IKFoM's interfaces over the oracle's kd-tree / update algebra, an Eigen stub for
the QR), so the arrays fall in two classes:
  PINNED to in-tree reference code: *_pose (State -> RotTransl), *_p_world_all (operator* on Points), *_H / *_h (the rows
      Localizator::calculate_H writes), *_dist and the gates Mapper::match / Plane::is_plane / on_plane apply, *_update_passes and
      *_update_n_valid (the loop and convergence rule as driven by the reference's call chain).
  THROUGH STAND-INS (the oracle checking itself behind the reference's interfaces): *_src (which map points the 5-NN returns),
      *_abcd (the QR solve), *_update_x / *_update_P / *_update_states (the IKFoM update algebra).  Those are pinned elsewhere:
      tests/test_oracle_pins.py (published algorithms, brute force, numpy QR / solves).
The npz carries the same lists under "pinned_fields" / "standin_fields".
Run from the repo root:  python tests/golden/make_golden_ref.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import lvamd  # noqa: E402

lvamd.load()
import lvref as lr  # noqa: E402
from limo_velo_amd import synth  # noqa: E402

assert lr.build() is not None, "the reference is not mounted and oracle/_ref is not built"
out = {}
for tag, extrinsics, est in (("id", "identity", 0), ("ext", "xaloc", 1)):
    sc = synth.make_scene(50_000, 2_000, extrinsics=extrinsics)
    lr.set_config(estimate_extrinsics=est)
    lr.reset()
    lr.map_add(sc["map_xyz"])
    for st, x in (("init", sc["x_init"]), ("true", sc["x_true"])):
        m = lr.match(x, sc["scan_xyz"])
        H, h, _ = lr.calculate_H(x, m["p_world"], m["abcd"])
        k = f"{tag}_{st}_"
        out[k + "pose"] = lr.state_to_pose(x)
        out[k + "p_world_all"] = lr.transform(x, sc["scan_xyz"])
        out[k + "src"], out[k + "abcd"], out[k + "dist"], out[k + "H"], out[k + "h"] = m["src"], m["abcd"], m["dist"], H, h
    x, P, n, tr, sums = lr.update(sc["x_init"], sc["P0"], sc["scan_xyz"])
    out[tag + "_update_x"], out[tag + "_update_P"], out[tag + "_update_passes"] = x, P, np.int64(n)
    out[tag + "_update_states"] = tr
    out[tag + "_update_n_valid"] = np.array([s["n_valid"] for s in sums], np.int64)
    out[tag + "_map_checksum"] = np.float64(sc["map_xyz"].astype(np.float64).sum())
lr.set_config()
lr.reset()
out["pinned_fields"] = np.array(["pose", "p_world_all", "H", "h", "dist", "update_passes", "update_n_valid"])
out["standin_fields"] = np.array(["src", "abcd", "update_x", "update_P", "update_states"])
path = os.path.join(ROOT, "tests", "golden", "ref_cfg0.npz")
np.savez_compressed(path, **out)
print("saved", path, os.path.getsize(path), "bytes;", {k: v.shape for k, v in list(out.items())[:8]})

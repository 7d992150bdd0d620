"""Host-side logic of the C++ shim that needs no GPU: the Accumulator's state look-up must pick the state the
reference's own index arithmetic picks (Accumulator.hpp:94-107 over Utils.hpp:9-23), quirks included — Compensator::path
starts its integration there, so a different choice de-skews differently."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "limo-velo_amd", "host")


def test_get_prev_state_follows_the_reference_index_arithmetic(lv, tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "limo-velo_amd", "csrc")])
    subprocess.check_call(["make", "-s", "-C", HOST])
    exe = tmp_path / "accum_check"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", HOST, "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "host", "accum_check.cpp"), "-o", str(exe),
                           "-L", os.path.join(ROOT, "limo-velo_amd"), "-llimovelo_shim", "-llimovelo_shim_config", "-llimovelo_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "limo-velo_amd")])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 mismatches" in r.stdout


def test_compensator_path_equals_the_reference_code(lv, oracle, tmp_path):
    """Compensator::path — the states and IMU samples the Accumulator hands out around [t1, t2] (get_states, get_prev_state with the
    reference's index quirks, get_imus, get_next_imu) and the up-sampling loop that integrates them (Compensator.cpp:35-102) — of the
    shim against the SAME function of the reference's compiled sources (oracle/_ref, lvr_path) on 300 random buffers: windows inside,
    before and beyond the buffered states, IMU rates above and below the state rate, duplicate stamps.  The number of states, every
    stamp and every f32 member must agree exactly when nothing rotates; with rotation the reference's libm sin / cos and the shim's
    polynomial (the device's) may differ in the last bit of R."""
    import struct
    import sys

    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import lvref

    if lvref.build() is None:
        import pytest

        pytest.skip("oracle/_ref is not built")
    lvref.set_config()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "limo-velo_amd", "csrc")])
    subprocess.check_call(["make", "-s", "-C", HOST])
    exe = tmp_path / "path_check"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", HOST, "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "host", "path_check.cpp"), "-o", str(exe),
                           "-L", os.path.join(ROOT, "limo-velo_amd"), "-llimovelo_shim", "-llimovelo_shim_config", "-llimovelo_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "limo-velo_amd")])
    rng = np.random.default_rng(17)
    cases = []
    for k in range(300):
        rotating = (k // 3) % 3 != 0
        n_states = int(rng.integers(2, 7))
        dt_s = float(rng.choice([0.01, 0.02, 0.1]))
        t_s = 5.0 + dt_s * np.arange(n_states)
        states = np.concatenate([oracle.motion_state(R=np.linalg.qr(rng.normal(size=(3, 3)))[0].astype(np.float32), pos=rng.uniform(-30, 30, 3),
                                                     vel=rng.uniform(-8, 8, 3), a=rng.normal(0, 1, 3) + [0, 0, 9.8],
                                                     w=rng.normal(0, 0.3, 3) if rotating else [0, 0, 0], time=float(t), bw=rng.normal(0, 0.01, 3) if rotating else [0, 0, 0],
                                                     ba=rng.normal(0, 0.05, 3)) for t in t_s])
        rate = float(rng.choice([0.0025, 0.005, 0.01, 0.025]))
        imu_t = np.arange(4.9, t_s[-1] + 0.15, rate)
        if k % 7 == 0:
            imu_t = np.sort(np.concatenate([imu_t, imu_t[::5]]))          # duplicate stamps
        imu_a = (rng.normal(0, 0.5, (len(imu_t), 3)) + [0, 0, 9.8]).astype(np.float32)
        imu_w = (rng.normal(0, 0.3, (len(imu_t), 3)) if rotating else np.zeros((len(imu_t), 3))).astype(np.float32)
        # as in the loop (src/main.cpp:66-73): the buffered states are the previous localisations, so at most ONE of them lies at or
        # after t1 (with two or more the reference's get_prev returns a default-constructed State whose a / w are uninitialised
        # memory: nothing to compare) — t1 exactly on the newest state, a little before it, or beyond it (a skipped cycle)
        kind = k % 3
        t1 = float(t_s[-1]) if kind == 0 else float(t_s[-1] - 0.4 * dt_s) if kind == 1 else float(t_s[-1] + rng.uniform(0.001, 0.04))
        t2 = float(t1 + rng.choice([0.01, 0.03, 0.1]))
        cases.append((t1, t2, states, imu_t, imu_a, imu_w, rotating))
    with open(tmp_path / "cases.bin", "wb") as f:
        f.write(struct.pack("<I", len(cases)))
        for t1, t2, states, imu_t, imu_a, imu_w, _ in cases:
            f.write(struct.pack("<ddI", t1, t2, len(states)))
            f.write(states.tobytes())
            f.write(struct.pack("<I", len(imu_t)))
            for t, a, w in zip(imu_t, imu_a, imu_w):
                f.write(struct.pack("<d", float(t)) + a.tobytes() + w.tobytes())
    xs = []
    for k in range(200):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        qo = rng.normal(size=4); qo /= np.linalg.norm(qo)
        x = np.zeros(26); x[0:3] = rng.uniform(-300, 300, 3); x[3:7] = q; x[7:11] = qo; x[11:14] = rng.uniform(-1, 1, 3); x[23:26] = [0, 0, -9.809]
        xs.append(x)
    with open(tmp_path / "cases.bin", "ab") as f:
        f.write(struct.pack("<I", len(xs)))
        for x in xs:
            f.write(x.tobytes())
    r = subprocess.run([str(exe), str(tmp_path / "cases.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = np.fromfile(tmp_path / "out.bin", np.uint8)
    off, exact, near = 0, 0, 0
    for t1, t2, states, imu_t, imu_a, imu_w, rotating in cases:
        n = int(raw[off:off + 4].view(np.uint32)[0]); off += 4
        got = raw[off:off + 184 * n].view(oracle.MOTION_DTYPE); off += 184 * n
        want = lvref.path(states, imu_a, imu_w, imu_t, t1, t2)
        assert len(got) == len(want), (t1, t2, len(got), len(want))
        assert np.array_equal(got["time"], want["time"])
        for fld in ("pos", "vel", "a", "w"):
            if not rotating:
                assert np.array_equal(got[fld].view(np.uint32), want[fld].view(np.uint32)), (fld, t1, t2)
            else:
                assert np.allclose(got[fld], want[fld], rtol=0, atol=1e-4), (fld, t1, t2)
        if not rotating:
            assert np.array_equal(got["R"].view(np.uint32), want["R"].view(np.uint32))
            exact += 1
        else:
            assert np.abs(got["R"] - want["R"]).max() < 5e-6
            near += 1
    assert exact >= 90 and near >= 190
    # State(const state_ikfom&, double): the f32 mirror the shim's State builds on the host against the reference's (lvr_state_to_pose)
    poses = raw[off:off + 96 * len(xs)].view(np.float32).reshape(len(xs), 24)
    off += 96 * len(xs)
    for x, got in zip(xs, poses):
        assert np.array_equal(got.view(np.uint32), lvref.state_to_pose(x).view(np.uint32))
    assert off == len(raw)

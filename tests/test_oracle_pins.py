"""Pins of the oracle's RECALLED parts that need no upstream source (VERDICT r03 "next" #4).

The esekf algebra of `oracle/lv_oracle.cpp::kf_step` (IKFoM's update_iterated_dyn_share_modified, absent from the reference
mount; call site /root/reference/src/Modules/Localizator.cpp:105-117,132) is checked here against properties that hold for
ANY correct iterated Kalman update on this manifold, derived independently in numpy:

* the fixed point of the iteration is the MAP estimate: the numerical gradient of
      J(x) = 1/2 |x [-] x_prop|^2_{P^-1} + 1/(2R) sum_i h_i(x)^2
  over the 23 tangent directions vanishes there, and the inverse of its numerical Hessian is the posterior covariance —
  with x far enough from x_prop that every manifold projection (A-matrix on the two SO3 blocks, Nx / Mx on S2) is far from
  the identity (tests/test_oracle.py::test_kf_step_matches_information_form only covers x = x_prop);
* the projection the update applies to dx = x [-] x_prop is the inverse of the numerical Jacobian of
  d -> (x [+] d) [-] x_prop (SO3 blocks; S2 block with upstream's documented integer-division quirk).

Row f-2: the pinned sin / cos polynomial (limo-velo_amd/csrc/lv_sincos.hpp, restated in the oracle) is the correctly rounded
f32 value; how often this container's glibc sinf / cosf — what the reference's std::sin(float) resolves to,
/root/reference/include/Headers/Utils.hpp:46 — differs from it is measured and bounded."""
import numpy as np
import pytest


def _rot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


class _Frozen:
    """A measurement model with FROZEN correspondences: the planes one matching pass of the oracle found, as f64 functions of
    the state — h_i(x) = -(n_i . p_world_i(x) + d_i), rows from the reference's own calculate_H formula
    (Localizator.cpp:29-57 through the oracle, itself checked by finite differences in test_oracle.py)."""

    def __init__(self, oracle, sc, x_lin, ext):
        tree = oracle.KdTree(sc["map_xyz"])
        it = oracle.iterate(x_lin, sc["map_xyz"], sc["scan_xyz"], tree=tree)
        v = it["valid"].astype(bool)
        self.o, self.ext = oracle, ext
        self.pl = sc["scan_xyz"][v].astype(np.float64)
        self.abcd32 = it["abcd"][v]
        self.abcd = self.abcd32.astype(np.float64)
        assert len(self.pl) > 500

    def world(self, x):
        return (self.pl @ _rot(x[7:11]).T + x[11:14]) @ _rot(x[3:7]).T + x[:3]

    def resid(self, x):
        return np.einsum("ij,ij->i", self.world(x), self.abcd[:, :3]) + self.abcd[:, 3]

    def jac(self, x, eps=1e-6):
        """d(residuals)/d(tangent) at x by central differences, 12 columns (pos, rot, offset_R_L_I, offset_T_L_I); without
        estimate_extrinsics the reference zeroes the last six (Localizator.cpp:52)."""
        Hn = np.zeros((len(self.pl), 12))
        for j in range(12 if self.ext else 6):
            e = np.zeros(23)
            e[j] = eps
            Hn[:, j] = (self.resid(self.o.boxplus(x, e)) - self.resid(self.o.boxplus(x, -e))) / (2 * eps)
        return Hn

    def sums(self, x):
        """H^T H / H^T h in f64 from the numerical Jacobian (h = -residual, Localizator.cpp:55): no f32 anywhere, so the
        fixed point of the iteration can be tested to rounding."""
        H, h = self.jac(x), -self.resid(x)
        return dict(HTH=H.T @ H, HTh=H.T @ h, sum_h2=float(h @ h), n_valid=len(h)), H

    def oracle_rows(self, x):
        """The same rows through the reference's calculate_H formula (f32 world point in, as the pipeline has it)."""
        pw, r = self.world(x), self.resid(x)
        H = np.zeros((len(r), 12))
        for i in range(len(r)):
            H[i], _ = self.o.calculate_H_row(x, pw[i].astype(np.float32), self.abcd32[i], np.float32(r[i]), estimate_extrinsics=self.ext)
        return H


def _prior(oracle, sc):
    """A propagated covariance with cross terms (twenty IMU predictions, as tests/test_gpu_parity.py::test_update_with_correlated_P)."""
    x, P = sc["x_init"].copy(), sc["P0"].copy()
    Q = np.diag([1e-4] * 3 + [1e-2] * 3 + [1e-5] * 3 + [1e-4] * 3)
    for _ in range(20):
        x, P = oracle.predict(x, P, 0.005, Q, [0.1, -0.05, 9.81], [0.01, 0.02, -0.01])
    return x, P


@pytest.mark.parametrize("ext", [False, True])
def test_fixed_point_of_the_iterated_update_is_the_map_estimate(oracle, scene_small, ext):
    sc = scene_small
    x_prop, P = _prior(oracle, sc)
    # pose prior of 0.1 m / 0.1 rad (the scene's default is 1: its gradient would drown in the differences' rounding), a loose
    # gravity direction
    S = np.ones(23)
    S[:6] = 0.1
    S[21:23] = 30.0
    P = P * np.outer(S, S)
    M = np.eye(23)           # ... and a shear that ties the gravity direction (unobserved by the planes) to the attitude, so
    M[21, 3], M[22, 4], M[21, 5] = 0.3, -0.25, 0.2   # that the S2 block moves with it: P <- M P M^T stays positive definite
    P = M @ P @ M.T
    prm = oracle.default_params(estimate_extrinsics=int(ext), lidar_noise=1e-3)
    R = 1e-3
    fz = _Frozen(oracle, sc, sc["x_true"], ext)
    # the propagated state lies 8 / 6 / 5 degrees and decimetres away from where the measurements want the pose (and, with
    # estimate_extrinsics, the extrinsics): at the fixed point x [-] x_prop is that large.  Without estimate_extrinsics the
    # reference zeroes the extrinsic columns of H (Localizator.cpp:52) and nothing correlates them with the rest: they stay put.
    off = np.zeros(23)
    off[:6] = [0.2, -0.15, 0.1, 0.14, -0.1, 0.09]
    if ext:
        off[6:12] = [0.1, 0.08, -0.09, 0.05, -0.04, 0.03]
    x_prop = oracle.boxplus(x_prop, off)
    x = x_prop.copy()
    for it in range(300):
        s, _ = fz.sums(x)
        x, dx, _, _ = oracle.kf_step(x, x_prop, P, s, params=prm, finalize=False)
        if np.abs(dx).max() < 1e-9:
            break
    assert np.abs(dx).max() < 1e-9, f"the frozen-correspondence iteration did not settle: {np.abs(dx).max()}"
    d_star = oracle.boxminus(x, x_prop)
    assert np.linalg.norm(d_star[3:6]) > 0.1 and np.linalg.norm(d_star[21:23]) > 1e-3   # projections far from the identity
    Pi = np.linalg.inv(P)

    # numerical Jacobians at the fixed point, per residual and per tangent coordinate (central differences at 1e-6: the
    # third derivatives are O(range) and O(1), truncation 1e-11; differencing the summed cost instead would carry the
    # cost's third derivative, ~1e10): D = d((x [+] d) [-] x_prop)/dd,  Hn = d(residuals)/dd
    D = np.zeros((23, 23))
    Hn = np.zeros((len(fz.pl), 23))
    e6 = 1e-6
    for j in range(23):
        e = np.zeros(23)
        e[j] = e6
        xp, xm = oracle.boxplus(x, e), oracle.boxplus(x, -e)
        D[:, j] = (oracle.boxminus(xp, x_prop) - oracle.boxminus(xm, x_prop)) / (2 * e6)
        Hn[:, j] = (fz.resid(xp) - fz.resid(xm)) / (2 * e6)
    g_p = D.T @ (Pi @ d_star)                 # gradient of 1/2 |x [-] x_prop|^2_{P^-1}
    g_m = Hn.T @ fz.resid(x) / R              # gradient of 1/(2R) sum h^2
    nd = 12 if ext else 6
    assert np.linalg.norm(g_p[:nd]) > 10.0 and np.abs(g_p[:3]).min() > 1.0 and np.abs(g_p[3:6]).max() > 1.0   # both terms pull ...
    # ... and cancel: the pose (and extrinsic) blocks with their A-matrix projections, the vector blocks through the prior's
    # cross terms.  Without estimate_extrinsics the extrinsic coordinates are not part of the minimisation.
    ok = np.arange(21) if ext else np.r_[np.arange(6), np.arange(12, 21)]
    scale = np.abs(g_p) + np.abs(g_m)
    rel = np.abs(g_p + g_m)[ok] / np.maximum(scale[ok], 1e-2 * scale.max())
    assert rel.max() < 1e-6, f"gradient / (|prior| + |measurement|) = {rel}"
    # (measured: 2e-8 / 1e-7 without / with estimate_extrinsics at |dx_rot| = 0.21 rad)
    # S2 block (gravity): no plane observes it, so its stationarity condition reads (P^-1 dx)_grav = 0 whatever the 2 x 2
    # projection is — it holds to rounding and pins nothing about Nx / Mx; the covariance below does see them
    rel_g = np.abs(g_p + g_m)[21:] / np.maximum(scale[21:], 1e-2 * scale.max())
    assert rel_g.max() < 1e-6, rel_g

    # the posterior covariance of a pass from the fixed point = inverse Gauss-Newton Hessian of J there, from the numerical
    # Jacobians only (the exact Hessian adds curvature terms of order |x [-] x_prop| that no Kalman update carries)
    s, H = fz.sums(x)
    Ho = fz.oracle_rows(x)      # (and the rows the pipeline would have used agree with the numerical ones to f32 rounding)
    assert np.abs(Ho - H).max() < 2e-5 * np.abs(H).max()
    _, _, _, P_post = oracle.kf_step(x, x_prop, P, s, params=prm, finalize=True)
    Hc = Hn.copy()
    Hc[:, 12:] = 0.0
    if not ext:
        Hc[:, 6:12] = 0.0           # estimate_extrinsics = false: the reference zeroes those columns (Localizator.cpp:52)
    C = np.linalg.inv(D.T @ Pi @ D + Hc.T @ Hc / R)
    sd = np.sqrt(np.diag(P_post))
    corr = np.abs(C - P_post) / np.outer(sd, sd)
    assert corr[np.ix_(ok, ok)].max() < 1e-6, corr[np.ix_(ok, ok)].max()       # (measured 4e-10 / 1.5e-8)
    # with the S2 rows / columns: upstream's Mx evaluates exp(.., 1/2) with an INTEGER 1/2 = 0 (the oracle's s2_Mx restates
    # it), a first-order projection; at |dx_grav| = 0.07 rad the gravity block of P is 1 % (in correlation units) away from
    # the Gauss-Newton value.  Recall-only: nothing here can tell upstream's quirk from a recall error of that size.
    assert np.linalg.norm(d_star[21:23]) > 0.05 and corr.max() < 0.03, corr.max()


def test_update_projection_is_the_inverse_jacobian_of_boxminus_after_boxplus(oracle, scene_small):
    """kf_step with NO measurement information (H^T H = 0, H^T h = 0) returns dx_ = -J dx with J the projection it applies to
    dx = x [-] x_prop; J must be the inverse of D = d((x [+] d) [-] x_prop)/dd at d = 0 (numerical, central differences)."""
    sc = scene_small
    x_prop = sc["x_init"]
    P = sc["P0"]
    rng = np.random.default_rng(11)
    zero = dict(HTH=np.zeros((12, 12)), HTh=np.zeros(12), sum_h2=0.0, n_valid=1)
    for trial in range(5):
        d0 = np.r_[rng.normal(scale=0.3, size=3), rng.normal(scale=0.25, size=6), rng.normal(scale=0.1, size=12), rng.normal(scale=0.02, size=2)]
        x = oracle.boxplus(x_prop, d0)
        dx = oracle.boxminus(x, x_prop)
        _, dxo, _, _ = oracle.kf_step(x, x_prop, P, zero, finalize=False)
        Jdx = -dxo                                   # dx_ = K_h + (K_x - I) J dx with K_h = K_x = 0
        D = np.zeros((23, 23))
        eps = 1e-6
        for j in range(23):
            e = np.zeros(23)
            e[j] = eps
            D[:, j] = (oracle.boxminus(oracle.boxplus(x, e), x_prop) - oracle.boxminus(oracle.boxplus(x, -e), x_prop)) / (2 * eps)
        want = np.linalg.solve(D, dx)
        assert np.abs(Jdx - want)[:21].max() < 1e-7, np.abs(Jdx - want)[:21].max()
        # S2: first-order agreement only (upstream's integer 1/2 in Mx; see the oracle's s2_Mx)
        assert np.abs(Jdx - want)[21:].max() < 0.6 * np.linalg.norm(dx[21:]) ** 2 + 1e-9, (Jdx[21:], want[21:])


def test_pinned_sincos_is_correctly_rounded_and_glibc_is_one_ulp_away_at_most(oracle):
    rng = np.random.default_rng(5)
    for lo, hi in ((0.0, 0.5), (-np.pi, np.pi)):      # de-skew angles |w| dt; a whole turn
        x = rng.uniform(lo, hi, 1_000_000).astype(np.float32)
        sn, cs = oracle.sincos_f32(x)
        x64 = x.astype(np.float64)
        # the polynomial = round-to-nearest of the f64 value: the platform-independent definition of sinf / cosf
        assert np.array_equal(sn.view(np.uint32), np.sin(x64).astype(np.float32).view(np.uint32))
        assert np.array_equal(cs.view(np.uint32), np.cos(x64).astype(np.float32).view(np.uint32))
        ns, nc, ulp = oracle.sincos_vs_libm(x)
        # glibc 2.35's sinf / cosf (0.56-ulp routines) miss that value in a few per cent of the arguments, never by more than
        # one ulp: DESIGN.md section 6 f-2 quotes the counts over 10^7 arguments
        assert ulp <= 1
        assert ns < 0.04 * len(x) and nc < 0.04 * len(x), (ns, nc)


# ---- "if the recall is wrong here, this is what moves"  -------------------------------------------------------
def _ulp_diff(a, b):
    """|a - b| in units of the last place of f32 values that share a sign (ABCD words do)."""
    ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    return np.abs(ia - ib)


def test_recall_residue_qr_back_substitution_order(oracle, lv, capsys):
    """The three restatements of Eigen's colPivHouseholderQr().solve() in this repository (oracle, stand-in Eigen, device) back-
    substitute ROW-wise; Eigen 3.3's triangular solve of a column-major matrix is COLUMN-oriented (a different association for ONE
    of the three unknowns).  Their agreement therefore checks nothing about that choice.  This test QUANTIFIES the difference
    instead of arguing about it: the headline scene's 65 536 planes and 3 000 seeded ones under both orders (the oracle's switch
    lvo_set_qr_backsub_columns).  What it asserts is the size of the effect — last-ulp changes in a minority of the planes, a
    handful of gate flips at most, a state difference far below the 1e-6 m / rad the parity tolerances allow — so that "bit for
    bit" in DESIGN reads as "bit for bit relative to the restatement, with THIS much riding on the recall"."""
    from limo_velo_amd import synth

    sc = synth.make_scene(1_048_576, 65_536)
    tree = oracle.KdTree(sc["map_xyz"])

    def run():
        o = oracle.iterate(sc["x_init"], sc["map_xyz"], sc["scan_xyz"], tree=tree, nthreads=8)
        x, P, passes, tr, sums = oracle.update(sc["x_init"], sc["P0"], sc["map_xyz"], sc["scan_xyz"], tree=tree, nthreads=8)
        return o, x, P, passes, tr, sums

    try:
        oracle.set_qr_backsub_columns(False)
        a = run()
        oracle.set_qr_backsub_columns(True)
        b = run()
        # 3 000 seeded planes: five points scattered around a random plane at map-like coordinates
        rng = np.random.default_rng(2024)
        n0 = rng.normal(size=(3000, 3)); n0 /= np.linalg.norm(n0, axis=1, keepdims=True)
        c0 = rng.uniform(-80, 80, (3000, 1, 3))
        u = rng.normal(size=(3000, 5, 3)) * 0.15
        pts = (c0 + u - (u @ n0[:, :, None]) * n0[:, None, :] + rng.normal(0, 0.01, (3000, 5, 3))).astype(np.float32)
        oracle.set_qr_backsub_columns(False)
        sq = np.full(5, 0.01, np.float32)
        pa = np.array([oracle.plane_fit(p, sq)[1] for p in pts])
        oracle.set_qr_backsub_columns(True)
        pb = np.array([oracle.plane_fit(p, sq)[1] for p in pts])
    finally:
        oracle.set_qr_backsub_columns(False)
    oa, ob = a[0], b[0]
    both = (oa["valid"] != 0) & (ob["valid"] != 0)
    ulp = _ulp_diff(np.ascontiguousarray(oa["abcd"][both], np.float32), np.ascontiguousarray(ob["abcd"][both], np.float32))
    changed_planes = int((ulp.max(axis=1) > 0).sum())
    flips = int(((oa["valid"] != 0) != (ob["valid"] != 0)).sum())
    ulp_s = _ulp_diff(np.ascontiguousarray(pa, np.float32), np.ascontiguousarray(pb, np.float32))
    dstate = [float(np.abs(a[4][i] - b[4][i]).max()) for i in range(min(len(a[4]), len(b[4])))]
    da = np.abs(oa["abcd"][both].astype(np.float64) - ob["abcd"][both].astype(np.float64))
    da[:, 3] /= np.maximum(np.abs(oa["abcd"][both][:, 3].astype(np.float64)), 1.0)   # (D relative to its size; the normal is a unit vector)
    das = np.abs(pa.astype(np.float64) - pb.astype(np.float64))
    das[:, 3] /= np.maximum(np.abs(pa[:, 3].astype(np.float64)), 1.0)
    report = {"headline_planes_compared": int(both.sum()), "planes_with_a_changed_word": changed_planes,
              "changed_words": int((ulp > 0).sum()), "median_ulp_of_changed_words": float(np.median(ulp[ulp > 0])),
              "max_abs_diff_normal_or_relD": float(da.max()), "gate_flips": flips,
              "seeded_planes_changed": int((ulp_s.max(axis=1) > 0).sum()), "seeded_max_abs_diff": float(das.max()),
              "passes": [int(a[3]), int(b[3])], "state_diff_per_pass": dstate, "final_state_diff": float(np.abs(a[1] - b[1]).max()),
              "final_cov_diff": float(np.abs(a[2] - b[2]).max())}
    with capsys.disabled():
        print("\nrecall residue, QR back substitution (row- vs column-oriented):", report)
    assert a[3] == b[3]
    assert 0 < changed_planes < 0.5 * both.sum()          # a real, minority effect
    assert da.max() < 2e-5 and das.max() < 2e-5           # one unknown's last bits, amplified by the conditioning of world-frame planes: f32 noise, not digits
    assert flips <= 8
    assert max(dstate) < 1e-7 and report["final_state_diff"] < 1e-7


def test_recall_residue_box_face_rule(lv, capsys):
    """ikd-Tree's Add_Points asks which points lie in the new point's 0.2 m box with an f32 RANGE test against
    [floor(v / len) * len, + len] ([UPSTREAM-RECALL]); this repository groups points by the integer key floor(v / len).  The two
    can only disagree for a coordinate that the f32 product floor(v / len) * len puts on the wrong side of v, or that sits exactly
    on a face.  Counted on the benchmark's 1 M-point map and four 64k-point scans inserted at the true pose (1.3 M points x 3
    coordinates): how many coordinates — and points — would land in a different box."""
    from limo_velo_amd import synth

    sc = synth.make_scene(1_048_576, 65_536)
    pts = [sc["map_xyz"]] + [synth.make_extra_scan(1_048_576, 65_536, k)["scan_xyz"] for k in range(4)]
    v = np.concatenate(pts).astype(np.float32)
    ln = np.float32(0.2)
    k = np.floor(v / ln)                       # f32 division, then floor: the key
    vmin = (k * ln).astype(np.float32)         # upstream's box: floor(v / len) * len, f32
    vmax = (vmin + ln).astype(np.float32)
    below = v < vmin                           # the point's own box does not contain it (lower face rounded above the point)
    above = v >= vmax                          # ... (upper face rounded to or below the point)
    on_face = (v == vmin) | (v == vmax)
    report = {"coordinates": int(v.size), "outside_own_box_below": int(below.sum()), "outside_own_box_above": int(above.sum()),
              "exactly_on_a_face": int(on_face.sum()), "points_affected": int((below | above | on_face).any(axis=1).sum()), "points": int(len(v))}
    with capsys.disabled():
        print("\nrecall residue, Add_Points box membership (f32 range test vs integer key):", report)
    # the effect is a few points per million at most: no result of this repository depends on it beyond which of two points
    # within rounding distance of a box face survives a down-sampling insert
    assert report["points_affected"] <= 1e-4 * len(v)

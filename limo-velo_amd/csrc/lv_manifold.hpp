// lv_manifold.hpp — device f64 manifold arithmetic of the IKFoM state (SO3 x2, S2, vect x5).
// The IKFoM submodule is absent from the reference mount; every function restates
// the published hku-mars/IKFoM (as vendored by FAST-LIO2) algorithm and is tagged [UPSTREAM-RECALL].
// dof layout: pos 0, rot 3, offset_R_L_I 6, offset_T_L_I 9, vel 12, bg 15, ba 18, grav 21 (2 dof)
// state doubles (lv_state): pos 0..2, rot 3..6 (x,y,z,w), offR 7..10, offT 11..13, vel 14..16,
//                           bg 17..19, ba 20..22, grav 23..25
#pragma once
#include "lv_device.hpp"

// From here on (this header and the rest of lv_solve.hip) FMA contraction is allowed: the 23-dof algebra
// is compared with the oracle at 1e-9, not bitwise, and fused code is ~35 % smaller — solve_kernel's
// time is instruction fetch of cold code.  The pose constants that feed the bit-exact f32 path
// (quat_to_rot, rt_compose, rt_inv in lv_device.hpp; finish_pose_consts) keep contraction OFF.
#pragma clang fp contract(fast)

namespace lv {

// ---- lean f64 elementary functions -----------------------------------------------------------------
// solve_kernel is one workgroup running cold code once per pass: its time is dominated by instruction
// fetch, so the ~40 KB of inlined libm range-reduction / correctly-rounded division code is replaced by
// compact versions accurate to a few ulp (the filter state is compared at 1e-9, see tests).
__device__ __forceinline__ double ddiv(double a, double b) {  // a / b, <= 2 ulp
    double r = __builtin_amdgcn_rcp(b);
    r = r * (2.0 - b * r);
    r = r * (2.0 - b * r);
    double q = a * r;
    return q + r * (a - b * q);
}
// sin and cos of x for |x| < ~1e5: Cody-Waite reduction to |r| <= pi/4, Taylor kernels
__device__ inline void dsincos(double x, double& sn, double& cs) {
    const double k = rint(x * 0.63661977236758134308);  // 2/pi
    double r = x - k * 1.57079632673412561417e+00;      // pi/2 split in three parts
    r = r - k * 6.07710050650619224932e-11;
    r = r - k * 2.02226624879595063154e-21;
    const double z = r * r;
    double ps = 1.58969099521155010221e-10;             // 1/13!
    ps = ps * z - 2.50507602534068634195e-08;           // -1/11!
    ps = ps * z + 2.75573137070700676789e-06;           // 1/9!
    ps = ps * z - 1.98412698298579493134e-04;           // -1/7!
    ps = ps * z + 8.33333333332248946124e-03;           // 1/5!
    ps = ps * z - 1.66666666666666324348e-01;           // -1/3!
    const double s0 = r + r * z * ps;
    double pc = -1.13596475577881948265e-11;            // -1/14!
    pc = pc * z + 2.08757232129817482790e-09;           // 1/12!
    pc = pc * z - 2.75573143513906633035e-07;           // -1/10!
    pc = pc * z + 2.48015872894767294178e-05;           // 1/8!
    pc = pc * z - 1.38888888888741095749e-03;           // -1/6!
    pc = pc * z + 4.16666666666666019037e-02;           // 1/4!
    const double c0 = 1.0 - 0.5 * z + z * z * pc;
    const int q = (int)k & 3;
    sn = (q == 0) ? s0 : (q == 1) ? c0 : (q == 2) ? -s0 : -c0;
    cs = (q == 0) ? c0 : (q == 1) ? -s0 : (q == 2) ? -c0 : s0;
}
// atan(t): two half-angle reductions t -> t / (1 + sqrt(1 + t^2)) bring |t| below tan(pi/16), then Taylor
__device__ inline double datan(double t) {
    const bool inv = fabs(t) > 1.0;
    double u = inv ? ddiv(1.0, t) : t;
    u = ddiv(u, 1.0 + sqrt(1.0 + u * u));
    u = ddiv(u, 1.0 + sqrt(1.0 + u * u));
    const double z = u * u;
    double p = 1.0 / 25.0;
    p = p * -z + 1.0 / 23.0;
    p = p * -z + 1.0 / 21.0;
    p = p * -z + 1.0 / 19.0;
    p = p * -z + 1.0 / 17.0;
    p = p * -z + 1.0 / 15.0;
    p = p * -z + 1.0 / 13.0;
    p = p * -z + 1.0 / 11.0;
    p = p * -z + 1.0 / 9.0;
    p = p * -z + 1.0 / 7.0;
    p = p * -z + 1.0 / 5.0;
    p = p * -z + 1.0 / 3.0;
    p = p * -z + 1.0;
    double a = 4.0 * (u * p);
    if (inv) a = (t > 0 ? 1.57079632679489661923 : -1.57079632679489661923) - a;
    return a;
}
__device__ inline double datan2(double y, double x) {
    if (x > 0) return datan(ddiv(y, x));
    if (x < 0) return datan(ddiv(y, x)) + (y >= 0 ? 3.14159265358979323846 : -3.14159265358979323846);
    return y > 0 ? 1.57079632679489661923 : (y < 0 ? -1.57079632679489661923 : 0.0);
}

constexpr double MTK_TOL = 1e-11;             // [UPSTREAM-RECALL MTK::tolerance<double>()]
constexpr double S2_LEN = 98090.0 / 10000.0;  // [UPSTREAM-RECALL typedef MTK::S2<double, 98090, 10000, 1> S2]

__device__ inline void d_quat_mul(const double a[4], const double b[4], double o[4]) {
    const double ax = a[0], ay = a[1], az = a[2], aw = a[3];
    const double bx = b[0], by = b[1], bz = b[2], bw = b[3];
    o[3] = aw * bw - ax * bx - ay * by - az * bz;
    o[0] = aw * bx + ax * bw + ay * bz - az * by;
    o[1] = aw * by + ay * bw + az * bx - ax * bz;
    o[2] = aw * bz + az * bw + ax * by - ay * bx;
}
__device__ inline void d_mat3_mul(const double* A, const double* B, double* C) {
    double T[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) T[i * 3 + j] = dot3d(A[i * 3], B[j], A[i * 3 + 1], B[3 + j], A[i * 3 + 2], B[6 + j]);
    for (int i = 0; i < 9; ++i) C[i] = T[i];
}
__device__ inline void d_mat3_vec(const double* A, const double* v, double* o) {
    double t0 = dot3d(A[0], v[0], A[1], v[1], A[2], v[2]);
    double t1 = dot3d(A[3], v[0], A[4], v[1], A[5], v[2]);
    double t2 = dot3d(A[6], v[0], A[7], v[1], A[8], v[2]);
    o[0] = t0; o[1] = t1; o[2] = t2;
}
__device__ inline void d_mat3_T(const double* A, double* o) {
    double t[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) t[i * 3 + j] = A[j * 3 + i];
    for (int i = 0; i < 9; ++i) o[i] = t[i];
}
__device__ inline void d_hat3(const double v[3], double H[9]) {
    H[0] = 0;     H[1] = -v[2]; H[2] = v[1];
    H[3] = v[2];  H[4] = 0;     H[5] = -v[0];
    H[6] = -v[1]; H[7] = v[0];  H[8] = 0;
}
// [UPSTREAM-RECALL MTK cos_sinc_sqrt]
__device__ inline void d_cos_sinc_sqrt(double x2, double& c, double& s) {
    const double taylor_n_bound = 1.220703125e-04;  // sqrt(sqrt(DBL_EPSILON)) = 2^-13
    if (x2 >= taylor_n_bound) {
        double x = sqrt(x2), sn;
        dsincos(x, sn, c);
        s = ddiv(sn, x);
        return;
    }
    const double inv[7] = {1 / 3., 1 / 4., 1 / 5., 1 / 6., 1 / 7., 1 / 8., 1 / 9.};
    double cosi = 1., sinc = 1.;
    double term = -1 / 2. * x2;
    for (int i = 0; i < 3; ++i) {
        cosi += term;
        term *= inv[2 * i];
        sinc += term;
        term *= -inv[2 * i + 1] * x2;
    }
    c = cosi;
    s = sinc;
}
// [UPSTREAM-RECALL MTK::exp<scalar,3>] + SO3::exp(vec, scale): q = (xyz, w)
__device__ inline void d_so3_exp(const double v[3], double scale, double q[4]) {
    const double half = scale / 2;
    double norm2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    double c, s;
    d_cos_sinc_sqrt(half * half * norm2, c, s);
    double mult = s * half;
    q[0] = mult * v[0]; q[1] = mult * v[1]; q[2] = mult * v[2];
    q[3] = c;
}
// [UPSTREAM-RECALL SO3::log -> MTK::log(res, w, vec, 2, true)]
__device__ inline void d_so3_log(const double q[4], double out[3]) {
    double nv = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    if (nv < MTK_TOL) nv = MTK_TOL;
    double s = ddiv(2.0, nv) * datan(ddiv(nv, q[3]));
    out[0] = s * q[0]; out[1] = s * q[1]; out[2] = s * q[2];
}
// [UPSTREAM-RECALL MTK::A_matrix]
__device__ inline void d_A_matrix(const double v[3], double A[9]) {
    double squaredNorm = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    double norm = sqrt(squaredNorm);
    for (int i = 0; i < 9; ++i) A[i] = (i % 4 == 0) ? 1.0 : 0.0;
    if (norm < MTK_TOL) return;
    double H[9], HH[9];
    d_hat3(v, H);
    d_mat3_mul(H, H, HH);
    double sn, cn;
    dsincos(norm, sn, cn);
    double c1 = ddiv(1 - cn, squaredNorm);
    double c2 = ddiv(1 - ddiv(sn, norm), squaredNorm);
    for (int i = 0; i < 9; ++i) A[i] = A[i] + c1 * H[i] + c2 * HH[i];
}
__device__ inline void d_s2_Bx(const double vec[3], double Bx[6]) {
    if (vec[0] + S2_LEN > MTK_TOL) {
        const double rd = ddiv(1.0, S2_LEN + vec[0]);
        Bx[0] = -vec[1];                           Bx[1] = -vec[2];
        Bx[2] = S2_LEN - vec[1] * vec[1] * rd;     Bx[3] = -vec[2] * vec[1] * rd;
        Bx[4] = -vec[2] * vec[1] * rd;             Bx[5] = S2_LEN - vec[2] * vec[2] * rd;
        for (int i = 0; i < 6; ++i) Bx[i] *= (1.0 / S2_LEN);
    } else {
        for (int i = 0; i < 6; ++i) Bx[i] = 0;
        Bx[3] = -1;
        Bx[4] = 1;
    }
}
__device__ inline void d_s2_boxplus(double vec[3], const double d[2]) {
    double Bx[6];
    d_s2_Bx(vec, Bx);
    double Bu[3] = {Bx[0] * d[0] + Bx[1] * d[1], Bx[2] * d[0] + Bx[3] * d[1], Bx[4] * d[0] + Bx[5] * d[1]};
    double q[4], R[9], o[3];
    d_so3_exp(Bu, 1.0, q);
    quat_to_rot(q, R);
    d_mat3_vec(R, vec, o);
    vec[0] = o[0]; vec[1] = o[1]; vec[2] = o[2];
}
__device__ inline void d_s2_boxminus(const double vec[3], const double other[3], double res[2]) {
    double H[9], hv[3];
    d_hat3(vec, H);
    d_mat3_vec(H, other, hv);
    double v_sin = sqrt(hv[0] * hv[0] + hv[1] * hv[1] + hv[2] * hv[2]);
    double v_cos = vec[0] * other[0] + vec[1] * other[1] + vec[2] * other[2];
    double theta = datan2(v_sin, v_cos);
    if (v_sin < MTK_TOL) {
        if (fabs(theta) > MTK_TOL) { res[0] = 3.1415926; res[1] = 0; }
        else { res[0] = 0; res[1] = 0; }
    } else {
        double Bx[6], Ho[9], t[3];
        d_s2_Bx(other, Bx);
        d_hat3(other, Ho);
        d_mat3_vec(Ho, vec, t);
        double f = ddiv(theta, v_sin);
        res[0] = f * (Bx[0] * t[0] + Bx[2] * t[1] + Bx[4] * t[2]);
        res[1] = f * (Bx[1] * t[0] + Bx[3] * t[1] + Bx[5] * t[2]);
    }
}
__device__ inline void d_s2_Nx_yy(const double vec[3], double Nx[6]) {
    double Bx[6], H[9];
    d_s2_Bx(vec, Bx);
    d_hat3(vec, H);
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j) {
            double a = Bx[0 * 2 + i] * H[0 * 3 + j] + Bx[1 * 2 + i] * H[1 * 3 + j] + Bx[2 * 2 + i] * H[2 * 3 + j];
            Nx[i * 3 + j] = 1 / S2_LEN / S2_LEN * a;
        }
}
// [UPSTREAM-RECALL quirk] upstream evaluates MTK::exp(..., scalar(1/2)) with an integer 1/2 == 0:
// exp_delta is the identity rotation.
__device__ inline void d_s2_Mx(const double vec[3], const double delta[2], double Mx[6]) {
    double Bx[6], H[9], T[9];
    d_s2_Bx(vec, Bx);
    d_hat3(vec, H);
    if (sqrt(delta[0] * delta[0] + delta[1] * delta[1]) < MTK_TOL) {
        for (int i = 0; i < 9; ++i) T[i] = -H[i];
    } else {
        double Bu[3] = {Bx[0] * delta[0] + Bx[1] * delta[1], Bx[2] * delta[0] + Bx[3] * delta[1],
                        Bx[4] * delta[0] + Bx[5] * delta[1]};
        double A[9], At[9];
        d_A_matrix(Bu, A);
        d_mat3_T(A, At);
        d_mat3_mul(H, At, T);
        for (int i = 0; i < 9; ++i) T[i] = -T[i];
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 2; ++j)
            Mx[i * 2 + j] = T[i * 3 + 0] * Bx[0 * 2 + j] + T[i * 3 + 1] * Bx[1 * 2 + j] + T[i * 3 + 2] * Bx[2 * 2 + j];
}
// T = Nx(x_grav) * Mx(xprop_grav, seg)   (2x2)
__device__ inline void d_s2_proj(const double xg[3], const double xpg[3], const double seg[2], double T[4]) {
    double Nx[6], Mx[6];
    d_s2_Nx_yy(xg, Nx);
    d_s2_Mx(xpg, seg, Mx);
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            T[i * 2 + j] = Nx[i * 3 + 0] * Mx[0 * 2 + j] + Nx[i * 3 + 1] * Mx[1 * 2 + j] + Nx[i * 3 + 2] * Mx[2 * 2 + j];
}

__device__ inline void d_state_boxplus(double* x, const double* d) {
    for (int i = 0; i < 3; ++i) x[0 + i] += d[0 + i];
    { double e[4], o[4]; d_so3_exp(d + 3, 1.0, e); d_quat_mul(x + 3, e, o); for (int i = 0; i < 4; ++i) x[3 + i] = o[i]; }
    { double e[4], o[4]; d_so3_exp(d + 6, 1.0, e); d_quat_mul(x + 7, e, o); for (int i = 0; i < 4; ++i) x[7 + i] = o[i]; }
    for (int i = 0; i < 3; ++i) x[11 + i] += d[9 + i];
    for (int i = 0; i < 3; ++i) x[14 + i] += d[12 + i];
    for (int i = 0; i < 3; ++i) x[17 + i] += d[15 + i];
    for (int i = 0; i < 3; ++i) x[20 + i] += d[18 + i];
    d_s2_boxplus(x + 23, d + 21);
}
__device__ inline void d_state_boxminus(const double* x, const double* o, double* d) {
    for (int i = 0; i < 3; ++i) d[0 + i] = x[0 + i] - o[0 + i];
    { double c[4] = {-o[3], -o[4], -o[5], o[6]}, q[4]; d_quat_mul(c, x + 3, q); d_so3_log(q, d + 3); }
    { double c[4] = {-o[7], -o[8], -o[9], o[10]}, q[4]; d_quat_mul(c, x + 7, q); d_so3_log(q, d + 6); }
    for (int i = 0; i < 3; ++i) d[9 + i] = x[11 + i] - o[11 + i];
    for (int i = 0; i < 3; ++i) d[12 + i] = x[14 + i] - o[14 + i];
    for (int i = 0; i < 3; ++i) d[15 + i] = x[17 + i] - o[17 + i];
    for (int i = 0; i < 3; ++i) d[18 + i] = x[20 + i] - o[20 + i];
    d_s2_boxminus(x + 23, o + 23, d + 21);
}

}  // namespace lv

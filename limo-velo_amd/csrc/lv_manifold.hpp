// lv_manifold.hpp — device f64 manifold arithmetic of the IKFoM state (SO3 x2, S2, vect x5).
// The IKFoM submodule is absent from the reference mount; every function restates
// the published hku-mars/IKFoM (as vendored by FAST-LIO2) algorithm and is tagged [UPSTREAM-RECALL].
// dof layout: pos 0, rot 3, offset_R_L_I 6, offset_T_L_I 9, vel 12, bg 15, ba 18, grav 21 (2 dof)
// state doubles (lv_state): pos 0..2, rot 3..6 (x,y,z,w), offR 7..10, offT 11..13, vel 14..16,
//                           bg 17..19, ba 20..22, grav 23..25
#pragma once
#include "lv_device.hpp"

namespace lv {

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
        double x = sqrt(x2);
        c = cos(x);
        s = sin(x) / x;
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
    double s = 2.0 / nv * atan(nv / q[3]);
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
    double c1 = (1 - cos(norm)) / squaredNorm;
    double c2 = (1 - sin(norm) / norm) / squaredNorm;
    for (int i = 0; i < 9; ++i) A[i] = A[i] + c1 * H[i] + c2 * HH[i];
}
__device__ inline void d_s2_Bx(const double vec[3], double Bx[6]) {
    if (vec[0] + S2_LEN > MTK_TOL) {
        Bx[0] = -vec[1];                                       Bx[1] = -vec[2];
        Bx[2] = S2_LEN - vec[1] * vec[1] / (S2_LEN + vec[0]);  Bx[3] = -vec[2] * vec[1] / (S2_LEN + vec[0]);
        Bx[4] = -vec[2] * vec[1] / (S2_LEN + vec[0]);          Bx[5] = S2_LEN - vec[2] * vec[2] / (S2_LEN + vec[0]);
        for (int i = 0; i < 6; ++i) Bx[i] /= S2_LEN;
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
    double theta = atan2(v_sin, v_cos);
    if (v_sin < MTK_TOL) {
        if (fabs(theta) > MTK_TOL) { res[0] = 3.1415926; res[1] = 0; }
        else { res[0] = 0; res[1] = 0; }
    } else {
        double Bx[6], Ho[9], t[3];
        d_s2_Bx(other, Bx);
        d_hat3(other, Ho);
        d_mat3_vec(Ho, vec, t);
        double f = theta / v_sin;
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

/*
 * lv_oracle.cpp — CPU ORACLE (test infrastructure; see lv_oracle.h header for the usage rule and
 * the statement of what its parity is pinned to).
 *
 * A dependency-free C++17 restatement of the LIMO-Velo iterated-KF-update hot path.  Every function
 * cites the reference file:line it follows (paths relative to /root/reference).  Pieces whose source
 * is NOT in the reference mount (ikd-Tree, IKFoM, Eigen internals) are tagged [UPSTREAM-RECALL].
 *
 * Build: g++ -O3 -fopenmp -ffp-contract=off, NO -march (reference CMakeLists.txt:8,16 builds -O3 for
 * baseline x86-64, i.e. no FMA contraction).
 */
#include "lv_oracle.h"

#include <algorithm>
#include <cmath>
#include <unordered_map>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ---------------------------------------------------------------------------------------------
// f32 3-vector algebra in Eigen-3.3 evaluation order.
// [UPSTREAM-RECALL Eigen 3.3 Redux.h redux_novec_unroller]: a fixed-size 3-term sum is evaluated as
// x0 + (x1 + x2) (recursive halving, first half = 1 element).  All fixed-size 3x3 * 3x3 and 3x3 * 3
// float products of State.cpp / RotTransl.cpp go through coeff-based lazy products whose coefficient
// is (lhs.row(i).transpose().cwiseProduct(rhs.col(j))).sum(), hence this order.
// ---------------------------------------------------------------------------------------------
inline float dot3f(float a0, float b0, float a1, float b1, float a2, float b2) {
    float p0 = a0 * b0, p1 = a1 * b1, p2 = a2 * b2;
    float t = p1 + p2;
    return p0 + t;
}
inline double dot3d(double a0, double b0, double a1, double b1, double a2, double b2) {
    double p0 = a0 * b0, p1 = a1 * b1, p2 = a2 * b2;
    double t = p1 + p2;
    return p0 + t;
}

struct RT32 {  // RotTransl (Objects.hpp:139-151): row-major R, t
    float R[9];
    float t[3];
};

// RotTransl operator*(RT1, RT2)  — RotTransl.cpp:36-41
inline RT32 rt_compose(const RT32& a, const RT32& b) {
    RT32 o;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            o.R[i * 3 + j] = dot3f(a.R[i * 3 + 0], b.R[0 * 3 + j], a.R[i * 3 + 1], b.R[1 * 3 + j],
                                   a.R[i * 3 + 2], b.R[2 * 3 + j]);
    for (int i = 0; i < 3; ++i)
        o.t[i] = dot3f(a.R[i * 3 + 0], b.t[0], a.R[i * 3 + 1], b.t[1], a.R[i * 3 + 2], b.t[2]) + a.t[i];
    return o;
}
// Point operator*(RT, p) — RotTransl.cpp:43-48
inline void rt_apply(const RT32& a, const float p[3], float out[3]) {
    for (int i = 0; i < 3; ++i)
        out[i] = dot3f(a.R[i * 3 + 0], p[0], a.R[i * 3 + 1], p[1], a.R[i * 3 + 2], p[2]) + a.t[i];
}
// RotTransl::inv() — RotTransl.cpp:29-34:  (R^T, -R^T * t)
inline RT32 rt_inv(const RT32& a) {
    RT32 o;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) o.R[i * 3 + j] = a.R[j * 3 + i];
    for (int i = 0; i < 3; ++i)
        o.t[i] = dot3f(-o.R[i * 3 + 0], a.t[0], -o.R[i * 3 + 1], a.t[1], -o.R[i * 3 + 2], a.t[2]);
    return o;
}

// ---------------------------------------------------------------------------------------------
// f64 quaternion / SO3 helpers.  Quaternion storage = Eigen coeffs order (x,y,z,w).
// ---------------------------------------------------------------------------------------------
// [UPSTREAM-RECALL Eigen Quaternion::toRotationMatrix]
inline void quat_to_rot(const double q[4], double R[9]) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
    R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}
inline void quat_conj(const double q[4], double o[4]) { o[0] = -q[0]; o[1] = -q[1]; o[2] = -q[2]; o[3] = q[3]; }
// [UPSTREAM-RECALL Eigen quaternion product, scalar path]
inline void quat_mul(const double a[4], const double b[4], double o[4]) {
    const double ax = a[0], ay = a[1], az = a[2], aw = a[3];
    const double bx = b[0], by = b[1], bz = b[2], bw = b[3];
    o[3] = aw * bw - ax * bx - ay * by - az * bz;
    o[0] = aw * bx + ax * bw + ay * bz - az * by;
    o[1] = aw * by + ay * bw + az * bx - ax * bz;
    o[2] = aw * bz + az * bw + ax * by - ay * bx;
}
inline void mat3_mul(const double* A, const double* B, double* C) {
    double T[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            T[i * 3 + j] = dot3d(A[i * 3], B[j], A[i * 3 + 1], B[3 + j], A[i * 3 + 2], B[6 + j]);
    std::memcpy(C, T, sizeof(T));
}
inline void mat3_vec(const double* A, const double* v, double* o) {
    double t[3];
    for (int i = 0; i < 3; ++i) t[i] = dot3d(A[i * 3], v[0], A[i * 3 + 1], v[1], A[i * 3 + 2], v[2]);
    o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
}
inline void mat3_T(const double* A, double* o) {
    double t[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) t[i * 3 + j] = A[j * 3 + i];
    std::memcpy(o, t, sizeof(t));
}
inline void hat3(const double v[3], double H[9]) {  // [UPSTREAM-RECALL MTK::hat]
    H[0] = 0;     H[1] = -v[2]; H[2] = v[1];
    H[3] = v[2];  H[4] = 0;     H[5] = -v[0];
    H[6] = -v[1]; H[7] = v[0];  H[8] = 0;
}
// Eigen cross: lhs x rhs
inline void cross3(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

constexpr double MTK_TOL = 1e-11;  // [UPSTREAM-RECALL MTK::tolerance<double>()]

// [UPSTREAM-RECALL MTK cos_sinc_sqrt]
inline void cos_sinc_sqrt(double x2, double& c, double& s) {
    static const double taylor_0_bound = std::numeric_limits<double>::epsilon();
    static const double taylor_2_bound = std::sqrt(taylor_0_bound);
    static const double taylor_n_bound = std::sqrt(taylor_2_bound);
    if (x2 >= taylor_n_bound) {
        double x = std::sqrt(x2);
        c = std::cos(x);
        s = std::sin(x) / x;
        return;
    }
    static const double inv[] = {1 / 3., 1 / 4., 1 / 5., 1 / 6., 1 / 7., 1 / 8., 1 / 9.};
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
// [UPSTREAM-RECALL MTK::exp<scalar,3>(result_vec, vec, scale)] -> returns w, writes xyz
inline double mtk_exp(double out_xyz[3], const double v[3], double scale) {
    double norm2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    double c, s;
    cos_sinc_sqrt(scale * scale * norm2, c, s);
    double mult = s * scale;
    out_xyz[0] = mult * v[0]; out_xyz[1] = mult * v[1]; out_xyz[2] = mult * v[2];
    return c;
}
// [UPSTREAM-RECALL SO3::exp(vec, scale)]: quaternion from rotation vector
inline void so3_exp(const double v[3], double scale, double q[4]) { q[3] = mtk_exp(q, v, scale / 2); }
// [UPSTREAM-RECALL SO3::log / MTK::log(result, w, vec, scale=2, plus_minus_periodicity=true)]
inline void so3_log(const double q[4], double out[3]) {
    double nv = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    if (nv < MTK_TOL) nv = MTK_TOL;  // (plus_minus_periodicity == true branch)
    double s = 2.0 / nv * std::atan(nv / q[3]);
    out[0] = s * q[0]; out[1] = s * q[1]; out[2] = s * q[2];
}
// [UPSTREAM-RECALL MTK::A_matrix]
inline void A_matrix(const double v[3], double A[9]) {
    double squaredNorm = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    double norm = std::sqrt(squaredNorm);
    for (int i = 0; i < 9; ++i) A[i] = (i % 4 == 0) ? 1.0 : 0.0;
    if (norm < MTK_TOL) return;
    double H[9], HH[9];
    hat3(v, H);
    mat3_mul(H, H, HH);
    double c1 = (1 - std::cos(norm)) / squaredNorm;
    double c2 = (1 - std::sin(norm) / norm) / squaredNorm;
    for (int i = 0; i < 9; ++i) A[i] = A[i] + c1 * H[i] + c2 * HH[i];
}

// S2 manifold, [UPSTREAM-RECALL typedef MTK::S2<double, 98090, 10000, 1> S2]: length 9.809, type 1.
constexpr double S2_LEN = 98090.0 / 10000.0;
inline void s2_Bx(const double vec[3], double Bx[6] /*3x2 row-major*/) {
    if (vec[0] + S2_LEN > MTK_TOL) {
        Bx[0] = -vec[1];                                         Bx[1] = -vec[2];
        Bx[2] = S2_LEN - vec[1] * vec[1] / (S2_LEN + vec[0]);    Bx[3] = -vec[2] * vec[1] / (S2_LEN + vec[0]);
        Bx[4] = -vec[2] * vec[1] / (S2_LEN + vec[0]);            Bx[5] = S2_LEN - vec[2] * vec[2] / (S2_LEN + vec[0]);
        for (int i = 0; i < 6; ++i) Bx[i] /= S2_LEN;
    } else {
        for (int i = 0; i < 6; ++i) Bx[i] = 0;
        Bx[1 * 2 + 1] = -1;
        Bx[2 * 2 + 0] = 1;
    }
}
inline void s2_boxplus(double vec[3], const double d[2]) {
    double Bx[6];
    s2_Bx(vec, Bx);
    double Bu[3] = {Bx[0] * d[0] + Bx[1] * d[1], Bx[2] * d[0] + Bx[3] * d[1], Bx[4] * d[0] + Bx[5] * d[1]};
    double q[4];
    so3_exp(Bu, 1.0, q);  // MTK::exp(res.vec(), Bu, scale/2) with scale = 1
    double R[9], o[3];
    quat_to_rot(q, R);
    mat3_vec(R, vec, o);
    vec[0] = o[0]; vec[1] = o[1]; vec[2] = o[2];
}
inline void s2_oplus3(double vec[3], const double d[3], double scale) {  // S2::oplus (used by predict)
    double q[4];
    so3_exp(d, scale, q);
    double R[9], o[3];
    quat_to_rot(q, R);
    mat3_vec(R, vec, o);
    vec[0] = o[0]; vec[1] = o[1]; vec[2] = o[2];
}
inline void s2_boxminus(const double vec[3], const double other[3], double res[2]) {
    double H[9], hv[3];
    hat3(vec, H);
    mat3_vec(H, other, hv);
    double v_sin = std::sqrt(hv[0] * hv[0] + hv[1] * hv[1] + hv[2] * hv[2]);
    double v_cos = vec[0] * other[0] + vec[1] * other[1] + vec[2] * other[2];
    double theta = std::atan2(v_sin, v_cos);
    if (v_sin < MTK_TOL) {
        if (std::fabs(theta) > MTK_TOL) { res[0] = 3.1415926; res[1] = 0; }
        else { res[0] = 0; res[1] = 0; }
    } else {
        double Bx[6];
        s2_Bx(other, Bx);
        double Ho[9], t[3];
        hat3(other, Ho);
        mat3_vec(Ho, vec, t);
        double f = theta / v_sin;
        res[0] = f * (Bx[0] * t[0] + Bx[2] * t[1] + Bx[4] * t[2]);
        res[1] = f * (Bx[1] * t[0] + Bx[3] * t[1] + Bx[5] * t[2]);
    }
}
// Nx = 1/len/len * Bx^T * hat(vec)   (2x3)
inline void s2_Nx_yy(const double vec[3], double Nx[6]) {
    double Bx[6], H[9];
    s2_Bx(vec, Bx);
    hat3(vec, H);
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j) {
            double a = Bx[0 * 2 + i] * H[0 * 3 + j] + Bx[1 * 2 + i] * H[1 * 3 + j] + Bx[2 * 2 + i] * H[2 * 3 + j];
            Nx[i * 3 + j] = 1 / S2_LEN / S2_LEN * a;
        }
}
// Mx (3x2).  NOTE [UPSTREAM-RECALL quirk]: upstream calls MTK::exp(..., scalar(1/2)) with an INTEGER
// 1/2 == 0, so exp_delta is the identity rotation; restated as such.
inline void s2_Mx(const double vec[3], const double delta[2], double Mx[6]) {
    double Bx[6], H[9];
    s2_Bx(vec, Bx);
    hat3(vec, H);
    double T[9];
    if (std::sqrt(delta[0] * delta[0] + delta[1] * delta[1]) < MTK_TOL) {
        for (int i = 0; i < 9; ++i) T[i] = -H[i];
    } else {
        double Bu[3] = {Bx[0] * delta[0] + Bx[1] * delta[1], Bx[2] * delta[0] + Bx[3] * delta[1],
                        Bx[4] * delta[0] + Bx[5] * delta[1]};
        double A[9], At[9];
        A_matrix(Bu, A);
        mat3_T(A, At);
        // -I * hat(vec) * A^T
        mat3_mul(H, At, T);
        for (int i = 0; i < 9; ++i) T[i] = -T[i];
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 2; ++j)
            Mx[i * 2 + j] = T[i * 3 + 0] * Bx[0 * 2 + j] + T[i * 3 + 1] * Bx[1 * 2 + j] + T[i * 3 + 2] * Bx[2 * 2 + j];
}

// ---------------------------------------------------------------------------------------------
// Small dense f64 linear algebra (row-major, n <= 24).
// ---------------------------------------------------------------------------------------------
constexpr int NS = 23;  // state dof

// inverse via LU with partial pivoting ([UPSTREAM-RECALL Eigen fixed-size .inverse() > 4x4 =
// PartialPivLU]); returns false if a pivot is exactly 0.
bool mat_inverse(const double* A, int n, double* Ainv) {
    std::vector<double> a(A, A + (size_t)n * n);
    std::vector<int> perm(n);
    for (int i = 0; i < n; ++i) perm[i] = i;
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = std::fabs(a[(size_t)k * n + k]);
        for (int i = k + 1; i < n; ++i) {
            double v = std::fabs(a[(size_t)i * n + k]);
            if (v > best) { best = v; p = i; }
        }
        if (best == 0.0) return false;
        if (p != k) {
            for (int j = 0; j < n; ++j) std::swap(a[(size_t)k * n + j], a[(size_t)p * n + j]);
            std::swap(perm[k], perm[p]);
        }
        double piv = a[(size_t)k * n + k];
        for (int i = k + 1; i < n; ++i) {
            double f = a[(size_t)i * n + k] / piv;
            a[(size_t)i * n + k] = f;
            for (int j = k + 1; j < n; ++j) a[(size_t)i * n + j] -= f * a[(size_t)k * n + j];
        }
    }
    // solve A X = I column by column: L U x = P e_c
    for (int c = 0; c < n; ++c) {
        double y[32];
        for (int i = 0; i < n; ++i) {
            double s = (perm[i] == c) ? 1.0 : 0.0;
            for (int j = 0; j < i; ++j) s -= a[(size_t)i * n + j] * y[j];
            y[i] = s;
        }
        for (int i = n - 1; i >= 0; --i) {
            double s = y[i];
            for (int j = i + 1; j < n; ++j) s -= a[(size_t)i * n + j] * y[j];
            y[i] = s / a[(size_t)i * n + i];
        }
        for (int i = 0; i < n; ++i) Ainv[(size_t)i * n + c] = y[i];
    }
    return true;
}

// ---------------------------------------------------------------------------------------------
// kNN
// ---------------------------------------------------------------------------------------------
// [UPSTREAM-RECALL ikd-Tree calc_dist(a,b)]: f32, left-to-right, unfused.
inline float calc_dist(const float* a, const float* b) {
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    float sx = dx * dx, sy = dy * dy, sz = dz * dz;
    float s = sx + sy;
    return s + sz;
}

struct Cand {
    float d;
    uint32_t i;
};
inline bool cand_less(const Cand& a, const Cand& b) { return a.d < b.d || (a.d == b.d && a.i < b.i); }

// bounded sorted list of the k best candidates under (d, idx) lexicographic order
struct TopK {
    int k, n;
    Cand c[16];
    explicit TopK(int k_) : k(k_), n(0) { for (auto& e : c) e = Cand{0.f, 0u}; }
    inline bool full() const { return n == k; }
    inline const Cand& worst() const { return c[n - 1]; }
    inline void push(Cand x) {
        if (n == k) {
            if (!cand_less(x, c[n - 1])) return;
            --n;
        }
        int j = n++;
        while (j > 0 && cand_less(x, c[j - 1])) { c[j] = c[j - 1]; --j; }
        c[j] = x;
    }
};

void write_topk(const TopK& t, int k, uint32_t* idx, float* d2, int32_t* found) {
    for (int j = 0; j < k; ++j) {
        idx[j] = j < t.n ? t.c[j].i : 0xFFFFFFFFu;
        d2[j] = j < t.n ? t.c[j].d : std::numeric_limits<float>::infinity();
    }
    if (found) *found = t.n;
}

// One point per node, pointer-linked, with bounding box (ikd-Tree node shape) [UPSTREAM-RECALL].
struct KdNode {
    float p[3];
    uint32_t idx;
    int axis;
    float bmin[3], bmax[3];
    KdNode* l;
    KdNode* r;
};
struct KdTree {
    std::vector<KdNode> pool;
    KdNode* root = nullptr;
    size_t m = 0;
};

KdNode* kd_build(KdTree& T, std::vector<uint32_t>& ids, size_t lo, size_t hi, const float* xyz) {
    if (lo >= hi) return nullptr;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = lo; i < hi; ++i)
        for (int a = 0; a < 3; ++a) {
            float v = xyz[(size_t)ids[i] * 3 + a];
            mn[a] = std::min(mn[a], v);
            mx[a] = std::max(mx[a], v);
        }
    int axis = 0;
    float best = mx[0] - mn[0];
    for (int a = 1; a < 3; ++a)
        if (mx[a] - mn[a] > best) { best = mx[a] - mn[a]; axis = a; }
    size_t mid = (lo + hi) / 2;
    std::nth_element(ids.begin() + lo, ids.begin() + mid, ids.begin() + hi, [&](uint32_t a, uint32_t b) {
        float va = xyz[(size_t)a * 3 + axis], vb = xyz[(size_t)b * 3 + axis];
        return va < vb || (va == vb && a < b);
    });
    T.pool.emplace_back();  // pool was reserved for m nodes: pointers stay valid
    KdNode* n = &T.pool.back();
    n->idx = ids[mid];
    for (int a = 0; a < 3; ++a) {
        n->p[a] = xyz[(size_t)ids[mid] * 3 + a];
        n->bmin[a] = mn[a];
        n->bmax[a] = mx[a];
    }
    n->axis = axis;
    n->l = kd_build(T, ids, lo, mid, xyz);
    n->r = kd_build(T, ids, mid + 1, hi, xyz);
    return n;
}

inline float box_dist(const KdNode* n, const float* q) {  // squared distance to node's bbox
    float s = 0.f;
    for (int a = 0; a < 3; ++a) {
        float d = 0.f;
        if (q[a] < n->bmin[a]) d = n->bmin[a] - q[a];
        else if (q[a] > n->bmax[a]) d = q[a] - n->bmax[a];
        s += d * d;
    }
    return s;
}

void kd_search(const KdNode* n, const float* q, TopK& best) {
    if (!n) return;
    // prune with a safety margin of a few ulps: box_dist is computed with a different rounding
    // sequence than calc_dist, so never prune a box that could hold an equal-distance point.
    if (best.full()) {
        float bd = box_dist(n, q);
        if (bd * 0.999999f > best.worst().d) return;
    }
    best.push(Cand{calc_dist(q, n->p), n->idx});
    const KdNode* first = n->l;
    const KdNode* second = n->r;
    if (q[n->axis] > n->p[n->axis]) std::swap(first, second);
    kd_search(first, q, best);
    kd_search(second, q, best);
}

// ---------------------------------------------------------------------------------------------
// Plane fit: R3Math::estimate_plane (Utils.cpp:32-57)
// ---------------------------------------------------------------------------------------------
static int g_qr_backsub_columns = 0;   // lvo_set_qr_backsub_columns: see the back substitution below
// Column-pivoted Householder QR least squares, f32, for an n x 3 system (n <= 8), restating the
// structure of Eigen 3.3 ColPivHouseholderQR::computeInPlace + _solve_impl [UPSTREAM-RECALL]:
// pivot on the largest running column norm, LAPACK-WN176 norm downdate, rank cut by
// nonzero_pivots.  Inner sums are plain left-to-right f32 (Eigen's dynamic-size reductions are
// SSE-packet ordered and alignment dependent; bit-identity with Eigen is NOT claimed).
void colpiv_qr_solve_f32(float A[][3], int rows, const float* b_in, float x[3]) {
    const int cols = 3, size = 3;
    const float eps = std::numeric_limits<float>::epsilon();
    float hCoeffs[3];
    int trans[3];
    float normsUpd[3], normsDir[3];
    for (int k = 0; k < cols; ++k) {
        float s = 0.f;
        for (int i = 0; i < rows; ++i) s += A[i][k] * A[i][k];
        normsDir[k] = normsUpd[k] = std::sqrt(s);
    }
    float maxn = std::max(normsUpd[0], std::max(normsUpd[1], normsUpd[2]));
    float th = maxn * eps / float(rows);
    const float threshold_helper = th * th;
    const float norm_downdate_threshold = std::sqrt(eps);
    int nonzero_pivots = size;
    for (int k = 0; k < size; ++k) {
        int bi = k;
        float bn = normsUpd[k];
        for (int j = k + 1; j < cols; ++j)
            if (normsUpd[j] > bn) { bn = normsUpd[j]; bi = j; }
        float biggest_sq = bn * bn;
        if (nonzero_pivots == size && biggest_sq < threshold_helper * float(rows - k)) nonzero_pivots = k;
        trans[k] = bi;
        if (k != bi) {
            for (int i = 0; i < rows; ++i) std::swap(A[i][k], A[i][bi]);
            std::swap(normsUpd[k], normsUpd[bi]);
            std::swap(normsDir[k], normsDir[bi]);
        }
        // makeHouseholderInPlace on A[k..rows-1][k]
        float tailSq = 0.f;
        for (int i = k + 1; i < rows; ++i) tailSq += A[i][k] * A[i][k];
        float c0 = A[k][k];
        float tau, beta;
        if (tailSq <= std::numeric_limits<float>::min()) {
            tau = 0.f;
            beta = c0;
            for (int i = k + 1; i < rows; ++i) A[i][k] = 0.f;
        } else {
            beta = std::sqrt(c0 * c0 + tailSq);
            if (c0 >= 0.f) beta = -beta;
            float den = c0 - beta;
            for (int i = k + 1; i < rows; ++i) A[i][k] = A[i][k] / den;
            tau = (beta - c0) / beta;
        }
        A[k][k] = beta;
        hCoeffs[k] = tau;
        // apply H_k to the trailing columns
        if (tau != 0.f) {
            for (int j = k + 1; j < cols; ++j) {
                float tmp = 0.f;
                for (int i = k + 1; i < rows; ++i) tmp += A[i][k] * A[i][j];
                tmp += A[k][j];
                A[k][j] -= tau * tmp;
                for (int i = k + 1; i < rows; ++i) A[i][j] -= tau * A[i][k] * tmp;
            }
        }
        // column-norm downdate
        for (int j = k + 1; j < cols; ++j) {
            if (normsUpd[j] != 0.f) {
                float temp = std::fabs(A[k][j]) / normsUpd[j];
                temp = (1.f + temp) * (1.f - temp);
                temp = temp < 0.f ? 0.f : temp;
                float r = normsUpd[j] / normsDir[j];
                float temp2 = temp * (r * r);
                if (temp2 <= norm_downdate_threshold) {
                    float s = 0.f;
                    for (int i = k + 1; i < rows; ++i) s += A[i][j] * A[i][j];
                    normsDir[j] = std::sqrt(s);
                    normsUpd[j] = normsDir[j];
                } else {
                    normsUpd[j] *= std::sqrt(temp);
                }
            }
        }
    }
    // solve
    x[0] = x[1] = x[2] = 0.f;
    if (nonzero_pivots == 0) return;
    float c[8];
    for (int i = 0; i < rows; ++i) c[i] = b_in[i];
    for (int k = 0; k < nonzero_pivots; ++k) {  // c = Q^T c
        float tau = hCoeffs[k];
        if (rows - k == 1) { c[k] *= (1.f - tau); continue; }
        if (tau != 0.f) {
            float tmp = 0.f;
            for (int i = k + 1; i < rows; ++i) tmp += A[i][k] * c[i];
            tmp += c[k];
            c[k] -= tau * tmp;
            for (int i = k + 1; i < rows; ++i) c[i] -= tau * A[i][k] * tmp;
        }
    }
    if (g_qr_backsub_columns) {
        // Eigen 3.3's `triangularView<Upper>().solveInPlace(vector)` on a column-major matrix goes through
        // triangular_solve_vector<.., OnTheLeft, Upper, false, ColMajor> [UPSTREAM-RECALL], which is COLUMN oriented: divide
        // rhs[i], then rhs.head(i) -= rhs[i] * col(i).head(i) — for x0 that is ((c0 - x2 A02) - x1 A01) / A00 where the
        // row-oriented form below computes ((c0 - A01 x1) - A02 x2) / A00: the same x1, x2, a different association for x0.
        // NOT the default: VERDICT r05 asked what moves if the recall of the row form is wrong (tests/test_oracle_pins.py,
        // DESIGN "if the recall is wrong here"); lvo_set_qr_backsub_columns(1) switches it on.
        for (int i = nonzero_pivots - 1; i >= 0; --i) {
            c[i] = c[i] / A[i][i];
            for (int j = 0; j < i; ++j) c[j] -= c[i] * A[j][i];
        }
    } else
    for (int i = nonzero_pivots - 1; i >= 0; --i) {  // back substitution on the upper triangle
        float s = c[i];
        for (int j = i + 1; j < nonzero_pivots; ++j) s -= A[i][j] * c[j];
        c[i] = s / A[i][i];
    }
    // undo the column permutation: P = T_0 T_1 T_2 ; dst.row(perm(i)) = c.row(i)
    int perm[3] = {0, 1, 2};  // m_colsPermutation: transpositions applied on the right, k = 0..size-1
    for (int k = 0; k < size; ++k) std::swap(perm[k], perm[trans[k]]);
    for (int i = 0; i < nonzero_pivots; ++i) x[perm[i]] = c[i];
}

// R3Math::estimate_plane — Utils.cpp:32-57
void estimate_plane(const float* near_xyz, int npts, float abcd[4]) {
    float A[8][3];
    float b[8];
    for (int j = 0; j < npts; ++j) {
        A[j][0] = near_xyz[j * 3 + 0];
        A[j][1] = near_xyz[j * 3 + 1];
        A[j][2] = near_xyz[j * 3 + 2];
        b[j] = -1.0f;
    }
    float nv[3];
    colpiv_qr_solve_f32(A, npts, b, nv);
    // normvec.norm(): fixed-size Vector3f -> sqrt(x0^2 + (x1^2 + x2^2))  (Eigen 3.3 redux order)
    float n = std::sqrt(dot3f(nv[0], nv[0], nv[1], nv[1], nv[2], nv[2]));
    abcd[0] = nv[0] / n;
    abcd[1] = nv[1] / n;
    abcd[2] = nv[2] / n;
    abcd[3] = (float)(1.0 / (double)n);  // Utils.cpp:54: `1.0 / n` is a double divide, stored to f32
}

// R3Math::is_plane — Utils.cpp:59-66
bool is_plane(const float abcd[4], const float* near_xyz, int npts, float threshold) {
    for (int j = 0; j < npts; ++j) {
        float res = abcd[0] * near_xyz[j * 3 + 0] + abcd[1] * near_xyz[j * 3 + 1] + abcd[2] * near_xyz[j * 3 + 2] + abcd[3];
        if (std::fabs(res) > threshold) return false;
    }
    return true;
}

// Plane::Plane(points, sq_dists) — Plane.cpp:19-25 with gates :36-43 and fit_plane :45-55
bool plane_ctor(const float* near_xyz, const float* sq_dists, int found, const lvo_params* prm, float abcd[4]) {
    abcd[0] = abcd[1] = abcd[2] = abcd[3] = 0.f;
    if (!(found >= prm->num_match_points)) return false;                                     // enough_points
    if (found < 1) return false;
    if (!((double)sq_dists[found - 1] < prm->max_dist_plane * prm->max_dist_plane)) return false;  // points_close_enough
    float est[4];
    estimate_plane(near_xyz, found, est);
    bool ok = is_plane(est, near_xyz, found, prm->planes_threshold);
    if (ok) { abcd[0] = est[0]; abcd[1] = est[1]; abcd[2] = est[2]; abcd[3] = est[3]; }
    return ok;
}

// Plane::dist_to_plane — Plane.cpp:27-29
inline float dist_to_plane(const float abcd[4], const float p[3]) {
    return abcd[0] * p[0] + abcd[1] * p[1] + abcd[2] * p[2] + abcd[3];
}

RT32 pose_X(const lvo_pose_f32& P) {
    RT32 x;
    std::memcpy(x.R, P.R, sizeof(x.R));
    std::memcpy(x.t, P.pos, sizeof(x.t));
    return x;
}
RT32 pose_LI(const lvo_pose_f32& P) {
    RT32 x;
    std::memcpy(x.R, P.RLI, sizeof(x.R));
    std::memcpy(x.t, P.tLI, sizeof(x.t));
    return x;
}

// ---------------------------------------------------------------------------------------------
// esekf pass algebra [UPSTREAM-RECALL esekfom.hpp update_iterated_dyn_share_modified]
// state dof layout: pos 0, rot 3, offset_R_L_I 6, offset_T_L_I 9, vel 12, bg 15, ba 18, grav 21
// ---------------------------------------------------------------------------------------------
void state_boxplus(lvo_state* x, const double* d) {
    for (int i = 0; i < 3; ++i) x->pos[i] += d[0 + i];
    { double e[4], o[4]; so3_exp(d + 3, 1.0, e); quat_mul(x->rot, e, o); std::memcpy(x->rot, o, sizeof(o)); }
    { double e[4], o[4]; so3_exp(d + 6, 1.0, e); quat_mul(x->offset_R_L_I, e, o); std::memcpy(x->offset_R_L_I, o, sizeof(o)); }
    for (int i = 0; i < 3; ++i) x->offset_T_L_I[i] += d[9 + i];
    for (int i = 0; i < 3; ++i) x->vel[i] += d[12 + i];
    for (int i = 0; i < 3; ++i) x->bg[i] += d[15 + i];
    for (int i = 0; i < 3; ++i) x->ba[i] += d[18 + i];
    s2_boxplus(x->grav, d + 21);
}
void state_boxminus(const lvo_state* x, const lvo_state* o, double* d) {
    for (int i = 0; i < 3; ++i) d[0 + i] = x->pos[i] - o->pos[i];
    { double c[4], q[4]; quat_conj(o->rot, c); quat_mul(c, x->rot, q); so3_log(q, d + 3); }
    { double c[4], q[4]; quat_conj(o->offset_R_L_I, c); quat_mul(c, x->offset_R_L_I, q); so3_log(q, d + 6); }
    for (int i = 0; i < 3; ++i) d[9 + i] = x->offset_T_L_I[i] - o->offset_T_L_I[i];
    for (int i = 0; i < 3; ++i) d[12 + i] = x->vel[i] - o->vel[i];
    for (int i = 0; i < 3; ++i) d[15 + i] = x->bg[i] - o->bg[i];
    for (int i = 0; i < 3; ++i) d[18 + i] = x->ba[i] - o->ba[i];
    s2_boxminus(x->grav, o->grav, d + 21);
}

// left-multiply rows [idx, idx+r) of the 23x23 matrix M by the r x r matrix T, reading from S
inline void rows_mul(double* M, const double* S, const double* T, int idx, int r) {
    for (int c = 0; c < NS; ++c) {
        double v[3];
        for (int i = 0; i < r; ++i) {
            double s = 0;
            if (r == 3) s = dot3d(T[i * 3], S[(idx + 0) * NS + c], T[i * 3 + 1], S[(idx + 1) * NS + c], T[i * 3 + 2], S[(idx + 2) * NS + c]);
            else s = T[i * 2] * S[(idx + 0) * NS + c] + T[i * 2 + 1] * S[(idx + 1) * NS + c];
            v[i] = s;
        }
        for (int i = 0; i < r; ++i) M[(idx + i) * NS + c] = v[i];
    }
}
// right-multiply columns [idx, idx+r) of M by T^T  (M.block<1,r>(i,idx) * T^T)
inline void cols_mulT(double* M, const double* T, int idx, int r) {
    for (int i = 0; i < NS; ++i) {
        double v[3];
        for (int j = 0; j < r; ++j) {
            double s = 0;
            if (r == 3) s = dot3d(M[i * NS + idx], T[j * 3], M[i * NS + idx + 1], T[j * 3 + 1], M[i * NS + idx + 2], T[j * 3 + 2]);
            else s = M[i * NS + idx] * T[j * 2] + M[i * NS + idx + 1] * T[j * 2 + 1];
            v[j] = s;
        }
        for (int j = 0; j < r; ++j) M[i * NS + idx + j] = v[j];
    }
}

// Degeneracy stage [UNKNOWN-FORK: the 4-argument update_iterated_dyn_share_modified of Huguet57/IKFoM is not in the
// mount; config/params.yaml:51-53 only says "eigenvalues" and "magnitude depends on delta"].  Restated as solution
// remapping (Zhang, Kaess, Singh: "On degeneracy of optimization-based state estimation problems", ICRA 2016) in
// information form: eigen-decomposition of the 6x6 pose block A of H^T H (cyclic Jacobi, fixed 8 sweeps), projector
// Pn = sum over eigenvalues >= threshold of v v^T; H^T H <- blockdiag(Pn, I) H^T H blockdiag(Pn, I),
// H^T h <- blockdiag(Pn, I) H^T h.  The prior is untouched, so degenerate directions keep their predicted value.
void jacobi6(double A[6][6], double V[6][6]) {
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 8; ++sweep)
        for (int p = 0; p < 5; ++p)
            for (int q = p + 1; q < 6; ++q) {
                const double apq = A[p][q];
                if (std::fabs(apq) < 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 6; ++k) {   // columns p, q
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 6; ++k) {   // rows p, q
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 6; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
}
void degeneracy_stage(lvo_iter_out* sums, const lvo_params* prm, double eig[6]) {
    double A[6][6], V[6][6];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) A[i][j] = sums->HTH[i * 12 + j];
    jacobi6(A, V);
    for (int i = 0; i < 6; ++i) eig[i] = A[i][i];
    if (prm->degeneracy_mode != 2) return;
    double Pn[6][6];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double s = 0.0;
            for (int e = 0; e < 6; ++e)
                if (eig[e] >= prm->degeneracy_threshold) s += V[i][e] * V[j][e];
            Pn[i][j] = s;
        }
    double H[12][12], T[12][12];
    for (int i = 0; i < 12; ++i)
        for (int j = 0; j < 12; ++j) H[i][j] = sums->HTH[i * 12 + j];
    for (int i = 0; i < 12; ++i)        // T = blockdiag(Pn, I) H
        for (int j = 0; j < 12; ++j) {
            if (i >= 6) { T[i][j] = H[i][j]; continue; }
            double s = 0.0;
            for (int e = 0; e < 6; ++e) s += Pn[i][e] * H[e][j];
            T[i][j] = s;
        }
    for (int i = 0; i < 12; ++i)        // H' = T blockdiag(Pn, I)
        for (int j = 0; j < 12; ++j) {
            if (j >= 6) { sums->HTH[i * 12 + j] = T[i][j]; continue; }
            double s = 0.0;
            for (int e = 0; e < 6; ++e) s += T[i][e] * Pn[e][j];
            sums->HTH[i * 12 + j] = s;
        }
    double h6[6];
    for (int i = 0; i < 6; ++i) {
        double s = 0.0;
        for (int e = 0; e < 6; ++e) s += Pn[i][e] * sums->HTh[e];
        h6[i] = s;
    }
    for (int i = 0; i < 6; ++i) sums->HTh[i] = h6[i];
}

int kf_step(lvo_state* x, const lvo_state* x_prop, const double* P_prop, const lvo_params* prm,
            const lvo_iter_out* sums_in, double* dx_out, int finalize, double* P_out) {
    lvo_iter_out sums_mod = *sums_in;
    if (prm->degeneracy_mode) {
        double eig[6];
        degeneracy_stage(&sums_mod, prm, eig);
    }
    const lvo_iter_out* sums = &sums_mod;
    const double R = prm->lidar_noise;
    double dx[NS], dx_new[NS];
    state_boxminus(x, x_prop, dx);
    std::memcpy(dx_new, dx, sizeof(dx));
    std::vector<double> P(P_prop, P_prop + NS * NS);

    const int so3_idx[2] = {3, 6};
    for (int b = 0; b < 2; ++b) {
        int idx = so3_idx[b];
        double A[9], At[9];
        A_matrix(dx + idx, A);
        mat3_T(A, At);  // res_temp_SO3 = A_matrix(seg).transpose()
        double t[3];
        mat3_vec(At, dx_new + idx, t);
        dx_new[idx] = t[0]; dx_new[idx + 1] = t[1]; dx_new[idx + 2] = t[2];
        rows_mul(P.data(), P.data(), At, idx, 3);
        cols_mulT(P.data(), At, idx, 3);
    }
    {
        int idx = 21;
        double Nx[6], Mx[6], T[4];
        s2_Nx_yy(x->grav, Nx);
        s2_Mx(x_prop->grav, dx + idx, Mx);
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                T[i * 2 + j] = Nx[i * 3 + 0] * Mx[0 * 2 + j] + Nx[i * 3 + 1] * Mx[1 * 2 + j] + Nx[i * 3 + 2] * Mx[2 * 2 + j];
        double t0 = T[0] * dx_new[idx] + T[1] * dx_new[idx + 1];
        double t1 = T[2] * dx_new[idx] + T[3] * dx_new[idx + 1];
        dx_new[idx] = t0; dx_new[idx + 1] = t1;
        rows_mul(P.data(), P.data(), T, idx, 2);
        cols_mulT(P.data(), T, idx, 2);
    }

    // n (=23) > dof_Measurement branch is only taken with < 23 matches; the algebra below is the
    // "else" branch  K = (H^T H + (P/R)^-1)^-1 H^T, which is the same estimator (matrix inversion
    // lemma) — this oracle always uses the normal-equation form and documents the deviation.
    std::vector<double> PR(NS * NS), P_temp(NS * NS), P_inv(NS * NS);
    for (int i = 0; i < NS * NS; ++i) PR[i] = P[i] / R;
    mat_inverse(PR.data(), NS, P_temp.data());
    for (int i = 0; i < 12; ++i)
        for (int j = 0; j < 12; ++j) P_temp[i * NS + j] += sums->HTH[i * 12 + j];
    mat_inverse(P_temp.data(), NS, P_inv.data());
    double K_h[NS];
    std::vector<double> K_x(NS * NS, 0.0);
    for (int i = 0; i < NS; ++i) {
        double s = 0;
        for (int j = 0; j < 12; ++j) s += P_inv[i * NS + j] * sums->HTh[j];
        K_h[i] = s;
        for (int c = 0; c < 12; ++c) {
            double t = 0;
            for (int j = 0; j < 12; ++j) t += P_inv[i * NS + j] * sums->HTH[j * 12 + c];
            K_x[i * NS + c] = t;
        }
    }
    double dxo[NS];
    for (int i = 0; i < NS; ++i) {
        double s = 0;
        for (int j = 0; j < NS; ++j) s += (K_x[i * NS + j] - (i == j ? 1.0 : 0.0)) * dx_new[j];
        dxo[i] = K_h[i] + s;
    }
    state_boxplus(x, dxo);
    if (dx_out) std::memcpy(dx_out, dxo, sizeof(dxo));
    int converge = 1;
    for (int i = 0; i < NS; ++i)
        if (std::fabs(dxo[i]) > prm->limits[i]) { converge = 0; break; }

    if (finalize && P_out) {
        std::vector<double> L(P);  // L_ = P_
        for (int b = 0; b < 2; ++b) {
            int idx = so3_idx[b];
            double A[9], At[9];
            A_matrix(dxo + idx, A);
            mat3_T(A, At);
            rows_mul(L.data(), P.data(), At, idx, 3);
            // K_x.block<3,1>(idx,i) = At * K_x.block<3,1>(idx,i), i < 12
            for (int c = 0; c < 12; ++c) {
                double v[3] = {K_x[(idx + 0) * NS + c], K_x[(idx + 1) * NS + c], K_x[(idx + 2) * NS + c]}, o[3];
                mat3_vec(At, v, o);
                for (int i = 0; i < 3; ++i) K_x[(idx + i) * NS + c] = o[i];
            }
            cols_mulT(L.data(), At, idx, 3);
            cols_mulT(P.data(), At, idx, 3);
        }
        {
            int idx = 21;
            double Nx[6], Mx[6], T[4];
            s2_Nx_yy(x->grav, Nx);
            s2_Mx(x_prop->grav, dxo + idx, Mx);
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j)
                    T[i * 2 + j] = Nx[i * 3 + 0] * Mx[0 * 2 + j] + Nx[i * 3 + 1] * Mx[1 * 2 + j] + Nx[i * 3 + 2] * Mx[2 * 2 + j];
            rows_mul(L.data(), P.data(), T, idx, 2);
            for (int c = 0; c < 12; ++c) {
                double v0 = K_x[(idx + 0) * NS + c], v1 = K_x[(idx + 1) * NS + c];
                K_x[(idx + 0) * NS + c] = T[0] * v0 + T[1] * v1;
                K_x[(idx + 1) * NS + c] = T[2] * v0 + T[3] * v1;
            }
            cols_mulT(L.data(), T, idx, 2);
            cols_mulT(P.data(), T, idx, 2);
        }
        for (int i = 0; i < NS; ++i)
            for (int j = 0; j < NS; ++j) {
                double s = 0;
                for (int c = 0; c < 12; ++c) s += K_x[i * NS + c] * P[c * NS + j];
                P_out[i * NS + j] = L[i * NS + j] - s;
            }
    }
    return converge;
}

}  // namespace

// =============================================================================================
// C interface
// =============================================================================================
extern "C" {

void lvo_default_params(lvo_params* p) {  // config/params.yaml:32,46-53; main.cpp:145
    p->max_num_iters = 3;
    p->num_match_points = 5;
    p->max_dist_plane = 2.0;
    p->planes_threshold = 5.e-2f;
    p->estimate_extrinsics = 0;
    p->lidar_noise = 0.001;
    for (int i = 0; i < 23; ++i) p->limits[i] = 0.001;
    p->degeneracy_mode = 0;            // the fork's stage is unknown: off unless a test asks for the restatement
    p->degeneracy_threshold = 5.0;     // config/params.yaml:52
}

// State::State(const state_ikfom&, double) — State.cpp:51-62 (f64 -> f32 casts)
void lvo_state_to_pose(const lvo_state* s, lvo_pose_f32* out) {
    double R[9], RLI[9];
    quat_to_rot(s->rot, R);
    quat_to_rot(s->offset_R_L_I, RLI);
    for (int i = 0; i < 9; ++i) { out->R[i] = (float)R[i]; out->RLI[i] = (float)RLI[i]; }
    for (int i = 0; i < 3; ++i) { out->pos[i] = (float)s->pos[i]; out->tLI[i] = (float)s->offset_T_L_I[i]; }
}

// Mapper.cpp:51   X * X.I_Rt_L() * p
void lvo_transform_scan(const lvo_pose_f32* pose, const float* scan_xyz, size_t n, float* out_xyz) {
    RT32 T = rt_compose(pose_X(*pose), pose_LI(*pose));  // State.cpp:83-85 -> RotTransl.cpp:36-41
    for (size_t i = 0; i < n; ++i) rt_apply(T, scan_xyz + 3 * i, out_xyz + 3 * i);
}

void lvo_knn_brute(const float* map_xyz, size_t m, const float* q_xyz, size_t n, int k,
                   uint32_t* idx, float* d2, int32_t* found, int64_t* n_ties, int nthreads) {
    int64_t ties = 0;
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads > 0 ? nthreads : 1) reduction(+ : ties)
    for (int64_t qi = 0; qi < (int64_t)n; ++qi) {
        TopK best(k + 1 <= 16 ? k + 1 : k);  // keep k+1 to detect a tie at rank k
        const float* q = q_xyz + 3 * qi;
        // a query with a NaN / infinite coordinate has no neighbours: the reference's tree search would compare NaN
        // distances (an unspecified candidate set) and then discard the match at Plane.cpp:42 (NaN < MAX^2 is false);
        // "never chosen" is the only defined outcome, found = 0 reproduces it
        if (!(std::isfinite(q[0]) && std::isfinite(q[1]) && std::isfinite(q[2]))) {
            write_topk(TopK(k), k, idx + (size_t)qi * k, d2 + (size_t)qi * k, found ? found + qi : nullptr);
            continue;
        }
        for (size_t j = 0; j < m; ++j) best.push(Cand{calc_dist(q, map_xyz + 3 * j), (uint32_t)j});
        if (best.n > k && best.c[k].d == best.c[k - 1].d) ++ties;
        TopK out(k);
        for (int j = 0; j < best.n && j < k; ++j) out.push(best.c[j]);
        write_topk(out, k, idx + (size_t)qi * k, d2 + (size_t)qi * k, found ? found + qi : nullptr);
    }
    if (n_ties) *n_ties = ties;
}

void* lvo_kdtree_build(const float* map_xyz, size_t m) {
    KdTree* T = new KdTree();
    T->m = m;
    T->pool.reserve(m + 1);
    std::vector<uint32_t> ids(m);
    for (size_t i = 0; i < m; ++i) ids[i] = (uint32_t)i;
    T->root = kd_build(*T, ids, 0, m, map_xyz);
    return T;
}
void lvo_kdtree_free(void* tree) { delete static_cast<KdTree*>(tree); }
size_t lvo_kdtree_size(const void* tree) { return static_cast<const KdTree*>(tree)->m; }

void lvo_kdtree_knn(const void* tree, const float* q_xyz, size_t n, int k, uint32_t* idx, float* d2,
                    int32_t* found, int nthreads) {
    const KdTree* T = static_cast<const KdTree*>(tree);
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads > 0 ? nthreads : 1)
    for (int64_t qi = 0; qi < (int64_t)n; ++qi) {
        TopK best(k);
        const float* q = q_xyz + 3 * qi;
        if (std::isfinite(q[0]) && std::isfinite(q[1]) && std::isfinite(q[2])) kd_search(T->root, q, best);   // see lvo_knn_brute
        write_topk(best, k, idx + (size_t)qi * k, d2 + (size_t)qi * k, found ? found + qi : nullptr);
    }
}

int lvo_plane_fit(const float* near_xyz, const float* sq_dists, int found, const lvo_params* prm, float abcd[4]) {
    return plane_ctor(near_xyz, sq_dists, found, prm, abcd) ? 1 : 0;
}

void lvo_plane_fit_f64(const float* near_xyz, int npts, double abcd[4]) {
    // normal equations in f64 on centred data are not what the reference does; solve the same
    // A n = -1 least-squares system by f64 Householder QR without pivoting subtleties (3 columns).
    double A[8][3], c[8];
    for (int i = 0; i < npts; ++i) {
        for (int j = 0; j < 3; ++j) A[i][j] = near_xyz[i * 3 + j];
        c[i] = -1.0;
    }
    for (int k = 0; k < 3; ++k) {
        double s = 0;
        for (int i = k; i < npts; ++i) s += A[i][k] * A[i][k];
        double nrm = std::sqrt(s);
        if (nrm == 0) continue;
        double alpha = A[k][k] >= 0 ? -nrm : nrm;
        double v[8];
        for (int i = k; i < npts; ++i) v[i] = A[i][k];
        v[k] -= alpha;
        double vs = 0;
        for (int i = k; i < npts; ++i) vs += v[i] * v[i];
        if (vs == 0) continue;
        for (int j = k; j < 3; ++j) {
            double d = 0;
            for (int i = k; i < npts; ++i) d += v[i] * A[i][j];
            d = 2 * d / vs;
            for (int i = k; i < npts; ++i) A[i][j] -= d * v[i];
        }
        double d = 0;
        for (int i = k; i < npts; ++i) d += v[i] * c[i];
        d = 2 * d / vs;
        for (int i = k; i < npts; ++i) c[i] -= d * v[i];
    }
    double x[3];
    for (int i = 2; i >= 0; --i) {
        double s = c[i];
        for (int j = i + 1; j < 3; ++j) s -= A[i][j] * x[j];
        x[i] = s / A[i][i];
    }
    double n = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    abcd[0] = x[0] / n; abcd[1] = x[1] / n; abcd[2] = x[2] / n; abcd[3] = 1.0 / n;
}

// Localizator::calculate_H — Localizator.cpp:29-57 (one row)
void lvo_calculate_H_row(const lvo_state* s, const float p_w[3], const float abcd[4], float dist,
                         int estimate_extrinsics, double Hrow[12], double* h) {
    lvo_pose_f32 S;
    lvo_state_to_pose(s, &S);                                   // State S(s, 0.)  :33
    RT32 X = pose_X(S), LI = pose_LI(S);
    RT32 back = rt_compose(rt_inv(LI), rt_inv(X));              // S.I_Rt_L().inv() * S.inv()  :38
    float p_lidar[3], p_imu[3];
    rt_apply(back, p_w, p_lidar);
    rt_apply(LI, p_lidar, p_imu);                               // :39
    double qc[4], R_inv[9], I_R_L_inv[9];
    quat_conj(s->rot, qc);            quat_to_rot(qc, R_inv);       // :43
    quat_conj(s->offset_R_L_I, qc);   quat_to_rot(qc, I_R_L_inv);   // :44
    double n[3] = {(double)abcd[0], (double)abcd[1], (double)abcd[2]};  // Normal.cpp:36-38
    double C[3], B[3], A[3], t[3];
    mat3_vec(R_inv, n, C);                                      // :47
    mat3_vec(I_R_L_inv, C, t);
    double pl[3] = {(double)p_lidar[0], (double)p_lidar[1], (double)p_lidar[2]};  // Point.cpp:132-135
    double pi[3] = {(double)p_imu[0], (double)p_imu[1], (double)p_imu[2]};
    cross3(pl, t, B);                                           // :48
    cross3(pi, C, A);                                           // :49
    for (int i = 0; i < 12; ++i) Hrow[i] = 0.0;                 // :31 Zero()
    Hrow[0] = abcd[0]; Hrow[1] = abcd[1]; Hrow[2] = abcd[2];    // :51
    Hrow[3] = A[0]; Hrow[4] = A[1]; Hrow[5] = A[2];
    if (estimate_extrinsics) {                                  // :52
        Hrow[6] = B[0]; Hrow[7] = B[1]; Hrow[8] = B[2];
        Hrow[9] = C[0]; Hrow[10] = C[1]; Hrow[11] = C[2];
    }
    *h = -(double)dist;                                         // :55
}

void lvo_iterate(const lvo_state* s, const lvo_params* prm, const void* tree, const float* map_xyz,
                 size_t m, const float* scan_xyz, size_t n, lvo_iter_out* out, uint32_t* knn_idx,
                 float* knn_d2, uint8_t* valid, float* abcd_out, float* dist_out, double* Hrows,
                 double* h_out, int nthreads) {
    const int k = prm->num_match_points;
    lvo_pose_f32 pose;
    lvo_state_to_pose(s, &pose);
    RT32 T = rt_compose(pose_X(pose), pose_LI(pose));
    std::vector<double> rows(n * 12, 0.0), hv(n, 0.0);
    std::vector<uint8_t> ok(n, 0);
    const KdTree* kd = static_cast<const KdTree*>(tree);
    const bool have_map = kd ? kd->m > 0 : m > 0;  // Mapper::match returns empty without a map (Mapper.cpp:42)
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads > 0 ? nthreads : 1)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        float pw[3];
        rt_apply(T, scan_xyz + 3 * i, pw);                      // Mapper.cpp:51
        TopK best(k);
        if (have_map) {
            if (kd) kd_search(kd->root, pw, best);              // Mapper.cpp:86 Nearest_Search
            else for (size_t j = 0; j < m; ++j) best.push(Cand{calc_dist(pw, map_xyz + 3 * j), (uint32_t)j});
        }
        float near[8 * 3], sq[8];
        for (int j = 0; j < best.n; ++j) {
            const float* mp = map_xyz + 3 * (size_t)best.c[j].i;
            near[3 * j] = mp[0]; near[3 * j + 1] = mp[1]; near[3 * j + 2] = mp[2];
            sq[j] = best.c[j].d;
        }
        if (knn_idx) for (int j = 0; j < k; ++j) knn_idx[i * k + j] = j < best.n ? best.c[j].i : 0xFFFFFFFFu;
        if (knn_d2) for (int j = 0; j < k; ++j) knn_d2[i * k + j] = j < best.n ? best.c[j].d : INFINITY;
        float abcd[4];
        bool chosen = have_map && plane_ctor(near, sq, best.n, prm, abcd);  // Mapper.cpp:89, Match::is_chosen
        float d = 0.f;
        if (chosen) {
            d = dist_to_plane(abcd, pw);                        // Match.cpp:21
            lvo_calculate_H_row(s, pw, abcd, d, prm->estimate_extrinsics, &rows[i * 12], &hv[i]);
            ok[i] = 1;
        } else {
            abcd[0] = abcd[1] = abcd[2] = abcd[3] = 0.f;
        }
        if (abcd_out) std::memcpy(abcd_out + 4 * i, abcd, sizeof(abcd));
        if (dist_out) dist_out[i] = d;
    }
    // H^T H and H^T h (esekf a-8; f64).  Fixed chunks of 1024 points are contracted independently (in
    // parallel) and the chunk partials are added in chunk order: deterministic for any thread count.
    std::memset(out, 0, sizeof(*out));
    const size_t CH = 1024, nch = (n + CH - 1) / CH;
    std::vector<lvo_iter_out> part(nch);
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (int64_t c = 0; c < (int64_t)nch; ++c) {
        lvo_iter_out& o = part[c];
        std::memset(&o, 0, sizeof(o));
        const size_t i1 = std::min(n, (size_t)(c + 1) * CH);
        for (size_t i = (size_t)c * CH; i < i1; ++i) {
            if (!ok[i]) continue;
            const double* r = &rows[i * 12];
            for (int a = 0; a < 12; ++a) {
                for (int b = 0; b < 12; ++b) o.HTH[a * 12 + b] += r[a] * r[b];
                o.HTh[a] += r[a] * hv[i];
            }
            o.sum_h2 += hv[i] * hv[i];
            o.n_valid += 1;
        }
    }
    for (size_t c = 0; c < nch; ++c) {
        for (int a = 0; a < 144; ++a) out->HTH[a] += part[c].HTH[a];
        for (int a = 0; a < 12; ++a) out->HTh[a] += part[c].HTh[a];
        out->sum_h2 += part[c].sum_h2;
        out->n_valid += part[c].n_valid;
    }
    if (valid) std::memcpy(valid, ok.data(), n);
    if (Hrows) std::memcpy(Hrows, rows.data(), n * 12 * sizeof(double));
    if (h_out) std::memcpy(h_out, hv.data(), n * sizeof(double));
}

int lvo_kf_step(lvo_state* x, const lvo_state* x_prop, const double* P_prop, const lvo_params* prm,
                const lvo_iter_out* sums, double dx_out[23], int finalize, double* P_out) {
    return kf_step(x, x_prop, P_prop, prm, sums, dx_out, finalize, P_out);
}

void lvo_degeneracy(lvo_iter_out* sums, const lvo_params* prm, double eig[6]) { degeneracy_stage(sums, prm, eig); }

void lvo_boxplus(lvo_state* x, const double dx[23]) { state_boxplus(x, dx); }
void lvo_boxminus(const lvo_state* x, const lvo_state* other, double dx[23]) { state_boxminus(x, other, dx); }

// esekf::update_iterated_dyn_share_modified [UPSTREAM-RECALL]; loop from -1.
int lvo_update(lvo_state* x, double* P, const lvo_params* prm, const void* tree, const float* map_xyz,
               size_t m, const float* scan_xyz, size_t n, double* trace, lvo_iter_out* per_pass_out,
               int nthreads) {
    const int maximum_iter = prm->max_num_iters;
    lvo_state x_prop = *x;
    std::vector<double> P_prop(P, P + NS * NS);
    int t = 0, passes = 0;
    for (int i = -1; i < maximum_iter; ++i) {
        lvo_iter_out sums;
        lvo_iterate(x, prm, tree, map_xyz, m, scan_xyz, n, &sums, nullptr, nullptr, nullptr, nullptr,
                    nullptr, nullptr, nullptr, nthreads);
        if (per_pass_out) per_pass_out[passes] = sums;
        bool valid = sums.n_valid > 0;  // h_share_model: valid=false when there are no matches
        if (!valid) {
            if (trace) {
                for (int j = 0; j < 23; ++j) trace[passes * 49 + j] = 0.0;
                std::memcpy(trace + passes * 49 + 23, x, 26 * sizeof(double));
            }
            ++passes;
            continue;
        }
        // the posterior P is only committed on the terminal pass (t > 1 or last iteration)
        double dxo[23];
        std::vector<double> P_post(NS * NS);
        int converge = kf_step(x, &x_prop, P_prop.data(), prm, &sums, dxo, 1, P_post.data());
        if (converge) t++;
        bool last = (t > 1 || i == maximum_iter - 1);
        if (last) std::memcpy(P, P_post.data(), NS * NS * sizeof(double));
        if (trace) {
            std::memcpy(trace + passes * 49, dxo, 23 * sizeof(double));
            std::memcpy(trace + passes * 49 + 23, x, 26 * sizeof(double));
        }
        ++passes;
        if (last) return passes;
    }
    return passes;
}

// esekf::predict [UPSTREAM-RECALL] with LIMO-Velo / FAST-LIO process model get_f, df_dx, df_dw.
void lvo_predict(lvo_state* x, double* P, double dt, const double* Q, const double acc[3], const double gyro[3]) {
    // flat (24-dim) indices: pos 0, rot 3, offR 6, offT 9, vel 12, bg 15, ba 18, grav 21(3)
    double f[24] = {0};
    double omega[3] = {gyro[0] - x->bg[0], gyro[1] - x->bg[1], gyro[2] - x->bg[2]};
    double am[3] = {acc[0] - x->ba[0], acc[1] - x->ba[1], acc[2] - x->ba[2]};
    double R[9], a_in[3];
    quat_to_rot(x->rot, R);
    mat3_vec(R, am, a_in);
    for (int i = 0; i < 3; ++i) { f[i] = x->vel[i]; f[3 + i] = omega[i]; f[12 + i] = a_in[i] + x->grav[i]; }
    std::vector<double> fx(24 * NS, 0.0), fw(24 * 12, 0.0);
    for (int i = 0; i < 3; ++i) fx[(0 + i) * NS + 12 + i] = 1.0;
    double Ha[9], RH[9];
    hat3(am, Ha);
    mat3_mul(R, Ha, RH);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            fx[(12 + i) * NS + 3 + j] = -RH[i * 3 + j];
            fx[(12 + i) * NS + 18 + j] = -R[i * 3 + j];
        }
    double zero2[2] = {0, 0}, Mx0[6];
    s2_Mx(x->grav, zero2, Mx0);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 2; ++j) fx[(12 + i) * NS + 21 + j] = Mx0[i * 2 + j];
    for (int i = 0; i < 3; ++i) fx[(3 + i) * NS + 15 + i] = -1.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) fw[(12 + i) * 12 + 3 + j] = -R[i * 3 + j];
    for (int i = 0; i < 3; ++i) { fw[(3 + i) * 12 + 0 + i] = -1.0; fw[(15 + i) * 12 + 6 + i] = 1.0; fw[(18 + i) * 12 + 9 + i] = 1.0; }

    lvo_state xb = *x;
    // x_.oplus(f_, dt)
    for (int i = 0; i < 3; ++i) x->pos[i] += f[i] * dt;
    { double e[4], o[4]; so3_exp(f + 3, dt, e); quat_mul(x->rot, e, o); std::memcpy(x->rot, o, sizeof(o)); }
    { double e[4], o[4]; so3_exp(f + 6, dt, e); quat_mul(x->offset_R_L_I, e, o); std::memcpy(x->offset_R_L_I, o, sizeof(o)); }
    for (int i = 0; i < 3; ++i) x->offset_T_L_I[i] += f[9 + i] * dt;
    for (int i = 0; i < 3; ++i) x->vel[i] += f[12 + i] * dt;
    for (int i = 0; i < 3; ++i) x->bg[i] += f[15 + i] * dt;
    for (int i = 0; i < 3; ++i) x->ba[i] += f[18 + i] * dt;
    s2_oplus3(x->grav, f + 21, dt);

    std::vector<double> F1(NS * NS, 0.0), fxf(NS * NS, 0.0), fwf(NS * 12, 0.0);
    for (int i = 0; i < NS; ++i) F1[i * NS + i] = 1.0;
    // vect states: (dof idx, flat dim) pairs
    const int vidx[5] = {0, 9, 12, 15, 18};
    for (int b = 0; b < 5; ++b)
        for (int j = 0; j < 3; ++j) {
            for (int c = 0; c < NS; ++c) fxf[(vidx[b] + j) * NS + c] = fx[(vidx[b] + j) * NS + c];
            for (int c = 0; c < 12; ++c) fwf[(vidx[b] + j) * 12 + c] = fw[(vidx[b] + j) * 12 + c];
        }
    const int sidx[2] = {3, 6};
    for (int b = 0; b < 2; ++b) {
        int idx = sidx[b], dim = sidx[b];
        double seg[3] = {-1 * f[dim] * dt, -1 * f[dim + 1] * dt, -1 * f[dim + 2] * dt};
        // F_x1 block = exp(seg, scalar(1/2)==0) = identity [UPSTREAM-RECALL quirk]; already identity.
        double A[9];
        A_matrix(seg, A);
        for (int c = 0; c < NS; ++c) {
            double v[3] = {fx[(dim) * NS + c], fx[(dim + 1) * NS + c], fx[(dim + 2) * NS + c]}, o[3];
            mat3_vec(A, v, o);
            for (int i = 0; i < 3; ++i) fxf[(idx + i) * NS + c] = o[i];
        }
        for (int c = 0; c < 12; ++c) {
            double v[3] = {fw[(dim) * 12 + c], fw[(dim + 1) * 12 + c], fw[(dim + 2) * 12 + c]}, o[3];
            mat3_vec(A, v, o);
            for (int i = 0; i < 3; ++i) fwf[(idx + i) * 12 + c] = o[i];
        }
    }
    {
        int idx = 21, dim = 21;
        double seg[3] = {f[dim] * dt, f[dim + 1] * dt, f[dim + 2] * dt};
        double Nx[6], Mx[6];
        s2_Nx_yy(x->grav, Nx);
        s2_Mx(xb.grav, zero2, Mx);
        // F_x1 block<2,2> = Nx * I * Mx  (exp with scalar(1/2)==0 is identity)
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                F1[(idx + i) * NS + idx + j] = Nx[i * 3] * Mx[0 * 2 + j] + Nx[i * 3 + 1] * Mx[1 * 2 + j] + Nx[i * 3 + 2] * Mx[2 * 2 + j];
        double Hb[9], A[9], At[9], HA[9];
        hat3(xb.grav, Hb);
        A_matrix(seg, A);
        mat3_T(A, At);
        mat3_mul(Hb, At, HA);
        double T[6];  // -Nx * I * hat(x_before) * A^T   (2x3)
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 3; ++j)
                T[i * 3 + j] = -(Nx[i * 3] * HA[0 * 3 + j] + Nx[i * 3 + 1] * HA[1 * 3 + j] + Nx[i * 3 + 2] * HA[2 * 3 + j]);
        for (int c = 0; c < NS; ++c)
            for (int i = 0; i < 2; ++i)
                fxf[(idx + i) * NS + c] = T[i * 3] * fx[(dim) * NS + c] + T[i * 3 + 1] * fx[(dim + 1) * NS + c] + T[i * 3 + 2] * fx[(dim + 2) * NS + c];
        for (int c = 0; c < 12; ++c)
            for (int i = 0; i < 2; ++i)
                fwf[(idx + i) * 12 + c] = T[i * 3] * fw[(dim) * 12 + c] + T[i * 3 + 1] * fw[(dim + 1) * 12 + c] + T[i * 3 + 2] * fw[(dim + 2) * 12 + c];
    }
    for (int i = 0; i < NS * NS; ++i) F1[i] += fxf[i] * dt;
    // P = F1 P F1^T + (dt fwf) Q (dt fwf)^T
    std::vector<double> FP(NS * NS, 0.0), Pn(NS * NS, 0.0), GQ(NS * 12, 0.0);
    for (int i = 0; i < NS; ++i)
        for (int j = 0; j < NS; ++j) {
            double s = 0;
            for (int c = 0; c < NS; ++c) s += F1[i * NS + c] * P[c * NS + j];
            FP[i * NS + j] = s;
        }
    for (int i = 0; i < NS; ++i)
        for (int j = 0; j < 12; ++j) {
            double s = 0;
            for (int c = 0; c < 12; ++c) s += (dt * fwf[i * 12 + c]) * Q[c * 12 + j];
            GQ[i * 12 + j] = s;
        }
    for (int i = 0; i < NS; ++i)
        for (int j = 0; j < NS; ++j) {
            double s = 0;
            for (int c = 0; c < NS; ++c) s += FP[i * NS + c] * F1[j * NS + c];
            double q = 0;
            for (int c = 0; c < 12; ++c) q += GQ[i * 12 + c] * (dt * fwf[j * 12 + c]);
            Pn[i * NS + j] = s + q;
        }
    std::memcpy(P, Pn.data(), NS * NS * sizeof(double));
}

// KD_TREE::Add_Points with downsampling [UPSTREAM-RECALL ikd-Tree], sequential restatement.
size_t lvo_map_add(const float* map_xyz, size_t m, const float* new_xyz, size_t k, int downsample, float box_length,
                   float* out_xyz) {
    struct P { float x, y, z; };
    std::vector<P> pts(m + k);
    std::vector<uint8_t> alive(m + k, 0);
    for (size_t i = 0; i < m; ++i) { pts[i] = {map_xyz[3 * i], map_xyz[3 * i + 1], map_xyz[3 * i + 2]}; alive[i] = 1; }
    for (size_t j = 0; j < k; ++j) pts[m + j] = {new_xyz[3 * j], new_xyz[3 * j + 1], new_xyz[3 * j + 2]};
    if (!downsample) {
        for (size_t j = 0; j < k; ++j) alive[m + j] = 1;
    } else {
        auto cell = [&](float v) { return (int64_t)std::floor(v / box_length); };
        auto key = [&](const P& p) {
            return (uint64_t)((cell(p.x) + (1 << 20)) & 0x1fffff) | ((uint64_t)((cell(p.y) + (1 << 20)) & 0x1fffff) << 21) |
                   ((uint64_t)((cell(p.z) + (1 << 20)) & 0x1fffff) << 42);
        };
        std::unordered_map<uint64_t, std::vector<uint32_t>> boxes;  // current occupants, in insertion order
        boxes.reserve(m + k);
        for (size_t i = 0; i < m; ++i) boxes[key(pts[i])].push_back((uint32_t)i);
        for (size_t j = 0; j < k; ++j) {
            const P& p = pts[m + j];
            // Box_of_Point / mid_point exactly as upstream: min = floor(x/len)*len, max = min+len, mid = min+(max-min)/2
            float mid[3];
            const float c[3] = {p.x, p.y, p.z};
            for (int a = 0; a < 3; ++a) {
                float vmin = std::floor(c[a] / box_length) * box_length;
                float vmax = vmin + box_length;
                mid[a] = (float)(vmin + (vmax - vmin) / 2.0);
            }
            std::vector<uint32_t>& occ = boxes[key(p)];
            const float pf[3] = {p.x, p.y, p.z};
            float min_dist = calc_dist(pf, mid);
            uint32_t best = (uint32_t)(m + j);
            for (uint32_t e : occ) {
                const float ef[3] = {pts[e].x, pts[e].y, pts[e].z};
                float d = calc_dist(ef, mid);
                if (d < min_dist) { min_dist = d; best = e; }
            }
            if (occ.size() > 1 || best == (uint32_t)(m + j)) {
                for (uint32_t e : occ) alive[e] = 0;  // Delete_by_range(Box_of_Point)
                occ.clear();
                occ.push_back(best);                  // Add_by_point(downsample_result)
                alive[best] = 1;
            }
        }
    }
    size_t n = 0;
    for (size_t i = 0; i < m + k; ++i)
        if (alive[i]) { out_xyz[3 * n] = pts[i].x; out_xyz[3 * n + 1] = pts[i].y; out_xyz[3 * n + 2] = pts[i].z; ++n; }
    return n;
}

// ---- row f-2 ------------------------------------------------------------------------------------------
// sin / cos for f32 arguments: evaluated through a fixed f64 polynomial and rounded to f32, so that the
// oracle and the device produce the same bits (libm's sinf / cosf differ between platforms by an ulp; the
// reference calls std::sin(float) in SO3Math::Exp, include/Headers/Utils.hpp:45).
static void sincos_f32(float xf, float& sn, float& cs) {
    const double x = (double)xf;
    const double k = std::rint(x * 0.63661977236758134308);
    double r = x - k * 1.57079632673412561417e+00;
    r = r - k * 6.07710050650619224932e-11;
    r = r - k * 2.02226624879595063154e-21;
    const double z = r * r;
    double ps = 1.58969099521155010221e-10;
    ps = ps * z - 2.50507602534068634195e-08;
    ps = ps * z + 2.75573137070700676789e-06;
    ps = ps * z - 1.98412698298579493134e-04;
    ps = ps * z + 8.33333333332248946124e-03;
    ps = ps * z - 1.66666666666666324348e-01;
    const double s0 = r + r * z * ps;
    double pc = -1.13596475577881948265e-11;
    pc = pc * z + 2.08757232129817482790e-09;
    pc = pc * z - 2.75573143513906633035e-07;
    pc = pc * z + 2.48015872894767294178e-05;
    pc = pc * z - 1.38888888888741095749e-03;
    pc = pc * z + 4.16666666666666019037e-02;
    const double c0 = 1.0 - 0.5 * z + z * z * pc;
    const int q = (int)k & 3;
    const double sd = (q == 0) ? s0 : (q == 1) ? c0 : (q == 2) ? -s0 : -c0;
    const double cd = (q == 0) ? c0 : (q == 1) ? -s0 : (q == 2) ? -c0 : s0;
    sn = (float)sd;
    cs = (float)cd;
}

extern "C" void lvo_sincos_f32(const float* x, size_t n, float* sn, float* cs) {
    for (size_t i = 0; i < n; ++i) sincos_f32(x[i], sn[i], cs[i]);
}
// How far the pinned polynomial is from THIS platform's libm: the reference evaluates std::sin(float) / std::cos(float)
// (include/Headers/Utils.hpp:46), i.e. glibc's sinf / cosf on its x86-64 build.  Counts the arguments whose f32 result differs
// and the largest difference in ulps — a measurement for DESIGN.md section 6 f-2, not a parity claim.
extern "C" void lvo_sincos_vs_libm(const float* x, size_t n, int64_t* n_sin_diff, int64_t* n_cos_diff, int32_t* max_ulp) {
    int64_t ds = 0, dc = 0;
    int32_t mu = 0;
    for (size_t i = 0; i < n; ++i) {
        float sn, cs;
        sincos_f32(x[i], sn, cs);
        const float ls = std::sin(x[i]), lc = std::cos(x[i]);   // float overloads: sinf / cosf
        int32_t a, b;
        std::memcpy(&a, &sn, 4); std::memcpy(&b, &ls, 4);
        if (a != b) { ++ds; const int32_t u = a > b ? a - b : b - a; if ((a ^ b) >= 0 && u > mu) mu = u; }
        std::memcpy(&a, &cs, 4); std::memcpy(&b, &lc, 4);
        if (a != b) { ++dc; const int32_t u = a > b ? a - b : b - a; if ((a ^ b) >= 0 && u > mu) mu = u; }
    }
    *n_sin_diff = ds; *n_cos_diff = dc; *max_ulp = mu;
}

// lvo_set_sincos_libm(1): SO3Math::Exp takes this platform's sinf / cosf instead of the pinned polynomial — the mode in which
// the oracle must equal the reference's own State::propagate_f compiled here (oracle/_ref, tests/test_oracle_ref.py) bit for bit;
// the default (0) is the polynomial the device evaluates as well (lvo_sincos_vs_libm measures the distance between the two).
static int g_sincos_libm = 0;
extern "C" void lvo_set_sincos_libm(int on) { g_sincos_libm = on; }
// lvo_set_qr_backsub_columns(1): the plane fit's back substitution in Eigen's column-oriented order (colpiv_qr_solve_f32): the
// "what moves if the recall is wrong" experiment of tests/test_oracle_pins.py; NOT the default, never the reference of a parity test
extern "C" void lvo_set_qr_backsub_columns(int on) { g_qr_backsub_columns = on; }

// SO3Math::Exp<float,float>(ang_vel, dt) — include/Headers/Utils.hpp:30-53
static void so3_exp_f32(const float w[3], float dt, float E[9]) {
    const float nrm = std::sqrt(dot3f(w[0], w[0], w[1], w[1], w[2], w[2]));
    for (int i = 0; i < 9; ++i) E[i] = (i % 4 == 0) ? 1.f : 0.f;
    if (!((double)nrm > 0.0000001)) return;
    const float r[3] = {w[0] / nrm, w[1] / nrm, w[2] / nrm};
    const float K[9] = {0.f, -r[2], r[1], r[2], 0.f, -r[0], -r[1], r[0], 0.f};
    const float r_ang = nrm * dt;
    float sn, cs;
    if (g_sincos_libm) { sn = std::sin(r_ang); cs = std::cos(r_ang); }   // (float overloads = sinf / cosf: what Utils.hpp:46 calls)
    else sincos_f32(r_ang, sn, cs);
    const float c = (float)(1.0 - (double)cs);
    float cK[9], cKK[9];
    for (int i = 0; i < 9; ++i) cK[i] = c * K[i];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) cKK[i * 3 + j] = dot3f(cK[i * 3], K[j], cK[i * 3 + 1], K[3 + j], cK[i * 3 + 2], K[6 + j]);
    for (int i = 0; i < 9; ++i) E[i] = (E[i] + sn * K[i]) + cKK[i];
}

// State::propagate_f + the time bookkeeping of State::update — src/Objects/State.cpp:94-121
void lvo_state_integrate(lvo_motion_state* s, const float a[3], const float w[3], double t) {
    const float dt = (float)(t - s->time);
    const float wm[3] = {w[0] - s->bw[0], w[1] - s->bw[1], w[2] - s->bw[2]};
    float E[9], Rn[9];
    so3_exp_f32(wm, dt, E);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rn[i * 3 + j] = dot3f(s->R[i * 3], E[j], s->R[i * 3 + 1], E[3 + j], s->R[i * 3 + 2], E[6 + j]);
    const float am[3] = {a[0] - s->ba[0], a[1] - s->ba[1], a[2] - s->ba[2]};
    float v[3];
    for (int i = 0; i < 3; ++i) v[i] = dot3f(s->R[i * 3], am[0], s->R[i * 3 + 1], am[1], s->R[i * 3 + 2], am[2]) - s->g[i];
    float veln[3], posn[3];
    for (int i = 0; i < 3; ++i) {
        veln[i] = s->vel[i] + v[i] * dt;
        posn[i] = s->pos[i] + (s->vel[i] * dt + ((0.5f * v[i]) * dt) * dt);
    }
    for (int i = 0; i < 9; ++i) s->R[i] = Rn[i];
    for (int i = 0; i < 3; ++i) { s->vel[i] = veln[i]; s->pos[i] = posn[i]; }
    s->time = t;
    for (int i = 0; i < 3; ++i) { s->a[i] = 0.5f * s->a[i] + 0.5f * a[i]; s->w[i] = 0.5f * s->w[i] + 0.5f * w[i]; }
}

void lvo_deskew(const float* xyz, const double* times, size_t n, const lvo_motion_state* states, size_t n_states,
                const lvo_motion_state* Xt2, float* out_xyz) {
    RT32 X2, LI2;
    std::memcpy(X2.R, Xt2->R, sizeof(X2.R)); std::memcpy(X2.t, Xt2->pos, sizeof(X2.t));
    std::memcpy(LI2.R, Xt2->RLI, sizeof(LI2.R)); std::memcpy(LI2.t, Xt2->tLI, sizeof(LI2.t));
    const RT32 back = rt_compose(rt_inv(LI2), rt_inv(X2));  // Xt2.I_Rt_L().inv() * Xt2.inv()  (Compensator.cpp:138)
    size_t p = 0;
    for (size_t s = 0; s + 1 < n_states; ++s) {             // the reference's two-pointer walk (:130-143)
        while (p < n && states[s].time <= times[p] && times[p] <= states[s + 1].time) {
            lvo_motion_state Xtp = states[s];
            lvo_state_integrate(&Xtp, states[s].a, states[s].w, times[p]);  // :133-134
            RT32 X, LI;
            std::memcpy(X.R, Xtp.R, sizeof(X.R)); std::memcpy(X.t, Xtp.pos, sizeof(X.t));
            std::memcpy(LI.R, Xtp.RLI, sizeof(LI.R)); std::memcpy(LI.t, Xtp.tLI, sizeof(LI.t));
            float g[3];
            rt_apply(rt_compose(X, LI), xyz + 3 * p, g);    // :137
            rt_apply(back, g, out_xyz + 3 * p);             // :138
            ++p;
        }
    }
    for (; p < n; ++p) { out_xyz[3 * p] = out_xyz[3 * p + 1] = out_xyz[3 * p + 2] = std::numeric_limits<float>::quiet_NaN(); }
}

size_t lvo_voxelgrid(const float* xyz, size_t n, float leaf, float* out_xyz) {
    if (n == 0) return 0;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], xyz[3 * i + a]); mx[a] = std::max(mx[a], xyz[3 * i + a]); }
    const float inv = 1.0f / leaf;
    int64_t minb[3], divb[3];
    for (int a = 0; a < 3; ++a) {
        minb[a] = (int64_t)std::floor(mn[a] * inv);
        divb[a] = (int64_t)std::floor(mx[a] * inv) - minb[a] + 1;
    }
    std::vector<std::pair<int64_t, uint32_t>> iv(n);
    for (size_t i = 0; i < n; ++i) {
        int64_t ijk[3];
        for (int a = 0; a < 3; ++a) ijk[a] = (int64_t)std::floor(xyz[3 * i + a] * inv) - minb[a];
        iv[i] = {ijk[0] + ijk[1] * divb[0] + ijk[2] * divb[0] * divb[1], (uint32_t)i};
    }
    std::stable_sort(iv.begin(), iv.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    size_t out = 0, i = 0;
    while (i < n) {
        size_t j = i;
        float sx = 0.f, sy = 0.f, sz = 0.f;
        while (j < n && iv[j].first == iv[i].first) {
            const float* p = xyz + 3 * (size_t)iv[j].second;
            sx += p[0]; sy += p[1]; sz += p[2];
            ++j;
        }
        const float cnt = (float)(j - i);
        out_xyz[3 * out] = sx / cnt; out_xyz[3 * out + 1] = sy / cnt; out_xyz[3 * out + 2] = sz / cnt;
        ++out;
        i = j;
    }
    return out;
}

}  // extern "C"

// ---- row f-4 ----------------------------------------------------------------------------------------------
namespace {
double o_microsec2sec(uint64_t t) {   // Conversions::microsec2Sec, src/Utils/Utils.cpp:18-23
    int order = 1e6;
    int secs = t / order;
    int musecs = t % order;
    return secs + musecs * 1e-6;
}
double o_nanosec2sec(uint32_t t) {    // Conversions::nanosec2Sec, :25-30
    int order = 1e9;
    int secs = t / order;
    int nsecs = t % order;
    return secs + nsecs * 1e-9;
}
double o_raw_time(const unsigned char* rec, const lvo_cloud_format& f) {
    if (f.time_type == 0) { float v; std::memcpy(&v, rec + f.off_time, 4); return (double)v; }
    if (f.time_type == 2) { uint32_t v; std::memcpy(&v, rec + f.off_time, 4); return o_nanosec2sec(v); }
    double v; std::memcpy(&v, rec + f.off_time, 8); return v;
}
}  // namespace

extern "C" size_t lvo_cloud_ingest(const void* data, size_t n, const lvo_cloud_format* f, const lvo_ingest_params* prm, lvo_point* out) {
    if (n == 0) return 0;
    const unsigned char* raw = static_cast<const unsigned char*>(data);
    // get_begin_time: relative stamps hang off the header stamp and the first (and last) raw point
    double begin = 0.0;
    if (f->relative_time) {
        const double front = o_raw_time(raw, *f), back = o_raw_time(raw + (n - 1) * (size_t)f->point_step, *f);
        begin = o_microsec2sec(prm->header_stamp_usec) + front;
        if (!prm->stamp_beginning) begin = begin - back;
    }
    std::vector<lvo_point> kept;
    int ds_counter = 0;
    for (size_t i = 0; i < n; ++i) {
        const unsigned char* p = raw + i * (size_t)f->point_step;
        lvo_point o;
        std::memcpy(&o.x, p + f->off_x, 4);
        std::memcpy(&o.y, p + f->off_y, 4);
        std::memcpy(&o.z, p + f->off_z, 4);
        o.pad_ = 0.f;
        const float nrm = std::sqrt(dot3f(o.x, o.x, o.y, o.y, o.z, o.z));   // Eigen Vector3f::norm()
        switch (f->intensity_type) {
            case 1: std::memcpy(&o.intensity, p + f->off_intensity, 4); break;
            case 2: o.intensity = (float)p[f->off_intensity]; break;
            case 3: { uint16_t v; std::memcpy(&v, p + f->off_intensity, 2); o.intensity = (float)v; break; }
            default: o.intensity = 0.f; break;
        }
        if (f->range_type == 4) { uint32_t v; std::memcpy(&v, p + f->off_range, 4); o.range = (float)v; }
        else o.range = nrm;
        double t = o_raw_time(p, *f);
        if (f->relative_time && !prm->offset_beginning) t = prm->full_rotation_time + t;
        o.time = t + begin;
        const bool keep_point = prm->downsample_rate <= 1 || ++ds_counter % prm->downsample_rate == 0;   // PointCloudProcessor.cpp:105
        if (keep_point && prm->min_dist < nrm) kept.push_back(o);
    }
    std::stable_sort(kept.begin(), kept.end(), [](const lvo_point& a, const lvo_point& b) { return a.time < b.time; });
    for (size_t i = 0; i < kept.size(); ++i) out[i] = kept[i];
    return kept.size();
}

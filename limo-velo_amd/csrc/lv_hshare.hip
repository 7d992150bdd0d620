// lv_hshare.hip — the Eigen-free half of `IKFoM::h_share_model` for a maintainer who keeps the UNMODIFIED esekf loop
// (reference src/Modules/Localizator.cpp:105-117: h_share_model is the callback update_iterated_dyn_share_modified calls,
// :112 init_dyn_share; :132 the update) and only replaces the measurement model: esekf depends on the N x 12 Jacobian H and
// the residual vector h only through H^T H and H^T h, so the callback hands it a PSEUDO measurement of at most 12 rows with
//      h_x^T h_x = H^T H        h_x^T h = H^T h
// and both gain branches of esekf reproduce the update of the true N-row measurement.
// Host arithmetic only (no device, no context): the record is 1.2 KB.
#include <cmath>
#include <cstring>

#include "../../include/limovelo_hip.h"

namespace {

// upper Cholesky factor of the leading n x n block of a 12 x 12 row-major SPD matrix: U^T U = A.  false if a pivot is not
// positive (the block is not numerically positive definite).
bool chol_upper(const double* A, int n, double* U /* n x n row-major */) {
    for (int i = 0; i < n * n; ++i) U[i] = 0.0;
    for (int j = 0; j < n; ++j) {
        double d = A[j * 12 + j];
        for (int k = 0; k < j; ++k) d -= U[k * n + j] * U[k * n + j];
        if (!(d > 0.0) || !std::isfinite(d)) return false;
        const double ujj = std::sqrt(d);
        U[j * n + j] = ujj;
        for (int c = j + 1; c < n; ++c) {
            double s = A[j * 12 + c];
            for (int k = 0; k < j; ++k) s -= U[k * n + j] * U[k * n + c];
            U[j * n + c] = s / ujj;
        }
    }
    return true;
}

// cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (n <= 12): A = V diag(lam) V^T, columns of V orthonormal
void jacobi_eig(const double* Ain, int n, double* lam, double* V /* n x n row-major, eigenvectors in columns */) {
    double A[144];
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) { A[i * n + j] = 0.5 * (Ain[i * 12 + j] + Ain[j * 12 + i]); V[i * n + j] = i == j ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; ++i) {
            diag += A[i * n + i] * A[i * n + i];
            for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j];
        }
        if (off <= 1e-32 * diag || off == 0.0) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[p * n + q];
                if (apq == 0.0) continue;
                const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; ++i) lam[i] = A[i * n + i];
}

}  // namespace

extern "C" int lv_pseudo_measurement(const lv_sums* sums, int estimate_extrinsics, double h_x[144], double h[12], int* rows) {
    if (!sums || !h_x || !h || !rows) return LV_EINVAL;
    std::memset(h_x, 0, 144 * sizeof(double));
    std::memset(h, 0, 12 * sizeof(double));
    *rows = 0;
    if (sums->n_valid <= 0) return LV_OK;   // h_share_model: dyn_share.valid = false (no matches): nothing to hand over
    const int n = estimate_extrinsics ? 12 : 6;
    if (!estimate_extrinsics) {
        // columns 6..11 of H are zero (Localizator.cpp:52): the leading 6 x 6 block carries everything and is positive
        // definite for any scan that constrains the pose: h_x = [U | 0] with U^T U = H^T H, h = U^-T H^T h
        double U[36];
        if (chol_upper(sums->HTH, 6, U)) {
            for (int r = 0; r < 6; ++r)
                for (int c = 0; c < 6; ++c) h_x[r * 12 + c] = U[r * 6 + c];
            for (int i = 0; i < 6; ++i) {   // forward substitution with L = U^T
                double s = sums->HTh[i];
                for (int k = 0; k < i; ++k) s -= U[k * 6 + i] * h[k];
                h[i] = s / U[i * 6 + i];
            }
            *rows = 6;
            return LV_OK;
        }
        // a degenerate scene (e.g. one plane only): fall through to the rank-revealing factor of the 6 x 6 block
    }
    // rank revealing: H^T H = V diag(lam) V^T  ->  h_x = sqrt(lam+) V^T, h = lam+^(-1/2) V^T H^T h; the rows of vanished
    // eigenvalues are zero (with estimate_extrinsics the 12 x 12 matrix is only semi-definite: `pos` and `offset_T_L_I` see
    // the same normal in two frames; H^T h lies in the range of H^T H)
    double lam[12], V[144];
    jacobi_eig(sums->HTH, n, lam, V);
    double lmax = 0.0;
    for (int i = 0; i < n; ++i) lmax = lam[i] > lmax ? lam[i] : lmax;
    const double cut = 1e-12 * lmax;
    for (int r = 0; r < n; ++r) {
        if (!(lam[r] > cut)) continue;   // (row r stays zero)
        const double s = std::sqrt(lam[r]);
        double vh = 0.0;
        for (int c = 0; c < n; ++c) {
            h_x[r * 12 + c] = s * V[c * n + r];
            vh += V[c * n + r] * sums->HTh[c];
        }
        h[r] = vh / s;
    }
    *rows = n;
    return LV_OK;
}

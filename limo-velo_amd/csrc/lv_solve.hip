// lv_solve.hip — the N-independent tail of each IKFoM pass, kept on the device so that the
// iterated update never leaves the GPU:
//   reduce_partials_kernel : block partials -> the 96-double record (fixed summation order)
//   solve_kernel           : esekf::update_iterated_dyn_share_modified's per-pass algebra
//                            [IKFoM absent from the reference mount; UPSTREAM-RECALL of
//                            hku-mars/IKFoM esekfom.hpp; call site reference
//                            src/Modules/Localizator.cpp:132] — manifold projections, the two
//                            23x23 inverses, K_h / K_x, boxplus, LIMITS test (src/main.cpp:145),
//                            posterior covariance on the terminal pass — and the f32 pose constants
//                            of the next pass (State(const state_ikfom&, double),
//                            reference src/Objects/State.cpp:51-62).
#include "lv_host.hpp"
#include "lv_manifold.hpp"

namespace lv {

__global__ void kf_begin_kernel(KfDev* kf) {
    if (threadIdx.x == 0) {
        kf->t = 0;
        kf->iter = -1;  // upstream loop starts at i = -1
        kf->done = 0;
        kf->passes = 0;
        kf->fallback_queries = 0;
        compute_pose_consts(kf->x, &kf->pose);
    }
}

constexpr int RED_SEGS = 8;
__global__ __launch_bounds__(SUMS_LEN* RED_SEGS) void reduce_partials_kernel(const double* __restrict__ partials, int nblocks,
                                                                              double* __restrict__ sums, const KfDev* kf) {
    __shared__ double s_seg[RED_SEGS][SUMS_LEN];
    if (kf->done) return;
    const int o = threadIdx.x % SUMS_LEN, seg = threadIdx.x / SUMS_LEN;
    double s = 0.0;
    for (int b = seg; b < nblocks; b += RED_SEGS) s += partials[(size_t)b * SUMS_LEN + o];
    s_seg[seg][o] = s;
    __syncthreads();
    if (seg == 0) {
        double r = s_seg[0][o];
#pragma unroll
        for (int g = 1; g < RED_SEGS; ++g) r += s_seg[g][o];
        sums[o] = r;
    }
}

constexpr int LD = NS + 1;  // padded leading dimension in LDS

// in-place Gauss-Jordan inverse of an SPD 23x23 matrix held in LDS, ping-pong between M and W;
// the result ends in M.  All 576 threads must call it; tid < 529 own element (i, j).
__device__ inline void gj_inverse(double (*M)[LD], double (*W)[LD], int tid) {
    const int i = tid / NS, j = tid % NS;
    const bool act = tid < NS * NS;
    double (*src)[LD] = M;
    double (*dst)[LD] = W;
    for (int k = 0; k < NS; ++k) {
        if (act) {
            const double p = src[k][k];
            double v;
            if (i == k) {
                v = (j == k) ? 1.0 / p : src[k][j] / p;
            } else {
                const double f = src[i][k];
                v = (j == k) ? -f / p : src[i][j] - f * (src[k][j] / p);
            }
            dst[i][j] = v;
        }
        __syncthreads();
        double (*t)[LD] = src;
        src = dst;
        dst = t;
    }
    // NS is odd: after 23 swaps the result lives in W; copy back to M
    if (act) M[i][j] = src[i][j];
    __syncthreads();
}

// out = J * in * J^T with J = identity except the SO3 blocks (3,6) and the S2 block (21)
__device__ inline void mm(double (*out)[LD], const double (*a)[LD], const double (*b)[LD], bool b_transposed, int tid) {
    if (tid < NS * NS) {
        const int i = tid / NS, j = tid % NS;
        double s = 0.0;
        for (int c = 0; c < NS; ++c) s += a[i][c] * (b_transposed ? b[j][c] : b[c][j]);
        out[i][j] = s;
    }
}

__device__ inline void set_identity(double (*J)[LD], int tid) {
    if (tid < NS * NS) J[tid / NS][tid % NS] = (tid / NS == tid % NS) ? 1.0 : 0.0;
}

// thread 0: J blocks from a tangent vector `seg` (23): A(seg[3:6])^T, A(seg[6:9])^T, Nx(x.grav) Mx(xprop.grav, seg[21:23])
__device__ inline void fill_projection(double (*J)[LD], const double* seg, const double* x, const double* xprop) {
    for (int b = 0; b < 2; ++b) {
        const int idx = b == 0 ? 3 : 6;
        double A[9];
        d_A_matrix(seg + idx, A);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) J[idx + r][idx + c] = A[c * 3 + r];  // transpose
    }
    double T[4];
    d_s2_proj(x + 23, xprop + 23, seg + 21, T);
    J[21][21] = T[0]; J[21][22] = T[1]; J[22][21] = T[2]; J[22][22] = T[3];
}

__global__ __launch_bounds__(576) void solve_kernel(KfDev* kf, const double* __restrict__ sums, SolveParams prm) {
    __shared__ double sP[NS][LD], sA[NS][LD], sB[NS][LD], sJ[NS][LD];
    __shared__ double sKx[NS][12], sHTH[12][12], sHTh[12], sdx[NS], sdxnew[NS], sdxo[NS], sKh[NS];
    __shared__ int s_last;
    const int tid = threadIdx.x;
    if (kf->done) return;
    const int pass = kf->passes;

    if (tid < 144) {
        const int a = tid / 12, b = tid % 12;
        const int lo = a < b ? a : b, hi = a < b ? b : a;
        const int idx = lo * 12 - lo * (lo - 1) / 2 + (hi - lo);
        sHTH[a][b] = sums[idx];
    }
    if (tid < 12) sHTh[tid] = sums[78 + tid];
    if (tid < SUMS_LEN && pass < MAX_PASSES) kf->sums_log[pass * SUMS_LEN + tid] = sums[tid];
    const double n_valid = sums[90];
    if (n_valid == 0.0) {  // h_share_model: dyn_share.valid = false -> `continue`
        if (tid == 0) {
            if (pass < MAX_PASSES) {
                for (int i = 0; i < NS; ++i) kf->trace[pass * 49 + i] = 0.0;
                for (int i = 0; i < NX; ++i) kf->trace[pass * 49 + NS + i] = kf->x[i];
            }
            kf->passes = pass + 1;
            kf->iter += 1;
            if (kf->iter >= prm.maximum_iter) kf->done = 1;
        }
        return;
    }

    set_identity(sJ, tid);
    if (tid < NS * NS) sB[tid / NS][tid % NS] = kf->P_prop[tid];
    __syncthreads();
    if (tid == 0) {
        double dx[NS];
        d_state_boxminus(kf->x, kf->x_prop, dx);
        fill_projection(sJ, dx, kf->x, kf->x_prop);
        for (int i = 0; i < NS; ++i) { sdx[i] = dx[i]; sdxnew[i] = dx[i]; }
    }
    __syncthreads();
    if (tid == 0) {  // dx_new blocks projected
        for (int b = 0; b < 3; ++b) {
            const int idx = b == 0 ? 3 : (b == 1 ? 6 : 21), r = b == 2 ? 2 : 3;
            double t[3];
            for (int i = 0; i < r; ++i) {
                double s = 0;
                if (r == 3) s = dot3d(sJ[idx + i][idx], sdx[idx], sJ[idx + i][idx + 1], sdx[idx + 1], sJ[idx + i][idx + 2], sdx[idx + 2]);
                else s = sJ[idx + i][idx] * sdx[idx] + sJ[idx + i][idx + 1] * sdx[idx + 1];
                t[i] = s;
            }
            for (int i = 0; i < r; ++i) sdxnew[idx + i] = t[i];
        }
    }
    // P_ = J P_prop J^T
    mm(sA, sJ, sB, false, tid);
    __syncthreads();
    mm(sP, sA, sJ, true, tid);
    __syncthreads();
    // P_temp = (P_/R)^-1 ; += HTH ; P_inv = P_temp^-1
    if (tid < NS * NS) sA[tid / NS][tid % NS] = sP[tid / NS][tid % NS] / prm.R;
    __syncthreads();
    gj_inverse(sA, sB, tid);
    if (tid < 144) sA[tid / 12][tid % 12] += sHTH[tid / 12][tid % 12];
    __syncthreads();
    gj_inverse(sA, sB, tid);  // sA = P_inv
    if (tid < NS) {
        double s = 0;
        for (int j = 0; j < 12; ++j) s += sA[tid][j] * sHTh[j];
        sKh[tid] = s;
    }
    if (tid >= 64 && tid < 64 + NS * 12) {
        const int e = tid - 64, i = e / 12, c = e % 12;
        double t = 0;
        for (int j = 0; j < 12; ++j) t += sA[i][j] * sHTH[j][c];
        sKx[i][c] = t;
    }
    __syncthreads();
    if (tid < NS) {
        double s = 0;
        for (int j = 0; j < NS; ++j) {
            const double kx = j < 12 ? sKx[tid][j] : 0.0;
            s += (kx - (tid == j ? 1.0 : 0.0)) * sdxnew[j];
        }
        sdxo[tid] = sKh[tid] + s;
    }
    __syncthreads();
    if (tid == 0) {
        double d[NS];
        for (int i = 0; i < NS; ++i) d[i] = sdxo[i];
        d_state_boxplus(kf->x, d);
        int converge = 1;
        for (int i = 0; i < NS; ++i)
            if (fabs(d[i]) > prm.limits[i]) { converge = 0; break; }
        int t = kf->t;
        if (converge) t++;
        kf->t = t;
        const int last = (t > 1 || kf->iter == prm.maximum_iter - 1) ? 1 : 0;
        s_last = last;
        if (pass < MAX_PASSES) {
            for (int i = 0; i < NS; ++i) kf->trace[pass * 49 + i] = d[i];
            for (int i = 0; i < NX; ++i) kf->trace[pass * 49 + NS + i] = kf->x[i];
        }
        kf->passes = pass + 1;
        kf->iter += 1;
        if (last) kf->done = 1;
        else compute_pose_consts(kf->x, &kf->pose);
    }
    __syncthreads();
    if (!s_last) return;

    // terminal pass: L_ = J2 P_ J2^T, K_x rows projected, P_ <- P_ J2^T, P = L_ - K_x[:, :12] P_[0:12, :]
    set_identity(sJ, tid);
    __syncthreads();
    if (tid == 0) fill_projection(sJ, sdxo, kf->x, kf->x_prop);
    __syncthreads();
    mm(sA, sJ, sP, false, tid);          // sA = J2 P_
    __syncthreads();
    mm(sB, sA, sJ, true, tid);           // sB = L_ = J2 P_ J2^T
    __syncthreads();
    mm(sA, sP, sJ, true, tid);           // sA = P_ J2^T
    __syncthreads();
    if (tid < NS * 12) {                 // K_x <- J2 K_x (rows)
        const int i = tid / 12, c = tid % 12;
        double s = 0;
        for (int r = 0; r < NS; ++r) s += sJ[i][r] * sKx[r][c];
        sP[i][c] = s;                    // stage projected K_x in sP (P_ no longer needed)
    }
    __syncthreads();
    if (tid < NS * NS) {
        const int i = tid / NS, j = tid % NS;
        double s = 0;
        for (int c = 0; c < 12; ++c) s += sP[i][c] * sA[c][j];
        kf->P_post[tid] = sB[i][j] - s;
    }
}

int launch_kf_begin(hipStream_t stream, KfDev* kf) {
    hipLaunchKernelGGL(kf_begin_kernel, dim3(1), dim3(64), 0, stream, kf);
    LV_HIP(hipGetLastError());
    return LV_OK;
}
int launch_reduce_partials(hipStream_t stream, const double* partials, int nblocks, double* sums, KfDev* kf) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(SUMS_LEN * RED_SEGS), 0, stream, partials, nblocks, sums, kf);
    LV_HIP(hipGetLastError());
    return LV_OK;
}
int launch_solve(hipStream_t stream, KfDev* kf, const double* sums, const SolveParams& prm) {
    hipLaunchKernelGGL(solve_kernel, dim3(1), dim3(576), 0, stream, kf, sums, prm);
    LV_HIP(hipGetLastError());
    return LV_OK;
}

}  // namespace lv

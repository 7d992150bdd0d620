// lv_solve_dev.hpp — device building blocks of the per-pass filter algebra (esekf::update_iterated_dyn_share_modified,
// [IKFoM absent from the reference mount; UPSTREAM-RECALL]; call site reference src/Modules/Localizator.cpp:132),
// shared by solve_kernel (lv_solve.hip) and by the record-independent half that rides along in
// fit_reduce_kernel (lv_match.hip): solve_prep.
#pragma once

#include "lv_device.hpp"
#include "lv_manifold.hpp"

namespace lv {

constexpr int LD = NS + 1;  // padded leading dimension in LDS

template <int NT>
__device__ inline void mm(double (*out)[LD], const double (*a)[LD], const double (*b)[LD], bool b_transposed, int tid) {
    for (int e = tid; e < NS * NS; e += NT) {
        const int i = e / NS, j = e % NS;
        double s = 0.0;
#pragma unroll 1
        for (int c = 0; c < NS; ++c) s += a[i][c] * (b_transposed ? b[j][c] : b[c][j]);
        out[i][j] = s;
    }
}
// Gauss-Jordan inverse of an SPD 6 x 6 matrix held one element per lane (lane = 6 i + j, lanes 0..35 of one wavefront;
// every lane of the wavefront must call): the pivot of step k is a lane broadcast, the pivot row / column elements
// come through two cross-lane permutes, so the six elimination steps need no LDS round trip in between.  Same
// operations in the same order as the ping-pong form below (one reciprocal per step).
__device__ __forceinline__ double lane_bcast_f64(double v, int src_lane) {   // src_lane uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_gather_f64(double v, int src_lane) {  // src_lane per lane
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double gj6_in_lanes(double w, int lane) {
    const int l = lane < 36 ? lane : 35;     // (idle lanes mirror a valid element: every permute source stays inside the matrix)
    const int i = l / 6, j = l % 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double rp = ddiv(1.0, lane_bcast_f64(w, k * 6 + k));
        const double wkj = lane_gather_f64(w, k * 6 + j);   // pivot row, my column
        const double wik = lane_gather_f64(w, i * 6 + k);   // my row, pivot column
        double v;
        if (i == k) {
            v = (j == k) ? rp : wkj * rp;
        } else {
            v = (j == k) ? -(wik * rp) : w - wik * (wkj * rp);
        }
        w = v;
    }
    return w;
}

// in-place (ping-pong) Gauss-Jordan inverse of an SPD NW x NW matrix (NW = 6 or 12); all threads call,
// tid < NW*NW work; one reciprocal per step
template <int NW>
__device__ inline void gj_spd(double (*W)[12][13], int& cur, int tid) {
    if (NW == 6) {
        // the whole matrix lives in the registers of wavefront 0 (the other wavefronts just wait at the one barrier below)
        if (tid < 64) {
            const int l = tid < 36 ? tid : 35;
            const double w = gj6_in_lanes(W[cur][l / 6][l % 6], tid);
            if (tid < 36) W[cur][tid / 6][tid % 6] = w;
        }
        __syncthreads();
        return;
    }
    for (int k = 0; k < NW; ++k) {
        if (tid < NW * NW) {
            const int i = tid / NW, j = tid % NW;
            const double rp = ddiv(1.0, W[cur][k][k]);
            double v;
            if (i == k) {
                v = (j == k) ? rp : W[cur][k][j] * rp;
            } else {
                const double f = W[cur][i][k];
                v = (j == k) ? -(f * rp) : W[cur][i][j] - f * (W[cur][k][j] * rp);
            }
            W[cur ^ 1][i][j] = v;
        }
        __syncthreads();
        cur ^= 1;
    }
}

// out = J in J^T for J = identity except the 3x3 blocks at 3 and 6 and the 2x2 block at 21: every output
// element touches at most 3 x 3 inputs, so the congruence is one pass (no intermediate product)
__device__ __forceinline__ void blk_range(int i, int& lo, int& n) {
    if (i >= 3 && i < 6) { lo = 3; n = 3; }
    else if (i >= 6 && i < 9) { lo = 6; n = 3; }
    else if (i >= 21) { lo = 21; n = 2; }
    else { lo = i; n = 1; }
}
template <int NT>
__device__ inline void congruence(double (*out)[LD], const double (*J)[LD], const double (*in)[LD], int tid) {
    for (int e = tid; e < NS * NS; e += NT) {
        const int i = e / NS, j = e % NS;
        int ia, na, jb, nb;
        blk_range(i, ia, na);
        blk_range(j, jb, nb);
        double s = 0.0;
        for (int a = 0; a < na; ++a) {
            double t = 0.0;
            for (int b = 0; b < nb; ++b) t += in[ia + a][jb + b] * J[j][jb + b];
            s += J[i][ia + a] * t;
        }
        out[i][j] = s;
    }
}

template <int NT>
__device__ inline void set_identity(double (*J)[LD], int tid) {
    for (int e = tid; e < NS * NS; e += NT) J[e / NS][e % NS] = (e / NS == e % NS) ? 1.0 : 0.0;
}

// compute_pose_consts (lv_device.hpp) with the four quaternion -> matrix conversions already done
__device__ inline void finish_pose_consts(const double* x, const double (*rot)[9], PoseConsts* out) {
#pragma clang fp contract(off)
    RT32 X, LI;
#pragma unroll
    for (int i = 0; i < 9; ++i) { X.R[i] = (float)rot[0][i]; LI.R[i] = (float)rot[1][i]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) { X.t[i] = (float)x[i]; LI.t[i] = (float)x[11 + i]; }
    out->Tc = rt_compose(X, LI);
    out->back = rt_compose(rt_inv(LI), rt_inv(X));
    out->LI = LI;
#pragma unroll
    for (int i = 0; i < 9; ++i) { out->R_inv[i] = rot[2][i]; out->I_R_L_inv[i] = rot[3][i]; }
}

// finish_pose_consts spread over lanes (same dot3f calls, same operand order => same bits): stage A by lanes
// 0..26 of the calling range (lane = tid - base), stage B by lanes 0..2 after a barrier.  tmp: 6 floats (the
// translations of inv(X), inv(LI)).
__device__ inline void pose_consts_stage_a(int lane, const double* x, const double (*rot)[9], PoseConsts* out, float* tmp) {
#pragma clang fp contract(off)
    if (lane < 12) {            // Tc = X * LI   (rt_compose)
        const int i = lane < 9 ? lane / 3 : lane - 9;
        const float a0 = (float)rot[0][i * 3 + 0], a1 = (float)rot[0][i * 3 + 1], a2 = (float)rot[0][i * 3 + 2];
        if (lane < 9) {
            const int j = lane % 3;
            out->Tc.R[lane] = dot3f(a0, (float)rot[1][0 * 3 + j], a1, (float)rot[1][1 * 3 + j], a2, (float)rot[1][2 * 3 + j]);
        } else {
            out->Tc.t[i] = dot3f(a0, (float)x[11], a1, (float)x[12], a2, (float)x[13]) + (float)x[i];
        }
    } else if (lane < 21) {     // back.R = LI^T * X^T
        const int e = lane - 12, i = e / 3, j = e % 3;
        // inv(LI).R[i][k] = LI.R[k][i], inv(X).R[k][j] = X.R[j][k]
        out->back.R[e] = dot3f((float)rot[1][0 * 3 + i], (float)rot[0][j * 3 + 0], (float)rot[1][1 * 3 + i], (float)rot[0][j * 3 + 1],
                               (float)rot[1][2 * 3 + i], (float)rot[0][j * 3 + 2]);
    } else if (lane < 27) {     // translations of inv(X) (tmp[0..2]) and inv(LI) (tmp[3..5])   (rt_inv)
        const int which = (lane - 21) / 3, i = (lane - 21) % 3;
        const double (*r) = rot[which];
        const int tb = which ? 11 : 0;
        tmp[which * 3 + i] = dot3f(-(float)r[0 * 3 + i], (float)x[tb], -(float)r[1 * 3 + i], (float)x[tb + 1], -(float)r[2 * 3 + i], (float)x[tb + 2]);
    } else if (lane < 39) {     // LI itself
        const int e = lane - 27;
        if (e < 9) out->LI.R[e] = (float)rot[1][e];
        else out->LI.t[e - 9] = (float)x[11 + e - 9];
    } else if (lane < 57) {
        const int e = lane - 39;
        if (e < 9) out->R_inv[e] = rot[2][e];
        else out->I_R_L_inv[e - 9] = rot[3][e - 9];
    }
}
__device__ inline void pose_consts_stage_b(int lane, const double (*rot)[9], PoseConsts* out, const float* tmp) {
#pragma clang fp contract(off)
    if (lane < 3) {             // back.t = inv(LI).R * inv(X).t + inv(LI).t
        const int i = lane;
        out->back.t[i] = dot3f((float)rot[1][0 * 3 + i], tmp[0], (float)rot[1][1 * 3 + i], tmp[1], (float)rot[1][2 * 3 + i], tmp[2]) + tmp[3 + i];
    }
}

// The three manifold blocks are independent: wave 0 / 1 / 2 (lane 0 of each) compute them concurrently
// on different SIMDs.  part 0: rot (dof 3), 1: offset_R_L_I (dof 6), 2: grav (dof 21).
// mode 0: seg = x [-] x_prop for this block (written to dx), then the projection block from seg
// mode 1: projection block from the given tangent `seg_in`
__device__ __forceinline__ void manifold_block(int part, int mode, const double* x, const double* xp, const double* seg_in, double* dx,
                                      double (*J)[LD]) {
    if (part < 2) {
        const int idx = part == 0 ? 3 : 6, q = part == 0 ? 3 : 7;
        double seg[3];
        if (mode == 0) {
            double c[4] = {-xp[q], -xp[q + 1], -xp[q + 2], xp[q + 3]}, qq[4];
            d_quat_mul(c, x + q, qq);
            d_so3_log(qq, seg);
            dx[idx] = seg[0]; dx[idx + 1] = seg[1]; dx[idx + 2] = seg[2];
        } else {
            seg[0] = seg_in[idx]; seg[1] = seg_in[idx + 1]; seg[2] = seg_in[idx + 2];
        }
        double A[9];
        d_A_matrix(seg, A);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) J[idx + r][idx + c] = A[c * 3 + r];  // res_temp_SO3 = A^T
    } else {
        double seg[2];
        if (mode == 0) {
            d_s2_boxminus(x + 23, xp + 23, seg);
            dx[21] = seg[0]; dx[22] = seg[1];
        } else {
            seg[0] = seg_in[21]; seg[1] = seg_in[22];
        }
        double T[4];
        d_s2_proj(x + 23, xp + 23, seg, T);
        J[21][21] = T[0]; J[21][22] = T[1]; J[22][21] = T[2]; J[22][22] = T[3];
    }
}

__device__ inline void boxplus_block(int part, double* x, const double* d) {
    if (part == 0) { double e[4], o[4]; d_so3_exp(d + 3, 1.0, e); d_quat_mul(x + 3, e, o); for (int i = 0; i < 4; ++i) x[3 + i] = o[i]; }
    else if (part == 1) { double e[4], o[4]; d_so3_exp(d + 6, 1.0, e); d_quat_mul(x + 7, e, o); for (int i = 0; i < 4; ++i) x[7 + i] = o[i]; }
    else d_s2_boxplus(x + 23, d + 21);
}

// Degeneracy stage of the fork's update_iterated_dyn_share_modified [UNKNOWN-FORK; see include/limovelo_hip.h
// degeneracy_mode]: eigen-decomposition of the 6x6 pose block of H^T H by cyclic Jacobi (fixed 8 sweeps, one lane: a
// serial f64 chain that only runs when the stage is switched on), eigenvalues to eig[6]; mode 2 projects the
// measurement information onto the eigen-directions at or above the threshold:
//   H^T H <- blockdiag(Pn, I) H^T H blockdiag(Pn, I),  H^T h <- blockdiag(Pn, I) H^T h,  Pn = sum_{lambda >= thr} v v^T.
__device__ inline void degeneracy_stage(double (*HTH)[12], double* HTh, int mode, double threshold, double* eig) {
    double A[6][6], V[6][6];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) { A[i][j] = HTH[i][j]; V[i][j] = i == j ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 8; ++sweep)
        for (int p = 0; p < 5; ++p)
            for (int q = p + 1; q < 6; ++q) {
                const double apq = A[p][q];
                if (fabs(apq) < 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 6; ++k) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 6; ++k) {
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
    for (int i = 0; i < 6; ++i) eig[i] = A[i][i];
    if (mode != 2) return;
    double Pn[6][6];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double s = 0.0;
            for (int e = 0; e < 6; ++e)
                if (A[e][e] >= threshold) s += V[i][e] * V[j][e];
            Pn[i][j] = s;
        }
    double T[6][12];
    for (int i = 0; i < 6; ++i)          // rows 0..5 of blockdiag(Pn, I) H
        for (int j = 0; j < 12; ++j) {
            double s = 0.0;
            for (int e = 0; e < 6; ++e) s += Pn[i][e] * HTH[e][j];
            T[i][j] = s;
        }
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 12; ++j) HTH[i][j] = T[i][j];
    for (int i = 0; i < 12; ++i) {       // columns 0..5 of (.) blockdiag(Pn, I)
        double r[6];
        for (int j = 0; j < 6; ++j) {
            double s = 0.0;
            for (int e = 0; e < 6; ++e) s += HTH[i][e] * Pn[e][j];
            r[j] = s;
        }
        for (int j = 0; j < 6; ++j) HTH[i][j] = r[j];
    }
    double h6[6];
    for (int i = 0; i < 6; ++i) {
        double s = 0.0;
        for (int e = 0; e < 6; ++e) s += Pn[i][e] * HTh[e];
        h6[i] = s;
    }
    for (int i = 0; i < 6; ++i) HTh[i] = h6[i];
}

// dof index -> offset in the 26-double state for the vect components
__device__ __forceinline__ int vect_state_index(int dof) {  // dof in {0..2, 9..20}
    return dof < 3 ? dof : dof + 2;                         // 9..11 -> 11..13, 12..14 -> 14..16, ...
}

// The half of a pass that does not depend on the measurement record: dx = x [-] x_prop with its projection J,
// dx_new = J dx, P_ = J P_prop J^T and A1 = (P_/R)_ww^-1.  It only needs the state the previous pass left in kf,
// so one extra workgroup of fit_reduce_kernel computes it while the other workgroups fit planes; solve_kernel
// picks the results up from kf->prep_* and starts at the record.  NT threads (>= 256), all must call.
template <int NW, int NT>
__device__ inline void solve_prep(KfDev* __restrict__ kf, double R_inv) {
    __shared__ double pP[NS][LD], pB[NS][LD], pJ[NS][LD];
    __shared__ double pW[2][12][13];
    __shared__ double px[NX], pxp[NX], pdx[NS];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    for (int e = tid; e < NS * NS; e += NT) pB[e / NS][e % NS] = kf->P_prop[e];
    if (tid < NX) { px[tid] = kf->x[tid]; pxp[tid] = kf->x_prop[tid]; }
    set_identity<NT>(pJ, tid);
    __syncthreads();
    // three manifold blocks on three waves, vect parts on wave 3
    if (wave < 3 && lane == 0) manifold_block(wave, 0, px, pxp, nullptr, pdx, pJ);
    if (wave == 3 && lane < 15) {
        const int dof = lane < 3 ? lane : lane + 6;  // 0..2, 9..20
        const int si = vect_state_index(dof);
        pdx[dof] = px[si] - pxp[si];
    }
    __syncthreads();
    if (tid < NS) {  // dx_new = J dx (identity outside the blocks)
        double s = 0.0;
        const int b = (tid >= 3 && tid < 6) ? 3 : (tid >= 6 && tid < 9) ? 6 : (tid >= 21) ? 21 : -1;
        if (b < 0) s = pdx[tid];
        else if (b == 21) s = pJ[tid][21] * pdx[21] + pJ[tid][22] * pdx[22];
        else s = dot3d(pJ[tid][b], pdx[b], pJ[tid][b + 1], pdx[b + 1], pJ[tid][b + 2], pdx[b + 2]);
        kf->prep_dxnew[tid] = s;
    }
    congruence<NT>(pP, pJ, pB, tid);  // P_ = J P_prop J^T
    __syncthreads();
    for (int e = tid; e < NS * NS; e += NT) kf->prep_P[e] = pP[e / NS][e % NS];
    if (tid < NW * NW) pW[0][tid / NW][tid % NW] = pP[tid / NW][tid % NW] * R_inv;
    __syncthreads();
    int cur = 0;
    gj_spd<NW>(pW, cur, tid);
    if (tid < NW * NW) kf->prep_A1[tid] = pW[cur][tid / NW][tid % NW];
}

}  // namespace lv

// lv_manifold.hpp switched FMA contraction on for the filter algebra above; the includer's own code goes back to
// the build default (off: the f32 path is bit-exact against the FMA-free reference arithmetic).  lv_solve.hip
// re-enables it for its kernels.
#pragma clang fp contract(off)

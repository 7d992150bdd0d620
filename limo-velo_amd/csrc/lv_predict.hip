// lv_predict.hip — row f-3: the filter state lives on the device between calls, and the IMU prediction
// esekf::predict(dt, Q, in) runs there too, so that propagate -> correct -> propagate never round-trips x, P.
//
// Reference: Localizator::propagate (src/Modules/Localizator.cpp:159-173: Q from cov_gyro / cov_acc /
// cov_bias_gyro / cov_bias_acc, in = {imu.a, imu.w}, IKFoM_KF.predict(dt, Q, in)).  IKFoM is absent from
// the reference mount: predict and the process model get_f / df_dx / df_dw are restated from the published
// hku-mars/IKFoM + FAST-LIO2 use-ikfom [UPSTREAM-RECALL], including its quirks (MTK::exp(..., scalar(1/2))
// with an integer 1/2 == 0, i.e. identity rotations inside F_x1).  N-independent, latency-only work: one
// workgroup, scalar manifold algebra on lane 0, the 23x23 products spread over 576 lanes.
#include "lv_host.hpp"
#include "lv_manifold.hpp"

namespace lv {

// up to PREDICT_BATCH consecutive IMU steps with the same Q in one launch (step: dt, acc, gyro): the reference propagates sample
// by sample (Localizator::propagate_to, src/Modules/Localizator.cpp:85-104) — 2.6 launches per 100 Hz update when each is its own
constexpr int PREDICT_BATCH_MAX = 8;
struct PredictArgs {
    double Q[144];
    int n;
    double step[PREDICT_BATCH_MAX][7];
};

// The scalar manifold algebra is spread over lanes of DIFFERENT wavefronts where its pieces are independent (three state
// [+] blocks, the df_dx / df_dw entries, three A matrices), and the per-column products of the SO3 / S2 rows over one lane
// per column: as two serial sections on lane 0 (round 1) the kernel took 14.5 us, 2.6 times per 100 Hz update.  Every entry
// is computed by the expression the serial form used: same bits.
__global__ __launch_bounds__(576) void predict_kernel(FilterDev* f, const KfDev* src, PredictArgs a) {
    __shared__ double sfx[24][NS], sfw[24][12];   // df_dx (24 x 23), df_dw (24 x 12) in the flat 24-dim layout
    __shared__ double sF1[NS][NS + 1], sfxf[NS][NS], sfwf[NS][12], sP[NS][NS + 1], sFP[NS][NS + 1], sGQ[NS][12];
    __shared__ double sx[NX], sxb[NX], sflat[24];
    __shared__ double sR[9], sam[3], sA[2][9], sT[6];
    const int tid = threadIdx.x;
    for (int st = 0; st < a.n; ++st) {   // (a step reads what the step before stored: every thread its own entries of f)
        if (tid < NS * NS) sP[tid / NS][tid % NS] = src ? src->P_post[tid] : f->P[tid];
        if (tid < NX) { const double v = src ? src->x[tid] : f->x[tid]; sx[tid] = v; sxb[tid] = v; }
        for (int e = tid; e < 24 * NS; e += 576) sfx[e / NS][e % NS] = 0.0;
        for (int e = tid; e < 24 * 12; e += 576) sfw[e / 12][e % 12] = 0.0;
        if (tid < NS * NS) { sF1[tid / NS][tid % NS] = (tid / NS == tid % NS) ? 1.0 : 0.0; sfxf[tid / NS][tid % NS] = 0.0; }
        if (tid < NS * 12) sfwf[tid / 12][tid % 12] = 0.0;
        __syncthreads();
        const double dt = a.step[st][0];
        const double acc_in[3] = {a.step[st][1], a.step[st][2], a.step[st][3]}, gyro_in[3] = {a.step[st][4], a.step[st][5], a.step[st][6]};
        // state doubles: pos 0..2, rot 3..6, offR 7..10, offT 11..13, vel 14..16, bg 17..19, ba 20..22, grav 23..25
        // flat dims:     pos 0, rot 3, offR 6, offT 9, vel 12, bg 15, ba 18, grav 21(3)
        if (tid == 0) {   // get_f: what everything else needs
            double fl[24];
            for (int i = 0; i < 24; ++i) fl[i] = 0.0;
            const double omega[3] = {gyro_in[0] - sxb[17], gyro_in[1] - sxb[18], gyro_in[2] - sxb[19]};
            const double am[3] = {acc_in[0] - sxb[20], acc_in[1] - sxb[21], acc_in[2] - sxb[22]};
            double R[9], a_in[3];
            quat_to_rot(sxb + 3, R);
            d_mat3_vec(R, am, a_in);
            for (int i = 0; i < 3; ++i) { fl[i] = sxb[14 + i]; fl[3 + i] = omega[i]; fl[12 + i] = a_in[i] + sxb[23 + i]; }
            for (int i = 0; i < 24; ++i) sflat[i] = fl[i];
            for (int i = 0; i < 9; ++i) sR[i] = R[i];
            for (int i = 0; i < 3; ++i) sam[i] = am[i];
        }
        __syncthreads();
        if (tid == 0) {            // x_.oplus(f_, dt): the vector parts
            for (int i = 0; i < 3; ++i) sx[i] += sflat[i] * dt;
            for (int i = 0; i < 3; ++i) sx[11 + i] += sflat[9 + i] * dt;
            for (int i = 0; i < 3; ++i) sx[14 + i] += sflat[12 + i] * dt;
            for (int i = 0; i < 3; ++i) sx[17 + i] += sflat[15 + i] * dt;
            for (int i = 0; i < 3; ++i) sx[20 + i] += sflat[18 + i] * dt;
        } else if (tid == 64 || tid == 128) {   // ... the two SO3 blocks
            const int q = tid == 64 ? 3 : 7, fi = tid == 64 ? 3 : 6;
            double e[4], o[4];
            d_so3_exp(sflat + fi, dt, e);
            d_quat_mul(sxb + q, e, o);
            for (int i = 0; i < 4; ++i) sx[q + i] = o[i];
        } else if (tid == 192) {   // ... the S2 block, then what the S2 rows of f_x_final need of it
            {   // S2::oplus(delta3, scale): vec = exp(delta, scale/2).toRotationMatrix() * vec
                double q[4], Rg[9], o[3];
                d_so3_exp(sflat + 21, dt, q);
                quat_to_rot(q, Rg);
                d_mat3_vec(Rg, sxb + 23, o);
                sx[23] = o[0]; sx[24] = o[1]; sx[25] = o[2];
            }
            const int idx = 21;
            const double seg[3] = {sflat[21] * dt, sflat[22] * dt, sflat[23] * dt};
            const double zero2[2] = {0.0, 0.0};
            double Nx[6], Mx[6];
            d_s2_Nx_yy(sx + 23, Nx);
            d_s2_Mx(sxb + 23, zero2, Mx);
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j)
                    sF1[idx + i][idx + j] = Nx[i * 3] * Mx[0 * 2 + j] + Nx[i * 3 + 1] * Mx[1 * 2 + j] + Nx[i * 3 + 2] * Mx[2 * 2 + j];
            double Hb[9], A[9], At[9], HA[9];
            d_hat3(sxb + 23, Hb);
            d_A_matrix(seg, A);
            d_mat3_T(A, At);
            d_mat3_mul(Hb, At, HA);
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 3; ++j) sT[i * 3 + j] = -(Nx[i * 3] * HA[j] + Nx[i * 3 + 1] * HA[3 + j] + Nx[i * 3 + 2] * HA[6 + j]);
        } else if (tid == 256) {   // df_dx, df_dw
            for (int i = 0; i < 3; ++i) sfx[i][12 + i] = 1.0;
            double Ha[9], RH[9];
            d_hat3(sam, Ha);
            d_mat3_mul(sR, Ha, RH);
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) { sfx[12 + i][3 + j] = -RH[i * 3 + j]; sfx[12 + i][18 + j] = -sR[i * 3 + j]; }
            const double zero2[2] = {0.0, 0.0};
            double Mx0[6];
            d_s2_Mx(sxb + 23, zero2, Mx0);
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 2; ++j) sfx[12 + i][21 + j] = Mx0[i * 2 + j];
            for (int i = 0; i < 3; ++i) sfx[3 + i][15 + i] = -1.0;
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) sfw[12 + i][3 + j] = -sR[i * 3 + j];
            for (int i = 0; i < 3; ++i) { sfw[3 + i][i] = -1.0; sfw[15 + i][6 + i] = 1.0; sfw[18 + i][9 + i] = 1.0; }
        } else if (tid == 320 || tid == 384) {   // SO3 states: A_matrix(-f*dt) (F_x1 block = exp(seg, 0) = identity: quirk)
            const int b = tid == 320 ? 0 : 1, idx = b == 0 ? 3 : 6;
            const double seg[3] = {-sflat[idx] * dt, -sflat[idx + 1] * dt, -sflat[idx + 2] * dt};
            double A[9];
            d_A_matrix(seg, A);
            for (int i = 0; i < 9; ++i) sA[b][i] = A[i];
        }
        __syncthreads();
        // vect states: rows copied through (dof index == flat index for pos, offT, vel, bg, ba)
        if (tid < 15 * NS) {
            const int b = tid / (3 * NS), r = (tid / NS) % 3, c = tid % NS;
            const int vidx[5] = {0, 9, 12, 15, 18};
            sfxf[vidx[b] + r][c] = sfx[vidx[b] + r][c];
        }
        if (tid >= 352 && tid < 352 + 15 * 12) {
            const int e = tid - 352, b = e / 36, r = (e / 12) % 3, c = e % 12;
            const int vidx[5] = {0, 9, 12, 15, 18};
            sfwf[vidx[b] + r][c] = sfw[vidx[b] + r][c];
        }
        // SO3 rows: f_x_final rows = A * f_x rows, one lane per (block, column); S2 rows likewise
        if (tid >= 64 && tid < 64 + 2 * (NS + 12)) {
            const int e = tid - 64, b = e / (NS + 12), c = e % (NS + 12), idx = b == 0 ? 3 : 6;
            if (c < NS) {
                const double v[3] = {sfx[idx][c], sfx[idx + 1][c], sfx[idx + 2][c]};
                double o[3];
                d_mat3_vec(sA[b], v, o);
                for (int i = 0; i < 3; ++i) sfxf[idx + i][c] = o[i];
            } else {
                const int cw = c - NS;
                const double v[3] = {sfw[idx][cw], sfw[idx + 1][cw], sfw[idx + 2][cw]};
                double o[3];
                d_mat3_vec(sA[b], v, o);
                for (int i = 0; i < 3; ++i) sfwf[idx + i][cw] = o[i];
            }
        }
        if (tid >= 192 && tid < 192 + NS + 12) {
            const int c = tid - 192, idx = 21;
            if (c < NS) {
                for (int i = 0; i < 2; ++i) sfxf[idx + i][c] = sT[i * 3] * sfx[21][c] + sT[i * 3 + 1] * sfx[22][c] + sT[i * 3 + 2] * sfx[23][c];
            } else {
                const int cw = c - NS;
                for (int i = 0; i < 2; ++i) sfwf[idx + i][cw] = sT[i * 3] * sfw[21][cw] + sT[i * 3 + 1] * sfw[22][cw] + sT[i * 3 + 2] * sfw[23][cw];
            }
        }
        __syncthreads();
        if (tid < NS * NS) sF1[tid / NS][tid % NS] += sfxf[tid / NS][tid % NS] * dt;   // F_x1 += f_x_final * dt
        __syncthreads();
        // P = F1 P F1^T + (dt fwf) Q (dt fwf)^T
        if (tid < NS * NS) {
            const int i = tid / NS, j = tid % NS;
            double s = 0.0;
            for (int c = 0; c < NS; ++c) s += sF1[i][c] * sP[c][j];
            sFP[i][j] = s;
        }
        if (tid < NS * 12) {
            const int i = tid / 12, j = tid % 12;
            double s = 0.0;
            for (int c = 0; c < 12; ++c) s += (dt * sfwf[i][c]) * a.Q[c * 12 + j];
            sGQ[i][j] = s;
        }
        __syncthreads();
        if (tid < NS * NS) {
            const int i = tid / NS, j = tid % NS;
            double s = 0.0, q = 0.0;
            for (int c = 0; c < NS; ++c) s += sFP[i][c] * sF1[j][c];
            for (int c = 0; c < 12; ++c) q += sGQ[i][c] * (dt * sfwf[j][c]);
            f->P[tid] = s + q;
        }
        if (tid < NX) f->x[tid] = sx[tid];
    __syncthreads();   // the LDS arrays are free again
    src = nullptr;     // (from the second step on the state is the one just stored in f)
    }
}

// resident filter state <-> the KfDev working copy of one iterated update
__global__ __launch_bounds__(576) void filter_to_kf_kernel(const FilterDev* f, KfDev* kf) {
    const int tid = threadIdx.x;
    if (tid < NS * NS) kf->P_prop[tid] = f->P[tid];
    if (tid < NX) kf->x[tid] = f->x[tid];
}
__global__ __launch_bounds__(576) void kf_to_filter_kernel(const KfDev* kf, FilterDev* f) {
    const int tid = threadIdx.x;
    if (tid < NS * NS) f->P[tid] = kf->P_post[tid];
    if (tid < NX) f->x[tid] = kf->x[tid];
}

int launch_predict(hipStream_t stream, FilterDev* f, const KfDev* src, const double* Q, int n, const double (*steps)[7]) {
    if (n <= 0) return LV_OK;
    PredictArgs a;
    for (int i = 0; i < 144; ++i) a.Q[i] = Q[i];
    a.n = n < PREDICT_BATCH_MAX ? n : PREDICT_BATCH_MAX;
    for (int i = 0; i < a.n; ++i)
        for (int j = 0; j < 7; ++j) a.step[i][j] = steps[i][j];
    hipLaunchKernelGGL(predict_kernel, dim3(1), dim3(576), 0, stream, f, src, a);
    LV_HIP(hipGetLastError());
    return LV_OK;
}
int launch_filter_to_kf(hipStream_t stream, const FilterDev* f, KfDev* kf) {
    hipLaunchKernelGGL(filter_to_kf_kernel, dim3(1), dim3(576), 0, stream, f, kf);
    LV_HIP(hipGetLastError());
    return LV_OK;
}
int launch_kf_to_filter(hipStream_t stream, const KfDev* kf, FilterDev* f) {
    hipLaunchKernelGGL(kf_to_filter_kernel, dim3(1), dim3(576), 0, stream, kf, f);
    LV_HIP(hipGetLastError());
    return LV_OK;
}

}  // namespace lv

// lv_rows.hip — Localizator::calculate_H (reference src/Modules/Localizator.cpp:29-57) as a standalone
// entry point for API parity: Jacobian rows for caller-supplied matches (world point, plane normal,
// distance), one lane per match.  The fused path (lv_match.hip) never materialises H.
#include "lv_host.hpp"

namespace lv {

__global__ void rows_from_matches_kernel(const KfDev* __restrict__ kf, const float* __restrict__ p_world,
                                         const float* __restrict__ abcd, const float* __restrict__ dist, uint32_t n,
                                         int estimate_extrinsics, double* __restrict__ H, double* __restrict__ h) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const PoseConsts& pc = kf->pose;
    const float qx = p_world[3 * i], qy = p_world[3 * i + 1], qz = p_world[3 * i + 2];
    float plx, ply, plz, pix, piy, piz;
    rt_apply(pc.back, qx, qy, qz, plx, ply, plz);                         // :38
    rt_apply(pc.LI, plx, ply, plz, pix, piy, piz);                        // :39
    const double n0 = (double)abcd[4 * i], n1 = (double)abcd[4 * i + 1], n2 = (double)abcd[4 * i + 2];
    const double* Ri = pc.R_inv;
    const double* Li = pc.I_R_L_inv;
    const double C0 = dot3d(Ri[0], n0, Ri[1], n1, Ri[2], n2);             // :47
    const double C1 = dot3d(Ri[3], n0, Ri[4], n1, Ri[5], n2);
    const double C2 = dot3d(Ri[6], n0, Ri[7], n1, Ri[8], n2);
    const double t0 = dot3d(Li[0], C0, Li[1], C1, Li[2], C2);
    const double t1 = dot3d(Li[3], C0, Li[4], C1, Li[5], C2);
    const double t2 = dot3d(Li[6], C0, Li[7], C1, Li[8], C2);
    const double lx = (double)plx, ly = (double)ply, lz = (double)plz;
    const double ix = (double)pix, iy = (double)piy, iz = (double)piz;
    double* r = H + (size_t)i * 12;
    r[0] = n0; r[1] = n1; r[2] = n2;                                      // :51
    r[3] = iy * C2 - iz * C1;                                             // :49
    r[4] = iz * C0 - ix * C2;
    r[5] = ix * C1 - iy * C0;
    if (estimate_extrinsics) {                                            // :52
        r[6] = ly * t2 - lz * t1;                                         // :48
        r[7] = lz * t0 - lx * t2;
        r[8] = lx * t1 - ly * t0;
        r[9] = C0; r[10] = C1; r[11] = C2;
    } else {
        r[6] = r[7] = r[8] = r[9] = r[10] = r[11] = 0.0;                  // :31 Zero()
    }
    h[i] = -(double)dist[i];                                              // :55
}

int launch_rows_from_matches(hipStream_t stream, const KfDev* kf, const float* p_world, const float* abcd, const float* dist,
                             uint32_t n, int estimate_extrinsics, double* H, double* h) {
    if (n == 0) return LV_OK;
    hipLaunchKernelGGL(rows_from_matches_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, kf, p_world, abcd, dist, n,
                       estimate_extrinsics, H, h);
    LV_HIP(hipGetLastError());
    return LV_OK;
}

}  // namespace lv

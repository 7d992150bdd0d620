// lv_match.hip — the measurement-model pass: world transform -> exact 5-NN in the voxel hash ->
// plane fit + gates -> point-to-plane residual + Jacobian row -> block-level H^T H / H^T h partials.
//
// One launch replaces, for all N scan points (reference call stack SURVEY.md §3.1):
//   Mapper::match            src/Modules/Mapper.cpp:40-56   (transform :51, match_plane :82-90)
//   KD_TREE::Nearest_Search  call site Mapper.cpp:86        [ikd-Tree, absent; exact kNN restated]
//   Plane::Plane / fit_plane src/Objects/Plane.cpp:19-55
//   R3Math::estimate_plane / is_plane   src/Utils/Utils.cpp:32-66
//   Match::Match             src/Objects/Match.cpp:18-22
//   Localizator::calculate_H src/Modules/Localizator.cpp:29-57
//   esekf h_x^T h_x, h_x^T h products   [IKFoM, absent]     -> never materialises the N x 12 H
//
// Work decomposition (wave64): S lanes of a wavefront cooperate on one scan point ("point tile"):
// they split the 27 neighbour voxels of the search, keep private sorted top-5 lists of packed
// (distance bits << 32 | map index) keys — one u64 compare is the reference's (d, index)
// lexicographic order — and merge them with xor-shuffles.  The plane fit / Jacobian is then
// evaluated redundantly by the S lanes (no divergence), group lane 0 stages the 12-wide row in
// LDS and the block contracts the staged rows into its 92 partial sums in f64, fixed order.
//
// Exactness of the voxel search: the query's level-l voxel and its 26 neighbours cover every point
// within (2^l * cell) of the query up to rounding of the voxel coordinates; a level is accepted only
// if 5 candidates were found and d5 < r_l^2 with r_l shrunk by a 1e-3 relative margin plus the f32
// rounding bound of the coordinates involved.  Otherwise the next (coarser) level is searched from
// scratch, and after the last level a brute-force scan of all M points decides.  Hence the result
// equals exhaustive search under the (d, index) order for every query.
#include "lv_host.hpp"

namespace lv {

constexpr uint64_t NONE_KEY = ~0ull;
constexpr int ROW_W = 14;  // r[0..11], h, valid

__device__ __forceinline__ void top5_insert(uint64_t (&k)[KNN], uint64_t x) {
#pragma unroll
    for (int j = 0; j < KNN; ++j) {
        uint64_t lo = k[j] < x ? k[j] : x;
        uint64_t hi = k[j] < x ? x : k[j];
        k[j] = lo;
        x = hi;
    }
}

// [UPSTREAM-RECALL ikd-Tree calc_dist]: (ax-bx)^2 + (ay-by)^2 + (az-bz)^2, f32, left to right, unfused
__device__ __forceinline__ float calc_dist(float qx, float qy, float qz, float4 m) {
    float dx = qx - m.x, dy = qy - m.y, dz = qz - m.z;
    float sx = dx * dx, sy = dy * dy, sz = dz * dz;
    float s = sx + sy;
    return s + sz;
}

__device__ __forceinline__ void scan_range(const float4* __restrict__ sorted, uint32_t start, uint32_t count, float qx,
                                           float qy, float qz, uint64_t (&k)[KNN]) {
    for (uint32_t j = 0; j < count; ++j) {
        float4 m = sorted[start + j];
        float d = calc_dist(qx, qy, qz, m);
        uint64_t key = ((uint64_t)__float_as_uint(d) << 32) | (uint64_t)__float_as_uint(m.w);
        if (key < k[KNN - 1]) top5_insert(k, key);
    }
}

template <int S>
__device__ __forceinline__ void merge_group(uint64_t (&k)[KNN]) {
#pragma unroll
    for (int off = S / 2; off >= 1; off >>= 1) {
        uint64_t o[KNN];
#pragma unroll
        for (int j = 0; j < KNN; ++j) o[j] = __shfl_xor(k[j], off);
#pragma unroll
        for (int j = 0; j < KNN; ++j)
            if (o[j] < k[KNN - 1]) top5_insert(k, o[j]);
    }
}

// Column-pivoted Householder QR least squares for the 5 x 3 system A n = -1, f32.  Same operation
// sequence as the oracle's restatement of R3Math::estimate_plane's
// `A.colPivHouseholderQr().solve(b)` (reference src/Utils/Utils.cpp:47).
__device__ inline void plane_qr_solve(float (&A)[KNN][3], float (&x)[3]) {
    constexpr int rows = KNN, cols = 3, size = 3;
    const float eps = 1.1920928955078125e-07f;
    float hC[3];
    int perm[3] = {0, 1, 2};
    float nU[3], nD[3];
#pragma unroll
    for (int k = 0; k < cols; ++k) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < rows; ++i) s += A[i][k] * A[i][k];
        nD[k] = nU[k] = sqrtf(s);
    }
    float maxn = fmaxf(nU[0], fmaxf(nU[1], nU[2]));
    float th = maxn * eps / (float)rows;
    const float threshold_helper = th * th;
    const float norm_downdate_threshold = sqrtf(eps);
    int nonzero_pivots = size;
#pragma unroll
    for (int k = 0; k < size; ++k) {
        int bi = k;
        float bn = nU[k];
#pragma unroll
        for (int j = k + 1; j < cols; ++j)
            if (nU[j] > bn) { bn = nU[j]; bi = j; }
        float biggest_sq = bn * bn;
        if (nonzero_pivots == size && biggest_sq < threshold_helper * (float)(rows - k)) nonzero_pivots = k;
#pragma unroll
        for (int j = k + 1; j < cols; ++j) {
            const bool sw = (bi == j);
#pragma unroll
            for (int i = 0; i < rows; ++i) {
                float a = A[i][k], b = A[i][j];
                A[i][k] = sw ? b : a;
                A[i][j] = sw ? a : b;
            }
            float a = nU[k], b = nU[j];
            nU[k] = sw ? b : a; nU[j] = sw ? a : b;
            a = nD[k]; b = nD[j];
            nD[k] = sw ? b : a; nD[j] = sw ? a : b;
            int pa = perm[k], pb = perm[j];
            perm[k] = sw ? pb : pa; perm[j] = sw ? pa : pb;
        }
        float tailSq = 0.f;
#pragma unroll
        for (int i = k + 1; i < rows; ++i) tailSq += A[i][k] * A[i][k];
        float c0 = A[k][k];
        float tau, beta;
        if (tailSq <= 1.17549435e-38f) {
            tau = 0.f;
            beta = c0;
#pragma unroll
            for (int i = k + 1; i < rows; ++i) A[i][k] = 0.f;
        } else {
            beta = sqrtf(c0 * c0 + tailSq);
            if (c0 >= 0.f) beta = -beta;
            float den = c0 - beta;
#pragma unroll
            for (int i = k + 1; i < rows; ++i) A[i][k] = A[i][k] / den;
            tau = (beta - c0) / beta;
        }
        A[k][k] = beta;
        hC[k] = tau;
        if (tau != 0.f) {
#pragma unroll
            for (int j = k + 1; j < cols; ++j) {
                float tmp = 0.f;
#pragma unroll
                for (int i = k + 1; i < rows; ++i) tmp += A[i][k] * A[i][j];
                tmp += A[k][j];
                A[k][j] -= tau * tmp;
#pragma unroll
                for (int i = k + 1; i < rows; ++i) A[i][j] -= tau * A[i][k] * tmp;
            }
        }
#pragma unroll
        for (int j = k + 1; j < cols; ++j) {
            if (nU[j] != 0.f) {
                float temp = fabsf(A[k][j]) / nU[j];
                temp = (1.f + temp) * (1.f - temp);
                temp = temp < 0.f ? 0.f : temp;
                float r = nU[j] / nD[j];
                float temp2 = temp * (r * r);
                if (temp2 <= norm_downdate_threshold) {
                    float s = 0.f;
#pragma unroll
                    for (int i = k + 1; i < rows; ++i) s += A[i][j] * A[i][j];
                    nD[j] = sqrtf(s);
                    nU[j] = nD[j];
                } else {
                    nU[j] *= sqrtf(temp);
                }
            }
        }
    }
    x[0] = x[1] = x[2] = 0.f;
    if (nonzero_pivots == 0) return;
    float c[KNN];
#pragma unroll
    for (int i = 0; i < rows; ++i) c[i] = -1.0f;
#pragma unroll
    for (int k = 0; k < size; ++k) {
        if (k < nonzero_pivots) {
            float tau = hC[k];
            if (tau != 0.f) {
                float tmp = 0.f;
#pragma unroll
                for (int i = k + 1; i < rows; ++i) tmp += A[i][k] * c[i];
                tmp += c[k];
                c[k] -= tau * tmp;
#pragma unroll
                for (int i = k + 1; i < rows; ++i) c[i] -= tau * A[i][k] * tmp;
            }
        }
    }
#pragma unroll
    for (int i = size - 1; i >= 0; --i) {
        if (i < nonzero_pivots) {
            float s = c[i];
#pragma unroll
            for (int j = i + 1; j < size; ++j)
                if (j < nonzero_pivots) s -= A[i][j] * c[j];
            c[i] = s / A[i][i];
        }
    }
#pragma unroll
    for (int i = 0; i < size; ++i) {
        if (i < nonzero_pivots) {
#pragma unroll
            for (int a = 0; a < 3; ++a)
                if (perm[i] == a) x[a] = c[i];
        }
    }
}

__device__ __forceinline__ void out_pair(int t, int& a, int& b) {
    if (t < 78) {
        int i = 0, rem = t;
        while (rem >= 12 - i) { rem -= 12 - i; ++i; }
        a = i;
        b = i + rem;
    } else if (t < 90) {
        a = t - 78;
        b = 12;
    } else if (t == 90) {
        a = 13;
        b = 13;
    } else {
        a = 12;
        b = 12;
    }
}

template <int S, bool DBG>
__global__ __launch_bounds__(256) void match_reduce_kernel(MapView map, const float4* __restrict__ scan, uint32_t n,
                                                           const KfDev* __restrict__ kf, MatchParams prm,
                                                           double* __restrict__ partials, DebugOut dbg,
                                                           int* __restrict__ fallback_counter) {
    constexpr int G = 256 / S;  // scan points per block iteration
    __shared__ double s_rows[G][ROW_W];
    if (kf->done) return;
    const int tid = threadIdx.x;
    const int gq = tid / S, gl = tid % S;
    const PoseConsts& pc = kf->pose;

    int oa = 0, ob = 0;
    if (tid < N_OUT) out_pair(tid, oa, ob);
    double acc = 0.0;

    const uint32_t per_iter = (uint32_t)G * gridDim.x;
    const uint32_t iters = (n + per_iter - 1) / per_iter;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t q = (it * gridDim.x + blockIdx.x) * (uint32_t)G + (uint32_t)gq;
        double row[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) row[i] = 0.0;
        double hres = 0.0;
        bool chosen = false;
        if (q < n) {
            const float4 sp = scan[q];
            const uint32_t oq = __float_as_uint(sp.w);
            float qx, qy, qz;
            rt_apply(pc.Tc, sp.x, sp.y, sp.z, qx, qy, qz);  // Mapper.cpp:51

            // ---- exact 5-NN ------------------------------------------------------------------
            uint64_t k[KNN];
#pragma unroll
            for (int j = 0; j < KNN; ++j) k[j] = NONE_KEY;
            int found = 0;
            if (map.m > 0) {
                const int c0x = cell_coord(qx, map.origin[0], map.inv_cell);
                const int c0y = cell_coord(qy, map.origin[1], map.inv_cell);
                const int c0z = cell_coord(qz, map.origin[2], map.inv_cell);
                const int ax = abs(c0x - CELL_OFFSET), ay = abs(c0y - CELL_OFFSET), az = abs(c0z - CELL_OFFSET);
                const int amax = max(ax, max(ay, az));
                const bool finite = (qx == qx) && (qy == qy) && (qz == qz);
                bool decided = false;
                int level = (finite && amax < CELL_FAR) ? 0 : map.n_levels;
                while (!decided) {
                    if (level < map.n_levels) {
                        const GridLevel gl_ = map.lv[level];
                        const int clx = c0x >> level, cly = c0y >> level, clz = c0z >> level;
                        for (int c = gl; c < 27; c += S) {
                            const int dz = c / 9 - 1, dy = (c / 3) % 3 - 1, dx = c % 3 - 1;
                            const uint32_t nx = (uint32_t)(clx + dx), ny = (uint32_t)(cly + dy), nz = (uint32_t)(clz + dz);
                            if (nx >= (1u << 21) || ny >= (1u << 21) || nz >= (1u << 21)) continue;
                            const uint64_t key = pack_cell(nx, ny, nz);
                            uint32_t slot = hash_cell(key, gl_.shift) & gl_.mask;
                            uint32_t start = 0, count = 0;
                            for (;;) {
                                const uint4 e = gl_.table[slot];
                                const uint64_t ek = (uint64_t)e.x | ((uint64_t)e.y << 32);
                                if (ek == key) { start = e.z; count = e.w; break; }
                                if (ek == EMPTY_KEY) break;
                                slot = (slot + 1) & gl_.mask;
                            }
                            scan_range(map.sorted, start, count, qx, qy, qz, k);
                        }
                        merge_group<S>(k);
                        const float scale = (float)(1 << level);
                        const float r = map.cell * (scale * 0.999f - 8.f * 1.1920928955078125e-07f * ((float)amax + 2.f * scale));
                        const float d5 = __uint_as_float((uint32_t)(k[KNN - 1] >> 32));
                        if (k[KNN - 1] != NONE_KEY && r > 0.f && d5 < r * r) {
                            decided = true;
                        } else {
#pragma unroll
                            for (int j = 0; j < KNN; ++j) k[j] = NONE_KEY;
                            ++level;
                        }
                    } else {
                        for (uint32_t j = gl; j < map.m; j += S) scan_range(map.sorted, j, 1, qx, qy, qz, k);
                        merge_group<S>(k);
                        decided = true;
                        level = map.n_levels + 1;
                    }
                }
                if (level > 0 && gl == 0 && fallback_counter) atomicAdd(fallback_counter, 1);
#pragma unroll
                for (int j = 0; j < KNN; ++j) found += (k[j] != NONE_KEY) ? 1 : 0;
            }

            // ---- Plane(near, sq_dists): gates, fit, is_plane ---------------------------------
            float abcd[4] = {0.f, 0.f, 0.f, 0.f};
            float dist = 0.f;
            if (found >= KNN) {                                                   // Plane.cpp:36-38
                const float d5 = __uint_as_float((uint32_t)(k[KNN - 1] >> 32));
                if ((double)d5 < prm.max_dist_plane_sq) {                         // Plane.cpp:40-43
                    float A[KNN][3], P[KNN][3];
#pragma unroll
                    for (int j = 0; j < KNN; ++j) {
                        const float4 nb = map.orig[(uint32_t)k[j]];
                        A[j][0] = P[j][0] = nb.x;
                        A[j][1] = P[j][1] = nb.y;
                        A[j][2] = P[j][2] = nb.z;
                    }
                    float nv[3];
                    plane_qr_solve(A, nv);                                        // Utils.cpp:47
                    const float nrm = sqrtf(dot3f(nv[0], nv[0], nv[1], nv[1], nv[2], nv[2]));  // Utils.cpp:50
                    float e0 = nv[0] / nrm, e1 = nv[1] / nrm, e2 = nv[2] / nrm;
                    float e3 = (float)(1.0 / (double)nrm);                        // Utils.cpp:54
                    bool ok = true;                                               // Utils.cpp:59-66
#pragma unroll
                    for (int j = 0; j < KNN; ++j) {
                        float res = e0 * P[j][0] + e1 * P[j][1] + e2 * P[j][2] + e3;
                        if (fabsf(res) > prm.planes_threshold) ok = false;
                    }
                    if (ok) {
                        chosen = true;
                        abcd[0] = e0; abcd[1] = e1; abcd[2] = e2; abcd[3] = e3;
                        dist = e0 * qx + e1 * qy + e2 * qz + e3;                  // Plane.cpp:27-29, Match.cpp:21
                    }
                }
            }

            // ---- Localizator::calculate_H row (Localizator.cpp:36-56) --------------------------
            if (chosen) {
                float plx, ply, plz, pix, piy, piz;
                rt_apply(pc.back, qx, qy, qz, plx, ply, plz);                     // :38
                rt_apply(pc.LI, plx, ply, plz, pix, piy, piz);                    // :39
                const double n0 = (double)abcd[0], n1 = (double)abcd[1], n2 = (double)abcd[2];
                const double* Ri = pc.R_inv;
                const double* Li = pc.I_R_L_inv;
                const double C0 = dot3d(Ri[0], n0, Ri[1], n1, Ri[2], n2);         // :47
                const double C1 = dot3d(Ri[3], n0, Ri[4], n1, Ri[5], n2);
                const double C2 = dot3d(Ri[6], n0, Ri[7], n1, Ri[8], n2);
                const double t0 = dot3d(Li[0], C0, Li[1], C1, Li[2], C2);
                const double t1 = dot3d(Li[3], C0, Li[4], C1, Li[5], C2);
                const double t2 = dot3d(Li[6], C0, Li[7], C1, Li[8], C2);
                const double lx = (double)plx, ly = (double)ply, lz = (double)plz;
                const double ix = (double)pix, iy = (double)piy, iz = (double)piz;
                row[0] = n0; row[1] = n1; row[2] = n2;                            // :51
                row[3] = iy * C2 - iz * C1;                                       // A = p_imu x C  :49
                row[4] = iz * C0 - ix * C2;
                row[5] = ix * C1 - iy * C0;
                if (prm.estimate_extrinsics) {                                    // :52
                    row[6] = ly * t2 - lz * t1;                                   // B = p_lidar x (I_R_L_inv C)  :48
                    row[7] = lz * t0 - lx * t2;
                    row[8] = lx * t1 - ly * t0;
                    row[9] = C0; row[10] = C1; row[11] = C2;
                }
                hres = -(double)dist;                                             // :55
            }

            if (DBG && gl == 0) {
                if (dbg.knn_idx) {
#pragma unroll
                    for (int j = 0; j < KNN; ++j) {
                        dbg.knn_idx[(size_t)oq * KNN + j] = (uint32_t)k[j];
                        dbg.knn_d2[(size_t)oq * KNN + j] = k[j] == NONE_KEY ? __uint_as_float(0x7f800000u)
                                                                           : __uint_as_float((uint32_t)(k[j] >> 32));
                    }
                }
                if (dbg.valid) dbg.valid[oq] = chosen ? 1 : 0;
                if (dbg.p_world) { dbg.p_world[(size_t)oq * 3] = qx; dbg.p_world[(size_t)oq * 3 + 1] = qy; dbg.p_world[(size_t)oq * 3 + 2] = qz; }
                if (dbg.abcd) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) dbg.abcd[(size_t)oq * 4 + j] = abcd[j];
                }
                if (dbg.dist) dbg.dist[oq] = dist;
                if (dbg.rows) {
#pragma unroll
                    for (int j = 0; j < 12; ++j) dbg.rows[(size_t)oq * 12 + j] = row[j];
                    dbg.h[oq] = hres;
                }
            }
        }
        if (gl == 0) {
#pragma unroll
            for (int j = 0; j < 12; ++j) s_rows[gq][j] = row[j];
            s_rows[gq][12] = hres;
            s_rows[gq][13] = chosen ? 1.0 : 0.0;
        }
        __syncthreads();
        if (tid < N_OUT) {
#pragma unroll 8
            for (int p = 0; p < G; ++p) acc += s_rows[p][oa] * s_rows[p][ob];
        }
        __syncthreads();
    }
    if (tid < SUMS_LEN) partials[(size_t)blockIdx.x * SUMS_LEN + tid] = tid < N_OUT ? acc : 0.0;
}

int match_grid_size(int S, uint32_t n, int max_blocks) {
    const uint32_t G = 256 / S;
    uint32_t need = (n + G - 1) / G;
    if (need < 1) need = 1;
    return (int)(need < (uint32_t)max_blocks ? need : (uint32_t)max_blocks);
}

template <int S>
static void launch_s(hipStream_t stream, bool dbg_on, int grid, const MapView& map, const float4* scan, uint32_t n,
                     const KfDev* kf, const MatchParams& prm, double* partials, const DebugOut& dbg, int* fb) {
    if (dbg_on)
        hipLaunchKernelGGL((match_reduce_kernel<S, true>), dim3(grid), dim3(256), 0, stream, map, scan, n, kf, prm, partials, dbg, fb);
    else
        hipLaunchKernelGGL((match_reduce_kernel<S, false>), dim3(grid), dim3(256), 0, stream, map, scan, n, kf, prm, partials, dbg, fb);
}

int launch_match_reduce(hipStream_t stream, int S, const MapView& map, const float4* scan_sorted, uint32_t n, const KfDev* kf,
                        const MatchParams& prm, double* partials, int grid, const DebugOut& dbg, int* fallback_counter) {
    const bool dbg_on = dbg.knn_idx || dbg.valid || dbg.p_world || dbg.abcd || dbg.dist || dbg.rows;
    switch (S) {
        case 1: launch_s<1>(stream, dbg_on, grid, map, scan_sorted, n, kf, prm, partials, dbg, fallback_counter); break;
        case 2: launch_s<2>(stream, dbg_on, grid, map, scan_sorted, n, kf, prm, partials, dbg, fallback_counter); break;
        case 4: launch_s<4>(stream, dbg_on, grid, map, scan_sorted, n, kf, prm, partials, dbg, fallback_counter); break;
        case 8: launch_s<8>(stream, dbg_on, grid, map, scan_sorted, n, kf, prm, partials, dbg, fallback_counter); break;
        case 16: launch_s<16>(stream, dbg_on, grid, map, scan_sorted, n, kf, prm, partials, dbg, fallback_counter); break;
        default: set_error("lanes_per_query must be 1,2,4,8 or 16 (got %d)", S); return LV_EINVAL;
    }
    LV_HIP(hipGetLastError());
    return LV_OK;
}

}  // namespace lv

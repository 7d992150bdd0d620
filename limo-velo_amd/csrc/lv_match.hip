// lv_match.hip — the measurement-model pass: world transform -> exact 5-NN in the voxel hash ->
// plane fit + gates -> point-to-plane residual + Jacobian row -> block-level H^T H / H^T h partials.
//
// Two launches replace, for all N scan points (reference call stack SURVEY.md §3.1):
//   search_kernel      Mapper::match            src/Modules/Mapper.cpp:40-56   (transform :51, match_plane :82-90)
//                      KD_TREE::Nearest_Search  call site Mapper.cpp:86        [ikd-Tree, absent; exact kNN restated]
//   fit_reduce_kernel  Plane::Plane / fit_plane src/Objects/Plane.cpp:19-55
//                      R3Math::estimate_plane / is_plane   src/Utils/Utils.cpp:32-66
//                      Match::Match             src/Objects/Match.cpp:18-22
//                      Localizator::calculate_H src/Modules/Localizator.cpp:29-57
//                      esekf h_x^T h_x, h_x^T h products   [IKFoM, absent]     -> never materialises the N x 12 H
//
// Work decomposition (wave64).  Search: S lanes of a wavefront cooperate on one scan point: they stream the
// neighbourhood bucket of its voxel 8 loads per lane in flight, keep private sorted top-5 lists of packed
// (distance bits << 32 | position) keys — one f64 min/max pair is the reference's (d, index) lexicographic
// compare-exchange — and merge them across lanes with DPP moves; the 5 winners go to a 128-byte record.
// Fit: one lane per scan point runs the QR plane fit, the gates and the Jacobian row, stages the row in LDS,
// and each wavefront contracts its 64 rows into the 29 (92) sums of the block partial in f64, fixed order.
//
// Exactness of the voxel search: the query's level-l voxel and its 26 neighbours cover every point
// within (2^l * cell) of the query up to rounding of the voxel coordinates; a level is accepted only
// if 5 candidates were found and d5 < r_l^2 with r_l shrunk by a 1e-3 relative margin plus the f32
// rounding bound of the coordinates involved.  Otherwise the next (coarser) level is searched from
// scratch: bucket levels 0 and 1 per lane group, level 2 by whole wavefronts, the level-3 block as the 216
// level-2 voxel lists that tile it, finally every id.  Hence the result equals exhaustive search under the
// (d, index) order for every query.  Non-capturing launches may stop once a rejected level proves
// d5 >= MAX_DIST_PLANE^2 (knn_search): the reference drops such a match whatever its neighbours are.
#include "lv_host.hpp"
#include "lv_pass_dev.hpp"    // solve_prep / solve_core / out_pair (restores -ffp-contract=off for everything below)

// A/B switches of two round-4 changes (scripts/build_variant.sh NAME -DLV_...=0 builds the form before them)
#ifndef LV_HALF_CHUNK
#define LV_HALF_CHUNK 1   // bucket_attempt: the tail of a level-0 bucket beyond its first chunk takes half a chunk when it fits
#endif
#ifndef LV_QR_SELECT
#define LV_QR_SELECT 1    // plane_qr_solve: selects instead of branches (tiny-tail case, norm downdate, Q^T c, back substitution)
#endif

namespace lv {

// Candidate keys.  A key packs (f32 distance bits << 32 | position) and is handled as an IEEE f64:
// for non-negative distances the f64 order of the bit pattern equals the unsigned order, so ONE
// v_min_f64 / v_max_f64 pair is a compare-exchange on the (distance, position) pair.  Inside a bucket
// the points are stored in ascending ORIGINAL index, so (distance, position) order == the reference's
// (distance, index) order, ties included; the generic path uses (distance, index) directly.
// NONE = largest finite f64 (its high word 0x7FEFFFFF is an f32 NaN pattern no distance produces;
// no key is ever an f64 NaN/inf because valid distance bits are <= 0x7F800000).
typedef double kkey;
__device__ __forceinline__ kkey make_key(float d, uint32_t low) {
    return __longlong_as_double((long long)(((uint64_t)__float_as_uint(d) << 32) | (uint64_t)low));
}
// the key of a candidate slot that may lie behind the end of its run: selects, not a branch around the distance arithmetic (the
// compiler turns `ok ? make_key(calc_dist(...), j) : none_key()` into an exec-mask region per candidate: save / branch / wait /
// restore around eight instructions).  Level-0 stream: search phase 13.4 -> 13.2 us per launch; level 1: see bucket_attempt
// (the loads have to be pinned in front of the arithmetic there).
__device__ __forceinline__ kkey make_key_if(bool ok, float d, uint32_t low) {
    const uint32_t hi = ok ? __float_as_uint(d) : 0x7FEFFFFFu;
    const uint32_t lo = ok ? low : 0xFFFFFFFFu;
    return __hiloint2double((int)hi, (int)lo);
}
__device__ __forceinline__ uint32_t key_lo(kkey k) { return (uint32_t)(uint64_t)__double_as_longlong(k); }
__device__ __forceinline__ uint32_t key_hi(kkey k) { return (uint32_t)((uint64_t)__double_as_longlong(k) >> 32); }
#define LV_NONE_BITS 0x7FEFFFFFFFFFFFFFll
__device__ __forceinline__ kkey none_key() { return __longlong_as_double(LV_NONE_BITS); }
__device__ __forceinline__ bool is_none(kkey k) { return __double_as_longlong(k) == LV_NONE_BITS; }


__device__ __forceinline__ kkey kmin(kkey a, kkey b) {
    kkey r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ kkey kmax(kkey a, kkey b) {
    kkey r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void cswap(kkey& a, kkey& b) {
    const kkey lo = kmin(a, b), hi = kmax(a, b);
    a = lo;
    b = hi;
}
// Batcher odd-even merge sort, 8 keys, 19 comparators (verified exhaustively with the 0-1 principle)
__device__ __forceinline__ void sort8(kkey (&c)[8]) {
    cswap(c[0], c[1]); cswap(c[2], c[3]); cswap(c[4], c[5]); cswap(c[6], c[7]);
    cswap(c[0], c[2]); cswap(c[1], c[3]); cswap(c[4], c[6]); cswap(c[5], c[7]);
    cswap(c[1], c[2]); cswap(c[5], c[6]);
    cswap(c[0], c[4]); cswap(c[1], c[5]); cswap(c[2], c[6]); cswap(c[3], c[7]);
    cswap(c[2], c[4]); cswap(c[3], c[5]);
    cswap(c[1], c[2]); cswap(c[3], c[4]); cswap(c[5], c[6]);
}
// k, o sorted ascending (o: at least 5 entries) -> k = the 5 smallest of the union, sorted:
// min(k[i], o[4-i]) selects exactly the 5 smallest (bitonic halving).  The selection is UNIMODAL — it follows the ascending k
// while k[i] is the smaller one, then the descending o[4-i] — and a unimodal sequence of 5 needs 5 comparators, not the 9 of
// a general sort5: (0,4) (1,3) (1,4) (2,4) (3,4), minimal by exhaustive search over all networks on the 0^p 1^m 0^q images
// (0-1 principle restricted to unimodal inputs; scripts/merge5_network.py re-derives and checks it).  Round 3: the search
// phase is 55 % VALU and the three DPP merge rounds were 99 of a task's ~450 instructions; they are 75 now.
__device__ __forceinline__ void order_unimodal5(kkey (&c)[5]) {
    cswap(c[0], c[4]); cswap(c[1], c[3]); cswap(c[1], c[4]); cswap(c[2], c[4]); cswap(c[3], c[4]);
}
// NUM_MATCH_POINTS other than 5 (3..8, the general-K build of the three-kernel pass: not a tuned path): the K selected keys
// padded with NONE to eight and sorted by the 19-comparator network
template <int K>
__device__ __forceinline__ void order_selected(kkey (&c)[K]) {
    if constexpr (K == 5) {
        order_unimodal5(c);
    } else {
        kkey t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = i < K ? c[i < K ? i : 0] : none_key();
        sort8(t);
#pragma unroll
        for (int i = 0; i < K; ++i) c[i] = t[i];
    }
}
// k, o sorted ascending (o: at least K entries) -> k = the K smallest of the union, sorted (K = 5: see above)
template <int K, typename T>
__device__ __forceinline__ void merge5(kkey (&k)[K], const T& o) {
#pragma unroll
    for (int i = 0; i < K; ++i) k[i] = kmin(o[K - 1 - i], k[i]);
    order_selected<K>(k);
}
// cross-lane exchange of a key through DPP (VALU data path, no LDS round trip):
//   0xB1 quad_perm(1,0,3,2) = lane ^ 1, 0x4E quad_perm(2,3,0,1) = lane ^ 2,
//   0x141 row_half_mirror = lane <-> 7 - lane (8-lane halves), 0x140 row_mirror = lane <-> 15 - lane.
// After the two quad rounds all four lanes of a quad hold the same list, so a mirror is as good as xor 4 / 8.
template <int CTRL>
__device__ __forceinline__ kkey dpp_key(kkey v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    // all source lanes of these permutations are enabled: bound_ctrl only spares the `old` operand its zeroing move
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int K>
__device__ __forceinline__ void merge_round(kkey (&k)[K]) {
    kkey o[K];
#pragma unroll
    for (int j = 0; j < K; ++j) o[j] = dpp_key<CTRL>(k[j]);
    merge5(k, o);
}
template <int S, int K>
__device__ __forceinline__ void merge_group(kkey (&k)[K]) {
    if (S >= 2) merge_round<0xB1>(k);
    if (S >= 4) merge_round<0x4E>(k);
    if (S >= 8) merge_round<0x141>(k);
    if (S >= 16) merge_round<0x140>(k);
}
// [UPSTREAM-RECALL ikd-Tree calc_dist]: (ax-bx)^2 + (ay-by)^2 + (az-bz)^2, f32, left to right, unfused
struct __attribute__((packed, aligned(4))) Xyz {   // one bucket point as streamed: 12 bytes
    float x, y, z;
};
__device__ __forceinline__ float calc_dist(float qx, float qy, float qz, Xyz m) {
    float dx = qx - m.x, dy = qy - m.y, dz = qz - m.z;
    float sx = dx * dx, sy = dy * dy, sz = dz * dz;
    float s = sx + sy;
    return s + sz;
}
__device__ __forceinline__ float calc_dist(float qx, float qy, float qz, float4 m) {
    float dx = qx - m.x, dy = qy - m.y, dz = qz - m.z;
    float sx = dx * dx, sy = dy * dy, sz = dz * dz;
    float s = sx + sy;
    return s + sz;
}

// Column-pivoted Householder QR least squares for the K x 3 system A n = -1 (K = NUM_MATCH_POINTS: 5 on the tuned paths), f32.  Same operation
// sequence as the oracle's restatement of R3Math::estimate_plane's
// `A.colPivHouseholderQr().solve(b)` (reference src/Utils/Utils.cpp:47).
// fast_fit (opt-in, lv_set_option "fast_fit"; OFF by default): the plane fit's divisions and square roots by the hardware's
// approximations — v_rcp_f32 + one Newton step on the quotient (2 fused multiply-adds), v_sqrt_f32 (1 ulp) — instead of the
// correctly rounded expansions the bit-exact default needs (26 divisions ~10 instructions each, 13 square roots ~12 each: a third
// of fit_row).  Results are within a few f32 ulps of the exact path: north_star's "residuals / state within a stated fp32
// tolerance", NOT the bit-exactness every parity test of the default path holds; a plane on the edge of PLANES_THRESHOLD may
// flip (tests/test_gpu_fast_fit.py counts the flips and bounds the band they come from).
template <bool FAST>
__device__ __forceinline__ float fit_div(float a, float b) {
    if constexpr (FAST) {
        const float r = __builtin_amdgcn_rcpf(b);
        const float q = a * r;
        return __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
    } else {
        return a / b;
    }
}
template <bool FAST>
__device__ __forceinline__ float fit_sqrt(float x) {
    if constexpr (FAST) return __builtin_amdgcn_sqrtf(x);
    else return sqrtf(x);
}
template <int K, bool FAST = false>
__device__ inline void plane_qr_solve(float (&A)[K][3], float (&x)[3]) {
    constexpr int rows = K, cols = 3, size = 3;
    const float eps = 1.1920928955078125e-07f;
    float hC[3];
    int perm[3] = {0, 1, 2};
    float nU[3], nD[3];
#pragma unroll
    for (int k = 0; k < cols; ++k) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < rows; ++i) s += A[i][k] * A[i][k];
        nD[k] = nU[k] = fit_sqrt<FAST>(s);
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
#if LV_QR_SELECT
        {   // both sides computed, selected (same operations on the side that counts => same bits; the discarded side may divide by zero)
            const bool tiny = tailSq <= 1.17549435e-38f;
            float b = fit_sqrt<FAST>(c0 * c0 + tailSq);
            if (c0 >= 0.f) b = -b;
            const float den = c0 - b;
#pragma unroll
            for (int i = k + 1; i < rows; ++i) {
                const float q = fit_div<FAST>(A[i][k], den);
                A[i][k] = tiny ? 0.f : q;
            }
            const float t = fit_div<FAST>(b - c0, b);
            tau = tiny ? 0.f : t;
            beta = tiny ? c0 : b;
        }
#else
        if (tailSq <= 1.17549435e-38f) {
            tau = 0.f;
            beta = c0;
#pragma unroll
            for (int i = k + 1; i < rows; ++i) A[i][k] = 0.f;
        } else {
            beta = fit_sqrt<FAST>(c0 * c0 + tailSq);
            if (c0 >= 0.f) beta = -beta;
            float den = c0 - beta;
#pragma unroll
            for (int i = k + 1; i < rows; ++i) A[i][k] = fit_div<FAST>(A[i][k], den);
            tau = fit_div<FAST>(beta - c0, beta);
        }
#endif
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
#if LV_QR_SELECT
            const bool nz = nU[j] != 0.f;
            float temp = fit_div<FAST>(fabsf(A[k][j]), nU[j]);
            temp = (1.f + temp) * (1.f - temp);
            temp = temp < 0.f ? 0.f : temp;
            const float r = fit_div<FAST>(nU[j], nD[j]);
            const float temp2 = temp * (r * r);
            const bool redo = temp2 <= norm_downdate_threshold;
            float s = 0.f;
#pragma unroll
            for (int i = k + 1; i < rows; ++i) s += A[i][j] * A[i][j];
            const float nd_new = fit_sqrt<FAST>(s);
            const float nu_scaled = nU[j] * fit_sqrt<FAST>(temp);
            nD[j] = (nz && redo) ? nd_new : nD[j];
            nU[j] = nz ? (redo ? nd_new : nu_scaled) : nU[j];
#else
            if (nU[j] != 0.f) {
                float temp = fit_div<FAST>(fabsf(A[k][j]), nU[j]);
                temp = (1.f + temp) * (1.f - temp);
                temp = temp < 0.f ? 0.f : temp;
                float r = fit_div<FAST>(nU[j], nD[j]);
                float temp2 = temp * (r * r);
                if (temp2 <= norm_downdate_threshold) {
                    float s = 0.f;
#pragma unroll
                    for (int i = k + 1; i < rows; ++i) s += A[i][j] * A[i][j];
                    nD[j] = fit_sqrt<FAST>(s);
                    nU[j] = nD[j];
                } else {
                    nU[j] *= fit_sqrt<FAST>(temp);
                }
            }
#endif
        }
    }
    x[0] = x[1] = x[2] = 0.f;
    if (nonzero_pivots == 0) return;
    float c[K];
#pragma unroll
    for (int i = 0; i < rows; ++i) c[i] = -1.0f;
#if LV_QR_SELECT
#pragma unroll
    for (int k = 0; k < size; ++k) {   // c = Q^T c, selects instead of branches (see the Householder step)
        const float tau = hC[k];
        const bool on = k < nonzero_pivots && tau != 0.f;
        float tmp = 0.f;
#pragma unroll
        for (int i = k + 1; i < rows; ++i) tmp += A[i][k] * c[i];
        tmp += c[k];
        const float ck = c[k] - tau * tmp;
        c[k] = on ? ck : c[k];
#pragma unroll
        for (int i = k + 1; i < rows; ++i) {
            const float ci = c[i] - tau * A[i][k] * tmp;
            c[i] = on ? ci : c[i];
        }
    }
#pragma unroll
    for (int i = size - 1; i >= 0; --i) {
        float sv = c[i];
#pragma unroll
        for (int j = i + 1; j < size; ++j) {
            const float sj = sv - A[i][j] * c[j];
            sv = (j < nonzero_pivots) ? sj : sv;
        }
        const float q = fit_div<FAST>(sv, A[i][i]);
        c[i] = (i < nonzero_pivots) ? q : c[i];
    }
#else
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
            c[i] = fit_div<FAST>(s, A[i][i]);
        }
    }
#endif
#pragma unroll
    for (int i = 0; i < size; ++i) {
        if (i < nonzero_pivots) {
#pragma unroll
            for (int a = 0; a < 3; ++a)
                if (perm[i] == a) x[a] = c[i];
        }
    }
}

// Voxel geometry of one query: continuous and integer level-0 voxel coordinates.
struct QGeom {
    float tx, ty, tz;
    int c0x, c0y, c0z, amax;
};
__device__ __forceinline__ QGeom make_geom(const MapView& map, float qx, float qy, float qz) {
    QGeom g;
    g.tx = (qx - map.origin[0]) * map.inv_cell;
    g.ty = (qy - map.origin[1]) * map.inv_cell;
    g.tz = (qz - map.origin[2]) * map.inv_cell;
    g.c0x = cell_coord(qx, map.origin[0], map.inv_cell);
    g.c0y = cell_coord(qy, map.origin[1], map.inv_cell);
    g.c0z = cell_coord(qz, map.origin[2], map.inv_cell);
    g.amax = max(abs(g.c0x - CELL_OFFSET), max(abs(g.c0y - CELL_OFFSET), abs(g.c0z - CELL_OFFSET)));
    return g;
}
// guaranteed search radius of the 27-voxel block at `lvl` for THIS query: one voxel edge plus the distance to
// the nearest wall of its own voxel, shrunk by 1e-3 relative and by the f32 rounding bound of the voxel
// coordinates (see file header).
__device__ __forceinline__ float search_radius(const MapView& map, const QGeom& g, int lvl) {
    const float scale = (float)(1 << lvl);
    const float bx = (float)((((g.c0x >> lvl) << lvl)) - CELL_OFFSET), by = (float)((((g.c0y >> lvl) << lvl)) - CELL_OFFSET),
                bz = (float)((((g.c0z >> lvl) << lvl)) - CELL_OFFSET);
    const float mx = fminf(g.tx - bx, scale - (g.tx - bx)), my = fminf(g.ty - by, scale - (g.ty - by)),
                mz = fminf(g.tz - bz, scale - (g.tz - bz));
    const float marg = fmaxf(fminf(mx, fminf(my, mz)), 0.f);
    return map.cell * ((scale + marg) * 0.999f - 8.f * 1.1920928955078125e-07f * ((float)g.amax + 2.f * scale));
}

__device__ __forceinline__ kkey shfl_xor_key(kkey v, int mask) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, mask);
    hi = __shfl_xor(hi, mask);
    return __hiloint2double(hi, lo);
}
// all-reduce of the sorted top-5 lists over a team of LANES consecutive lanes (LANES <= 16: merge_group over
// DPP; 64: the whole wavefront, the two cross-row rounds go through ds_bpermute)
template <int LANES, int K>
__device__ __forceinline__ void merge_team(kkey (&k)[K]) {
    if (LANES <= 16) {
        merge_group<LANES>(k);
    } else {
        merge_group<16>(k);
#pragma unroll
        for (int mask = 16; mask < LANES; mask <<= 1) {
            kkey o[K];
#pragma unroll
            for (int j = 0; j < K; ++j) o[j] = shfl_xor_key(k[j], mask);
            merge5(k, o);
        }
    }
}

// a candidate key stands for a real point iff its distance is finite (NONE and the +inf distance of deleted
// entries / ids are not)
__device__ __forceinline__ bool key_real(kkey k) { return key_hi(k) < 0x7F800000u; }

// k, o sorted ascending -> k = the K smallest of the union (merge5), drop = min(drop, the distances that left)
template <int K, typename T>
__device__ __forceinline__ void merge5_drop(kkey (&k)[K], const T& o, uint32_t& drop) {
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const kkey lo = kmin(o[K - 1 - i], k[i]), hi = kmax(o[K - 1 - i], k[i]);
        k[i] = lo;
        drop = min(drop, key_hi(hi));   // (NONE's high word is above every distance: it never lowers the minimum)
    }
    order_selected<K>(k);
}
template <int CTRL, int K>
__device__ __forceinline__ void merge_round_drop(kkey (&k)[K], uint32_t& drop) {
    kkey o[K];
#pragma unroll
    for (int j = 0; j < K; ++j) o[j] = dpp_key<CTRL>(k[j]);
    const uint32_t od = (uint32_t)__builtin_amdgcn_update_dpp((int)drop, (int)drop, CTRL, 0xF, 0xF, true);
    drop = min(drop, od);
    merge5_drop(k, o, drop);
}
template <int LANES, int K>
__device__ __forceinline__ void merge_team_drop(kkey (&k)[K], uint32_t& drop) {   // (the DPP rounds of merge_group)
    static_assert(LANES <= 16, "lane groups of up to 16 lanes");
    if (LANES >= 2) merge_round_drop<0xB1>(k, drop);
    if (LANES >= 4) merge_round_drop<0x4E>(k, drop);
    if (LANES >= 8) merge_round_drop<0x141>(k, drop);
    if (LANES >= 16) merge_round_drop<0x140>(k, drop);
}
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One bucket-level attempt by a team of LANES lanes (tl = lane in team): ONE probe of the level's bucket table
// and ONE coalesced stream over the neighbourhood bucket of the query's voxel (a miss means the whole
// 27-voxel block is empty).  k is all-NONE on entry; returns true with the sorted result in k (low words =
// positions inside the bucket starting at bstart) iff 5 candidates were found inside the guaranteed radius,
// otherwise false with k reset to NONE.  Deleted entries stay in place with x = +inf: their distance is +inf,
// which loses against every real candidate and fails the radius test.
// bl == 1 (round 6): the level-1 block in level-0 storage.  Its 27 voxels are tiled exactly by the eight level-0 buckets of one tile
// GROUP (lv_device.hpp REPL_LEVELS), which a (re)build lays out side by side: ONE probe of the group table and ONE stream over the
// region {start, extent} — runs with their slack, the slack filled with +inf (distance +inf, like a deleted entry) — give the
// candidate set a replicated level-1 bucket gave (rounds 1-5) for 27 more copies of every map point and two thirds of every
// insert / deletion.  A group that an insert broke up has extent 0: the point is not decided here and goes on to the lists.
// Order.  Inside ONE bucket position order is id order; across the eight runs it is not, so here the key's low word (position in
// the region) orders equal distances differently from the reference's (distance, index).  That can only change the result when
// two candidates at bit-equal distance compete — among the five winners, or the fifth winner with a candidate that was dropped.
// Both are detected (the smallest dropped distance is tracked through every selection: `drop`) and such a point is NOT decided
// here either: the lists carry ids in their keys.  Exact for every input; the detour is taken by points with an exact f32
// distance tie among their six nearest neighbours.
template <int LANES, int K>
__device__ __forceinline__ bool bucket_attempt(const MapView& map, int bl, const QGeom& geo, float qx, float qy, float qz, int tl,
                                               kkey (&k)[K], uint32_t& bstart, long long* clk, Xyz* stage = nullptr, uint32_t* bound = nullptr) {
    // bound (optional): from a level that is NOT accepted although it saw K living candidates, the bits of the K-th smallest distance
    // among them — K living points lie within it, so no point farther than that is among the query's K nearest (left alone otherwise)
    // stage (LDS, LANES * 8 entries of this team, or nullptr): the first chunk's candidates are kept there by
    // position, so that the caller can pick the winners up without another trip to memory
    const GridLevel g = bl == 0 ? map.bt[0] : map.gt;
    const uint64_t key = pack_cell((uint32_t)(geo.c0x >> bl), (uint32_t)(geo.c0y >> bl), (uint32_t)(geo.c0z >> bl));
    uint32_t drop = 0x7FEFFFFFu;   // (bl == 1) smallest distance that left a selection
    uint32_t slot = hash_cell(key, g.shift) & g.mask;
    uint32_t bcount = 0;
    for (;;) {
        const uint4 e = g.table[slot];
        // (all four words are wanted NOW: left alone the compiler fetches the key's 8 bytes, compares, and only then goes back
        // for {start, count} — a second dependent trip to memory on the hit path of every probe)
        asm volatile("" :: "v"(e.x), "v"(e.y), "v"(e.z), "v"(e.w));
        const uint64_t ek = (uint64_t)e.x | ((uint64_t)e.y << 32);
        if (ek == key) { bstart = e.z; bcount = e.w; break; }
        if (ek == EMPTY_KEY) break;
        slot = (slot + 1) & g.mask;
    }
    if (clk) { asm volatile("" :: "v"(bcount)); clk[2] = clock64(); }
    if (bcount < K) return false;
    constexpr int U = 8;
    const Xyz* __restrict__ bp = reinterpret_cast<const Xyz*>(map.bxyz[0]) + bstart;
    uint32_t base = 0;
    do {
        // (round 4) A bucket of the benchmark's map holds 62 candidates on average and more than the 64 of a chunk one time in four:
        // two in three of the 8-point tasks went through a second chunk for the handful of candidates beyond — at the full price of
        // eight loads, eight distances and the 19-comparator sort.  When what is left fits (at most 4 per lane for every lane group
        // of the wavefront) the tail takes HALF a chunk: four loads, four distances, a 5-comparator sort, the same merge.  Level 0
        // only: on the level-1 stream (seven chunks) the wavefront-uniform loop it needs cost more than the half chunk saves.
        if constexpr (LV_HALF_CHUNK && K == KNN) {
            if (bl == 0 && base != 0 && __builtin_amdgcn_ballot_w64(base + (uint32_t)(LANES * 4) < bcount) == 0ull) {
                Xyz mp4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t j = base + (uint32_t)(u * LANES + tl);
                    mp4[u] = bp[j < bcount ? j : 0];
                }
                kkey c4[5];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t j = base + (uint32_t)(u * LANES + tl);
                    c4[u] = make_key_if(j < bcount, calc_dist(qx, qy, qz, mp4[u]), j);
                }
                c4[4] = none_key();
                cswap(c4[0], c4[1]); cswap(c4[2], c4[3]); cswap(c4[0], c4[2]); cswap(c4[1], c4[3]); cswap(c4[1], c4[2]);
                merge5(k, c4);
                break;   // (nothing is left behind a half chunk)
            }
        }
        Xyz mpt[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t j = base + (uint32_t)(u * LANES + tl);
            mpt[u] = bp[j < bcount ? j : 0];
        }
        if (stage && base == 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) stage[u * LANES + tl] = mpt[u];
        }
        kkey ck[U];
        // Keys by selects on every level (make_key_if).  Behind level 0 the eight loads of a chunk are pinned in front of the
        // arithmetic: left to the scheduler, the select form gets them ONE AT A TIME there (`s_waitcnt vmcnt(0)` after each
        // load, to stay inside 128 registers) — eight dependent round trips per chunk, the first launch 38 -> 59 us; pinned,
        // it beats the exec-mask form: first launch 37.6 -> 35.8 us, second 28.2 -> 26.5 (29.0 -> 29.7 k it/s, A/B in one box).
        if (bl != 0) asm volatile("" :: "v"(mpt[0].x), "v"(mpt[1].x), "v"(mpt[2].x), "v"(mpt[3].x), "v"(mpt[4].x), "v"(mpt[5].x), "v"(mpt[6].x), "v"(mpt[7].x));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t j = base + (uint32_t)(u * LANES + tl);
            ck[u] = make_key_if(j < bcount, calc_dist(qx, qy, qz, mpt[u]), j);
        }
        sort8(ck);
#ifndef LV_DIAG_NODROP
#define LV_DIAG_NODROP 0   // (diagnostic build only: level 1 without the tie tracking — NOT exact)
#endif
        if (bl != 0 && K < U && !LV_DIAG_NODROP) drop = min(drop, key_hi(ck[K < U ? K : 0]));
        if (base == 0) {   // k is still all-NONE: the union's five smallest are the chunk's
#pragma unroll
            for (int i = 0; i < K; ++i) k[i] = ck[i];
        } else if (bl != 0 && !LV_DIAG_NODROP) {
            merge5_drop(k, ck, drop);
        } else {
            merge5(k, ck);
        }
        base += LANES * U;
    } while ((LV_HALF_CHUNK && K == KNN && bl == 0) ? __builtin_amdgcn_ballot_w64(base < bcount) != 0ull : base < bcount);
    if (bl != 0 && !LV_DIAG_NODROP) merge_team_drop<LANES>(k, drop);
    else merge_team<LANES>(k);
    const float r = search_radius(map, geo, bl);
    const float d5 = __uint_as_float(key_hi(k[K - 1]));
    bool ok = !is_none(k[K - 1]) && r > 0.f && d5 < r * r;
    if (bl != 0) {   // (see "Order" above)
        ok = ok && drop != key_hi(k[K - 1]);
#pragma unroll
        for (int j = 0; j + 1 < K; ++j) ok = ok && key_hi(k[j]) != key_hi(k[j + 1]);
    }
    if (ok) return true;
    if (bound && key_real(k[K - 1])) *bound = min(*bound, key_hi(k[K - 1]));
#pragma unroll
    for (int j = 0; j < K; ++j) k[j] = none_key();
    return false;
}

// A block of SIDE x SIDE x SIDE level-2 voxels (origin (bx, by, bz) in level-2 voxel coordinates) searched by a WHOLE wavefront over
// the level-2 voxel lists: every lane probes one voxel (SIDE^3 <= 64: one round; the 6 x 6 x 6 = 216 lists of the level-3 block:
// four), a wave scan turns the list lengths into one virtual candidate array, and the lanes stream it 8 loads per lane in flight
// (each load finds its list by a 6-step binary search over the prefix sums in LDS).  Keys are (distance, id); r = the radius the
// block guarantees for this query (block_radius); on success k holds the sorted result on every lane.  s_pref / s_start: 64 words
// each, private to this wavefront.
// Blocks in use: the 27 lists around the query's level-2 voxel (SIDE 3, the level-2 block); the 64 lists of the 4 x 4 x 4 block
// that extends it, on every axis, on the side of the voxel wall the query is nearer to (SIDE 4: covers 3 m where the 3-block
// covers between 2 and 3 — taken by the timed launches when the 3-block cannot certify MAX_DIST_PLANE); the 216 lists that tile
// the level-3 block (SIDE 6, capturing launches).
// guaranteed radius of a block of `side` level-`lvl` voxels per axis whose lower corner is voxel (bx, by, bz) (level-lvl voxel
// coordinates) for THIS query: the distance to the nearest face, shrunk like search_radius
__device__ __forceinline__ float block_radius(const MapView& map, const QGeom& g, int lvl, int bx, int by, int bz, int side) {
    const float lox = (float)((bx << lvl) - CELL_OFFSET), loy = (float)((by << lvl) - CELL_OFFSET), loz = (float)((bz << lvl) - CELL_OFFSET);
    const float ext = (float)(side << lvl);
    const float mx = fminf(g.tx - lox, lox + ext - g.tx), my = fminf(g.ty - loy, loy + ext - g.ty), mz = fminf(g.tz - loz, loz + ext - g.tz);
    const float m = fmaxf(fminf(mx, fminf(my, mz)), 0.f);
    return map.cell * (m * 0.999f - 8.f * 1.1920928955078125e-07f * ((float)g.amax + 2.f * ext));
}
template <int K>
__device__ __forceinline__ bool cells_attempt(const MapView& map, int bx, int by, int bz, int SIDE, float r, float qx, float qy, float qz, int lane,
                                              kkey (&k)[K], uint32_t* s_pref, uint32_t* s_start, const QGeom* geo = nullptr, uint32_t bound = 0x7FFFFFFFu) {
    // geo + bound (optional; round 6): what the bucket levels proved on their way here — K living points lie within the distance
    // whose bits are `bound` — so a list whose VOXEL is farther from the query than that holds none of the K nearest and is neither
    // probed nor streamed: of the 27 voxels of a block a sphere of 1.2 .. 1.5 m touches about eight
    const int NC = SIDE * SIDE * SIDE;
    constexpr int U = 8;
    const GridLevel g = map.ct;
#pragma unroll
    for (int j = 0; j < K; ++j) k[j] = none_key();
    for (int r0 = 0; r0 < NC; r0 += 64) {
        const int ci = r0 + lane;
        uint32_t start = 0, cnt = 0;
        if (ci < NC) {
            // (SIDE is 3, 4 or 6: divisions by constants, selected)
            const int dz = SIDE == 3 ? ci / 9 : (SIDE == 4 ? ci >> 4 : ci / 36), dy = SIDE == 3 ? (ci / 3) % 3 : (SIDE == 4 ? (ci >> 2) & 3 : (ci / 6) % 6),
                      dx = SIDE == 3 ? ci % 3 : (SIDE == 4 ? ci & 3 : ci % 6);
            const uint32_t nx = (uint32_t)(bx + dx), ny = (uint32_t)(by + dy), nz = (uint32_t)(bz + dz);
            bool wanted = nx < (1u << 19) && ny < (1u << 19) && nz < (1u << 19);
            if (wanted && geo && bound < 0x7F800000u) {
                // distance from the query to the voxel's box in level-0 voxel units, made smaller by the rounding of the voxel
                // coordinates (as search_radius) and by 2e-3 relative before it is compared with the proven distance
                const float err = 8.f * 1.1920928955078125e-07f * ((float)geo->amax + 8.f);
                const float lx = (float)((int)(nx << 2) - CELL_OFFSET), ly = (float)((int)(ny << 2) - CELL_OFFSET), lz = (float)((int)(nz << 2) - CELL_OFFSET);
                const float ex = fmaxf(fmaxf(lx - geo->tx, geo->tx - (lx + 4.f)) - err, 0.f);
                const float ey = fmaxf(fmaxf(ly - geo->ty, geo->ty - (ly + 4.f)) - err, 0.f);
                const float ez = fmaxf(fmaxf(lz - geo->tz, geo->tz - (lz + 4.f)) - err, 0.f);
                const float d2 = (ex * ex + ey * ey + ez * ez) * (map.cell * map.cell) * 0.998f;
                wanted = !(d2 > __uint_as_float(bound));
            }
            if (wanted) {
                const uint64_t key = pack_cell(nx, ny, nz);
                uint32_t slot = hash_cell(key, g.shift) & g.mask;
                for (;;) {
                    const uint4 e = g.table[slot];
                    asm volatile("" :: "v"(e.x), "v"(e.y), "v"(e.z), "v"(e.w));   // (one 16-byte load: see bucket_attempt)
                    const uint64_t ek = (uint64_t)e.x | ((uint64_t)e.y << 32);
                    if (ek == key) { start = e.z; cnt = e.w; break; }
                    if (ek == EMPTY_KEY) break;
                    slot = (slot + 1) & g.mask;
                }
            }
        }
        uint32_t incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        const uint32_t total = __shfl(incl, 63);
        if (total == 0) continue;
        wave_lds_fence();   // the previous round's readers are done
        s_pref[lane] = incl - cnt;
        s_start[lane] = start;
        wave_lds_fence();
        for (uint32_t base = 0; base < total; base += 64 * U) {
            // the chunk's positions FIRST — the eight look-ups advance together: eight independent LDS reads per step of the
            // binary search (six steps), then the eight loads back to back.  One look-up after the other (round 2's form) was
            // 48 dependent LDS round trips per chunk: what made "a list block cost a wavefront ~2x a bucket" (experiments_r03)
            uint32_t at[U];
            {
                uint32_t vv[U];
                int Lr[U];   // the last list whose first virtual index is <= vv (empty lists share their successor's)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t v = base + (uint32_t)(u * 64 + lane);
                    vv[u] = v < total ? v : 0u;
                    Lr[u] = 0;
                }
#pragma unroll
                for (int step = 32; step >= 1; step >>= 1) {
                    uint32_t pv[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) pv[u] = s_pref[Lr[u] + step];
#pragma unroll
                    for (int u = 0; u < U; ++u) Lr[u] += pv[u] <= vv[u] ? step : 0;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) at[u] = s_start[Lr[u]] + (vv[u] - s_pref[Lr[u]]);
            }
            float4 mpt[U];
#pragma unroll
            for (int u = 0; u < U; ++u) mpt[u] = map.cell4[at[u]];
            kkey ck[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t v = base + (uint32_t)(u * 64 + lane);
                ck[u] = (u < U && v < total) ? make_key(calc_dist(qx, qy, qz, mpt[u < U ? u : 0]), __float_as_uint(mpt[u < U ? u : 0].w)) : none_key();
            }
            sort8(ck);
            merge5(k, ck);
        }
    }
    merge_team<64>(k);
    const float d5 = __uint_as_float(key_hi(k[K - 1]));
    if (!is_none(k[K - 1]) && r > 0.f && d5 < r * r) return true;
#pragma unroll
    for (int j = 0; j < K; ++j) k[j] = none_key();
    return false;
}

// exhaustive scan of every id by a whole wavefront, (distance, id) keys; deleted ids sit at +inf
template <int K>
__device__ __forceinline__ void brute_attempt(const MapView& map, float qx, float qy, float qz, int lane, kkey (&k)[K]) {
    constexpr int U = 8;
#pragma unroll
    for (int j = 0; j < K; ++j) k[j] = none_key();
    for (uint32_t base = 0; base < map.n_ids; base += 64 * U) {
        float4 mpt[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t j = base + (uint32_t)(u * 64 + lane);
            mpt[u] = map.orig[j < map.n_ids ? j : 0];
        }
        kkey ck[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t j = base + (uint32_t)(u * 64 + lane);
            ck[u] = j < map.n_ids ? make_key(calc_dist(qx, qy, qz, mpt[u]), j) : none_key();
        }
        sort8(ck);
        merge5(k, ck);
    }
    merge_team<64>(k);
}

// One scan point that the per-lane-group bucket levels left undecided, searched by a WHOLE wavefront: level-2 bucket, the
// level-3 block as 216 voxel lists, finally every id (see knn_search).  On return every lane holds the sorted result in kw
// (keys carry point ids); returns what decided (2, 3 voxel levels; 4 every id; 5 bounded stop: no neighbours reported).
#ifndef LV_COARSE_INLINE
#define LV_COARSE_INLINE __forceinline__
#endif
template <bool DBG, int K>
__device__ LV_COARSE_INLINE int knn_coarse(const MapView& map, KfDev* __restrict__ kf, float wx, float wy, float wz, int lane, kkey (&kw)[K],
                                          double max_dist_sq, uint32_t* s_pref, uint32_t* s_start, uint32_t bound = 0x7FFFFFFFu) {
#ifdef LV_DIAG_NO_COARSE
    return 5;
#endif
    const QGeom wgeo = make_geom(map, wx, wy, wz);
    const bool w_in_range = wgeo.amax < CELL_FAR;
#pragma unroll
    for (int j = 0; j < K; ++j) kw[j] = none_key();
    int wbin = 5;
    bool done = false;
    if (w_in_range) {
        bool stop = false;
        const int c2x = wgeo.c0x >> 2, c2y = wgeo.c0y >> 2, c2z = wgeo.c0z >> 2;
        if constexpr (DBG) {   // capturing launches: the level-2 block, then the level-3 block, to the exact answer
            done = cells_attempt(map, c2x - 1, c2y - 1, c2z - 1, 3, search_radius(map, wgeo, 2), wx, wy, wz, lane, kw, s_pref, s_start);
            if (done) wbin = 2;
            if (!done) {
                if (lane == 0) atomicAdd(&kf->fallback_queries, 1);
                done = cells_attempt(map, ((wgeo.c0x >> 3) - 1) * 2, ((wgeo.c0y >> 3) - 1) * 2, ((wgeo.c0z >> 3) - 1) * 2, 6, search_radius(map, wgeo, 3),
                                     wx, wy, wz, lane, kw, s_pref, s_start);
                if (done) wbin = 3;
            }
        } else {
            // timed launches: ONE list block.  A block that is not accepted proves d5 >= r^2 (f32), and the reference's gate is
            // (double)d5 < MAX_DIST_PLANE^2 (Plane.cpp:42): once r^2 reaches that, the point has no match whatever its
            // neighbours are.  The level-2 block guarantees between 2 and 3 m (cell 0.5); when that is not enough for the gate
            // (MAX_DIST_PLANE = 2 and the query within millimetres of a wall of its level-2 voxel; larger MAX_DIST_PLANE) the block
            // grows to 4 x 4 x 4 voxels, extended on every axis towards the wall the query is nearer to: >= 3 m.  One call site
            // with run-time geometry: a second inlined copy (the level-3 block, rounds 1-5) cost the one-launch pass 1.5 us per
            // launch in register allocation alone (profiles/experiments_r06).
            float r = search_radius(map, wgeo, 2);
            int side = 3, bx = c2x - 1, by = c2y - 1, bz = c2z - 1;
            if (!(r > 0.f && (double)(r * r) >= max_dist_sq)) {
                side = 4;
                bx = c2x - 1 - ((wgeo.tx - (float)((c2x << 2) - CELL_OFFSET)) < 2.f ? 1 : 0);
                by = c2y - 1 - ((wgeo.ty - (float)((c2y << 2) - CELL_OFFSET)) < 2.f ? 1 : 0);
                bz = c2z - 1 - ((wgeo.tz - (float)((c2z << 2) - CELL_OFFSET)) < 2.f ? 1 : 0);
                r = block_radius(map, wgeo, 2, bx, by, bz, 4);
                if (lane == 0) atomicAdd(&kf->fallback_queries, 1);
            }
            done = cells_attempt(map, bx, by, bz, side, r, wx, wy, wz, lane, kw, s_pref, s_start, &wgeo, bound);
            if (done) wbin = side == 3 ? 2 : 3;
            stop = r > 0.f && (double)(r * r) >= max_dist_sq;
        }
        if (!done && !stop) { brute_attempt(map, wx, wy, wz, lane, kw); wbin = 4; }
    } else {   // outside the voxel range (2^19 voxels from the map origin): no structure to lean on
        if (lane == 0) atomicAdd(&kf->fallback_queries, 1);
        brute_attempt(map, wx, wy, wz, lane, kw);
        wbin = 4;
    }
    return wbin;
}

// Exact 5-NN of the world point (qx, qy, qz): executed by the S lanes of a lane group (gl = lane in group); all 64
// lanes of the wavefront must be active (`live` = false marks padding lanes that only lend a hand).
// On return every lane of the group holds the same sorted keys k[]; src >= 0: bucket level the winners came
// from (key low word = position inside the bucket starting at bstart), src < 0: low word = point id.
// Ladder: bucket levels 0 and 1 per lane group, 64 / S scan points at a time; the few points still undecided are
// then taken one at a time by the WHOLE wavefront: level-2 bucket (~1000 candidates: 20 dependent load-sort steps
// for a lane group, 3 for a wavefront), level-3 block (the 216 level-2 voxel lists that tile it), finally every id.  A block of level l covers every point within r_l of the query (search_radius), so a level
// that is not accepted proves d5 >= r_l^2.  BOUNDED launches (the timed path) stop there as soon as
// r_l^2 >= MAX_DIST_PLANE^2: the reference discards such a match at Plane.cpp:40-43 whatever its 5 neighbours are,
// so the point is reported without neighbours (found = 0) and the update is unchanged; capturing launches (API
// parity: lv_iterate / lv_fetch_knn) always continue to the exact answer.
// clk != nullptr (DBG builds): phase-stamp slot of this workgroup.
template <int S, bool DBG, int K>
__device__ __forceinline__ void knn_search(const MapView& map, KfDev* __restrict__ kf, float qx, float qy, float qz, int gl,
                                           kkey (&k)[K], uint32_t& bstart, int& src, long long* clk, bool hist,
                                           bool live, Xyz* stage0, double max_dist_sq, uint32_t* s_pref, uint32_t* s_start,
                                           bool* undecided = nullptr, uint32_t* bound_out = nullptr) {
    // undecided != nullptr: stop after the bucket levels 0 / 1 and report whether the point is still open (the caller
    // then runs knn_coarse on it)
    if (undecided) *undecided = false;
    if (map.m == 0) return;
    const QGeom geo = make_geom(map, qx, qy, qz);
    const bool finite = __builtin_isfinite(qx) && __builtin_isfinite(qy) && __builtin_isfinite(qz);
    const bool in_range = finite && geo.amax < CELL_FAR;
    // a point with a NaN / infinite coordinate has no neighbours (k stays NONE, found = 0): the reference discards such
    // a match at Plane.cpp:42 whatever its tree search returned
    bool decided = !live || !finite;
    int hist_bin = -1;   // what decided (instrumentation): 0, 1 bucket level; 2, 3 voxel lists; 4 every id; 5 bounded stop
    uint32_t bound = 0x7FFFFFFFu;   // what the bucket levels that were not accepted proved (bucket_attempt): prunes the list levels
    if (live && in_range) {
#pragma unroll
#ifndef LV_DIAG_LEVELS
#define LV_DIAG_LEVELS 2
#endif
        for (int bl = 0; bl < LV_DIAG_LEVELS; ++bl) {   // level 0: the query's bucket; level 1: the region of its tile group (bucket_attempt)
            if (!decided) {
                decided = bucket_attempt<S>(map, bl, geo, qx, qy, qz, gl, k, bstart, (DBG && bl == 0) ? clk : nullptr,
                                            bl == 0 ? stage0 : nullptr, bl == 1 ? &bound : nullptr);   // (level 1's bound: its region holds the level-0 bucket's neighbourhood and more)
                if (decided) { src = bl; hist_bin = bl; }
            }
        }
    }
    if (undecided) {   // the caller shares the coarse levels out among the wavefronts of its workgroup (pass_kernel)
        *undecided = !decided;
        if (bound_out) *bound_out = bound;
        return;
    }
    const int lane = (int)(threadIdx.x & 63u);
    unsigned long long pending = __ballot(!decided && gl == 0);
    while (pending) {
        const int L = __ffsll((long long)pending) - 1;   // leader lane of the scan point served now
        pending &= pending - 1;
        const float wx = __shfl(qx, L), wy = __shfl(qy, L), wz = __shfl(qz, L);
        const uint32_t wbound = (uint32_t)__shfl((int)bound, L);
        kkey kw[K];
        const int wbin = knn_coarse<DBG>(map, kf, wx, wy, wz, lane, kw, max_dist_sq, s_pref, s_start, wbound);
        if (lane / S == L / S) {   // (keys carry point ids: src stays -1)
#pragma unroll
            for (int j = 0; j < K; ++j) k[j] = kw[j];
            hist_bin = wbin;
            decided = true;
        }
    }
    if (DBG && hist && live && finite && gl == 0) atomicAdd(&kf->level_hist[hist_bin >= 0 ? hist_bin : 5], 1);
}

// Plane fit + gates + Jacobian row of ONE scan point (one lane): Plane.cpp:19-55, Utils.cpp:32-66,
// Match.cpp:18-22, Localizator.cpp:36-56.  P / nidx / dbits: the 5 nearest map points in (distance, index)
// order; found < 0 marks a padding lane.  The row {J[0..W), h, valid} goes to srow (LDS).
// KEEP: the row stays in the caller's registers (keep[0 .. W + 2)) instead of going to LDS — the multi-round
// estimate_extrinsics pass stages a wavefront's 64 rows of 14 doubles in two halves (pass_kernel).
template <int W, bool EXT, bool DBG, int K, bool KEEP = false, bool FAST = false>
__device__ __forceinline__ void fit_row(const PoseConsts& pc, const MatchParams& prm, const DebugOut& dbg, int found,
                                        const float (&P)[K][3], const uint32_t (&nidx)[K], const uint32_t (&dbits)[K],
                                        float qx, float qy, float qz, uint32_t oq, double* srow, double* keep = nullptr) {
    double row[W];
#pragma unroll
    for (int i = 0; i < W; ++i) row[i] = 0.0;
    double hres = 0.0;
    bool chosen = false;
    float abcd[4] = {0.f, 0.f, 0.f, 0.f};
    float dist = 0.f;
    // the point in the LiDAR and IMU frames (Localizator.cpp:38-39): independent of the plane, so computed HERE, where the
    // scheduler can slot it between the dependent steps of the QR (behind the `chosen` gate it was a serial tail of its own)
    float plx, ply, plz, pix, piy, piz;
    rt_apply(pc.back, qx, qy, qz, plx, ply, plz);                         // :38
    rt_apply(pc.LI, plx, ply, plz, pix, piy, piz);                        // :39
    if (found >= K) {                                                   // Plane.cpp:36-38
        const float d5 = __uint_as_float(dbits[K - 1]);
        if ((double)d5 < prm.max_dist_plane_sq) {                         // Plane.cpp:40-43
            float A[K][3];
#pragma unroll
            for (int j = 0; j < K; ++j) { A[j][0] = P[j][0]; A[j][1] = P[j][1]; A[j][2] = P[j][2]; }
            float nv[3];
            plane_qr_solve<K, FAST>(A, nv);                               // Utils.cpp:47
            const float nrm = fit_sqrt<FAST>(dot3f(nv[0], nv[0], nv[1], nv[1], nv[2], nv[2]));  // Utils.cpp:50
            const float e0 = fit_div<FAST>(nv[0], nrm), e1 = fit_div<FAST>(nv[1], nrm), e2 = fit_div<FAST>(nv[2], nrm);
            const float e3 = FAST ? fit_div<true>(1.0f, nrm) : (float)(1.0 / (double)nrm);   // Utils.cpp:54 (exact path: f64 divide)
            bool ok = true;                                               // Utils.cpp:59-66
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const float res = e0 * P[j][0] + e1 * P[j][1] + e2 * P[j][2] + e3;
                if (fabsf(res) > prm.planes_threshold) ok = false;
            }
            if (ok) {
                chosen = true;
                abcd[0] = e0; abcd[1] = e1; abcd[2] = e2; abcd[3] = e3;
                dist = e0 * qx + e1 * qy + e2 * qz + e3;                  // Plane.cpp:27-29, Match.cpp:21
            }
        }
    }
    // ---- Localizator::calculate_H row (Localizator.cpp:36-56) ------------------------------
    if (chosen) {
        const double n0 = (double)abcd[0], n1 = (double)abcd[1], n2 = (double)abcd[2];
        const double* Ri = pc.R_inv;
        const double C0 = dot3d(Ri[0], n0, Ri[1], n1, Ri[2], n2);         // :47
        const double C1 = dot3d(Ri[3], n0, Ri[4], n1, Ri[5], n2);
        const double C2 = dot3d(Ri[6], n0, Ri[7], n1, Ri[8], n2);
        const double ix = (double)pix, iy = (double)piy, iz = (double)piz;
        row[0] = n0; row[1] = n1; row[2] = n2;                            // :51
        row[3] = iy * C2 - iz * C1;                                       // A = p_imu x C  :49
        row[4] = iz * C0 - ix * C2;
        row[5] = ix * C1 - iy * C0;
        if (EXT) {                                                        // :52
            const double* Li = pc.I_R_L_inv;
            const double t0 = dot3d(Li[0], C0, Li[1], C1, Li[2], C2);
            const double t1 = dot3d(Li[3], C0, Li[4], C1, Li[5], C2);
            const double t2 = dot3d(Li[6], C0, Li[7], C1, Li[8], C2);
            const double lx = (double)plx, ly = (double)ply, lz = (double)plz;
            row[W - 6] = ly * t2 - lz * t1;                               // B = p_lidar x (I_R_L_inv C)  :48
            row[W - 5] = lz * t0 - lx * t2;
            row[W - 4] = lx * t1 - ly * t0;
            row[W - 3] = C0; row[W - 2] = C1; row[W - 1] = C2;
        }
        hres = -(double)dist;                                             // :55
    }
    if constexpr (KEEP) {
#pragma unroll
        for (int j = 0; j < W; ++j) keep[j] = row[j];
        keep[W] = hres;
        keep[W + 1] = chosen ? 1.0 : 0.0;
    } else {
#pragma unroll
        for (int j = 0; j < W; ++j) srow[j] = row[j];
        srow[W] = hres;
        srow[W + 1] = chosen ? 1.0 : 0.0;
    }
    if (DBG && found >= 0) {
        if (dbg.knn_idx) {
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const bool have = j < found;
                dbg.knn_idx[(size_t)oq * K + j] = have ? nidx[j] : 0xFFFFFFFFu;
                dbg.knn_d2[(size_t)oq * K + j] = have ? __uint_as_float(dbits[j]) : __uint_as_float(0x7f800000u);
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
            for (int j = 0; j < 12; ++j) dbg.rows[(size_t)oq * 12 + j] = (j < W) ? row[j < W ? j : 0] : 0.0;
            dbg.h[oq] = hres;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// The pass runs as two kernels: the search and the fit have very different register needs (the search
// wants many waves in flight to hide its dependent loads, the QR fit wants ~100 VGPRs), so each gets its own
// kernel and register budget.  search_kernel hands one 128-byte record per scan point to fit_reduce_kernel
// through HBM/L2, slot-major (8 float4 planes of qstride entries) so both sides are coalesced:
//   slots 0-4  the 5 nearest map points {x, y, z, original index}   (fetched here, while L2-hot)
//   slot  5    {world x, y, z, original scan index}
//   slot  6    squared distances 0..3 (bits)      slot 7  {distance 4 (bits), found, -, -}
// General K (NUM_MATCH_POINTS 3..8, three-kernel pass only): slots 0..K-1 the neighbours, slot K the world point, then the K
// distance bits followed by `found`, four words to a slot — the layout above for K = 5.
constexpr int QREC_SLOTS = qrec_slots(KNN);
static_assert(QREC_SLOTS == 8, "the K = 5 record is 128 bytes");
#ifdef LV_SEARCH_WAVES
#define LV_SEARCH_BOUNDS __launch_bounds__(256, LV_SEARCH_WAVES)
#else
#define LV_SEARCH_BOUNDS __launch_bounds__(256)
#endif

// FIRST: the update starts here (no begin kernel): the pass constants come with the kernel arguments, and one extra
// workgroup (the last) installs the state, covariance, constants and loop counters in kf / the mailbox for the kernels
// that follow.
template <int S, bool DBG, bool FIRST, int K = KNN>
__global__ LV_SEARCH_BOUNDS void search_kernel(MapView map, const float4* __restrict__ scan, uint32_t n,
                                                     KfDev* __restrict__ kf, float4* __restrict__ qrec, uint32_t qstride,
                                                     const uint32_t* __restrict__ tile_order, uint32_t n_tiles, double max_dist_sq,
                                                     DebugOut dbg, BeginArg begin, KfHostIO* io) {
    constexpr int GS = 256 / S;
    constexpr int STAGE = S * 8;   // candidates of a lane group's first level-0 chunk, kept in LDS by position
    __shared__ Xyz s_stage[GS][STAGE];
    __shared__ uint32_t s_pref[4][64], s_start[4][64];   // wavefront-cooperative levels: per-wave prefix sums / list starts
    if (FIRST) {
        if (blockIdx.x == gridDim.x - 1u) {   // the installing workgroup
            for (int i = threadIdx.x; i < NS * NS; i += 256) {
                const double p = begin.P[i];
                kf->P_prop[i] = p;
                kf->P_post[i] = p;
                io->P_post[i] = p;   // an update without a terminal pass returns the propagated covariance
            }
            if (threadIdx.x < NX) {
                const double v = begin.x[threadIdx.x];
                kf->x[threadIdx.x] = v;
                kf->x_prop[threadIdx.x] = v;
                io->x[threadIdx.x] = v;
            }
            constexpr int NW32 = (int)(sizeof(PoseConsts) / 4);
            if (threadIdx.x < NW32) reinterpret_cast<uint32_t*>(&kf->pose)[threadIdx.x] = reinterpret_cast<const uint32_t*>(&begin.pose)[threadIdx.x];
            if (threadIdx.x == 0) {
                io->passes = 0;
                kf->t = 0;
                kf->iter = -1;  // upstream loop starts at i = -1
                kf->done = 0;
                kf->passes = 0;
            }
            return;
        }
    } else {
        if (kf->done) return;
    }
    const int tid = threadIdx.x;
    const int gq = tid / S, gl = tid % S;
    // tile order: farthest-from-sensor tiles first (ScanStore::order_tiles).  An XCD-aware order (contiguous
    // Morton runs per XCD) measured no different from plain round-robin here.
    const uint32_t vb = (tile_order && blockIdx.x < n_tiles) ? tile_order[blockIdx.x] : blockIdx.x;
    const uint32_t q = vb * (uint32_t)GS + (uint32_t)gq;
    long long* stamp_slot = (DBG && dbg.clk && tid == 0 && blockIdx.x < (uint32_t)dbg.clk_blocks) ? dbg.clk + (size_t)blockIdx.x * 8 : nullptr;
    if (DBG && stamp_slot) { stamp_slot[0] = clock64(); dbg.clk[(size_t)(dbg.clk_blocks + blockIdx.x) * 8 + 0] = wall_clock64(); }
    if (n == 0) return;
    // no lane leaves early: the coarse levels are searched by whole wavefronts (knn_search);
    // padding lanes (q >= n) carry a copy of the last point, lend a hand and store nothing
    const bool live = q < n;
    kkey k[K];
#pragma unroll
    for (int j = 0; j < K; ++j) k[j] = none_key();
    uint32_t bstart = 0;
    int src = -1;
    const float4 sp = scan[live ? q : n - 1];
    float qx, qy, qz;
    rt_apply(FIRST ? begin.pose.Tc : kf->pose.Tc, sp.x, sp.y, sp.z, qx, qy, qz);  // Mapper.cpp:51
    if (DBG && stamp_slot) { asm volatile("" :: "v"(qx), "v"(qy), "v"(qz)); stamp_slot[1] = clock64(); }
    knn_search<S, DBG>(map, kf, qx, qy, qz, gl, k, bstart, src, stamp_slot, DBG && !dbg.clk, live, s_stage[gq], max_dist_sq,
                       s_pref[tid >> 6], s_start[tid >> 6]);
    if (DBG && stamp_slot) { asm volatile("" :: "v"(k[0]), "v"(k[4])); stamp_slot[3] = clock64(); }
    int found = 0;
#pragma unroll
    for (int j = 0; j < K; ++j) found += key_real(k[j]) ? 1 : 0;
    if (live) {
#pragma unroll
        for (int slot0 = 0; slot0 < qrec_slots(K); slot0 += S) {
            const int slot = slot0 + gl;
            if (slot >= qrec_slots(K)) break;
            float4 v;
            if (slot < K) {
                kkey kk = k[0];
#pragma unroll
                for (int j = 1; j < K; ++j) kk = (slot == j) ? k[j] : kk;
                v = make_float4(0.f, 0.f, 0.f, __uint_as_float(0xFFFFFFFFu));
                if (key_real(kk)) {
                    const uint32_t pos = key_lo(kk);
                    if (src == 0 && pos < (uint32_t)STAGE && !DBG) {
                        // decided at level 0 inside its first chunk (the common case): the point is still in LDS
                        // (same wavefront wrote it, LDS operations of a wavefront execute in order); its id is
                        // only reported by capturing (DBG) launches
                        const Xyz w = s_stage[gq][pos];
                        v = make_float4(w.x, w.y, w.z, __uint_as_float(0xFFFFFFFFu));
                    } else if (src >= 0) {
                        const Xyz w = reinterpret_cast<const Xyz*>(map.bxyz[0])[(size_t)bstart + pos];
                        v = make_float4(w.x, w.y, w.z, __uint_as_float(map.bidx[0][(size_t)bstart + pos]));
                    } else {
                        v = map.orig[pos];
                        v.w = __uint_as_float(pos);
                    }
                }
            } else if (slot == K) {
                v = make_float4(qx, qy, qz, sp.w);
            } else if constexpr (K == KNN) {
                if (slot == 6) {
                    v = make_float4(__uint_as_float(key_hi(k[0])), __uint_as_float(key_hi(k[1])), __uint_as_float(key_hi(k[2])),
                                    __uint_as_float(key_hi(k[3])));
                } else {
                    v = make_float4(__uint_as_float(key_hi(k[4])), __int_as_float(found), 0.f, 0.f);
                }
            } else {   // word w of the distance slots: distance bits of neighbour w, then `found`
                float wv[4];
#pragma unroll
                for (int cw = 0; cw < 4; ++cw) {
                    const int w = (slot - K - 1) * 4 + cw;
                    float val = w == K ? __int_as_float(found) : 0.f;
#pragma unroll
                    for (int j = 0; j < K; ++j) val = (w == j) ? __uint_as_float(key_hi(k[j])) : val;
                    wv[cw] = val;
                }
                v = make_float4(wv[0], wv[1], wv[2], wv[3]);
            }
            qrec[(size_t)slot * qstride + q] = v;
        }
    }
    if (DBG && stamp_slot) { stamp_slot[4] = clock64(); }
    if (DBG && stamp_slot) { dbg.clk[(size_t)(dbg.clk_blocks + blockIdx.x) * 8 + 1] = wall_clock64(); }
}

// One lane per scan point: fit + row (fit_row); each of the workgroup's 4 wavefronts then contracts its own 64
// staged rows in f64, and the 4 wave sums are combined in a fixed order into the block partial: FIT_POINTS
// scan points per workgroup iteration, so a 64k-point scan leaves 256 partials that solve_kernel folds itself.
constexpr int FIT_POINTS = 256;
template <bool EXT, bool DBG, int K = KNN>
__global__ __launch_bounds__(FIT_POINTS) void fit_reduce_kernel(const float4* __restrict__ qrec, uint32_t qstride, uint32_t n,
                                                                KfDev* __restrict__ kf, MatchParams prm,
                                                                double* __restrict__ partials, DebugOut dbg) {
    constexpr int G = FIT_POINTS;
    constexpr int NWAVE = G / 64;
    constexpr int W = EXT ? 12 : 6;
    constexpr int ROW_W = W + 2;
    constexpr int NOUT = W * (W + 1) / 2 + W + 2;
    constexpr int NACC = (NOUT + 63) / 64;
    __shared__ double s_rows[G][ROW_W];
    __shared__ double s_out[NWAVE][2][SUMS_LEN];
    if (kf->done) return;
    if (blockIdx.x == 0) {   // rides along: the half of the coming solve that does not need this pass' record
        solve_prep<W, FIT_POINTS>(kf, prm.R_inv);
        return;
    }
    const uint32_t bid = blockIdx.x - 1u, nfit = gridDim.x - 1u;   // plane-fitting workgroups
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const PoseConsts& pc = kf->pose;
    int oa[NACC], ob[NACC], orec[NACC];
    double acc[NACC];
    // NOUT <= 32 (no extrinsics): the two half-wavefronts each contract 32 of the wavefront's 64 rows
    constexpr bool HALVES = NOUT <= 32;
    const int olane = HALVES ? (lane & 31) : lane;       // output owned by this lane
    const int prow0 = HALVES ? (lane >> 5) * 32 : 0;     // first row of its share
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
        oa[a] = ob[a] = orec[a] = 0;
        acc[a] = 0.0;
        if (olane + a * 64 < NOUT) out_pair<W>(olane + a * 64, oa[a], ob[a], orec[a]);
    }
    for (int t = tid; t < 2 * NWAVE * SUMS_LEN; t += G) (&s_out[0][0][0])[t] = 0.0;
    const uint32_t vb = (bid % 8u) * (nfit / 8u) + bid / 8u;
    const uint32_t per_iter = (uint32_t)G * nfit;
    const uint32_t iters = (n + per_iter - 1) / per_iter;
    long long* stamp_slot = (DBG && dbg.clk && tid == 0 && bid < (uint32_t)dbg.clk_blocks) ? dbg.clk + (size_t)bid * 8 : nullptr;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t q = (it * nfit + vb) * (uint32_t)G + (uint32_t)tid;
        if (DBG && stamp_slot && it == 0) stamp_slot[5] = clock64();
        float P[K][3];
        uint32_t nidx[K], dbits[K];
        float qx = 0.f, qy = 0.f, qz = 0.f;
        uint32_t oq = 0;
        int found = -1;
#pragma unroll
        for (int j = 0; j < K; ++j) { P[j][0] = P[j][1] = P[j][2] = 0.f; nidx[j] = 0xFFFFFFFFu; dbits[j] = 0x7f800000u; }
        if (q < n) {
            float4 r[qrec_slots(K)];
#pragma unroll
            for (int sl = 0; sl < qrec_slots(K); ++sl) r[sl] = qrec[(size_t)sl * qstride + q];
#pragma unroll
            for (int j = 0; j < K; ++j) { P[j][0] = r[j].x; P[j][1] = r[j].y; P[j][2] = r[j].z; nidx[j] = __float_as_uint(r[j].w); }
            qx = r[K].x; qy = r[K].y; qz = r[K].z; oq = __float_as_uint(r[K].w);
            const auto dword = [&](int w) { const float4 v = r[K + 1 + w / 4]; return (w & 3) == 0 ? v.x : (w & 3) == 1 ? v.y : (w & 3) == 2 ? v.z : v.w; };
#pragma unroll
            for (int j = 0; j < K; ++j) dbits[j] = __float_as_uint(dword(j));
            found = __float_as_int(dword(K));
        }
        fit_row<W, EXT, DBG>(pc, prm, dbg, found, P, nidx, dbits, qx, qy, qz, oq, s_rows[tid]);
        __syncthreads();
        if (DBG && stamp_slot && it == 0) stamp_slot[6] = clock64();
        const double (*rows)[ROW_W] = s_rows + wave * 64;
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            if (olane + a * 64 < NOUT) {
                double sacc = acc[a];
#pragma unroll 8
                for (int p = 0; p < (HALVES ? 32 : 64); ++p) sacc += rows[prow0 + p][oa[a]] * rows[prow0 + p][ob[a]];
                acc[a] = sacc;
            }
        }
        __syncthreads();
        if (DBG && stamp_slot && it == 0) stamp_slot[7] = clock64();
    }
#pragma unroll
    for (int a = 0; a < NACC; ++a)
        if (olane + a * 64 < NOUT) s_out[wave][HALVES ? (lane >> 5) : 0][orec[a]] = acc[a];
    __syncthreads();
    if (tid < SUMS_LEN) {   // fixed order: (half 0 + half 1) of wave 0, then the other waves likewise
        double s = s_out[0][0][tid] + s_out[0][1][tid];
#pragma unroll
        for (int w = 1; w < NWAVE; ++w) s += s_out[w][0][tid] + s_out[w][1][tid];
        partials[(size_t)bid * SUMS_LEN + tid] = s;
    }
}

// workgroups of fit_reduce_kernel that fit planes (one more is launched for solve_prep)
int fit_grid_size(uint32_t n, int max_blocks) {
    uint32_t need = (n + (uint32_t)FIT_POINTS - 1) / (uint32_t)FIT_POINTS;
    if (need < 1) need = 1;
    uint32_t grid = need < (uint32_t)max_blocks ? need : (uint32_t)max_blocks;
    grid = (grid + 7u) & ~7u;  // multiple of 8 (XCD-aware tile order)
    return (int)grid;
}

template <int S, int K = KNN>
static void launch_search(hipStream_t stream, bool dbg_on, const MapView& map, const float4* scan, uint32_t n, KfDev* kf,
                          float4* qrec, uint32_t qstride, const uint32_t* tile_order, uint32_t n_tiles, double max_dist_sq,
                          const DebugOut& dbg, const BeginArg* begin, KfHostIO* io) {
    constexpr uint32_t GS = 256 / S;
    uint32_t grid = (n + GS - 1) / GS;
    grid = (grid + 7u) & ~7u;
    if (grid == 0) grid = 8;
    if (begin) {   // first launch of an update: one more workgroup installs the state
        if (dbg_on) hipLaunchKernelGGL((search_kernel<S, true, true, K>), dim3(grid + 1), dim3(256), 0, stream, map, scan, n, kf, qrec, qstride, tile_order, n_tiles, max_dist_sq, dbg, *begin, io);
        else hipLaunchKernelGGL((search_kernel<S, false, true, K>), dim3(grid + 1), dim3(256), 0, stream, map, scan, n, kf, qrec, qstride, tile_order, n_tiles, max_dist_sq, dbg, *begin, io);
        return;
    }
    static const BeginArg none{};
    if (dbg_on) hipLaunchKernelGGL((search_kernel<S, true, false, K>), dim3(grid), dim3(256), 0, stream, map, scan, n, kf, qrec, qstride, tile_order, n_tiles, max_dist_sq, dbg, none, io);
    else hipLaunchKernelGGL((search_kernel<S, false, false, K>), dim3(grid), dim3(256), 0, stream, map, scan, n, kf, qrec, qstride, tile_order, n_tiles, max_dist_sq, dbg, none, io);
}

static bool debug_requested(const DebugOut& dbg) {
    return dbg.knn_idx || dbg.valid || dbg.p_world || dbg.abcd || dbg.dist || dbg.rows || dbg.clk;
}

// split form, kernel 1: exact 5-NN of every scan point -> qrec (max_dist_sq: MAX_DIST_PLANE^2, the radius beyond
// which a non-capturing launch may stop — see knn_search)
int launch_search(hipStream_t stream, int S, const MapView& map, const float4* scan_sorted, uint32_t n, KfDev* kf, float4* qrec,
                  uint32_t qstride, const uint32_t* tile_order, uint32_t n_tiles, double max_dist_sq, const DebugOut& dbg,
                  const BeginArg* begin, KfHostIO* io, int num_match) {
    const bool dbg_on = debug_requested(dbg);
    if (num_match != KNN) {   // NUM_MATCH_POINTS other than 5: the general-K build, eight lanes per point
#define LV_K(K_) case K_: launch_search<8, K_>(stream, dbg_on, map, scan_sorted, n, kf, qrec, qstride, tile_order, n_tiles, max_dist_sq, dbg, begin, io); break
        switch (num_match) {
            LV_K(3); LV_K(4); LV_K(6); LV_K(7); LV_K(8);
            default: set_error("NUM_MATCH_POINTS must be 3..8 (got %d)", num_match); return LV_EINVAL;
        }
#undef LV_K
        LV_HIP(hipGetLastError());
        return LV_OK;
    }
    switch (S) {
        case 1: launch_search<1>(stream, dbg_on, map, scan_sorted, n, kf, qrec, qstride, tile_order, n_tiles, max_dist_sq, dbg, begin, io); break;
        case 2: launch_search<2>(stream, dbg_on, map, scan_sorted, n, kf, qrec, qstride, tile_order, n_tiles, max_dist_sq, dbg, begin, io); break;
        case 4: launch_search<4>(stream, dbg_on, map, scan_sorted, n, kf, qrec, qstride, tile_order, n_tiles, max_dist_sq, dbg, begin, io); break;
        case 8: launch_search<8>(stream, dbg_on, map, scan_sorted, n, kf, qrec, qstride, tile_order, n_tiles, max_dist_sq, dbg, begin, io); break;
        case 16: launch_search<16>(stream, dbg_on, map, scan_sorted, n, kf, qrec, qstride, tile_order, n_tiles, max_dist_sq, dbg, begin, io); break;
        default: set_error("lanes_per_query must be 1,2,4,8 or 16 (got %d)", S); return LV_EINVAL;
    }
    LV_HIP(hipGetLastError());
    return LV_OK;
}

// split form, kernel 2: qrec -> plane fits, Jacobian rows, `grid` block partials
int launch_fit_reduce(hipStream_t stream, const float4* qrec, uint32_t qstride, uint32_t n, KfDev* kf, const MatchParams& prm,
                      double* partials, int grid, const DebugOut& dbg, int num_match) {
    const bool dbg_on = debug_requested(dbg);
    const bool ext = prm.estimate_extrinsics != 0;
#define LV_LAUNCH(EXT_, DBG_, K_) \
    hipLaunchKernelGGL((fit_reduce_kernel<EXT_, DBG_, K_>), dim3(grid + 1), dim3(FIT_POINTS), 0, stream, qrec, qstride, n, kf, prm, partials, dbg)
#define LV_LAUNCH_K(K_) \
    do { if (ext) { if (dbg_on) LV_LAUNCH(true, true, K_); else LV_LAUNCH(true, false, K_); } \
         else { if (dbg_on) LV_LAUNCH(false, true, K_); else LV_LAUNCH(false, false, K_); } } while (0)
    switch (num_match) {
        case 3: LV_LAUNCH_K(3); break;
        case 4: LV_LAUNCH_K(4); break;
        case KNN: LV_LAUNCH_K(KNN); break;
        case 6: LV_LAUNCH_K(6); break;
        case 7: LV_LAUNCH_K(7); break;
        case 8: LV_LAUNCH_K(8); break;
        default: set_error("NUM_MATCH_POINTS must be 3..8 (got %d)", num_match); return LV_EINVAL;
    }
#undef LV_LAUNCH_K
#undef LV_LAUNCH
    LV_HIP(hipGetLastError());
    return LV_OK;
}


// ------------------------------------------------------------------------------------------------------
// ONE launch per measurement pass (lv_pass_dev.hpp): every workgroup of 1024 threads (one per CU)
//   1. prologue: re-derives the state and the f32 constants of this pass — the solve of the PREVIOUS pass from the
//      workgroup partials that launch left behind (mode 1), or takes them from the kernel arguments (mode 0: the update
//      starts here) / from kf (mode 2: a begin kernel installed a device-resident state);
//   2. per round searches 8 tiles of 32 scan points (two steps of its four 256-thread units; S = 8 lanes per point,
//      the same knn_search as search_kernel) and keeps the 128-byte hand-over records in LDS instead of HBM;
//   3. four of its wavefronts (64 points each, one lane per point) run the plane fits, rows and the contraction of
//      fit_reduce_kernel on those records;
//   4. writes one compact partial (PassDims::OW doubles) for the next launch's prologue.
// The grid is sized so that every workgroup is resident at once (one per CU): there are no residency rounds whose
// workgroups would each pay the prologue, and nothing in the kernel waits for another workgroup.
// The last workgroup of the grid (its tiles are the nearest ones = the lightest search share) also keeps the books
// after its own search and fits (bookkeeping / prepare_next / terminal pass): that work hides behind the heavier
// workgroups' searches.  A launch with rounds = 0 and one workgroup is the closing launch of an update (the solve of
// its last pass, nothing to search).
struct PassArgs {
    MapView map;
    const float4* scan;
    const uint32_t* tile_order;   // 32-point tiles, farthest first (or nullptr)
    uint32_t n, n_tiles32;
    KfDev* kf;
    KfHostIO* io;
    const double* recs_in;        // compact partials of the previous launch (mode 1)
    double* part_out;             // compact partials of this launch
    double* sums_out;             // optional: the folded 96-double record of the previous pass
    float4* qrec;                 // optional (lv_set_record_dump): the hand-over records also go to memory (lv_fetch_neighbors)
    long long* clk;               // optional (instrumentation): 16 shader-clock + 16 wall-clock stamps per workgroup
    uint32_t qstride;
    int nrec, mode, rounds, launch;   // launch: index of this launch in the update (parity selects KfDev::ps)
    const uint32_t* cost_in;          // per searching workgroup: wall-clock ticks its search + fits took in the previous launch (or nullptr)
    uint32_t* cost_out;               // ... in this launch
    int steps;                        // search steps per round: 2, or 1 for scans small enough to spread over the CUs in one step
    uint32_t nsearch;                 // searching workgroups: the grid, or the grid minus a dedicated bookkeeping workgroup
    MatchParams mp;
    SolveParams sp;
};
constexpr int PK_UNITS = PK_THREADS / 256;        // 256-thread units: one 32-point tile each per step
constexpr int PK_GROUPS = PK_THREADS / 8;         // lane groups (= scan points in flight) per workgroup: 128
constexpr int PK_STAGE = 64;                      // candidates of a lane group's first level-0 chunk kept in LDS
constexpr int PK_STEPS = 2048 / PK_THREADS;       // search steps per round: 2 (256 scan points per workgroup and round either way)
constexpr int PK_FITW = PK_STEPS * PK_GROUPS / 64;   // fit wavefronts: one per 64 points of a round (4)
constexpr int PK_RPTS_MAX = PK_STEPS * PK_GROUPS;    // scan points per workgroup and round (256)
constexpr int PK_CLK = 32;                        // stamp words per workgroup
constexpr size_t PK_REGION0 = sizeof(Xyz) * PK_GROUPS * PK_STAGE < 65536 ? 65536 : sizeof(Xyz) * PK_GROUPS * PK_STAGE;   // 98304: stage | solve scratch | rows
constexpr size_t PK_OFF_REC = PK_REGION0;                                             // float4 [steps][8 slots][128]
constexpr size_t PK_OFF_PREF = PK_OFF_REC + sizeof(float4) * PK_STEPS * QREC_SLOTS * PK_GROUPS;
constexpr size_t PK_OFF_POSE = PK_OFF_PREF + sizeof(uint32_t) * 2 * (PK_THREADS / 64) * 64;
constexpr size_t PK_OFF_OUT = PK_OFF_POSE + ((sizeof(PoseConsts) + 15) / 16) * 16;
constexpr size_t PK_OFF_KEEP = PK_OFF_OUT + sizeof(double) * PK_FITW * 2 * SUMS_LEN;
constexpr size_t PK_OFF_QN = PK_OFF_KEEP + ((sizeof(KeepLds) + 15) / 16) * 16;
constexpr size_t PK_OFF_QUEUE = PK_OFF_QN + 32;                              // float4 [256] + uint32 [256]: points left to the coarse levels
constexpr size_t PK_OFF_QUEUEQ = PK_OFF_QUEUE + sizeof(float4) * PK_RPTS_MAX;
constexpr size_t PK_OFF_QUEUEB = PK_OFF_QUEUEQ + sizeof(uint32_t) * PK_RPTS_MAX;   // uint32 [256]: what the bucket levels proved about each queued point
constexpr size_t PK_LDS_BYTES = PK_OFF_QUEUEB + sizeof(uint32_t) * PK_RPTS_MAX;
constexpr size_t PK_OFF_BOOK = 32 * 1024;   // the books' scratch inside region 0: above the solve scratch and above the staged rows
static_assert(sizeof(SolveLds) <= PK_OFF_BOOK && PK_OFF_BOOK + sizeof(BookLds) <= PK_REGION0, "solve / books scratch must fit under the stage");
static_assert(sizeof(double) * PK_FITW * 64 * 14 <= PK_OFF_BOOK, "staged rows must stay below the books' scratch");
static_assert(PK_LDS_BYTES <= 160 * 1024, "one workgroup per CU");

// The bookkeeping workgroup's extra work in a searching launch, by T threads that synchronise through bar (tid = 0 .. T - 1):
// mode 1: the books of the pass the prologue solved (K) and the preparation of the next solve; mode 0 / 2: the update
// starts in this launch: install the state (mode 0: from the kernel arguments; mode 2: a begin kernel did) and prepare
// the first solve.
template <int W, int T, class Bar>
__device__ __forceinline__ void keeper_books(const PassArgs& a, const BeginArg& begin, KeepLds& K, BookLds& Bk, KfDev* __restrict__ kf,
                                             const KfDev::PassState* __restrict__ ps_in, KfDev::PassState* __restrict__ ps_out,
                                             const PoseConsts* pose, int tid, Bar& bar, long long* clk) {
    if (a.mode == 1) {
        bookkeeping<W, T>(K, Bk, kf, ps_in, ps_out, a.io, a.sums_out, a.sp.R_inv, a.sp.seq, pose, K.Pprop, K.xp, false, tid, bar, clk);
        return;
    }
    constexpr int NW32 = (int)(sizeof(PoseConsts) / 4);
    for (int i = tid; i < NS * NS; i += T) {
        double p;
        if (a.mode == 0) {
            p = begin.P[i];
            kf->P_prop[i] = p;
            kf->P_post[i] = p;
            a.io->P_post[i] = p;   // an update without a terminal pass returns the propagated covariance
        } else {
            p = kf->P_prop[i];
        }
        Bk.B[i / NS][i % NS] = p;
    }
    if (tid < NX) {
        double v;
        if (a.mode == 0) {
            v = begin.x[tid];
            kf->x[tid] = v;
            kf->x_prop[tid] = v;
            a.io->x[tid] = v;
        } else {
            v = kf->x[tid];
        }
        ps_out->x[tid] = v;
        K.x[tid] = v;
        K.xp[tid] = a.mode == 0 ? v : kf->x_prop[tid];
    }
    if (a.mode == 0 && tid >= 128 && tid < 128 + NW32)
        reinterpret_cast<uint32_t*>(&kf->pose)[tid - 128] = reinterpret_cast<const uint32_t*>(&begin.pose)[tid - 128];
    if (tid == 0) {
        if (a.mode == 0) {
            a.io->passes = 0;
            kf->t = 0;
            kf->iter = -1;  // upstream loop starts at i = -1
            kf->done = 0;
            kf->passes = 0;
        }
        ps_out->t = 0;
        ps_out->iter = -1;
        ps_out->passes = 0;
    }
    bar();   // K.x / K.xp are read by other wavefronts next
    prepare_next<W, T>(Bk, ps_out, K.x, K.xp, a.sp.R_inv, tid, bar, clk);
}

constexpr int PK_BOOKW = PK_THREADS / 64 - PK_FITW;   // wavefronts of a workgroup that do not fit planes (12)

// CLOSING: the closing launch of an update as its own (search-free) kernel: the solve of the last pass and the terminal
// books by one workgroup.
// MULTI: the instantiation for scans of more than one round per workgroup (round 4): a round's plane fits run BESIDE the next
// round's search instead of between two barriers.  Scans of one round — the headline — keep the instantiation without it: the same
// source compiled with the overlap logic in place fitted planes 0.4 us slower per launch (register allocation / loop peeling).
template <bool EXT, bool CLOSING, bool MULTI = false, bool FAST = false>
__global__ __launch_bounds__(PK_THREADS, PK_THREADS / 256) void pass_kernel(PassArgs a, BeginArg begin) {
    constexpr int S = 8;
    constexpr int W = EXT ? 12 : 6;
    constexpr int ROW_W = W + 2;
    constexpr int NOUT = PassDims<W>::NOUT, OW = PassDims<W>::OW;
    constexpr int NACC = (NOUT + 63) / 64;
    constexpr bool HALVES = NOUT <= 32;
    __shared__ __attribute__((aligned(16))) unsigned char smem[PK_LDS_BYTES];
    SolveLds& L = *reinterpret_cast<SolveLds*>(smem);
    BookLds& Bk = *reinterpret_cast<BookLds*>(smem + PK_OFF_BOOK);
    Xyz (*s_stage)[PK_STAGE] = reinterpret_cast<Xyz (*)[PK_STAGE]>(smem);
    // staged Jacobian rows of a fit wavefront.  Multi-round scans fit a round's planes BESIDE the next round's search (below), so
    // there a fit wavefront's rows must lie inside its OWN candidate-stage area (wavefront w: [w * 6 KB, w * 6 KB + 4 KB)) — packed
    // at a 4 KB stride, wavefront 1's rows would overlap wavefront 0's stage, which wavefront 0 refills as soon as it rejoins the
    // search.  The single-round / EXT instantiations fit between two barriers and keep the packed layout.
    // With estimate_extrinsics a wavefront's 64 rows of 14 doubles are 7 KB — more than its 6 KB of stage: the multi-round EXT
    // instantiation (round 5) stages them in TWO HALVES of 32 rows (3.5 KB), the rows waiting in registers (fit_row<.., KEEP>);
    // the contraction runs over rows 0..31, then 32..63: the same order, the same bits as the one-piece form.
    constexpr bool TWO_PHASE = MULTI && EXT;
    constexpr int ROWS_STAGED = TWO_PHASE ? 32 : 64;
    constexpr size_t ROWS_STRIDE = MULTI ? sizeof(Xyz) * 8 * PK_STAGE : sizeof(double) * 64 * ROW_W;
    static_assert(sizeof(double) * ROWS_STAGED * ROW_W <= ROWS_STRIDE && ROWS_STRIDE * PK_FITW <= PK_OFF_BOOK, "staged rows: inside the stride, below the books");
    float4* s_rec = reinterpret_cast<float4*>(smem + PK_OFF_REC);
    uint32_t (*s_pref)[64] = reinterpret_cast<uint32_t (*)[64]>(smem + PK_OFF_PREF);
    uint32_t (*s_start)[64] = s_pref + PK_THREADS / 64;
    PoseConsts& s_pose = *reinterpret_cast<PoseConsts*>(smem + PK_OFF_POSE);
    double (*s_out)[2][SUMS_LEN] = reinterpret_cast<double (*)[2][SUMS_LEN]>(smem + PK_OFF_OUT);

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const uint32_t nwg = a.nsearch, bid = blockIdx.x;
    // the bookkeeping workgroup: a dedicated one (scans that leave a CU free: its books run beside the others' searches);
    // otherwise a searching one, after its own fits: the one that was quickest in the previous launch (its books then
    // hide behind the slower workgroups' searches), without a history the last one (nearest = cheapest tiles)
    const bool dedicated = gridDim.x > nwg;
    bool keeper = bid == gridDim.x - 1u;
    const long long t_begin = wall_clock64();
    KfDev* __restrict__ kf = a.kf;
    const KfDev::PassState* __restrict__ ps_in = &kf->ps[a.launch & 1];
    KfDev::PassState* __restrict__ ps_out = &kf->ps[(a.launch + 1) & 1];
    KeepLds& K = *reinterpret_cast<KeepLds*>(smem + PK_OFF_KEEP);
    int* s_qn = reinterpret_cast<int*>(smem + PK_OFF_QN);   // entries in the queue of points left to the coarse levels
    float4* s_queue = reinterpret_cast<float4*>(smem + PK_OFF_QUEUE);
    uint32_t* s_queueq = reinterpret_cast<uint32_t*>(smem + PK_OFF_QUEUEQ);
    uint32_t* s_queueb = reinterpret_cast<uint32_t*>(smem + PK_OFF_QUEUEB);
    // [0] / [3] queue entries and [1] / [4] search-task counter of the even / odd rounds, [2] the books' sub-barrier, [5] fit
    // wavefronts that have taken their records out of the buffer (multi-round scans) — the prologue's barriers publish them
    if (threadIdx.x < 8) s_qn[threadIdx.x] = 0;
    bool books_done = false;
    constexpr int NW32 = (int)(sizeof(PoseConsts) / 4);
    long long* clk = a.clk ? a.clk + (size_t)bid * PK_CLK : nullptr;
#define PK_STAMP(i, cond) do { if (clk && (cond)) { clk[i] = clock64(); clk[16 + (i)] = wall_clock64(); } } while (0)
    PK_STAMP(0, tid == 0);
    if (clk && tid == 0) clk[16 + 10] = 0;   // (stamp 10 marks the launch's bookkeeping workgroup)

    // ---- 1. prologue ------------------------------------------------------------------------------------
    const bool searching = !CLOSING && a.rounds > 0 && bid < nwg;
    if (a.mode == 1) {
        if (CLOSING) {   // the terminal pass: what it needs besides the solve, fetched while the prologue's own loads are in flight
            for (int e = tid; e < NS * NS; e += PK_THREADS) Bk.P[e / NS][e % NS] = ps_in->prep_P[e];
            if (tid < NX) Bk.xp[tid] = kf->x_prop[tid];
            set_identity<PK_THREADS>(Bk.J, tid);   // (the projection blocks are filled in by solve_core, beside the boxplus)
        }
        const bool choose = !CLOSING && !dedicated && a.cost_in != nullptr;
        if (!solve_core<W, !CLOSING>(L, kf, ps_in, a.recs_in, a.nrec, a.sp, &s_pose, tid, clk, choose ? a.cost_in : nullptr, choose ? (int)nwg : 0,
                                     CLOSING ? &Bk : nullptr))
            return;   // the update ended in an earlier launch
        if (choose && L.cheapest >= 0) keeper = (int)bid == L.cheapest;
        if (keeper) {   // region 0 is about to become the candidate stage: remember what the books need
            keep_solve(K, L, tid);
            if (!CLOSING) {   // (the propagated covariance / state arrive while this workgroup searches)
                for (int e = tid; e < NS * NS; e += PK_THREADS) K.Pprop[e] = kf->P_prop[e];
                if (tid < NX) K.xp[tid] = kf->x_prop[tid];
            }
        }
        const int ended = L.last;
        __syncthreads();
        if (ended || !searching) {   // that solve ended the update (or this is the closing launch): nothing to search
            if (!keeper) return;
            // the books by the whole workgroup
            WgBar bar;
            const bool closing = a.rounds == 0;
            bookkeeping<W, PK_THREADS>(K, Bk, kf, ps_in, ps_out, a.io, a.sums_out, a.sp.R_inv, a.sp.seq, &s_pose, closing ? kf->P_prop : K.Pprop,
                                       closing ? kf->x_prop : K.xp, closing, tid, bar, clk, CLOSING);
            PK_STAMP(10, tid == 0);
            return;
        }
    } else {
        if (a.mode == 2 && kf->done) return;
        if (tid < NW32)
            reinterpret_cast<uint32_t*>(&s_pose)[tid] =
                a.mode == 0 ? reinterpret_cast<const uint32_t*>(&begin.pose)[tid] : reinterpret_cast<const uint32_t*>(&kf->pose)[tid];
        __syncthreads();
    }
    // (from here on region 0 is the candidate stage / the staged rows; s_pose and K stay)
    PK_STAMP(3, tid == 0);

    if constexpr (!CLOSING) {
    if (searching) {
    // ---- 2. / 3. search and fit rounds ------------------------------------------------------------------
    const int gq = tid / S, gl = tid % S;           // lane group (0..127) = position of its point among the 128 of a step; lane in group
    // fit wavefronts: wavefront f < PK_FITW takes the 64 points [fbase, fbase + 64) of step fstep (wavefronts 0..3 of a
    // workgroup sit on the four SIMDs of the CU)
    const int fstep = wave / (PK_GROUPS / 64), fbase = (wave % (PK_GROUPS / 64)) * 64;
    const bool fitter = wave < PK_FITW && fstep < a.steps;
    int oa[NACC], ob[NACC];
    double acc[NACC];
    const int olane = HALVES ? (lane & 31) : lane;       // output owned by this lane
    const int prow0 = HALVES ? (lane >> 5) * 32 : 0;     // first row of its share
#pragma unroll
    for (int c = 0; c < NACC; ++c) {
        int orec_unused = 0;
        oa[c] = ob[c] = 0;
        acc[c] = 0.0;
        if (olane + c * 64 < NOUT) out_pair<W>(olane + c * 64, oa[c], ob[c], orec_unused);
    }
    for (int t = tid; t < PK_FITW * 2 * SUMS_LEN; t += PK_THREADS) (&s_out[0][0][0])[t] = 0.0;
    const DebugOut nodbg{};
    auto do_fits = [&](bool stamp, bool release_records) __attribute__((always_inline)) {
        const float4* rec = s_rec + (size_t)fstep * QREC_SLOTS * PK_GROUPS + fbase;
        float4 r[QREC_SLOTS];
#pragma unroll
        for (int sl = 0; sl < QREC_SLOTS; ++sl) r[sl] = rec[sl * PK_GROUPS + lane];
        if (release_records) {   // (multi-round scans) the records are in registers: the buffer may take the next round's
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) atomicAdd(s_qn + 5, 1);
        }
        float P[KNN][3];
        uint32_t nidx[KNN], dbits[KNN];
#pragma unroll
        for (int j = 0; j < KNN; ++j) { P[j][0] = r[j].x; P[j][1] = r[j].y; P[j][2] = r[j].z; nidx[j] = __float_as_uint(r[j].w); }
        dbits[0] = __float_as_uint(r[6].x); dbits[1] = __float_as_uint(r[6].y); dbits[2] = __float_as_uint(r[6].z);
        dbits[3] = __float_as_uint(r[6].w); dbits[4] = __float_as_uint(r[7].x);
        const int found = __float_as_int(r[7].y);
        double (*rows_w)[ROW_W] = reinterpret_cast<double (*)[ROW_W]>(smem + (size_t)wave * ROWS_STRIDE);
        const double (*rows)[ROW_W] = rows_w;
        if constexpr (TWO_PHASE) {
            double keep[ROW_W];
            fit_row<W, EXT, false, KNN, true>(s_pose, a.mp, nodbg, found, P, nidx, dbits, r[5].x, r[5].y, r[5].z, __float_as_uint(r[5].w), nullptr, keep);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if ((lane >> 5) == half) {
#pragma unroll
                    for (int j = 0; j < ROW_W; ++j) rows_w[lane & 31][j] = keep[j];
                }
                wave_lds_fence();   // this half's 32 rows are staged
                if (stamp && half == 1) PK_STAMP(7, tid == 0);
#pragma unroll
                for (int c = 0; c < NACC; ++c) {
                    if (olane + c * 64 < NOUT) {
                        double sacc = acc[c];
#pragma unroll 8
                        for (int p = 0; p < 32; ++p) sacc += rows[p][oa[c]] * rows[p][ob[c]];
                        acc[c] = sacc;
                    }
                }
                wave_lds_fence();   // every lane has read this half before the next one overwrites it
            }
        } else {
        double* srow = &rows_w[lane][0];
        fit_row<W, EXT, false, KNN, false, FAST>(s_pose, a.mp, nodbg, found, P, nidx, dbits, r[5].x, r[5].y, r[5].z, __float_as_uint(r[5].w), srow);
        wave_lds_fence();   // this wavefront's 64 rows are staged
        if (stamp) PK_STAMP(7, tid == 0);
#pragma unroll
        for (int c = 0; c < NACC; ++c) {
            if (olane + c * 64 < NOUT) {
                double sacc = acc[c];
#pragma unroll 8
                for (int p = 0; p < (HALVES ? 32 : 64); ++p) sacc += rows[prow0 + p][oa[c]] * rows[prow0 + p][ob[c]];
                acc[c] = sacc;
            }
        }
        }
    };
    for (int round = 0; round < a.rounds; ++round) {
        // The round's wave-level tasks — one per (step, wavefront slot): 8 points, 8 lanes each — are handed out through an
        // LDS counter instead of two fixed tasks per wavefront: a wavefront that drew short buckets takes a third task instead
        // of waiting at the barrier (tile_order is farthest first, so the expensive tasks go out first).  Which wavefront
        // serves a task changes nothing: a task's points and record slots are fixed.  Measured (A/B of two builds in one
        // box): 148.3-150.2 -> 146.8-147.9 us per update, all of it in the first two passes (spans 41.3 -> 40.2, 31.9 -> 29.2).
        const int ntask = a.steps * (PK_THREADS / 64);
        // (counters by round parity: the other parity's are reset while this round runs — see the end of the round)
        int* const s_task = s_qn + ((MULTI && (round & 1)) ? 4 : 1);
        int* const s_qcnt = s_qn + ((MULTI && (round & 1)) ? 3 : 0);
#pragma unroll 1
        for (;;) {
            int task = 0;
            if (lane == 0) task = atomicAdd(s_task, 1);
            task = __builtin_amdgcn_readfirstlane(task);
            if (task >= ntask) break;
            if (MULTI && round > 0) {
                // the previous round's fits may still be running beside this search (below): its records must have left the
                // buffer before this round's are written — every fit wavefront counts itself in s_qn[5] once it holds its 64
                // records in registers, microseconds before the first task of this round gets here (bounded: never a hang)
                const int want = a.steps * (PK_GROUPS / 64) * round;
                int spin = 0;
                for (; spin < (1 << 20) && __hip_atomic_load(s_qn + 5, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < want; ++spin)
                    __builtin_amdgcn_s_sleep(1);
                // (never seen; if it ever expires the records below overwrite ones still being read: tell the host — the flag rides
                // in the counter the finishing pass posts, lv_update / lv_filter_get fail with LV_EHIP)
                if (spin == (1 << 20) && lane == 0) atomicOr(reinterpret_cast<unsigned int*>(&kf->fallback_queries), KF_FAULT_BIT);
            }
            const int step = task / (PK_THREADS / 64);
            const int gqv = (task % (PK_THREADS / 64)) * 8 + (lane >> 3);   // position among the step's 128 points
            // strided assignment: every workgroup gets far and near tiles
            // (snake: odd bands are dealt backwards, so that no workgroup gets the farther tile of every band — the first launch's
            // span 39.9 -> 38.2 us, A/B in one box)
            const uint32_t band = (uint32_t)((round * a.steps + step) * PK_UNITS + (gqv >> 5));
            const uint32_t vbi = band * nwg + ((band & 1u) ? nwg - 1u - bid : bid);
            const bool tile_ok = vbi < a.n_tiles32;
            float4* rec = s_rec + (size_t)step * QREC_SLOTS * PK_GROUPS;
            if (!tile_ok) {   // (whole 256-thread unit: uniform per wavefront)
                if (gl == 7) rec[7 * PK_GROUPS + gqv] = make_float4(0.f, __int_as_float(-1), 0.f, 0.f);
                continue;
            }
            const uint32_t tile = a.tile_order ? a.tile_order[vbi] : vbi;
            const uint32_t q = tile * 32u + (uint32_t)(gqv & 31);
            const bool live = q < a.n;
            kkey k[KNN];
#pragma unroll
            for (int j = 0; j < KNN; ++j) k[j] = none_key();
            uint32_t bstart = 0;
            int src = -1;
            const float4 sp = a.scan[live ? q : a.n - 1];
            float qx, qy, qz;
            rt_apply(s_pose.Tc, sp.x, sp.y, sp.z, qx, qy, qz);  // Mapper.cpp:51
            // bucket levels 0 / 1 here; a point they leave open (a handful per scan once the pose is close, a few per cent of the
            // far tiles while it is off) goes to the workgroup's queue: the coarse levels take a whole wavefront per point,
            // and sixteen wavefronts share them out after the barrier instead of one wavefront serving its own in turn
            bool open_pt = false;
            uint32_t open_bound = 0x7FFFFFFFu;
            knn_search<S, false>(a.map, kf, qx, qy, qz, gl, k, bstart, src, nullptr, false, live, s_stage[gq], a.mp.max_dist_plane_sq,
                                 s_pref[wave], s_start[wave], &open_pt, &open_bound);
            if (open_pt && gl == 0) {
                const int qi = atomicAdd(s_qcnt, 1);
                s_queue[qi] = make_float4(qx, qy, qz, __int_as_float(step * PK_GROUPS + gqv));
                s_queueq[qi] = q;
                s_queueb[qi] = open_bound;
            }
            int found = 0;
#pragma unroll
            for (int j = 0; j < KNN; ++j) found += key_real(k[j]) ? 1 : 0;
            {
                const int slot = gl;
                float4 v;
                if (slot < KNN) {
                    kkey kk = k[0];
#pragma unroll
                    for (int j = 1; j < KNN; ++j) kk = (slot == j) ? k[j] : kk;
                    v = make_float4(0.f, 0.f, 0.f, __uint_as_float(0xFFFFFFFFu));
                    if (live && key_real(kk)) {
                        const uint32_t pos = key_lo(kk);
                        if (src == 0 && pos < (uint32_t)PK_STAGE) {
                            const Xyz w = s_stage[gq][pos];   // still in LDS (same wavefront wrote it)
                            v = make_float4(w.x, w.y, w.z, __uint_as_float(0xFFFFFFFFu));
                        } else if (src >= 0) {
                            const Xyz w = reinterpret_cast<const Xyz*>(a.map.bxyz[0])[(size_t)bstart + pos];
                            v = make_float4(w.x, w.y, w.z, __uint_as_float(0xFFFFFFFFu));
                        } else {
                            v = a.map.orig[pos];
                            v.w = __uint_as_float(pos);
                        }
                    }
                } else if (slot == 5) {
                    v = make_float4(qx, qy, qz, sp.w);
                } else if (slot == 6) {
                    v = make_float4(__uint_as_float(key_hi(k[0])), __uint_as_float(key_hi(k[1])), __uint_as_float(key_hi(k[2])),
                                    __uint_as_float(key_hi(k[3])));
                } else {
                    v = make_float4(__uint_as_float(key_hi(k[4])), __int_as_float(live ? found : -1), 0.f, 0.f);
                }
                rec[slot * PK_GROUPS + gqv] = v;
                if (a.qrec && live) a.qrec[(size_t)slot * a.qstride + q] = v;
            }
            if (round == 0 && step == 0) PK_STAMP(4, tid == 0);   // (wavefront 0: the end of its last task of the first 16)
        }
        if (round == 0) PK_STAMP(5, tid == 0);                    // (wavefront 0 has no task left)
        __syncthreads();   // the records of both steps are complete; nobody reads the candidate stage any more
        if (tid == 0) *s_task = 0;   // (this parity serves the round after next: published by the next round's barrier)
        if (round == 0) PK_STAMP(6, tid == 0);
        const int nq = *s_qcnt;
        // (`unlikely` is for the register allocator, not the branch: spill weights follow the estimated block frequencies, and without
        // the hint this loop — a handful of points per launch — looked hotter than the plane fits behind it: the allocator kept the
        // loop clean and put nine scratch reloads, each an exposed trip to memory, into the middle of the QR: fits 3.7 -> 4.45 us)
#ifndef LV_COARSE_EXPECT
#define LV_COARSE_EXPECT 0
#endif
#ifndef LV_COARSE_PROB
#define LV_COARSE_PROB 0.0005
#endif
        if (__builtin_expect_with_probability(nq > 0, 1, LV_COARSE_PROB)) {   // (uniform) the open points, one per wavefront at a time: the level-2 / level-3 lists / every id
            // (the lane index through an opaque move: everything this rare path derives from it — the lists' voxel offsets by lane, the
            // record addresses — is loop-invariant, and hoisted out of the round loop it cost the kernel nine spilled registers,
            // stored by every thread of every launch: 9.4 MB of scratch writes per launch, converged launches 24.1 -> 25.0 us)
            int clane = lane;
            asm volatile("" : "+v"(clane));
            for (int i = wave; i < nq; i += PK_THREADS / 64) {
                const float4 e = s_queue[i];
                const int p = __float_as_int(e.w);
                const uint32_t q = s_queueq[i];
                kkey kw[KNN];
                knn_coarse<false>(a.map, kf, e.x, e.y, e.z, clane, kw, a.mp.max_dist_plane_sq, s_pref[wave], s_start[wave], s_queueb[i]);
                int found = 0;
#pragma unroll
                for (int j = 0; j < KNN; ++j) found += key_real(kw[j]) ? 1 : 0;
                if (clane < QREC_SLOTS && clane != 5) {   // (slot 5, the world point, is in the record already)
                    const int slot = clane;
                    float4 v;
                    if (slot < KNN) {
                        kkey kk = kw[0];
#pragma unroll
                        for (int j = 1; j < KNN; ++j) kk = (slot == j) ? kw[j] : kk;
                        v = make_float4(0.f, 0.f, 0.f, __uint_as_float(0xFFFFFFFFu));
                        if (key_real(kk)) {   // (keys carry point ids)
                            const uint32_t pos = key_lo(kk);
                            v = a.map.orig[pos];
                            v.w = __uint_as_float(pos);
                        }
                    } else if (slot == 6) {
                        v = make_float4(__uint_as_float(key_hi(kw[0])), __uint_as_float(key_hi(kw[1])), __uint_as_float(key_hi(kw[2])),
                                        __uint_as_float(key_hi(kw[3])));
                    } else {
                        v = make_float4(__uint_as_float(key_hi(kw[4])), __int_as_float(found), 0.f, 0.f);
                    }
                    s_rec[((p / PK_GROUPS) * QREC_SLOTS + slot) * PK_GROUPS + (p % PK_GROUPS)] = v;
                    if (a.qrec) a.qrec[(size_t)slot * a.qstride + q] = v;
                }
            }
            __syncthreads();
            if (tid == 0) *s_qcnt = 0;   // (this parity serves the round after next)
        }
        if (round + 1 < a.rounds) {   // (not the last round: its fits are part of the loop; the last round's follow the loop)
            if constexpr (!MULTI) {
                // (not instantiated for multi-round scans any more: kept as the A/B form, LV_FUSED_MULTI=0 limits it to three rounds)
                if (fitter) do_fits(false, false);
                __syncthreads();
            } else if (fitter) {
                // Round 4: NO barrier behind these fits — the twelve wavefronts that do not fit planes go straight on to the next
                // round's search tasks while the four fit (rows staged in the fit wavefronts' OWN candidate-stage areas, region 0
                // below 4 x 6 KB, which nobody else writes; the record buffer is released as soon as the records are in
                // registers) and join the search when they are done.  Before, the twelve idled through every round's fits but
                // the last: 131 072 points 202 us per update, 262 144 points 332.
                do_fits(false, true);
            }
        }
    }
    // ---- 3b. the fits of the last round — and, in the bookkeeping workgroup of a launch whose prologue solved a pass that does
    // not end the update (mode 1), BESIDE them the books, by the twelve wavefronts that do not fit planes: scratch above the
    // staged rows, nothing in the books depends on this launch's search.  Out here, after the round loop, the books' register
    // appetite competes with nothing that is live (inlined INTO the loop it spilled the search: round 2, 60 % slower).
    {
        const bool beside = a.mode == 1 && keeper && 64 * PK_BOOKW >= NS * W;   // (the books' widest step takes NS * W threads)
        if (fitter) {
            do_fits(true, false);
        } else if (beside && wave >= PK_FITW) {
            SubBar bar(s_qn + 2, PK_BOOKW);
            bookkeeping<W, 64 * PK_BOOKW>(K, Bk, kf, ps_in, ps_out, a.io, a.sums_out, a.sp.R_inv, a.sp.seq, &s_pose, K.Pprop, K.xp, false,
                                          tid - 64 * PK_FITW, bar, clk);
            PK_STAMP(10, tid == 64 * PK_FITW);
        }
        books_done = beside;
        PK_STAMP(8, tid == 0);
        __syncthreads();
    }
    // ---- 4. the workgroup partial, compact layout, fixed order
    if (fitter) {
#pragma unroll
        for (int c = 0; c < NACC; ++c)
            if (olane + c * 64 < NOUT) s_out[wave][HALVES ? (lane >> 5) : 0][olane + c * 64] = acc[c];
    }
    __syncthreads();
    if (tid < OW) {
        double s = 0.0;
        if (tid < NOUT) {
            s = s_out[0][0][tid] + s_out[0][1][tid];
#pragma unroll
            for (int f = 1; f < PK_FITW; ++f) s += s_out[f][0][tid] + s_out[f][1][tid];
        }
        a.part_out[(size_t)bid * OW + tid] = s;
    }
    }   // searching
    if (a.cost_out && tid == 0 && bid < nwg) a.cost_out[bid] = (uint32_t)(wall_clock64() - t_begin);
    }   // !CLOSING
    PK_STAMP(9, tid == 0);
    if (!keeper || books_done) return;
    // ---- 5. the books (one workgroup, after its own search and fits; its scratch lies above the staged rows) ---------
    // (Running them on the twelve wavefronts that do not fit planes, beside the fits, over an LDS-counter barrier was tried:
    // inlined into the round loop the books' register appetite spilled the search, the whole kernel ran 60 % longer.)
    {
        WgBar bar;
        keeper_books<W, PK_THREADS>(a, begin, K, Bk, kf, ps_in, ps_out, &s_pose, tid, bar, clk);
    }
    PK_STAMP(10, tid == 0);
#undef PK_STAMP
}

// Geometry of pass_kernel for an n-point scan (32-point tiles; a workgroup takes PK_UNITS tiles per step): the searching
// workgroups, search steps per round (1 when one step per workgroup covers the scan), rounds, and whether a CU is left over
// for a dedicated bookkeeping workgroup (its books then run beside the others' searches instead of after its own).
void pass_grid_size(uint32_t n, int max_wg, int* nsearch, int* steps, int* rounds, int* dedicated) {
    const uint32_t nt = (n + 31u) / 32u, U = (uint32_t)PK_UNITS, M = (uint32_t)(max_wg > 1 ? max_wg : 2);
    uint32_t st, g;
    if (nt <= U * M) { st = 1; g = (nt + U - 1u) / U; }
    else { st = (uint32_t)PK_STEPS; g = (nt + st * U - 1u) / (st * U); }
    const bool ded = g <= M - 1u;
    if (g > M) g = M;
    if (g < 1) g = 1;
    // A/B knob (profiles/experiments_r05/half_rounds_ab.txt): a scan of ONE two-step round per workgroup as TWO one-step rounds
    // through the MULTI instantiation, so that the first half's plane fits run beside the second half's search
    static const bool split = [] { const char* e = getenv("LV_PASS_SPLIT"); return e && atoi(e) != 0; }();
    if (split && st == (uint32_t)PK_STEPS && (nt + st * U * g - 1u) / (st * U * g) == 1u && nt > U * g) st = 1;
    *nsearch = (int)g;
    *steps = (int)st;
    *rounds = (int)((nt + st * U * g - 1u) / (st * U * g));
    *dedicated = ded ? 1 : 0;
}
int pass_clock_words() { return PK_CLK; }

int launch_pass(hipStream_t stream, const PassLaunch& pl, const BeginArg* begin) {
    PassArgs a;
    a.map = *pl.map;
    a.scan = pl.scan;
    a.tile_order = pl.tile_order;
    a.n = pl.n;
    a.n_tiles32 = (pl.n + 31u) / 32u;
    a.kf = pl.kf;
    a.io = pl.io;
    a.recs_in = pl.recs_in;
    a.part_out = pl.part_out;
    a.sums_out = pl.sums_out;
    a.qrec = pl.qrec;
    a.clk = pl.clk;
    a.qstride = pl.qstride;
    a.nrec = pl.nrec;
    a.mode = pl.mode;
    a.rounds = pl.rounds;
    a.launch = pl.launch;
    a.steps = pl.steps;
    a.cost_in = pl.cost_in;
    a.cost_out = pl.cost_out;
    a.nsearch = (uint32_t)pl.nwg;
    a.mp = pl.mp;
    a.sp = pl.sp;
    static const BeginArg none{};
    const BeginArg& b = begin ? *begin : none;
    const dim3 grid((unsigned)(pl.rounds > 0 ? pl.nwg + (pl.dedicated ? 1 : 0) : 1)), block(PK_THREADS);
    const bool closing = pl.rounds == 0;
    if (pl.mp.estimate_extrinsics) {
        if (closing) hipLaunchKernelGGL((pass_kernel<true, true>), grid, block, 0, stream, a, b);
        else if (pl.rounds > 1 && pl.multi_overlap) hipLaunchKernelGGL((pass_kernel<true, false, true>), grid, block, 0, stream, a, b);
        else hipLaunchKernelGGL((pass_kernel<true, false>), grid, block, 0, stream, a, b);
    } else {
        if (closing) hipLaunchKernelGGL((pass_kernel<false, true>), grid, block, 0, stream, a, b);
        else if (pl.rounds > 1 && pl.multi_overlap) {
            if (pl.mp.fast_fit) hipLaunchKernelGGL((pass_kernel<false, false, true, true>), grid, block, 0, stream, a, b);
            else hipLaunchKernelGGL((pass_kernel<false, false, true>), grid, block, 0, stream, a, b);
        } else {
            // (fast_fit: the opt-in approximate plane fit exists for the default 6-column configuration only)
            if (pl.mp.fast_fit) hipLaunchKernelGGL((pass_kernel<false, false, false, true>), grid, block, 0, stream, a, b);
            else hipLaunchKernelGGL((pass_kernel<false, false>), grid, block, 0, stream, a, b);
        }
    }
    LV_HIP(hipGetLastError());
    return LV_OK;
}

}  // namespace lv

// lv_scan.hip — scan residency.  `points2match = points` (reference src/Modules/Localizator.cpp:131)
// becomes: upload once per correct(), Morton-sort in the LiDAR frame so that consecutive lanes /
// wavefronts query neighbouring voxels (a rigid transform preserves neighbourhoods, so the order
// stays coherent for every IKFoM pass).  Results are always reported in ORIGINAL scan order: each
// record carries its original index in .w.
#include "lv_host.hpp"
#include "lv_ldssort.hpp"

#include <cstdlib>
#include <cstring>

#include <hipcub/hipcub.hpp>

namespace lv {

__device__ __forceinline__ uint32_t spread10(uint32_t v) {
    v &= 0x3ff;
    v = (v | (v << 16)) & 0x030000ff;
    v = (v | (v << 8)) & 0x0300f00f;
    v = (v | (v << 4)) & 0x030c30c3;
    v = (v | (v << 2)) & 0x09249249;
    return v;
}

__global__ void scan_keys_kernel(const float4* __restrict__ pts, uint32_t n, float ox, float oy, float oz, float inv_cell,
                                 uint32_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 p = pts[i];
    float fx = fminf(fmaxf((p.x - ox) * inv_cell, 0.f), 1023.f);
    float fy = fminf(fmaxf((p.y - oy) * inv_cell, 0.f), 1023.f);
    float fz = fminf(fmaxf((p.z - oz) * inv_cell, 0.f), 1023.f);
    keys[i] = spread10((uint32_t)fx) | (spread10((uint32_t)fy) << 1) | (spread10((uint32_t)fz) << 2);
    idx[i] = i;
}

__global__ void scan_gather_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ idx_sorted, uint32_t n,
                                   float4* __restrict__ sorted) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    sorted[i] = pts[idx_sorted[i]];
}

// ---- row f-2: Compensator::compensate + voxelgrid_downsample on the device ------------------------------
// de-skew: one lane per raw point (reference src/Modules/Compensator.cpp:123-146)
__device__ __forceinline__ float4 deskew_point(const float4 p, const double t, uint32_t i, const MotionState* __restrict__ states,
                                               uint32_t n_states, const MotionState& xt2) {
    const float qnan = __uint_as_float(0x7fc00000u);
    float4 o = make_float4(qnan, qnan, qnan, __uint_as_float(i));
    // the reference's two-pointer walk puts a point into the FIRST interval [states[s].time, states[s+1].time]
    // that contains its time
    if (n_states >= 2 && t >= states[0].time && t <= states[n_states - 1].time) {
        uint32_t lo = 0, hi = n_states - 2;  // smallest s with t <= states[s+1].time
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (t <= states[mid + 1].time) hi = mid; else lo = mid + 1;
        }
        const MotionState st = states[lo];
        const float dt = (float)(t - st.time);
        RT32 X, LI;
        motion_integrate_pose(st, dt, X.R, X.t);                         // Xtp += IMU(states[s].a, states[s].w, t)  :133-134
#pragma unroll
        for (int k = 0; k < 9; ++k) LI.R[k] = st.RLI[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) LI.t[k] = st.tLI[k];
        float gx, gy, gz;
        const RT32 T = rt_compose(X, LI);
        rt_apply(T, p.x, p.y, p.z, gx, gy, gz);                          // global_p = Xtp * Xtp.I_Rt_L() * p   :137
        RT32 X2, LI2;
#pragma unroll
        for (int k = 0; k < 9; ++k) { X2.R[k] = xt2.R[k]; LI2.R[k] = xt2.RLI[k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) { X2.t[k] = xt2.pos[k]; LI2.t[k] = xt2.tLI[k]; }
        const RT32 back = rt_compose(rt_inv(LI2), rt_inv(X2));            // Xt2.I_Rt_L().inv() * Xt2.inv()      :138
        rt_apply(back, gx, gy, gz, o.x, o.y, o.z);
    }
    return o;
}
__global__ void deskew_kernel(const float4* __restrict__ raw, const double* __restrict__ times, uint32_t n,
                              const MotionState* __restrict__ states, uint32_t n_states, MotionState xt2,
                              float4* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = deskew_point(raw[i], times[i], i, states, n_states, xt2);
}

// pcl::VoxelGrid [UPSTREAM-RECALL PCL 1.8 applyFilter]: bounds -> leaf index -> sort -> centroid per leaf
__device__ __forceinline__ unsigned flip_f32(float f) {  // order-preserving float -> uint
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unflip_f32(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
__global__ __launch_bounds__(256) void vg_bounds_kernel(const float4* __restrict__ pts, uint32_t n, unsigned* __restrict__ bounds /*min xyz, max xyz*/) {
    // bounds per workgroup in LDS first (six same-address device atomics per POINT were 14 us for an 11k-point window)
    __shared__ unsigned s_b[6];
    if (threadIdx.x < 3) s_b[threadIdx.x] = 0xFFFFFFFFu;
    else if (threadIdx.x < 6) s_b[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t stride = gridDim.x * blockDim.x;
    unsigned lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float4 p = pts[i];
        if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) continue;
        const unsigned f[3] = {flip_f32(p.x), flip_f32(p.y), flip_f32(p.z)};
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = f[a] < lo[a] ? f[a] : lo[a]; hi[a] = f[a] > hi[a] ? f[a] : hi[a]; }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        unsigned l = lo[a], h = hi[a];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned l2 = __shfl_xor(l, o), h2 = __shfl_xor(h, o);
            l = l2 < l ? l2 : l;
            h = h2 > h ? h2 : h;
        }
        if ((threadIdx.x & 63) == 0) { atomicMin(&s_b[a], l); atomicMax(&s_b[3 + a], h); }
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&bounds[threadIdx.x], s_b[threadIdx.x]);
    else if (threadIdx.x < 6) atomicMax(&bounds[threadIdx.x], s_b[threadIdx.x]);
}
// bounds scratch back to "nothing seen" (its last two words: the result record of voxel_and_sort)
__global__ void vg_bounds_init_kernel(unsigned* __restrict__ bounds) {
    if (threadIdx.x < 8) bounds[threadIdx.x] = threadIdx.x < 3 ? 0xFFFFFFFFu : 0u;
}
__global__ void vg_keys_kernel(const float4* __restrict__ pts, uint32_t n, const unsigned* __restrict__ bounds, float inv_leaf,
                               uint64_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    idx[i] = i;
    if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) { keys[i] = ~0ull >> 1; return; }  // dropped (sorted last)
    long long minb[3], divb[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        minb[a] = (long long)floorf(unflip_f32(bounds[a]) * inv_leaf);
        divb[a] = (long long)floorf(unflip_f32(bounds[3 + a]) * inv_leaf) - minb[a] + 1;
    }
    const long long i0 = (long long)floorf(p.x * inv_leaf) - minb[0], i1 = (long long)floorf(p.y * inv_leaf) - minb[1],
                    i2 = (long long)floorf(p.z * inv_leaf) - minb[2];
    keys[i] = (uint64_t)(i0 + i1 * divb[0] + i2 * divb[0] * divb[1]);
}
// head of every leaf: sequential f32 sum of its points in input order (stable sort), centroid = sum / count
__global__ void vg_heads_kernel(const uint64_t* __restrict__ keys_sorted, uint32_t n, uint32_t* __restrict__ heads) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = keys_sorted[i];
    heads[i] = (k != (~0ull >> 1) && (i == 0 || keys_sorted[i - 1] != k)) ? 1u : 0u;
}
__global__ void vg_centroid_kernel(const float4* __restrict__ pts, const uint64_t* __restrict__ keys_sorted,
                                   const uint32_t* __restrict__ idx_sorted, const uint32_t* __restrict__ heads,
                                   const uint32_t* __restrict__ hpos, uint32_t n, float4* __restrict__ out,
                                   unsigned* __restrict__ n_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (i == n - 1) *n_out = hpos[i] + heads[i];   // leaves in all (read back with the bounds)
    if (!heads[i]) return;
    const uint64_t k = keys_sorted[i];
    float sx = 0.f, sy = 0.f, sz = 0.f;
    // the leaf's points in input order, eight loads in flight at a time (one dependent key -> index -> point chain per
    // member made this kernel 26 us for an 11k-point window); the SUM stays sequential: it is the oracle's order
    constexpr int C = 8;
    uint32_t j = i;
    bool more = true;
    while (more) {
        uint64_t kk[C];
        uint32_t id[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const uint32_t jj = j + c < n ? j + c : n - 1;
            kk[c] = keys_sorted[jj];
            id[c] = idx_sorted[jj];
        }
        float4 pp[C];
#pragma unroll
        for (int c = 0; c < C; ++c) pp[c] = pts[id[c]];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            if (more && j + c < n && kk[c] == k) { sx += pp[c].x; sy += pp[c].y; sz += pp[c].z; }
            else if (more) { more = false; j += c; }
        }
        if (more) j += C;
    }
    const float cnt = (float)(j - i);
    const uint32_t o = hpos[i];
    out[o] = make_float4(sx / cnt, sy / cnt, sz / cnt, __uint_as_float(o));
}

// ---- small windows: the whole chain in ONE launch --------------------------------------------------------------------
// A 0.01 s window of a stream holds 1-2 thousand raw points.  De-skew, bounds, leaf keys, stable sort, leaf heads, scan,
// centroids, Morton keys, sort, gather, tile order were twelve launches (+ two library sorts) of a few microseconds of work
// each; here ONE workgroup runs the same stages over LDS, separated by workgroup barriers.  Every stage computes what its
// stand-alone kernel computes (same expressions, same order of every sum: the leaf centroid stays the sequential f32 sum
// in input order), so the result is bit-identical to the general path (tests/test_gpu_deskew.py runs both).
// bounds[0..5]: min / max of the de-skewed points (flipped floats), [6]: points out, [7]: 1 = the leaf grid does not fit the
// packed sort key (the caller then takes the general path).
constexpr int SMALL_WINDOW = 2048;
constexpr int SW_THREADS = 1024;
__device__ __forceinline__ void lds_bitonic_u64(uint64_t* s_key, uint32_t n, int tid) {
    lds_bitonic_sort_u64<SW_THREADS>(s_key, lds_sort_len(n), tid);   // (padding sorts last; lv_ldssort.hpp)
}
// raw and out_raw are NOT restrict-qualified: lv_scan_downsample with leaf <= 0 hands the same buffer in for both (every read
// of raw precedes the barriers in front of the first store to out_raw, but `restrict` would promise the compiler more).
// LDS: 88 KB static here, 156 KB in window_tail_kernel — more than the 64 KB of older CDNA parts: this file is gfx950 code.
static_assert(sizeof(float4) * 2 * 2048 + sizeof(uint64_t) * 2048 + sizeof(float) * 2048 <= 160 * 1024, "window_small_kernel: one workgroup's LDS (gfx950: 160 KB per CU)");
__global__ __launch_bounds__(SW_THREADS) void window_small_kernel(const float4* raw, const double* __restrict__ times,
                                                                  uint32_t n_in, const MotionState* __restrict__ states,
                                                                  uint32_t n_states, MotionState xt2, int do_deskew, float leaf,
                                                                  float inv_sort_cell, uint32_t tile_points,
                                                                  float4* out_raw, float4* __restrict__ out_sorted,
                                                                  uint32_t* __restrict__ tile_order, unsigned* __restrict__ bounds,
                                                                  unsigned long long* __restrict__ note, uint32_t seq) {
    __shared__ float4 s_pt[SMALL_WINDOW];       // de-skewed input
    __shared__ float4 s_out[SMALL_WINDOW];      // output points (voxel-grid order)
    __shared__ uint64_t s_key[SMALL_WINDOW];
    __shared__ float s_r2[SMALL_WINDOW];
    __shared__ unsigned s_b[8];
    __shared__ uint32_t s_wsum[SW_THREADS / 64 + 1];
    const int tid = threadIdx.x;
    if (tid < 3) s_b[tid] = 0xFFFFFFFFu;
    else if (tid < 8) s_b[tid] = 0u;
    // ---- de-skew (or take the points as they are: Compensator::downsample on its own)
    for (uint32_t i = tid; i < (uint32_t)SMALL_WINDOW; i += SW_THREADS) {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n_in) {
            const float4 p = raw[i];
            o = do_deskew ? deskew_point(p, times[i], i, states, n_states, xt2) : p;
        }
        s_pt[i] = o;
    }
    __syncthreads();
    // ---- bounds of the finite points (vg_bounds_kernel)
    {
        unsigned lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
        for (uint32_t i = tid; i < n_in; i += SW_THREADS) {
            const float4 p = s_pt[i];
            if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) continue;
            const unsigned f[3] = {flip_f32(p.x), flip_f32(p.y), flip_f32(p.z)};
#pragma unroll
            for (int a = 0; a < 3; ++a) { lo[a] = f[a] < lo[a] ? f[a] : lo[a]; hi[a] = f[a] > hi[a] ? f[a] : hi[a]; }
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            unsigned l = lo[a], h = hi[a];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned l2 = __shfl_xor(l, o), h2 = __shfl_xor(h, o);
                l = l2 < l ? l2 : l;
                h = h2 > h ? h2 : h;
            }
            if ((tid & 63) == 0) { atomicMin(&s_b[a], l); atomicMax(&s_b[3 + a], h); }
        }
    }
    __syncthreads();
    uint32_t n_out = n_in;
    if (s_b[0] == 0xFFFFFFFFu) {   // no finite point: nothing to match
        if (tid < 8) bounds[tid] = tid < 6 ? s_b[tid] : 0u;
        if (tid == 0) { note_post(note, seq, 0u); note_post(note + 1, seq, 0u); }
        return;
    }
    if (leaf > 0.f) {
        // ---- leaf keys (vg_keys_kernel), packed with the input index so that one u64 order == the stable sort by key
        const float inv_leaf = 1.0f / leaf;
        long long minb[3], divb[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            minb[a] = (long long)floorf(unflip_f32(s_b[a]) * inv_leaf);
            divb[a] = (long long)floorf(unflip_f32(s_b[3 + a]) * inv_leaf) - minb[a] + 1;
        }
        // (the packed key needs leaf index < 2^52: the extent of a window in leaves; otherwise the general path)
        const double cells = (double)divb[0] * (double)divb[1] * (double)divb[2];
        if (!(cells < 4.0e15)) {
            if (tid < 8) bounds[tid] = tid < 6 ? s_b[tid] : (tid == 7 ? 1u : 0u);
            if (tid == 0) { note_post(note, seq, 0u); note_post(note + 1, seq, 1u); }
            return;
        }
        for (uint32_t i = tid; i < (uint32_t)SMALL_WINDOW; i += SW_THREADS) {
            uint64_t k = ~0ull;                                    // padding and dropped points sort last
            if (i < n_in) {
                const float4 p = s_pt[i];
                if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
                    const long long i0 = (long long)floorf(p.x * inv_leaf) - minb[0], i1 = (long long)floorf(p.y * inv_leaf) - minb[1],
                                    i2 = (long long)floorf(p.z * inv_leaf) - minb[2];
                    k = ((uint64_t)(i0 + i1 * divb[0] + i2 * divb[0] * divb[1]) << 11) | (uint64_t)i;
                }
            }
            s_key[i] = k;
        }
        __syncthreads();
        lds_bitonic_u64(s_key, n_in, tid);
        // ---- leaf heads + exclusive scan (vg_heads_kernel, DeviceScan) — two elements per thread
        uint32_t h0 = 0, h1 = 0;
        {
            const uint32_t i = 2u * (uint32_t)tid;
            if (i < n_in) { const uint64_t k = s_key[i]; h0 = (k != ~0ull && (i == 0 || (s_key[i - 1] >> 11) != (k >> 11))) ? 1u : 0u; }
            if (i + 1 < n_in) { const uint64_t k = s_key[i + 1]; h1 = (k != ~0ull && ((s_key[i] >> 11) != (k >> 11))) ? 1u : 0u; }
        }
        uint32_t incl = h0 + h1;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o);
            if ((tid & 63) >= o) incl += v;
        }
        if ((tid & 63) == 63) s_wsum[tid >> 6] = incl;
        __syncthreads();
        if (tid == 0) {
            uint32_t run = 0;
            for (int w = 0; w < SW_THREADS / 64; ++w) { const uint32_t v = s_wsum[w]; s_wsum[w] = run; run += v; }
            s_wsum[SW_THREADS / 64] = run;
        }
        __syncthreads();
        const uint32_t excl = s_wsum[tid >> 6] + incl - (h0 + h1);
        n_out = s_wsum[SW_THREADS / 64];
        // ---- centroid per leaf: the sequential f32 sum of its points in input order (vg_centroid_kernel)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const uint32_t i = 2u * (uint32_t)tid + (uint32_t)e;
            const bool head = e == 0 ? h0 != 0u : h1 != 0u;
            if (!head) continue;
            const uint64_t leafk = s_key[i] >> 11;
            float sx = 0.f, sy = 0.f, sz = 0.f;
            uint32_t j = i;
            while (j < n_in && (s_key[j] >> 11) == leafk && s_key[j] != ~0ull) {
                const float4 p = s_pt[(uint32_t)(s_key[j] & 2047u)];
                sx += p.x; sy += p.y; sz += p.z;
                ++j;
            }
            const float cnt = (float)(j - i);
            const uint32_t o = excl + (e == 1 ? h0 : 0u);
            s_out[o] = make_float4(sx / cnt, sy / cnt, sz / cnt, __uint_as_float(o));
        }
        __syncthreads();
    } else {
        for (uint32_t i = tid; i < n_in; i += SW_THREADS) s_out[i] = s_pt[i];
        __syncthreads();
    }
    for (uint32_t i = tid; i < n_out; i += SW_THREADS) out_raw[i] = s_out[i];
    if (tid < 8) bounds[tid] = tid < 6 ? s_b[tid] : (tid == 6 ? n_out : 0u);
    // (the note goes out here: the host continues while the Morton order below is still being computed — everything that
    // reads it is ordered behind this kernel on the stream)
    if (tid == 0) { note_post(note, seq, n_out); note_post(note + 1, seq, 0u); }
    if (n_out == 0) return;
    // ---- Morton order of the output, tile ranges, tile order (scan_sort_small_kernel)
    const float ox = unflip_f32(s_b[0]), oy = unflip_f32(s_b[1]), oz = unflip_f32(s_b[2]);
    for (uint32_t i = tid; i < (uint32_t)SMALL_WINDOW; i += SW_THREADS) {
        uint64_t k = ~0ull;
        if (i < n_out) {
            const float4 p = s_out[i];
            const float fx = fminf(fmaxf((p.x - ox) * inv_sort_cell, 0.f), 1023.f);
            const float fy = fminf(fmaxf((p.y - oy) * inv_sort_cell, 0.f), 1023.f);
            const float fz = fminf(fmaxf((p.z - oz) * inv_sort_cell, 0.f), 1023.f);
            const uint32_t key = spread10((uint32_t)fx) | (spread10((uint32_t)fy) << 1) | (spread10((uint32_t)fz) << 2);
            k = ((uint64_t)key << 32) | i;
        }
        s_key[i] = k;
    }
    __syncthreads();
    lds_bitonic_u64(s_key, n_out, tid);
    for (uint32_t i = tid; i < n_out; i += SW_THREADS) {
        const float4 p = s_out[(uint32_t)s_key[i]];
        out_sorted[i] = p;
        s_r2[i] = p.x * p.x + p.y * p.y + p.z * p.z;
    }
    __syncthreads();
    if (tile_points == 0) return;
    const uint32_t nt = (n_out + tile_points - 1) / tile_points;
    uint64_t mine = ~0ull;
    if ((uint32_t)tid < nt) {
        float r2 = 0.f;
        const uint32_t b = (uint32_t)tid * tile_points;
        for (uint32_t i = 0; i < tile_points && b + i < n_out; ++i) r2 = fmaxf(r2, s_r2[b + i]);
        mine = ((uint64_t)(~__float_as_uint(r2)) << 32) | (uint32_t)tid;   // complement ascending = range descending; ties by tile
    }
    __syncthreads();
    if ((uint32_t)tid < nt) s_key[tid] = mine;   // (nt <= 1024 whenever tile_points >= 2)
    __syncthreads();
    if ((uint32_t)tid < nt) {
        uint32_t rank = 0;
        for (uint32_t u = 0; u < nt; ++u) rank += s_key[u] < mine ? 1u : 0u;
        tile_order[rank] = (uint32_t)tid;
    }
}

// ---- larger windows (a 0.01 s window of a 64-ring sensor holds ~13 k raw points): two launches + the library sort -------------
// The general chain below is thirteen launches and two host round trips for ~40 us of work.  Here: (1) de-skew straight from the
// LiDAR buffer + bounds + sort keys, (2) the library's stable sort, (3) ONE workgroup for everything behind the sort — leaf
// heads, their scan, the centroids (one leaf per thread: the sequential f32 sum in input order, eight loads in flight), the
// Morton order of the few hundred output points, tile ranges and tile order — which posts (points out, status) as a note
// (lv_note.hpp).  Every stage computes what its stand-alone kernel computes, in the same order: bit-identical results
// (tests/test_gpu_cloud.py runs both).  Limits: WT_IN points in, WT_OUT out; beyond them the tail declines (status 1) and the
// caller takes the general chain.
constexpr int WT_OUT = 4096;
constexpr int WT_IN = 16384;
constexpr long long WT_HALF = 1ll << 20;   // leaf coordinates of a window are taken relative to -2^20 (21-bit fields of the sort key)
// (1) de-skew + bounds + sort keys.  The voxel grid's leaf index i0 + i1 * div0 + i2 * div0 * div1 (vg_keys_kernel) orders the
// leaves exactly like the triple (i2, i1, i0) does — a mixed-radix number — and subtracting the grid's minimum from every
// coordinate changes neither order nor equality: the key (i2, i1, i0) in three 21-bit fields relative to a FIXED offset gives
// the same sorted sequence and the same leaf boundaries without knowing the bounds first, which is what cost a launch of its
// own.  A coordinate outside the fields (|x| >= 2^20 leaves: 500 km at 0.5 m) raises bounds[7]: the tail then declines.
__global__ __launch_bounds__(256) void deskew_keys_kernel(const CloudPoint* __restrict__ cloud, uint32_t n, const MotionState* __restrict__ states,
                                                          uint32_t n_states, MotionState xt2, float inv_leaf, float4* __restrict__ out,
                                                          uint64_t* __restrict__ keys, uint32_t* __restrict__ idx, unsigned* __restrict__ bounds) {
    __shared__ unsigned s_b[6];
    if (threadIdx.x < 3) s_b[threadIdx.x] = 0xFFFFFFFFu;
    else if (threadIdx.x < 6) s_b[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    if (i < n) {
        const CloudPoint c = cloud[i];
        const float4 o = deskew_point(make_float4(c.x, c.y, c.z, 0.f), c.time, i, states, n_states, xt2);
        out[i] = o;
        uint64_t key = ~0ull >> 1;   // dropped (sorted last): vg_keys_kernel
        if (isfinite(o.x) && isfinite(o.y) && isfinite(o.z)) {
            lo[0] = hi[0] = flip_f32(o.x); lo[1] = hi[1] = flip_f32(o.y); lo[2] = hi[2] = flip_f32(o.z);
            const long long i0 = (long long)floorf(o.x * inv_leaf), i1 = (long long)floorf(o.y * inv_leaf), i2 = (long long)floorf(o.z * inv_leaf);
            const long long lim = WT_HALF - 2;
            if (i0 < -lim || i0 > lim || i1 < -lim || i1 > lim || i2 < -lim || i2 > lim) atomicOr(&bounds[7], 1u);
            else key = ((uint64_t)(i2 + WT_HALF) << 42) | ((uint64_t)(i1 + WT_HALF) << 21) | (uint64_t)(i0 + WT_HALF);
        }
        keys[i] = key;
        idx[i] = i;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        unsigned l = lo[a], h = hi[a];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned l2 = __shfl_xor(l, o), h2 = __shfl_xor(h, o);
            l = l2 < l ? l2 : l;
            h = h2 > h ? h2 : h;
        }
        if ((threadIdx.x & 63) == 0) { atomicMin(&s_b[a], l); atomicMax(&s_b[3 + a], h); }
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&bounds[threadIdx.x], s_b[threadIdx.x]);
    else if (threadIdx.x < 6) atomicMax(&bounds[threadIdx.x], s_b[threadIdx.x]);
}
// (3) everything behind the sort, by one workgroup.  The sorted keys are only needed to tell where a leaf begins: they are read
// once, coalesced, and what stays in LDS is one word per sorted element — its input index, bit 31 set on the first element of
// a leaf (all ones: a dropped point) — from which the heads are counted and scanned (a run of consecutive elements per thread)
// and the leaves' members are walked.
__global__ __launch_bounds__(SW_THREADS) void window_tail_kernel(const float4* __restrict__ desk, const uint64_t* __restrict__ keys_sorted,
                                                                 const uint32_t* __restrict__ idx_sorted, uint32_t n_in, float inv_sort_cell,
                                                                 uint32_t tile_points, float4* __restrict__ out_raw,
                                                                 float4* __restrict__ out_sorted, uint32_t* __restrict__ tile_order,
                                                                 unsigned* __restrict__ bounds, unsigned long long* __restrict__ note,
                                                                 uint32_t seq, long long* __restrict__ clk) {
#define WT_STAMP(i) do { if (clk && threadIdx.x == 0) { clk[i] = wall_clock64(); if ((i) == 4) clk[7] = clock64(); if ((i) == 5) clk[7] = clock64() - clk[7]; } } while (0)
    WT_STAMP(0);
    constexpr int WT_STAGE = 6144;   // sorted elements whose points are staged in LDS at a time (3 x 24 KB)
    __shared__ __attribute__((aligned(16))) unsigned char s_region[WT_IN * sizeof(uint32_t) + 3 * WT_STAGE * sizeof(float)];
    __shared__ uint32_t s_headpos[WT_OUT];
    __shared__ uint32_t s_wsum[SW_THREADS / 64 + 1], s_gsum[SW_THREADS / 64];
    static_assert(sizeof(s_region) >= WT_OUT * (sizeof(float4) + sizeof(uint64_t) + sizeof(float)), "the output arrays overlay the element words and the stage");
    static_assert(sizeof(s_region) + sizeof(uint32_t) * WT_OUT <= 160 * 1024 - 1024, "window_tail_kernel: one workgroup's LDS (gfx950: 160 KB per CU)");
    uint32_t* s_hi = reinterpret_cast<uint32_t*>(s_region);
    float* s_px = reinterpret_cast<float*>(s_region + WT_IN * sizeof(uint32_t));
    float* s_py = s_px + WT_STAGE;
    float* s_pz = s_py + WT_STAGE;
    const int tid = threadIdx.x;
    auto report = [&](uint32_t n_out, uint32_t status) {
        if (tid == 0) { bounds[6] = n_out; bounds[7] = status; note_post(note, seq, n_out); note_post(note + 1, seq, status); }
    };
    if (bounds[7] != 0u || n_in > (uint32_t)WT_IN) { report(0u, 1u); return; }   // (a coordinate outside the key's fields; too many points)
    if (bounds[0] == 0xFFFFFFFFu) { report(0u, 0u); return; }                     // no finite point: nothing to match
    constexpr uint64_t DROPPED = ~0ull >> 1;
    constexpr uint32_t HEAD = 0x80000000u, GONE = 0xFFFFFFFFu;
#pragma unroll 4
    for (uint32_t i = tid; i < n_in; i += SW_THREADS) {
        const uint64_t k = keys_sorted[i];
        const uint64_t kp = i ? keys_sorted[i - 1] : 0ull;
        const uint32_t id = idx_sorted[i];
        s_hi[i] = k == DROPPED ? GONE : ((i == 0 || k != kp) ? (HEAD | id) : id);
    }
    __syncthreads();
    WT_STAMP(1);
    // ---- leaf heads (vg_heads_kernel) of this thread's run of consecutive sorted elements, then their exclusive scan
    const uint32_t per = (n_in + SW_THREADS - 1) / SW_THREADS;
    const uint32_t b0 = per * (uint32_t)tid, b1 = b0 + per < n_in ? b0 + per : n_in;
    uint32_t hcnt = 0, gone = 0;
    for (uint32_t i = b0; i < b1; ++i) { const uint32_t v = s_hi[i]; hcnt += (v & HEAD) && v != GONE ? 1u : 0u; gone += v == GONE ? 1u : 0u; }
    uint32_t incl = hcnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o);
        if ((tid & 63) >= o) incl += v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) gone += __shfl_xor(gone, o);
    if ((tid & 63) == 63) s_wsum[tid >> 6] = incl;
    if ((tid & 63) == 0) s_gsum[tid >> 6] = gone;
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0, g = 0;
        for (int w = 0; w < SW_THREADS / 64; ++w) { const uint32_t v = s_wsum[w]; s_wsum[w] = run; run += v; g += s_gsum[w]; }
        s_wsum[SW_THREADS / 64] = run;
        s_gsum[0] = g;
    }
    __syncthreads();
    const uint32_t n_out = s_wsum[SW_THREADS / 64];
    const uint32_t n_valid = n_in - s_gsum[0];   // (the dropped points sort last)
    if (n_out > (uint32_t)WT_OUT) { report(0u, 1u); return; }
    {
        uint32_t o = s_wsum[tid >> 6] + incl - hcnt;
        for (uint32_t i = b0; i < b1; ++i) { const uint32_t v = s_hi[i]; if ((v & HEAD) && v != GONE) s_headpos[o++] = i; }
    }
    __syncthreads();
    WT_STAMP(2);
    // ---- centroid per leaf (vg_centroid_kernel): the sequential f32 sum of its points in input order.  A leaf's members are
    // consecutive sorted elements; their points are staged in LDS WT_STAGE elements at a time by the whole workgroup (six loads
    // in flight per thread, one round trip per stage), and every thread adds up the members of its leaves that lie in the stage,
    // in order, out of LDS — the longest leaf of a window (hundreds of points next to the sensor) was a chain of 8-load round
    // trips of ONE thread before, most of this kernel's 25 us.
    constexpr int OPT = WT_OUT / SW_THREADS;
    float4 cen[OPT];
    uint32_t lj[OPT], lend[OPT];
#pragma unroll
    for (int r = 0; r < OPT; ++r) {
        const uint32_t o = (uint32_t)tid + (uint32_t)r * SW_THREADS;
        cen[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        lj[r] = lend[r] = 0u;
        if (o < n_out) { lj[r] = s_headpos[o]; lend[r] = o + 1u < n_out ? s_headpos[o + 1u] : n_valid; }
    }
    for (uint32_t base = 0; base < n_valid; base += (uint32_t)WT_STAGE) {
        const uint32_t top = base + (uint32_t)WT_STAGE < n_valid ? base + (uint32_t)WT_STAGE : n_valid;
#pragma unroll 6
        for (uint32_t e = base + (uint32_t)tid; e < top; e += SW_THREADS) {
            const float4 p = desk[s_hi[e] & ~HEAD];
            s_px[e - base] = p.x; s_py[e - base] = p.y; s_pz[e - base] = p.z;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < OPT; ++r) {
            const uint32_t stop = lend[r] < top ? lend[r] : top;
            float sx = cen[r].x, sy = cen[r].y, sz = cen[r].z;
            uint32_t j = lj[r];
            // (eight members' coordinates read ahead of the adds: one LDS latency per eight members instead of per member)
            for (; j + 8u <= stop; j += 8u) {
                float vx[8], vy[8], vz[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) { vx[c] = s_px[j - base + c]; vy[c] = s_py[j - base + c]; vz[c] = s_pz[j - base + c]; }
#pragma unroll
                for (int c = 0; c < 8; ++c) { sx += vx[c]; sy += vy[c]; sz += vz[c]; }
            }
            for (; j < stop; ++j) { sx += s_px[j - base]; sy += s_py[j - base]; sz += s_pz[j - base]; }
            cen[r].x = sx; cen[r].y = sy; cen[r].z = sz;
            lj[r] = j;
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < OPT; ++r) {
        const uint32_t o = (uint32_t)tid + (uint32_t)r * SW_THREADS;
        if (o >= n_out) continue;
        const float cnt = (float)(lend[r] - s_headpos[o]);
        cen[r] = make_float4(cen[r].x / cnt, cen[r].y / cnt, cen[r].z / cnt, __uint_as_float(o));
    }
    WT_STAMP(3);
    __syncthreads();   // nobody reads the element words any more: the region becomes output points | Morton keys | ranges
    float4* s_out = reinterpret_cast<float4*>(s_region);
    uint64_t* s_mkey = reinterpret_cast<uint64_t*>(s_region + WT_OUT * sizeof(float4));
    float* s_r2 = reinterpret_cast<float*>(s_region + WT_OUT * (sizeof(float4) + sizeof(uint64_t)));
#pragma unroll
    for (int r = 0; r < OPT; ++r) {
        const uint32_t o = (uint32_t)tid + (uint32_t)r * SW_THREADS;
        if (o < n_out) { s_out[o] = cen[r]; out_raw[o] = cen[r]; }
    }
    __syncthreads();
    if (n_out == 0) { report(0u, 0u); return; }
    // ---- Morton order of the output, tile ranges, tile order (scan_sort_small_kernel)
    const float ox = unflip_f32(bounds[0]), oy = unflip_f32(bounds[1]), oz = unflip_f32(bounds[2]);
    const uint32_t len2 = lds_sort_len(n_out);
    for (uint32_t i = tid; i < len2; i += SW_THREADS) {
        uint64_t k = ~0ull;
        if (i < n_out) {
            const float4 p = s_out[i];
            const float fx = fminf(fmaxf((p.x - ox) * inv_sort_cell, 0.f), 1023.f);
            const float fy = fminf(fmaxf((p.y - oy) * inv_sort_cell, 0.f), 1023.f);
            const float fz = fminf(fmaxf((p.z - oz) * inv_sort_cell, 0.f), 1023.f);
            const uint32_t key = spread10((uint32_t)fx) | (spread10((uint32_t)fy) << 1) | (spread10((uint32_t)fz) << 2);
            k = ((uint64_t)key << 32) | i;
        }
        s_mkey[i] = k;
    }
    __syncthreads();
    WT_STAMP(4);
    lds_bitonic_sort_u64<SW_THREADS>(s_mkey, len2, tid);
    WT_STAMP(5);
    for (uint32_t i = tid; i < n_out; i += SW_THREADS) {
        const float4 p = s_out[(uint32_t)s_mkey[i]];
        out_sorted[i] = p;
        s_r2[i] = p.x * p.x + p.y * p.y + p.z * p.z;
    }
    __syncthreads();
    if (tile_points != 0) {
        const uint32_t nt = (n_out + tile_points - 1) / tile_points;   // <= 1024 (the caller checks tile_points >= 4)
        uint64_t mine = ~0ull;
        if ((uint32_t)tid < nt) {
            float r2 = 0.f;
            const uint32_t b = (uint32_t)tid * tile_points;
            for (uint32_t i = 0; i < tile_points && b + i < n_out; ++i) r2 = fmaxf(r2, s_r2[b + i]);
            mine = ((uint64_t)(~__float_as_uint(r2)) << 32) | (uint32_t)tid;   // complement ascending = range descending; ties by tile
        }
        __syncthreads();
        if ((uint32_t)tid < nt) s_mkey[tid] = mine;
        __syncthreads();
        if ((uint32_t)tid < nt) {
            uint32_t rank = 0;
            for (uint32_t u = 0; u < nt; ++u) rank += s_mkey[u] < mine ? 1u : 0u;
            tile_order[rank] = (uint32_t)tid;
        }
    }
    WT_STAMP(6);
    report(n_out, 0u);
#undef WT_STAMP
}

int ScanStore::reserve_raw(size_t cap, size_t n_states) {
    if (cap > raw_cap) {
        size_t ncap = raw_cap ? raw_cap : 4096;
        while (ncap < cap) ncap *= 2;
        hipFree(d_in); hipFree(d_times); hipFree(d_desk); hipFree(d_vkeys); hipFree(d_vkeys_sorted); hipFree(d_vidx);
        hipFree(d_vidx_sorted); hipFree(d_heads); hipFree(d_hpos); hipFree(d_vsort_tmp); hipFree(d_vscan_tmp);
        d_in = d_desk = nullptr; d_times = nullptr; d_vkeys = d_vkeys_sorted = nullptr;
        d_vidx = d_vidx_sorted = d_heads = d_hpos = nullptr; d_vsort_tmp = d_vscan_tmp = nullptr;
        LV_HIP(hipMalloc(&d_in, ncap * sizeof(float4)));
        LV_HIP(hipMalloc(&d_times, ncap * sizeof(double)));
        LV_HIP(hipMalloc(&d_desk, ncap * sizeof(float4)));
        LV_HIP(hipMalloc(&d_vkeys, ncap * sizeof(uint64_t)));
        LV_HIP(hipMalloc(&d_vkeys_sorted, ncap * sizeof(uint64_t)));
        LV_HIP(hipMalloc(&d_vidx, ncap * sizeof(uint32_t)));
        LV_HIP(hipMalloc(&d_vidx_sorted, ncap * sizeof(uint32_t)));
        LV_HIP(hipMalloc(&d_heads, ncap * sizeof(uint32_t)));
        LV_HIP(hipMalloc(&d_hpos, ncap * sizeof(uint32_t)));
        vsort_tmp_bytes = 0;
        LV_HIP((hipError_t)hipcub::DeviceRadixSort::SortPairs(nullptr, vsort_tmp_bytes, d_vkeys, d_vkeys_sorted, d_vidx, d_vidx_sorted,
                                                               (int)ncap, 0, 63, (hipStream_t)0));
        LV_HIP(hipMalloc(&d_vsort_tmp, vsort_tmp_bytes));
        vscan_tmp_bytes = 0;
        LV_HIP((hipError_t)hipcub::DeviceScan::ExclusiveSum(nullptr, vscan_tmp_bytes, d_heads, d_hpos, (int)ncap, (hipStream_t)0));
        LV_HIP(hipMalloc(&d_vscan_tmp, vscan_tmp_bytes));
        raw_cap = ncap;
    }
    if (n_states > states_cap) {
        hipFree(d_states);
        d_states = nullptr;
        LV_HIP(hipMalloc(&d_states, n_states * sizeof(MotionState)));
        states_cap = n_states;
    }
    if (!d_bounds) LV_HIP(hipMalloc(&d_bounds, 8 * sizeof(unsigned)));
    return LV_OK;
}

// d_in / d_times / d_states hold the uploaded raw scan; leaves the de-skewed (and, if leaf > 0, voxel-grid
// down-sampled) points in d_raw[0 .. n) in output order and Morton-sorts them into d_sorted
int ScanStore::deskew_downsample(hipStream_t stream, uint32_t n_in, uint32_t n_states, const MotionState& xt2, float leaf,
                                 float sort_cell) {
    const int B = 256;
    const uint32_t grid = (n_in + B - 1) / B;
    int rc = reserve(n_in);
    if (rc) return rc;
    if (small_window_applies(n_in)) {
        bool fell_back = false;
        rc = window_small(stream, d_in, n_in, n_states, &xt2, leaf, sort_cell, &fell_back);
        if (rc || !fell_back) return rc;
    }
    float4* desk = leaf > 0.f ? d_desk : d_raw;
    hipLaunchKernelGGL(deskew_kernel, dim3(grid), dim3(B), 0, stream, d_in, d_times, n_in, d_states, n_states, xt2, desk);
    return voxel_and_sort(stream, n_in, leaf, sort_cell);
}

bool ScanStore::small_window_applies(uint32_t n_in) const {
    return small_enabled && n_in > 0 && n_in <= (uint32_t)SMALL_WINDOW && tile_points >= 2;
}

// the one-launch chain for windows of up to SMALL_WINDOW points; src = d_in with time stamps (xt2 != nullptr: de-skew) or
// the points as they are.  *fell_back: the kernel declined (leaf grid too large for the packed key): nothing was produced.
int ScanStore::window_small(hipStream_t stream, const float4* src, uint32_t n_in, uint32_t n_states, const MotionState* xt2, float leaf,
                            float sort_cell, bool* fell_back) {
    *fell_back = false;
    int rc = reserve(n_in);
    if (rc) return rc;
    rc = reserve_tiles((n_in + tile_points - 1) / tile_points);
    if (rc) return rc;
    n_tiles = 0;
    static const MotionState none{};
    LV_HIP(note_alloc(notes));
    const uint32_t seq = notes.next();
    hipLaunchKernelGGL(window_small_kernel, dim3(1), dim3(SW_THREADS), 0, stream, src, d_times, n_in, d_states, n_states,
                       xt2 ? *xt2 : none, xt2 ? 1 : 0, leaf, 1.0f / sort_cell, tile_points, d_raw, d_sorted, d_tile_order, d_bounds,
                       notes.d, seq);
    LV_HIP(hipGetLastError());
    uint32_t v[2] = {0, 0};   // points out (0: no finite point), status (1: declined)
    if (!note_wait(notes, 0, 2, seq, v, stream)) { set_error("window kernel did not report"); return LV_EHIP; }
    if (v[1]) { *fell_back = true; return LV_OK; }
    n = v[0];
    n_tiles = n ? (n + tile_points - 1) / tile_points : 0;
    return LV_OK;
}

bool ScanStore::large_window_applies(uint32_t n_in, uint32_t n_states, float leaf) const {
    return large_enabled && n_in > 0 && n_in <= (uint32_t)WT_IN && leaf > 0.f && tile_points >= 4 && n_states >= 2;
}

// windows beyond SMALL_WINDOW points (up to WT_IN), straight from the LiDAR buffer (cloud = its first point): deskew_keys_kernel,
// the library sort, window_tail_kernel.  d_bounds must hold "nothing seen" (CloudStore::window resets it when asked to).
// *fell_back: the tail declined (more than WT_OUT points out): nothing was produced, d_desk holds the de-skewed points.
int ScanStore::window_large(hipStream_t stream, const CloudPoint* cloud, uint32_t n_in, uint32_t n_states, const MotionState& xt2, float leaf,
                            float sort_cell, bool* fell_back) {
    *fell_back = false;
    int rc = reserve(n_in);
    if (rc) return rc;
    rc = reserve_tiles(((uint32_t)WT_OUT + tile_points - 1) / tile_points);
    if (rc) return rc;
    n_tiles = 0;
    LV_HIP(note_alloc(notes));
    if (!d_tail_clk && getenv("LV_TAIL_CLK")) { LV_HIP(hipMalloc(&d_tail_clk, 8 * sizeof(long long))); LV_HIP(hipMemset(d_tail_clk, 0, 8 * sizeof(long long))); }
    const uint32_t seq = notes.next();
    const int B = 256;
    const uint32_t grid = (n_in + B - 1) / B;
    hipLaunchKernelGGL(deskew_keys_kernel, dim3(grid), dim3(B), 0, stream, cloud, n_in, d_states, n_states, xt2, 1.0f / leaf, d_desk, d_vkeys, d_vidx,
                       d_bounds);
    size_t tmp = vsort_tmp_bytes;
    LV_HIP((hipError_t)hipcub::DeviceRadixSort::SortPairs(d_vsort_tmp, tmp, d_vkeys, d_vkeys_sorted, d_vidx, d_vidx_sorted, (int)n_in, 0, 63,
                                                           stream));
    hipLaunchKernelGGL(window_tail_kernel, dim3(1), dim3(SW_THREADS), 0, stream, d_desk, d_vkeys_sorted, d_vidx_sorted, n_in, 1.0f / sort_cell,
                       tile_points, d_raw, d_sorted, d_tile_order, d_bounds, notes.d, seq, d_tail_clk);
    LV_HIP(hipGetLastError());
    uint32_t v[2] = {0, 0};
    if (!note_wait(notes, 0, 2, seq, v, stream)) { set_error("window kernel did not report"); return LV_EHIP; }
    if (d_tail_clk) {   // LV_TAIL_CLK=1: phase stamps of the tail kernel (100 MHz wall clock), printed
        long long h[8];
        LV_HIP(hipMemcpy(h, d_tail_clk, sizeof(h), hipMemcpyDeviceToHost));
        fprintf(stderr, "window_tail us: keys->words %.2f  heads+scan %.2f  centroids %.2f  morton keys %.2f  sort %.2f  gather+tiles %.2f  (n_in %u, out %u; the sort: %lld shader cycles = %.0f MHz)\n",
                (h[1] - h[0]) / 100.0, (h[2] - h[1]) / 100.0, (h[3] - h[2]) / 100.0, (h[4] - h[3]) / 100.0, (h[5] - h[4]) / 100.0, (h[6] - h[5]) / 100.0, n_in, v[0], h[7], h[7] / ((h[5] - h[4]) / 100.0 + 1e-9));
    }
    if (v[1]) { *fell_back = true; return LV_OK; }
    n = v[0];
    n_tiles = n ? (n + tile_points - 1) / tile_points : 0;
    return LV_OK;
}

// the points (de-skewed, or uploaded as they are: Compensator::downsample on its own) wait in d_desk (leaf > 0) or
// d_raw (leaf <= 0): voxel grid, then the Morton order the search kernel wants
int ScanStore::voxel_and_sort(hipStream_t stream, uint32_t n_in, float leaf, float sort_cell, bool try_small) {
    if (try_small && small_window_applies(n_in)) {   // (Compensator::downsample on its own: the points wait in d_desk / d_raw)
        bool fell_back = false;
        int rc = window_small(stream, leaf > 0.f ? d_desk : d_raw, n_in, 0, nullptr, leaf, sort_cell, &fell_back);
        if (rc || !fell_back) return rc;
    }
    const int B = 256;
    const uint32_t grid = (n_in + B - 1) / B;
    float4* desk = leaf > 0.f ? d_desk : d_raw;
    hipLaunchKernelGGL(vg_bounds_init_kernel, dim3(1), dim3(64), 0, stream, d_bounds);
    const uint32_t bgrid = grid < 64u ? grid : 64u;   // (grid-stride: at most 64 workgroups touch the six device words)
    hipLaunchKernelGGL(vg_bounds_kernel, dim3(bgrid), dim3(B), 0, stream, desk, n_in, d_bounds);
    uint32_t n_out = n_in;
    if (leaf > 0.f) {
        hipLaunchKernelGGL(vg_keys_kernel, dim3(grid), dim3(B), 0, stream, d_desk, n_in, d_bounds, 1.0f / leaf, d_vkeys, d_vidx);
        size_t tmp = vsort_tmp_bytes;
        LV_HIP((hipError_t)hipcub::DeviceRadixSort::SortPairs(d_vsort_tmp, tmp, d_vkeys, d_vkeys_sorted, d_vidx, d_vidx_sorted, (int)n_in,
                                                               0, 63, stream));
        hipLaunchKernelGGL(vg_heads_kernel, dim3(grid), dim3(B), 0, stream, d_vkeys_sorted, n_in, d_heads);
        size_t stmp = vscan_tmp_bytes;
        LV_HIP((hipError_t)hipcub::DeviceScan::ExclusiveSum(d_vscan_tmp, stmp, d_heads, d_hpos, (int)n_in, stream));
        hipLaunchKernelGGL(vg_centroid_kernel, dim3(grid), dim3(B), 0, stream, d_desk, d_vkeys_sorted, d_vidx_sorted, d_heads, d_hpos,
                           n_in, d_raw, d_bounds + 6);
    }
    unsigned hb[8];   // bounds (6 words) + the number of leaves: one copy back
    LV_HIP(hipMemcpyAsync(hb, d_bounds, sizeof(hb), hipMemcpyDeviceToHost, stream));
    LV_HIP(hipStreamSynchronize(stream));
    if (leaf > 0.f) n_out = hb[6];
    n = n_out;
    if (hb[0] == 0xFFFFFFFFu) n = 0;  // no finite point survived
    if (n == 0) return LV_OK;
    auto unflip = [](unsigned u) { unsigned v = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u; float f; memcpy(&f, &v, 4); return f; };
    const float bmin[3] = {unflip(hb[0]), unflip(hb[1]), unflip(hb[2])};
    return sort(stream, bmin, sort_cell);   // Morton order for the match kernel, exactly as lv_scan_set does
}

int ScanStore::reserve(size_t cap) {
    if (cap <= capacity) return LV_OK;
    size_t ncap = capacity ? capacity : 4096;
    while (ncap < cap) ncap *= 2;
    hipFree(d_raw); hipFree(d_sorted); hipFree(d_keys); hipFree(d_keys_sorted); hipFree(d_idx); hipFree(d_idx_sorted);
    hipFree(d_sort_tmp);
    d_raw = d_sorted = nullptr; d_keys = d_keys_sorted = d_idx = d_idx_sorted = nullptr; d_sort_tmp = nullptr;
    capacity = 0;
    LV_HIP(hipMalloc(&d_raw, ncap * sizeof(float4)));
    LV_HIP(hipMalloc(&d_sorted, ncap * sizeof(float4)));
    LV_HIP(hipMalloc(&d_keys, ncap * sizeof(uint32_t)));
    LV_HIP(hipMalloc(&d_keys_sorted, ncap * sizeof(uint32_t)));
    LV_HIP(hipMalloc(&d_idx, ncap * sizeof(uint32_t)));
    LV_HIP(hipMalloc(&d_idx_sorted, ncap * sizeof(uint32_t)));
    LV_HIP((hipError_t)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_tmp_bytes, d_keys, d_keys_sorted, d_idx, d_idx_sorted,
                                                           (int)ncap, 0, 30, (hipStream_t)0));
    LV_HIP(hipMalloc(&d_sort_tmp, sort_tmp_bytes));
    capacity = ncap;
    return LV_OK;
}

void ScanStore::release() {
    hipFree(d_raw); hipFree(d_sorted); hipFree(d_keys); hipFree(d_keys_sorted); hipFree(d_idx); hipFree(d_idx_sorted);
    hipFree(d_sort_tmp);
    hipFree(d_in); hipFree(d_times); hipFree(d_desk); hipFree(d_vkeys); hipFree(d_vkeys_sorted); hipFree(d_vidx);
    hipFree(d_vidx_sorted); hipFree(d_heads); hipFree(d_hpos); hipFree(d_vsort_tmp); hipFree(d_vscan_tmp); hipFree(d_states);
    hipFree(d_bounds); hipFree(d_tile_order); hipFree(d_tail_clk);
    note_free(notes);
    *this = ScanStore();
}

// Small scans (the 0.01 s windows of a stream: a few hundred points after the voxel grid): keys, sort, gather, tile ranges
// and tile order by ONE workgroup — the five-launch chain below it costs ~35 us whatever the size.  Same keys, same
// (key, index) order as the stable radix sort, same tile keys: the result is the one the general path produces.
constexpr int SMALL_SCAN = 2048;
__global__ __launch_bounds__(1024) void scan_sort_small_kernel(const float4* __restrict__ pts, uint32_t n, float ox, float oy, float oz,
                                                               float inv_cell, uint32_t tile_points, float4* __restrict__ sorted,
                                                               uint32_t* __restrict__ tile_order) {
    __shared__ uint64_t s_key[SMALL_SCAN];
    __shared__ float s_r2[SMALL_SCAN];
    const int tid = threadIdx.x;
#pragma unroll
    for (int h = 0; h < SMALL_SCAN / 1024; ++h) {
        const uint32_t i = (uint32_t)tid + (uint32_t)h * 1024u;
        uint64_t k = ~0ull;
        if (i < n) {
            const float4 p = pts[i];
            const float fx = fminf(fmaxf((p.x - ox) * inv_cell, 0.f), 1023.f);
            const float fy = fminf(fmaxf((p.y - oy) * inv_cell, 0.f), 1023.f);
            const float fz = fminf(fmaxf((p.z - oz) * inv_cell, 0.f), 1023.f);
            const uint32_t key = spread10((uint32_t)fx) | (spread10((uint32_t)fy) << 1) | (spread10((uint32_t)fz) << 2);
            k = ((uint64_t)key << 32) | i;
        }
        s_key[i] = k;
    }
    __syncthreads();
    // bitonic network over the smallest power of two that holds n (padding sorts last)
    lds_bitonic_sort_u64<1024>(s_key, lds_sort_len(n), tid);
#pragma unroll
    for (int h = 0; h < SMALL_SCAN / 1024; ++h) {
        const uint32_t i = (uint32_t)tid + (uint32_t)h * 1024u;
        if (i < n) {
            const float4 p = pts[(uint32_t)s_key[i]];
            sorted[i] = p;
            s_r2[i] = p.x * p.x + p.y * p.y + p.z * p.z;
        }
    }
    __syncthreads();
    if (tile_points == 0) return;
    const uint32_t nt = (n + tile_points - 1) / tile_points;   // <= SMALL_SCAN / tile_points
    uint64_t mine = ~0ull;
    if ((uint32_t)tid < nt) {
        float r2 = 0.f;
        const uint32_t b = (uint32_t)tid * tile_points;
        for (uint32_t i = 0; i < tile_points && b + i < n; ++i) r2 = fmaxf(r2, s_r2[b + i]);
        mine = ((uint64_t)(~__float_as_uint(r2)) << 32) | (uint32_t)tid;   // complement ascending = range descending; ties by tile
    }
    __syncthreads();
    if ((uint32_t)tid < nt) s_key[tid] = mine;   // (nt <= 1024 whenever tile_points >= 2)
    __syncthreads();
    if ((uint32_t)tid < nt) {
        uint32_t rank = 0;
        for (uint32_t u = 0; u < nt; ++u) rank += s_key[u] < mine ? 1u : 0u;
        tile_order[rank] = (uint32_t)tid;
    }
}

int ScanStore::sort(hipStream_t stream, const float bbox_min[3], float cell) {
    if (n == 0) return LV_OK;
    if (n <= (uint32_t)SMALL_SCAN && tile_points != 1) {
        n_tiles = 0;
        const uint32_t nt = tile_points ? (n + tile_points - 1) / tile_points : 0;
        int rc = reserve_tiles(nt);
        if (rc) return rc;
        hipLaunchKernelGGL(scan_sort_small_kernel, dim3(1), dim3(1024), 0, stream, d_raw, n, bbox_min[0], bbox_min[1], bbox_min[2],
                           1.0f / cell, tile_points, d_sorted, d_tile_order);
        LV_HIP(hipGetLastError());
        n_tiles = nt;
        return LV_OK;
    }
    const int B = 256;
    const uint32_t grid = (n + B - 1) / B;
    hipLaunchKernelGGL(scan_keys_kernel, dim3(grid), dim3(B), 0, stream, d_raw, n, bbox_min[0], bbox_min[1], bbox_min[2],
                       1.0f / cell, d_keys, d_idx);
    size_t tmp = sort_tmp_bytes;
    LV_HIP((hipError_t)hipcub::DeviceRadixSort::SortPairs(d_sort_tmp, tmp, d_keys, d_keys_sorted, d_idx, d_idx_sorted, (int)n, 0,
                                                           30, stream));
    hipLaunchKernelGGL(scan_gather_kernel, dim3(grid), dim3(B), 0, stream, d_raw, d_idx_sorted, n, d_sorted);
    LV_HIP(hipGetLastError());
    return order_tiles(stream, tile_points);
}

int ScanStore::reserve_tiles(uint32_t nt) {
    if (nt <= tile_cap) return LV_OK;
    hipFree(d_tile_order);
    d_tile_order = nullptr;
    tile_cap = 0;
    uint32_t cap = 1024;
    while (cap < nt) cap *= 2;
    LV_HIP(hipMalloc(&d_tile_order, cap * sizeof(uint32_t)));
    tile_cap = cap;
    return LV_OK;
}

// Dispatch order of the search kernel's point tiles: tiles whose points lie farthest from the sensor first.
// A pose error moves a point by (translation error + rotation error x range), so the far tiles are the ones
// that miss bucket level 0 in the first pass and run several times longer; started first they overlap with the
// cheap tiles instead of forming the tail of the launch (longest-processing-time-first; a scheduling hint only,
// results do not depend on it).
__global__ void tile_range_keys_kernel(const float4* __restrict__ sorted, uint32_t n, uint32_t tile_points, uint32_t ntiles,
                                       uint32_t* __restrict__ keys, uint32_t* __restrict__ ids) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    float r2 = 0.f;
    const uint32_t b = t * tile_points;
    for (uint32_t i = 0; i < tile_points && b + i < n; ++i) {
        const float4 p = sorted[b + i];
        r2 = fmaxf(r2, p.x * p.x + p.y * p.y + p.z * p.z);
    }
    keys[t] = ~__float_as_uint(r2);   // ascending sort of the complement = descending range (r2 >= 0)
    ids[t] = t;
}

int ScanStore::order_tiles(hipStream_t stream, uint32_t tile_points) {
    n_tiles = 0;
    if (n == 0 || tile_points == 0) return LV_OK;
    const uint32_t nt = (n + tile_points - 1) / tile_points;
    int rc = reserve_tiles(nt);
    if (rc) return rc;
    // d_keys / d_idx / d_keys_sorted are free again once the points are gathered (nt <= n <= capacity)
    hipLaunchKernelGGL(tile_range_keys_kernel, dim3((nt + 255) / 256), dim3(256), 0, stream, d_sorted, n, tile_points, nt, d_keys, d_idx);
    size_t tmp = sort_tmp_bytes;
    LV_HIP((hipError_t)hipcub::DeviceRadixSort::SortPairs(d_sort_tmp, tmp, d_keys, d_keys_sorted, d_idx, d_tile_order, (int)nt, 0, 32,
                                                           stream));
    n_tiles = nt;
    return LV_OK;
}

}  // namespace lv

// lv_map.hip — map residency: multi-level voxel hash over Morton-sorted points in HBM.
//
// Replaces ikd-Tree's Build (call site reference src/Modules/Mapper.cpp:68-71): instead of a
// pointer kd-tree (one point per node, ~100 B/node, log2(M) dependent loads per query) the map is
// held as
//   * `sorted`: float4 {x,y,z, original index} in Morton order of the level-0 voxel coordinates —
//     every voxel of every level (edge voxel_size * 2^l) is one contiguous, coalesced range;
//   * per level l an open-addressing hash table {packed voxel coords -> (start, count)}, 16 B per
//     slot so that one probe is one dwordx4 load, load factor <= 0.25;
//   * `orig`: float4 in insertion order (index space of the kNN results).
// Exactness of the search that uses these tables is argued in lv_match.hip.
#include "lv_host.hpp"

#include <hipcub/hipcub.hpp>

namespace lv {

__global__ void map_keys_kernel(const float4* __restrict__ pts, uint32_t m, float ox, float oy, float oz,
                                float inv_cell, uint64_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    float4 p = pts[i];
    uint32_t cx = (uint32_t)cell_coord(p.x, ox, inv_cell);
    uint32_t cy = (uint32_t)cell_coord(p.y, oy, inv_cell);
    uint32_t cz = (uint32_t)cell_coord(p.z, oz, inv_cell);
    keys[i] = morton3(cx, cy, cz);
    idx[i] = i;
}

__global__ void map_gather_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ idx_sorted, uint32_t m,
                                  float4* __restrict__ sorted) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    uint32_t j = idx_sorted[i];
    float4 p = pts[j];
    p.w = __uint_as_float(j);
    sorted[i] = p;
}

// number of levels (from 0) at which element i starts a new voxel
__device__ __forceinline__ int head_levels(const uint64_t* keys, uint32_t i, int n_levels) {
    if (i == 0) return n_levels;
    uint64_t x = keys[i] ^ keys[i - 1];
    if (x == 0) return 0;
    int hb = 63 - __clzll((long long)x);
    int nl = hb / 3 + 1;
    return nl < n_levels ? nl : n_levels;
}

__global__ void map_count_heads_kernel(const uint64_t* __restrict__ keys, uint32_t m, int n_levels,
                                       uint32_t* __restrict__ counts) {
    __shared__ uint32_t s_cnt[MAX_LEVELS];
    if (threadIdx.x < MAX_LEVELS) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) {
        int nl = head_levels(keys, i, n_levels);
        for (int l = 0; l < nl; ++l) atomicAdd(&s_cnt[l], 1u);
    }
    __syncthreads();
    if (threadIdx.x < n_levels && s_cnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x], s_cnt[threadIdx.x]);
}

struct GridLevelW {
    uint4* table;
    uint32_t mask;
    uint32_t shift;
};

struct TablePtrs {
    uint4* table[MAX_LEVELS];
    uint32_t mask[MAX_LEVELS];
    uint32_t shift[MAX_LEVELS];
};

__global__ void map_insert_kernel(const uint64_t* __restrict__ keys, uint32_t m, int n_levels, TablePtrs tp) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    int nl = head_levels(keys, i, n_levels);
    uint64_t k0 = keys[i];
    for (int l = 0; l < nl; ++l) {
        uint64_t prefix = k0 >> (3 * l);
        // end = first j > i whose level-l prefix differs (keys are sorted)
        uint32_t lo = i + 1, hi = m;
        while (lo < hi) {
            uint32_t mid = lo + ((hi - lo) >> 1);
            if ((keys[mid] >> (3 * l)) == prefix) lo = mid + 1;
            else hi = mid;
        }
        uint32_t cx = compact21(k0) >> l, cy = compact21(k0 >> 1) >> l, cz = compact21(k0 >> 2) >> l;
        uint64_t key = pack_cell(cx, cy, cz);
        uint32_t slot = hash_cell(key, tp.shift[l]) & tp.mask[l];
        uint4* tbl = tp.table[l];
        for (;;) {
            unsigned long long* kp = reinterpret_cast<unsigned long long*>(&tbl[slot]);
            unsigned long long old = atomicCAS(kp, (unsigned long long)EMPTY_KEY, (unsigned long long)key);
            if (old == (unsigned long long)EMPTY_KEY) {
                tbl[slot].z = i;
                tbl[slot].w = lo - i;
                break;
            }
            slot = (slot + 1) & tp.mask[l];
        }
    }
}


// ---- neighbourhood buckets -----------------------------------------------------------------------
// For every level-l voxel c whose 3x3x3 block holds at least one point (the occupied voxels dilated by
// one), the points of that block are copied into one contiguous run ("bucket").  A query then needs ONE
// hash probe and ONE coalesced stream per level instead of 27 probes + 27 short dependent gathers; the
// search region, and therefore the exactness argument of lv_match.hip, is unchanged.  Every point lands
// in 27 buckets per level: HBM capacity (288 GB) is traded for latency.
__device__ __forceinline__ bool probe_cell(const GridLevelW& g, uint64_t key, uint32_t& start, uint32_t& count) {
    uint32_t slot = hash_cell(key, g.shift) & g.mask;
    for (;;) {
        const uint4 e = g.table[slot];
        const uint64_t ek = (uint64_t)e.x | ((uint64_t)e.y << 32);
        if (ek == key) { start = e.z; count = e.w; return true; }
        if (ek == EMPTY_KEY) { start = 0; count = 0; return false; }
        slot = (slot + 1) & g.mask;
    }
}

// pass 1: every occupied voxel registers its 27 neighbours (incl. itself) in the bucket table
__global__ void bucket_register_kernel(GridLevelW occ, uint32_t occ_slots, GridLevelW bt, uint32_t* __restrict__ cell_slots,
                                       uint32_t cell_cap, uint32_t* __restrict__ flags) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t slot = t / 27, nb = t % 27;
    if (slot >= occ_slots) return;
    const uint4 e = occ.table[slot];
    const uint64_t key = (uint64_t)e.x | ((uint64_t)e.y << 32);
    if (key == EMPTY_KEY) return;
    const uint32_t cx = (uint32_t)(key & 0x1fffff), cy = (uint32_t)((key >> 21) & 0x1fffff), cz = (uint32_t)((key >> 42) & 0x1fffff);
    const int dz = (int)nb / 9 - 1, dy = ((int)nb / 3) % 3 - 1, dx = (int)nb % 3 - 1;
    const uint32_t nx = cx + dx, ny = cy + dy, nz = cz + dz;
    if (nx >= (1u << 21) || ny >= (1u << 21) || nz >= (1u << 21)) return;
    const uint64_t nkey = pack_cell(nx, ny, nz);
    uint32_t bs = hash_cell(nkey, bt.shift) & bt.mask;
    for (uint32_t probes = 0; probes <= bt.mask; ++probes) {
        unsigned long long* kp = reinterpret_cast<unsigned long long*>(&bt.table[bs]);
        const unsigned long long old = atomicCAS(kp, (unsigned long long)EMPTY_KEY, (unsigned long long)nkey);
        if (old == (unsigned long long)EMPTY_KEY) {
            const uint32_t idx = atomicAdd(&flags[0], 1u);
            if (idx < cell_cap) cell_slots[idx] = bs;
            else flags[1] = 1;
            return;
        }
        if (old == (unsigned long long)nkey) return;
        bs = (bs + 1) & bt.mask;
    }
    flags[1] = 1;
}

// pass 2 (count) / pass 3 (fill): one 64-lane workgroup per bucket voxel; lane c < 27 owns neighbour c
template <bool FILL>
__global__ __launch_bounds__(64) void map_bucket_kernel(GridLevelW occ, GridLevelW bt, const uint32_t* __restrict__ cell_slots,
                                                        uint32_t n_cells, const float4* __restrict__ sorted,
                                                        uint32_t* __restrict__ bcount, const uint32_t* __restrict__ boff,
                                                        float4* __restrict__ bucket) {
    const uint32_t cell = blockIdx.x;
    if (cell >= n_cells) return;
    const int lane = threadIdx.x;
    const uint32_t slot = cell_slots[cell];
    const uint4 e = bt.table[slot];
    const uint64_t key = (uint64_t)e.x | ((uint64_t)e.y << 32);
    const uint32_t cx = (uint32_t)(key & 0x1fffff), cy = (uint32_t)((key >> 21) & 0x1fffff), cz = (uint32_t)((key >> 42) & 0x1fffff);
    uint32_t start = 0, count = 0;
    if (lane < 27) {
        const int dz = lane / 9 - 1, dy = (lane / 3) % 3 - 1, dx = lane % 3 - 1;
        const uint32_t nx = cx + dx, ny = cy + dy, nz = cz + dz;
        if (nx < (1u << 21) && ny < (1u << 21) && nz < (1u << 21)) probe_cell(occ, pack_cell(nx, ny, nz), start, count);
    }
    uint32_t incl = count;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    const uint32_t total = __shfl(incl, 63);
    if (!FILL) {
        if (lane == 0) bcount[cell] = total;
    } else {
        const uint32_t base = boff[cell] + (incl - count);
        for (uint32_t j = 0; j < count; ++j) bucket[(size_t)base + j] = sorted[start + j];
        if (lane == 0) { bt.table[slot].z = boff[cell]; bt.table[slot].w = total; }
    }
}

// Sort every bucket by ORIGINAL index (ascending).  The match kernel orders candidates by
// (distance, position-in-bucket); with this layout that equals the reference's (distance, index)
// order, ties included.  Bitonic network in the "flip" form (always-ascending compare-exchange,
// partners beyond n skipped == padding with +inf), valid for any n.  One workgroup per bucket.
// small buckets (<= 64 points, the bulk at level 0): one wavefront per bucket, bitonic network through
// cross-lane shuffles, no LDS, no barriers
__global__ __launch_bounds__(256) void bucket_sort_wave_kernel(const uint32_t* __restrict__ bcount,
                                                               const uint32_t* __restrict__ boff, uint32_t n_cells,
                                                               float4* __restrict__ bucket) {
    const uint32_t cell = blockIdx.x * 4u + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (cell >= n_cells) return;
    const uint32_t n = bcount[cell];
    if (n < 2 || n > 64) return;
    float4* g = bucket + boff[cell];
    float4 v = (uint32_t)lane < n ? g[lane] : make_float4(0.f, 0.f, 0.f, __uint_as_float(0xFFFFFFFFu));
    for (int k = 2; k <= 64; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int partner = lane ^ j;
            float4 o;
            o.x = __shfl(v.x, partner); o.y = __shfl(v.y, partner); o.z = __shfl(v.z, partner); o.w = __shfl(v.w, partner);
            const bool take_min = ((lane & k) == 0) == ((lane & j) == 0);
            const uint32_t a = __float_as_uint(v.w), b = __float_as_uint(o.w);
            const bool use_other = take_min ? (b < a) : (b > a);
            if (use_other) v = o;
        }
    }
    if ((uint32_t)lane < n) g[lane] = v;
}

constexpr int BSORT_THREADS = 256;
constexpr uint32_t BSORT_LDS = 2048;
__global__ __launch_bounds__(BSORT_THREADS) void bucket_sort_kernel(const uint32_t* __restrict__ bcount,
                                                                    const uint32_t* __restrict__ boff, uint32_t n_cells,
                                                                    float4* __restrict__ bucket) {
    __shared__ float4 s_pts[BSORT_LDS];
    const uint32_t cell = blockIdx.x;
    if (cell >= n_cells) return;
    const uint32_t n = bcount[cell];
    if (n <= 64) return;  // handled by bucket_sort_wave_kernel
    float4* g = bucket + boff[cell];
    if (n <= BSORT_LDS) {
        // rank by counting: original indices are unique, so the rank of a point is the number of points with a
        // smaller index — n compares per point out of LDS (broadcast reads), no barriers between steps; for the
        // few-hundred-point buckets this beats the bitonic network (log^2 n barrier-separated stages) severalfold
        __shared__ uint32_t s_idx[BSORT_LDS];
        for (uint32_t i = threadIdx.x; i < n; i += BSORT_THREADS) {
            const float4 p = g[i];
            s_pts[i] = p;
            s_idx[i] = __float_as_uint(p.w);
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += BSORT_THREADS) {
            const uint32_t mine = s_idx[i];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < n; ++j) rank += s_idx[j] < mine ? 1u : 0u;
            g[rank] = s_pts[i];
        }
        return;
    }
    float4* a = g;
    for (uint32_t k = 2; (k >> 1) < n; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            const bool flip = (j == (k >> 1));
            for (uint32_t i = threadIdx.x; i < n; i += BSORT_THREADS) {
                const uint32_t l = flip ? (i ^ (k - 1)) : (i ^ j);
                if (l > i && l < n) {
                    const float4 x = a[i], y = a[l];
                    if (__float_as_uint(x.w) > __float_as_uint(y.w)) { a[i] = y; a[l] = x; }
                }
            }
            __syncthreads();
        }
    }
}

// ---- KD_TREE::Add_Points(points, downsample = true) ------------------------------------------------
// [UPSTREAM-RECALL ikd-Tree; call site reference src/Modules/Mapper.cpp:73-76, box_length 0.2 m :65].
// Upstream processes the new points one by one: the point nearest to the centre of p's 0.2 m box among
// {p} U (current occupants) survives if the box held more than one point or p is that nearest point
// (occupants must be STRICTLY closer to beat p), otherwise the box is left alone.  The batch form below
// reproduces exactly that sequence: all points are keyed by their box, a stable radix sort groups each box
// with its old occupants first (old order) and the new points after (input order), and one lane replays
// the sequential rule inside its box.  Boxes are independent, so the result equals the sequential one.
__device__ __forceinline__ int box_coord(float v, float len) {
    float f = floorf(v / len);
    f = fminf(fmaxf(f, -1048000.0f), 1048000.0f);
    return (int)f + CELL_OFFSET;
}
__device__ __forceinline__ float box_center_dist(float4 p, float len) {
    const float c[3] = {p.x, p.y, p.z};
    float mid[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float vmin = floorf(c[a] / len) * len;
        const float vmax = vmin + len;
        mid[a] = (float)((double)vmin + (double)(vmax - vmin) / 2.0);
    }
    const float dx = p.x - mid[0], dy = p.y - mid[1], dz = p.z - mid[2];  // calc_dist(point, mid_point)
    const float sx = dx * dx, sy = dy * dy, sz = dz * dz;
    const float s = sx + sy;
    return s + sz;
}

__global__ void box_keys_kernel(const float4* __restrict__ pts, uint32_t total, float len, uint64_t* __restrict__ keys,
                                uint32_t* __restrict__ idx, uint32_t* __restrict__ alive, uint32_t m_old) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float4 p = pts[i];
    keys[i] = pack_cell((uint32_t)box_coord(p.x, len), (uint32_t)box_coord(p.y, len), (uint32_t)box_coord(p.z, len));
    idx[i] = i;
    alive[i] = i < m_old ? 1u : 0u;
}

__global__ void box_rule_kernel(const float4* __restrict__ pts, const uint64_t* __restrict__ keys_sorted,
                                const uint32_t* __restrict__ idx_sorted, uint32_t total, uint32_t m_old, float len,
                                uint32_t* __restrict__ alive) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const uint64_t key = keys_sorted[i];
    if (i > 0 && keys_sorted[i - 1] == key) return;  // not the head of its box
    uint32_t end = i + 1;
    while (end < total && keys_sorted[end] == key) ++end;
    // old occupants come first (ascending index), new points after
    uint32_t nE = 0;
    while (i + nE < end && idx_sorted[i + nE] < m_old) ++nE;
    if (i + nE == end) return;  // box untouched by new points
    bool multi = nE > 1;
    uint32_t cur = nE == 1 ? idx_sorted[i] : 0xFFFFFFFFu;
    float dcur = nE == 1 ? box_center_dist(pts[cur], len) : 0.f;
    for (uint32_t j = i + nE; j < end; ++j) {
        const uint32_t p = idx_sorted[j];
        uint32_t best = p;
        float md = box_center_dist(pts[p], len);
        if (multi) {
            for (uint32_t e = i; e < i + nE; ++e) {
                const uint32_t ei = idx_sorted[e];
                const float d = box_center_dist(pts[ei], len);
                if (d < md) { md = d; best = ei; }
            }
        } else if (cur != 0xFFFFFFFFu) {
            if (dcur < md) { md = dcur; best = cur; }
        }
        if (multi || best == p) {
            if (multi) { for (uint32_t e = i; e < i + nE; ++e) alive[idx_sorted[e]] = 0u; }
            else if (cur != 0xFFFFFFFFu) alive[cur] = 0u;
            alive[best] = 1u;
            cur = best;
            dcur = md;
            multi = false;
        }
    }
}

__global__ void box_compact_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ alive,
                                   const uint32_t* __restrict__ apos, uint32_t total, float4* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    if (alive[i]) out[apos[i]] = pts[i];
}

int MapStore::add_downsample(hipStream_t stream, uint32_t k, float box_length) {
    const uint32_t total = m + k;
    if (k == 0) return LV_OK;
    const int B = 256;
    const uint32_t grid = (total + B - 1) / B;
    hipLaunchKernelGGL(box_keys_kernel, dim3(grid), dim3(B), 0, stream, d_orig, total, box_length, d_keys, d_idx, d_alive, m);
    size_t tmp = sort_tmp_bytes;
    LV_HIP((hipError_t)hipcub::DeviceRadixSort::SortPairs(d_sort_tmp, tmp, d_keys, d_keys_sorted, d_idx, d_idx_sorted, (int)total,
                                                           0, 63, stream));
    hipLaunchKernelGGL(box_rule_kernel, dim3(grid), dim3(B), 0, stream, d_orig, d_keys_sorted, d_idx_sorted, total, m, box_length,
                       d_alive);
    size_t stmp = ascan_tmp_bytes;
    LV_HIP((hipError_t)hipcub::DeviceScan::ExclusiveSum(d_ascan_tmp, stmp, d_alive, d_apos, (int)total, stream));
    hipLaunchKernelGGL(box_compact_kernel, dim3(grid), dim3(B), 0, stream, d_orig, d_alive, d_apos, total, d_orig2);
    uint32_t last_pos = 0, last_alive = 0;
    LV_HIP(hipMemcpyAsync(&last_pos, d_apos + (total - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    LV_HIP(hipMemcpyAsync(&last_alive, d_alive + (total - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    LV_HIP(hipStreamSynchronize(stream));
    float4* t = d_orig;
    d_orig = d_orig2;
    d_orig2 = t;
    m = last_pos + last_alive;
    return LV_OK;
}

static inline uint32_t next_pow2(uint64_t v) {
    uint32_t p = 64;
    while (p < v) p <<= 1;
    return p;
}

int MapStore::reserve(size_t cap) {
    if (cap <= capacity) return LV_OK;
    size_t ncap = capacity ? capacity : 4096;
    while (ncap < cap) ncap *= 2;
    float4* n_orig = nullptr;
    LV_HIP(hipMalloc(&n_orig, ncap * sizeof(float4)));
    if (d_orig && m) LV_HIP(hipMemcpyAsync(n_orig, d_orig, (size_t)m * sizeof(float4), hipMemcpyDeviceToDevice, 0));
    LV_HIP(hipDeviceSynchronize());
    if (d_orig) hipFree(d_orig);
    d_orig = n_orig;
    if (d_sorted) hipFree(d_sorted);
    if (d_keys) hipFree(d_keys);
    if (d_keys_sorted) hipFree(d_keys_sorted);
    if (d_idx) hipFree(d_idx);
    if (d_idx_sorted) hipFree(d_idx_sorted);
    if (d_orig2) hipFree(d_orig2);
    if (d_alive) hipFree(d_alive);
    if (d_apos) hipFree(d_apos);
    if (d_ascan_tmp) hipFree(d_ascan_tmp);
    d_ascan_tmp = nullptr;
    LV_HIP(hipMalloc(&d_orig2, ncap * sizeof(float4)));
    LV_HIP(hipMalloc(&d_alive, ncap * sizeof(uint32_t)));
    LV_HIP(hipMalloc(&d_apos, ncap * sizeof(uint32_t)));
    ascan_tmp_bytes = 0;
    LV_HIP((hipError_t)hipcub::DeviceScan::ExclusiveSum(nullptr, ascan_tmp_bytes, d_alive, d_apos, (int)ncap, (hipStream_t)0));
    LV_HIP(hipMalloc(&d_ascan_tmp, ascan_tmp_bytes));
    LV_HIP(hipMalloc(&d_sorted, ncap * sizeof(float4)));
    LV_HIP(hipMalloc(&d_keys, ncap * sizeof(uint64_t)));
    LV_HIP(hipMalloc(&d_keys_sorted, ncap * sizeof(uint64_t)));
    LV_HIP(hipMalloc(&d_idx, ncap * sizeof(uint32_t)));
    LV_HIP(hipMalloc(&d_idx_sorted, ncap * sizeof(uint32_t)));
    if (d_sort_tmp) hipFree(d_sort_tmp);
    d_sort_tmp = nullptr;
    sort_tmp_bytes = 0;
    LV_HIP((hipError_t)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_tmp_bytes, d_keys, d_keys_sorted, d_idx, d_idx_sorted,
                                                           (int)ncap, 0, 63, (hipStream_t)0));
    LV_HIP(hipMalloc(&d_sort_tmp, sort_tmp_bytes));
    capacity = ncap;
    return LV_OK;
}

void MapStore::release() {
    hipFree(d_orig); hipFree(d_sorted); hipFree(d_keys); hipFree(d_keys_sorted); hipFree(d_idx); hipFree(d_idx_sorted);
    hipFree(d_sort_tmp); hipFree(d_counts);
    hipFree(d_orig2); hipFree(d_alive); hipFree(d_apos); hipFree(d_ascan_tmp);
    hipFree(d_cell_slots); hipFree(d_flags); hipFree(d_bcount); hipFree(d_boff); hipFree(d_scan_tmp);
    for (int l = 0; l < MAX_BUCKET_LEVELS; ++l) { hipFree(d_btable[l]); hipFree(d_bxyz[l]); hipFree(d_bidx[l]); hipFree(d_bucket4[l]); }
    hipFree(d_bucket_tmp);
    for (int l = 0; l < MAX_LEVELS; ++l) hipFree(d_tables[l]);
    *this = MapStore();
}

// Rebuild the search structure over d_orig[0..m).  bbox = host-computed bounds of the points.
int MapStore::rebuild(hipStream_t stream, float cell, const float bbox_min[3], const float bbox_max[3]) {
    view = MapView();
    view.m = m;
    view.cell = cell;
    view.inv_cell = 1.0f / cell;
    if (m == 0) return LV_OK;
    if (!origin_set) {  // origin = bbox centre snapped to the level-0 lattice; kept for the map's lifetime
        for (int a = 0; a < 3; ++a) origin[a] = floorf(0.5f * (bbox_min[a] + bbox_max[a]) / cell) * cell;
        origin_set = true;
    }
    float ext_cells = 1.f;
    for (int a = 0; a < 3; ++a) {
        view.origin[a] = origin[a];
        float lo = floorf((bbox_min[a] - origin[a]) / cell), hi = floorf((bbox_max[a] - origin[a]) / cell);
        if (!(fabsf(lo) < (float)CELL_FAR) || !(fabsf(hi) < (float)CELL_FAR)) {
            set_error("map extent exceeds +-%d voxels of %.3f m around the map origin", CELL_FAR, cell);
            return LV_ERANGE;
        }
        ext_cells = fmaxf(ext_cells, hi - lo + 1.f);
    }
    int n_levels = 1;
    while ((1 << (n_levels - 1)) < (int)ext_cells && n_levels < MAX_LEVELS) ++n_levels;
    view.n_levels = n_levels;

    const int B = 256;
    const uint32_t grid = (m + B - 1) / B;
    hipLaunchKernelGGL(map_keys_kernel, dim3(grid), dim3(B), 0, stream, d_orig, m, origin[0], origin[1], origin[2],
                       view.inv_cell, d_keys, d_idx);
    size_t tmp = sort_tmp_bytes;
    LV_HIP((hipError_t)hipcub::DeviceRadixSort::SortPairs(d_sort_tmp, tmp, d_keys, d_keys_sorted, d_idx, d_idx_sorted, (int)m,
                                                           0, 63, stream));
    hipLaunchKernelGGL(map_gather_kernel, dim3(grid), dim3(B), 0, stream, d_orig, d_idx_sorted, m, d_sorted);
    if (!d_counts) LV_HIP(hipMalloc(&d_counts, MAX_LEVELS * sizeof(uint32_t)));
    LV_HIP(hipMemsetAsync(d_counts, 0, MAX_LEVELS * sizeof(uint32_t), stream));
    hipLaunchKernelGGL(map_count_heads_kernel, dim3(grid), dim3(B), 0, stream, d_keys_sorted, m, n_levels, d_counts);
    uint32_t counts[MAX_LEVELS];
    LV_HIP(hipMemcpyAsync(counts, d_counts, sizeof(counts), hipMemcpyDeviceToHost, stream));
    LV_HIP(hipStreamSynchronize(stream));

    TablePtrs tp{};
    for (int l = 0; l < n_levels; ++l) {
        uint32_t size = next_pow2((uint64_t)counts[l] * 4);
        if (size > table_size[l]) {
            if (d_tables[l]) hipFree(d_tables[l]);
            LV_HIP(hipMalloc(&d_tables[l], (size_t)size * sizeof(uint4)));
            table_size[l] = size;
        }
        size = table_size[l];
        LV_HIP(hipMemsetAsync(d_tables[l], 0xFF, (size_t)size * sizeof(uint4), stream));
        int lg = 0;
        while ((1u << lg) < size) ++lg;
        tp.table[l] = d_tables[l];
        tp.mask[l] = size - 1;
        tp.shift[l] = 64 - lg;
        view.lv[l].table = d_tables[l];
        view.lv[l].mask = size - 1;
        view.lv[l].shift = 64 - lg;
        n_cells[l] = counts[l];
    }
    hipLaunchKernelGGL(map_insert_kernel, dim3(grid), dim3(B), 0, stream, d_keys_sorted, m, n_levels, tp);
    LV_HIP(hipGetLastError());
    view.sorted = d_sorted;
    view.orig = d_orig;
    view.n_bucket_levels = n_levels < MAX_BUCKET_LEVELS ? n_levels : MAX_BUCKET_LEVELS;
    for (int l = 0; l < view.n_bucket_levels; ++l) {
        int rc = build_buckets(stream, l, counts[l]);
        if (rc) return rc;
    }
    return LV_OK;
}

// the sorted float4 buckets -> 12-byte points (what the search kernel streams) + a parallel index array
__global__ void bucket_pack_kernel(const float4* __restrict__ in, uint32_t n, float* __restrict__ xyz, uint32_t* __restrict__ idx) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = in[i];
    xyz[(size_t)i * 3] = p.x;
    xyz[(size_t)i * 3 + 1] = p.y;
    xyz[(size_t)i * 3 + 2] = p.z;
    idx[i] = __float_as_uint(p.w);
}

int MapStore::build_buckets(hipStream_t stream, int level, uint32_t n_occupied) {
    if (!d_flags) LV_HIP(hipMalloc(&d_flags, 2 * sizeof(uint32_t)));
    GridLevelW occ{d_tables[level], view.lv[level].mask, view.lv[level].shift};
    const uint32_t occ_slots = view.lv[level].mask + 1;
    uint32_t size = next_pow2((uint64_t)n_occupied * 16);  // dilation factor <= 8 keeps the load <= 0.5
    if (size < btable_size[level]) size = btable_size[level];
    for (;;) {
        if (size > btable_size[level]) {
            hipFree(d_btable[level]);
            d_btable[level] = nullptr;
            LV_HIP(hipMalloc(&d_btable[level], (size_t)size * sizeof(uint4)));
            btable_size[level] = size;
        }
        const size_t need_cells = (size_t)size / 2;
        if (need_cells > cells_cap) {
            hipFree(d_cell_slots); hipFree(d_bcount); hipFree(d_boff); hipFree(d_scan_tmp);
            d_cell_slots = d_bcount = d_boff = nullptr;
            d_scan_tmp = nullptr;
            LV_HIP(hipMalloc(&d_cell_slots, need_cells * sizeof(uint32_t)));
            LV_HIP(hipMalloc(&d_bcount, need_cells * sizeof(uint32_t)));
            LV_HIP(hipMalloc(&d_boff, need_cells * sizeof(uint32_t)));
            scan_tmp_bytes = 0;
            LV_HIP((hipError_t)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_tmp_bytes, d_bcount, d_boff, (int)need_cells, (hipStream_t)0));
            LV_HIP(hipMalloc(&d_scan_tmp, scan_tmp_bytes));
            cells_cap = need_cells;
        }
        LV_HIP(hipMemsetAsync(d_btable[level], 0xFF, (size_t)size * sizeof(uint4), stream));
        LV_HIP(hipMemsetAsync(d_flags, 0, 2 * sizeof(uint32_t), stream));
        int lg = 0;
        while ((1u << lg) < size) ++lg;
        GridLevelW bt{d_btable[level], size - 1, (uint32_t)(64 - lg)};
        const uint64_t threads = (uint64_t)occ_slots * 27;
        hipLaunchKernelGGL(bucket_register_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, stream, occ, occ_slots, bt,
                           d_cell_slots, (uint32_t)(size / 2), d_flags);
        uint32_t flags[2];
        LV_HIP(hipMemcpyAsync(flags, d_flags, sizeof(flags), hipMemcpyDeviceToHost, stream));
        LV_HIP(hipStreamSynchronize(stream));
        if (flags[1] || flags[0] > size / 2) {  // table too small for this map's dilation factor: double and retry
            size *= 2;
            continue;
        }
        const uint32_t nb = flags[0];
        n_bcells[level] = nb;
        hipLaunchKernelGGL((map_bucket_kernel<false>), dim3(nb), dim3(64), 0, stream, occ, bt, d_cell_slots, nb, d_sorted, d_bcount,
                           d_boff, (float4*)nullptr);
        size_t stmp = scan_tmp_bytes;
        LV_HIP((hipError_t)hipcub::DeviceScan::ExclusiveSum(d_scan_tmp, stmp, d_bcount, d_boff, (int)nb, stream));
        uint32_t last_off = 0, last_cnt = 0;
        LV_HIP(hipMemcpyAsync(&last_off, d_boff + (nb - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        LV_HIP(hipMemcpyAsync(&last_cnt, d_bcount + (nb - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        LV_HIP(hipStreamSynchronize(stream));
        const uint64_t total = (uint64_t)last_off + last_cnt;
        if (total > 0xFFFFFFF0ull) { set_error("bucket array exceeds 2^32 entries at level %d", level); return LV_ERANGE; }
        bucket_points[level] = (size_t)total;
        if (level < SORTED_BUCKET_LEVELS) {   // sorted by original index, then packed to 12-byte points + index array
            if (bucket_points[level] > bucket_tmp_cap) {
                hipFree(d_bucket_tmp);
                d_bucket_tmp = nullptr;
                bucket_tmp_cap = 0;
                size_t cap = bucket_points[level] + bucket_points[level] / 8;
                LV_HIP(hipMalloc(&d_bucket_tmp, cap * sizeof(float4)));
                bucket_tmp_cap = cap;
            }
            if (bucket_points[level] > bucket_cap[level]) {
                hipFree(d_bxyz[level]); hipFree(d_bidx[level]);
                d_bxyz[level] = nullptr; d_bidx[level] = nullptr;
                bucket_cap[level] = 0;
                size_t cap = bucket_points[level] + bucket_points[level] / 8;
                LV_HIP(hipMalloc(&d_bxyz[level], (cap * 3 + 4) * sizeof(float)));
                LV_HIP(hipMalloc(&d_bidx[level], cap * sizeof(uint32_t)));
                bucket_cap[level] = cap;
            }
            hipLaunchKernelGGL((map_bucket_kernel<true>), dim3(nb), dim3(64), 0, stream, occ, bt, d_cell_slots, nb, d_sorted, d_bcount,
                               d_boff, d_bucket_tmp);
            hipLaunchKernelGGL(bucket_sort_wave_kernel, dim3((nb + 3) / 4), dim3(256), 0, stream, d_bcount, d_boff, nb, d_bucket_tmp);
            hipLaunchKernelGGL(bucket_sort_kernel, dim3(nb), dim3(BSORT_THREADS), 0, stream, d_bcount, d_boff, nb, d_bucket_tmp);
            if (total > 0)
                hipLaunchKernelGGL(bucket_pack_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, stream, d_bucket_tmp,
                                   (uint32_t)total, d_bxyz[level], d_bidx[level]);
        } else {            // coarse levels: candidates are ordered by (distance, index) keys, the bucket stays unsorted
            if (bucket_points[level] > bucket_cap[level]) {
                hipFree(d_bucket4[level]);
                d_bucket4[level] = nullptr;
                bucket_cap[level] = 0;
                size_t cap = bucket_points[level] + bucket_points[level] / 8;
                LV_HIP(hipMalloc(&d_bucket4[level], cap * sizeof(float4)));
                bucket_cap[level] = cap;
            }
            hipLaunchKernelGGL((map_bucket_kernel<true>), dim3(nb), dim3(64), 0, stream, occ, bt, d_cell_slots, nb, d_sorted, d_bcount,
                               d_boff, d_bucket4[level]);
        }
        LV_HIP(hipGetLastError());
        view.bt[level].table = d_btable[level];
        view.bt[level].mask = size - 1;
        view.bt[level].shift = (uint32_t)(64 - lg);
        view.bxyz[level] = d_bxyz[level];
        view.bidx[level] = d_bidx[level];
        view.bucket4[level] = d_bucket4[level];
        return LV_OK;
    }
}

}  // namespace lv

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
    return LV_OK;
}

}  // namespace lv

// lv_scan.hip — scan residency.  `points2match = points` (reference src/Modules/Localizator.cpp:131)
// becomes: upload once per correct(), Morton-sort in the LiDAR frame so that consecutive lanes /
// wavefronts query neighbouring voxels (a rigid transform preserves neighbourhoods, so the order
// stays coherent for every IKFoM pass).  Results are always reported in ORIGINAL scan order: each
// record carries its original index in .w.
#include "lv_host.hpp"

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

int ScanStore::reserve(size_t cap) {
    if (cap <= capacity) return LV_OK;
    size_t ncap = capacity ? capacity : 4096;
    while (ncap < cap) ncap *= 2;
    release();
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
    *this = ScanStore();
}

int ScanStore::sort(hipStream_t stream, const float bbox_min[3], float cell) {
    if (n == 0) return LV_OK;
    const int B = 256;
    const uint32_t grid = (n + B - 1) / B;
    hipLaunchKernelGGL(scan_keys_kernel, dim3(grid), dim3(B), 0, stream, d_raw, n, bbox_min[0], bbox_min[1], bbox_min[2],
                       1.0f / cell, d_keys, d_idx);
    size_t tmp = sort_tmp_bytes;
    LV_HIP((hipError_t)hipcub::DeviceRadixSort::SortPairs(d_sort_tmp, tmp, d_keys, d_keys_sorted, d_idx, d_idx_sorted, (int)n, 0,
                                                           30, stream));
    hipLaunchKernelGGL(scan_gather_kernel, dim3(grid), dim3(B), 0, stream, d_raw, d_idx_sorted, n, d_sorted);
    LV_HIP(hipGetLastError());
    return LV_OK;
}

}  // namespace lv

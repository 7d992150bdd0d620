// lv_ldssort.hpp — bitonic sorting networks over LDS for the one-workgroup kernels of the 100 Hz cycle (window_small_kernel,
// window_tail_kernel, scan_sort_small_kernel, inc_small_front_kernel, inc_sort_small_kernel).
//
// `len` (a power of two >= 64) keys, T threads, ascending.  Pair t of a step (k2, j) is
// lo = ((t >> lj) << (lj + 1)) | (t & (j - 1)), hi = lo | j with j = 1 << lj — shifts, not the division / remainder by a run-time
// j the first version of these loops paid twice per pair.  With j <= 64 the 64 pairs of a wavefront lie inside ITS OWN 128
// consecutive elements, and so do those of the step before and after as long as that one has j <= 64 too: such steps are
// separated by a wavefront-level fence instead of a workgroup barrier (49 of the 55 steps of a 1024-key sort).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace lv {

__device__ __forceinline__ void ldssort_wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// CMP(lo, hi) -> true when the pair is out of ascending order; SWAP(lo, hi) exchanges the two entries
template <int T, class Cmp, class Swap>
__device__ __forceinline__ void lds_bitonic_network(uint32_t len, int tid, Cmp out_of_order, Swap swap) {
    int lk = 1;
    for (uint32_t k2 = 2; k2 <= len; k2 <<= 1, ++lk) {
        for (int lj = lk - 1; lj >= 0; --lj) {
            const uint32_t j = 1u << lj;
            for (uint32_t t = (uint32_t)tid; t < len / 2; t += (uint32_t)T) {
                const uint32_t lo = ((t >> lj) << (lj + 1)) | (t & (j - 1u)), hi = lo | j;
                const bool up = (lo & k2) == 0u;
                if (out_of_order(lo, hi) == up) swap(lo, hi);
            }
            const uint32_t next_j = lj > 0 ? (j >> 1) : k2;   // (the next level starts at j = its k2 / 2 = this k2)
            if (j > 64u || next_j > 64u || (k2 == len && lj == 0)) __syncthreads();
            else ldssort_wave_fence();
        }
    }
}

template <int T>
__device__ __forceinline__ void lds_bitonic_sort_u64(uint64_t* s_key, uint32_t len, int tid) {
    lds_bitonic_network<T>(len, tid, [&](uint32_t lo, uint32_t hi) { return s_key[lo] > s_key[hi]; },
                           [&](uint32_t lo, uint32_t hi) { const uint64_t a = s_key[lo]; s_key[lo] = s_key[hi]; s_key[hi] = a; });
}

// composite order (key, index): what a stable sort of the keys gives when the indices start ascending
template <int T>
__device__ __forceinline__ void lds_bitonic_sort_u64_u32(uint64_t* s_key, uint32_t* s_idx, uint32_t len, int tid) {
    lds_bitonic_network<T>(len, tid,
                           [&](uint32_t lo, uint32_t hi) {
                               const uint64_t a = s_key[lo], b = s_key[hi];
                               return a > b || (a == b && s_idx[lo] > s_idx[hi]);
                           },
                           [&](uint32_t lo, uint32_t hi) {
                               const uint64_t a = s_key[lo]; s_key[lo] = s_key[hi]; s_key[hi] = a;
                               const uint32_t ia = s_idx[lo]; s_idx[lo] = s_idx[hi]; s_idx[hi] = ia;
                           });
}

__device__ __forceinline__ uint32_t lds_sort_len(uint32_t n) {   // the smallest power of two >= max(n, 64)
    uint32_t len = 64;
    while (len < n) len <<= 1;
    return len;
}

}  // namespace lv

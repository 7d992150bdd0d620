// tests/emu/hip/hip_runtime.h — a HOST stand-in for <hip/hip_runtime.h>, TEST INFRASTRUCTURE ONLY.
//
// It lets tests/emu/mapinc_emu.cpp compile the one-thread-per-item kernels of limo-velo_amd/csrc/lv_mapinc.hpp with
// g++ and run them as plain loops (one "thread" after another, optionally in reverse or shuffled order), so the
// bookkeeping of the incremental map can be checked against the oracle in the GPU-less container.  Nothing here is
// linked into, loaded by or reachable from the product library; the product runs these kernels on the GPU only.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __restrict__
#define __launch_bounds__(...)

struct float4 { float x, y, z, w; };
struct uint4 { unsigned int x, y, z, w; };
struct uint2 { unsigned int x, y; };
struct emu_dim3 { unsigned int x = 1, y = 1, z = 1; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

extern emu_dim3 threadIdx, blockIdx, blockDim, gridDim;

static inline unsigned int __float_as_uint(float f) { unsigned int u; std::memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned int u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }

template <typename T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = (T)(o + v); return o; }
template <typename T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <typename T> static inline T atomicCAS(T* p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }
template <typename T> static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }

using std::abs;
using std::max;
using std::min;

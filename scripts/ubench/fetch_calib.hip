// fetch_calib.hip — calibration of rocprofv3's FETCH_SIZE on gfx950 against KNOWN byte counts, for the access
// shapes the search kernel uses.  MI355X_MICROARCH.md prescribes x2 for wide (16 B/lane) coalesced reads; the
// candidate stream of lv::search_kernel is 12-byte records read by 8-lane groups, so the factor is measured here
// instead of assumed.  Every kernel reads each byte of a 1 GiB buffer (>> 256 MB MALL) exactly once.
//   calib_x4      : global_load_dwordx4, fully coalesced (1 KiB per wavefront instruction)
//   calib_x3      : global_load_dwordx3 of packed 12-byte records, fully coalesced
//   calib_bucket  : the search kernel's shape — 8-lane groups, each streaming its own run of 56 packed 12-byte
//                   records (8 loads per lane in flight, tail predicated), runs visited in a scattered order
//   calib_x1      : global_load_dword, coalesced (256 B per wavefront instruction)
// Build:  hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
// Run:    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -o calib -- ./fetch_calib
// The program prints the bytes each kernel reads per launch; scripts/ubench/fetch_calib_summ.py divides.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); exit(1); } } while (0)

struct __attribute__((packed, aligned(4))) Xyz { float x, y, z; };

__global__ __launch_bounds__(256) void calib_x4(const float4* __restrict__ p, size_t n, float* __restrict__ out) {
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) void calib_x3(const Xyz* __restrict__ p, size_t n, float* __restrict__ out) {
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const Xyz v = p[i];
        s += v.x + v.y + v.z;
    }
    if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) void calib_x1(const float* __restrict__ p, size_t n, float* __restrict__ out) {
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s == 123.456f) out[0] = s;
}
// runs of RUN records; group g (8 lanes) visits run perm(g)
constexpr int RUN = 56;
__global__ __launch_bounds__(256) void calib_bucket(const Xyz* __restrict__ p, uint32_t n_runs, float* __restrict__ out) {
    const uint32_t g = (blockIdx.x * 256u + threadIdx.x) >> 3, tl = threadIdx.x & 7u;
    if (g >= n_runs) return;
    const uint32_t run = (uint32_t)(((uint64_t)g * 2654435761ull) % n_runs);   // n_runs odd => a permutation
    const Xyz* bp = p + (size_t)run * RUN;
    float s = 0.f;
    Xyz m[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const uint32_t j = (uint32_t)u * 8u + tl;
        m[u] = bp[j < (uint32_t)RUN ? j : 0];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += m[u].x + m[u].y + m[u].z;
    if (s == 123.456f) out[0] = s;
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    void* buf;
    float* out;
    CK(hipMalloc(&buf, bytes + 4096));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 0, bytes + 4096));
    CK(hipDeviceSynchronize());
    const int reps = 5;
    for (int r = 0; r < reps; ++r) {
        const size_t n4 = bytes / 16, n3 = bytes / 12, n1 = bytes / 4;
        hipLaunchKernelGGL(calib_x4, dim3(4096), dim3(256), 0, 0, (const float4*)buf, n4, out);
        hipLaunchKernelGGL(calib_x3, dim3(4096), dim3(256), 0, 0, (const Xyz*)buf, n3, out);
        hipLaunchKernelGGL(calib_x1, dim3(4096), dim3(256), 0, 0, (const float*)buf, n1, out);
        uint32_t n_runs = (uint32_t)(bytes / (12 * RUN));
        if ((n_runs & 1u) == 0) --n_runs;
        while (n_runs % 2654435761ull == 0) n_runs -= 2;
        hipLaunchKernelGGL(calib_bucket, dim3((n_runs * 8 + 255) / 256), dim3(256), 0, 0, (const Xyz*)buf, n_runs, out);
        CK(hipDeviceSynchronize());
        if (r == 0)
            printf("{\"calib_x4\": %zu, \"calib_x3\": %zu, \"calib_x1\": %zu, \"calib_bucket\": %zu}\n", n4 * 16, n3 * 12, n1 * 4,
                   (size_t)n_runs * RUN * 12);
    }
    CK(hipFree(buf));
    CK(hipFree(out));
    return 0;
}

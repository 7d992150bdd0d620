// cu_mask_starve.hip — a workgroup that needs a WHOLE compute unit (1024 threads, 128 VGPRs, 150 KB LDS: pass_kernel's shape)
// cannot be placed while another stream keeps every CU topped up with small workgroups; does a CU mask on that stream help?
// Stream A: round trips of one such workgroup.  Stream B: a flood of 64-thread workgroups with ~20 us of work each, on
// (a) a low-priority stream, (b) a stream created by hipExtStreamCreateWithCUMask over every 4th CU.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(1024) void whole_cu(uint32_t* p) {
    __shared__ uint32_t s[150 * 256];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) p[0] += s[5];
}
__global__ void flood(uint32_t* out, int spin) {
    uint32_t v = blockIdx.x;
    for (int i = 0; i < spin; ++i) v = v * 1664525u + 1013904223u;
    if (v == 0xdeadbeefu) out[0] = v;
}
static double pct(std::vector<double>& v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)((v.size() - 1) * q)]; }
int main() {
    hipStream_t a, b_lo, b_mask = nullptr;
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    hipStreamCreateWithPriority(&b_lo, hipStreamNonBlocking, lo);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
    for (int i = 0; i < ncu; i += 4) mask[(size_t)i / 32] |= 1u << (i % 32);
    hipError_t em = hipExtStreamCreateWithCUMask(&b_mask, (uint32_t)mask.size(), mask.data());
    printf("CUs %d; hipExtStreamCreateWithCUMask: %s\n", ncu, hipGetErrorString(em));
    uint32_t *p, *out; hipMalloc(&p, 4096); hipMalloc(&out, 4096); hipMemset(p, 0, 4096);
    hipFuncSetAttribute(reinterpret_cast<const void*>(whole_cu), hipFuncAttributeMaxDynamicSharedMemorySize, 0);
    const char* names[] = {"stream B idle", "B low priority: flood of small workgroups", "B CU-masked (every 4th CU): same flood"};
    for (int mode = 0; mode < 3; ++mode) {
        hipStream_t b = mode == 2 ? b_mask : b_lo;
        if (mode == 2 && !b_mask) break;
        std::vector<double> lat;
        hipEvent_t ev; hipEventCreate(&ev);
        bool pending = false; int floods = 0;
        double flood_ms = 0;
        const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(800);
        auto tf = std::chrono::steady_clock::now();
        while (std::chrono::steady_clock::now() < t_end) {
            if (mode && (!pending || hipEventQuery(ev) == hipSuccess)) {
                if (pending) flood_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tf).count();
                hipLaunchKernelGGL(flood, dim3(4000000), dim3(64), 0, b, out, 2000);
                hipEventRecord(ev, b); pending = true; ++floods; tf = std::chrono::steady_clock::now();
            }
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(whole_cu, dim3(1), dim3(1024), 0, a, p);
            hipStreamSynchronize(a);
            lat.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        }
        hipDeviceSynchronize();
        printf("%-44s: %6zu round trips, median %9.1f us  p99 %9.1f us  max %9.1f us  (floods %d, ~%.1f ms each)\n", names[mode], lat.size(), pct(lat, 0.5),
               pct(lat, 0.99), pct(lat, 1.0), floods, floods > 1 ? flood_ms / (floods - 1) : 0.0);
    }
    return 0;
}

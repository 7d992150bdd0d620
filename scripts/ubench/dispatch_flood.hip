// dispatch_flood.hip — does a kernel of very many tiny workgroups on a LOW-priority (optionally CU-masked) stream delay small
// launches on another stream?  (profiles/experiments_r05/async_rebuild.txt: the background map rebuild's build kernels launch
// up to 1.7e9 threads as 6.7e6 workgroups.)  Prints the round-trip latency of a 1-workgroup kernel on stream A (launch +
// hipStreamSynchronize) while stream B runs (a) nothing, (b) a flood of tiny workgroups, (c) the same work as a grid-stride
// kernel of 2048 workgroups.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void tiny(uint32_t* p) { if (threadIdx.x == 0) p[blockIdx.x & 1023] += 1; }
__global__ void flood(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && in[i & 0xFFFFF] == 0xdeadbeefu) out[i & 1023] = 1;
}
__global__ void strided(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (in[i & 0xFFFFF] == 0xdeadbeefu) out[i & 1023] = 1;
}
static double pct(std::vector<double>& v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)((v.size() - 1) * q)]; }
int main() {
    hipStream_t a, b;
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&b, hipStreamNonBlocking, lo));
    uint32_t *p, *in, *out;
    CK(hipMalloc(&p, 4096)); CK(hipMalloc(&in, 4 << 20)); CK(hipMalloc(&out, 4096));
    CK(hipMemset(p, 0, 4096)); CK(hipMemset(in, 0, 4 << 20)); CK(hipMemset(out, 0, 4096));
    const size_t n = 1700000000ull;
    for (int mode = 0; mode < 3; ++mode) {
        std::vector<double> lat;
        const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(600);
        int floods = 0;
        hipEvent_t ev; CK(hipEventCreate(&ev));
        bool pending = false;
        while (std::chrono::steady_clock::now() < t_end) {
            if (mode && (!pending || hipEventQuery(ev) == hipSuccess)) {
                if (mode == 1) hipLaunchKernelGGL(flood, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, b, in, out, n);
                else hipLaunchKernelGGL(strided, dim3(2048), dim3(256), 0, b, in, out, n);
                CK(hipEventRecord(ev, b)); pending = true; ++floods;
            }
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, p);
            CK(hipStreamSynchronize(a));
            lat.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        }
        CK(hipDeviceSynchronize());
        printf("%-38s: %6zu round trips, median %8.1f us  p99 %9.1f us  max %9.1f us  (background kernels launched: %d)\n",
               mode == 0 ? "stream B idle" : mode == 1 ? "B: 6.6e6 workgroups of 256 threads" : "B: grid-stride, 2048 workgroups", lat.size(), pct(lat, 0.5), pct(lat, 0.99),
               pct(lat, 1.0), floods);
    }
    return 0;
}

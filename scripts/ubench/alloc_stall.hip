// alloc_stall.hip — does hipMalloc / hipFree of large buffers in ANOTHER THREAD stall small launches on this one?
// (profiles/experiments_r05/async_rebuild.txt.)  Thread A: round trips of a 1-workgroup kernel (launch + hipStreamSynchronize) on
// its own stream.  Thread B: (a) nothing, (b) hipMalloc + hipFree of 256 MB in a loop, (c) hipMalloc only (freed at the end),
// (d) hipMemsetAsync of 1 GB + synchronise on its own stream, (e) a device-to-host copy of 4 bytes + synchronise in a loop.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void tiny(uint32_t* p) { if (threadIdx.x == 0) p[blockIdx.x & 1023] += 1; }
static double pct(std::vector<double>& v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)((v.size() - 1) * q)]; }
int main() {
    hipStream_t a, b;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    uint32_t* p; hipMalloc(&p, 4096); hipMemset(p, 0, 4096);
    void* big; hipMalloc(&big, 1ull << 30);
    const char* names[] = {"thread B idle", "B: hipMalloc + hipFree of 256 MB", "B: hipMalloc of 256 MB (kept)", "B: hipMemsetAsync 1 GB + sync", "B: 4-byte D2H copy + sync"};
    for (int mode = 0; mode < 5; ++mode) {
        std::atomic<bool> stop{false};
        std::atomic<int> ops{0};
        std::thread tb([&] {
            hipSetDevice(0);
            std::vector<void*> kept;
            uint32_t h = 0;
            while (!stop.load()) {
                if (mode == 1) { void* q = nullptr; if (hipMalloc(&q, 256u << 20) == hipSuccess) hipFree(q); }
                else if (mode == 2) { void* q = nullptr; if (kept.size() < 40 && hipMalloc(&q, 256u << 20) == hipSuccess) kept.push_back(q); else std::this_thread::sleep_for(std::chrono::milliseconds(1)); }
                else if (mode == 3) { hipMemsetAsync(big, 0xFF, 1ull << 30, b); hipStreamSynchronize(b); }
                else if (mode == 4) { hipMemcpyAsync(&h, p, 4, hipMemcpyDeviceToHost, b); hipStreamSynchronize(b); }
                else std::this_thread::sleep_for(std::chrono::milliseconds(1));
                ++ops;
            }
            for (void* q : kept) hipFree(q);
        });
        std::vector<double> lat;
        const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(700);
        while (std::chrono::steady_clock::now() < t_end) {
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, p);
            hipStreamSynchronize(a);
            lat.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        }
        stop = true;
        tb.join();
        hipDeviceSynchronize();
        printf("%-36s: %6zu round trips, median %8.1f us  p99 %9.1f us  max %9.1f us  (B's operations: %d)\n", names[mode], lat.size(), pct(lat, 0.5),
               pct(lat, 0.99), pct(lat, 1.0), ops.load());
    }
    return 0;
}

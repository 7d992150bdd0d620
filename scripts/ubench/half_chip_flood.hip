// half_chip_flood.hip — follow-up of cu_mask_starve.hip: a whole-CU workgroup (pass_kernel's shape) starves behind a flood of
// small workgroups because the dispatcher tops every CU up.  Does it still starve when the other stream's kernel is a
// PERSISTENT grid of at most half as many 1024-thread workgroups as there are CUs (each looping over the same work)?
// Stream A: round trips of one whole-CU workgroup (and of 6 / 128 / 256 of them).  Stream B (low priority): the same total
// arithmetic as cu_mask_starve's flood, as G x 1024 threads looping.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
// (112 values live at once: the compiler gives the kernel ~128 VGPRs, so that four wavefronts fill a SIMD's register file as
// pass_kernel's do — with few registers the probe would fit BESIDE a persistent workgroup and prove nothing)
__global__ __launch_bounds__(1024) void whole_cu(uint32_t* p, const uint32_t* q) {
    __shared__ uint32_t s[150 * 256];
    uint32_t r[112];
#pragma unroll
    for (int i = 0; i < 112; ++i) r[i] = q[(threadIdx.x + 64 * i) & 1023];
#pragma unroll
    for (int i = 0; i < 112; ++i) asm volatile("" : "+v"(r[i]));
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 112; ++i) acc = acc * 31u + r[i];
    s[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) p[blockIdx.x] += s[5];
}
__global__ __launch_bounds__(1024) void persistent(uint32_t* out, int spin, uint32_t vblocks) {
    extern __shared__ uint32_t pad[];
    const uint32_t wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    uint32_t acc = 0;
    for (uint32_t vb = blockIdx.x * nw + wave; vb < vblocks; vb += gridDim.x * nw) {
        uint32_t v = vb;
        for (int i = 0; i < spin; ++i) v = v * 1664525u + 1013904223u;
        acc ^= v;
    }
    if (acc == 0xdeadbeefu) out[0] = acc + pad[0];
}
static double pct(std::vector<double>& v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)((v.size() - 1) * q)]; }
int main() {
    hipStream_t a, b;
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    hipStreamCreateWithPriority(&b, hipStreamNonBlocking, lo);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    uint32_t *p, *out; hipMalloc(&p, 4096); hipMalloc(&out, 4096); hipMemset(p, 0, 4096);
    hipFuncSetAttribute(reinterpret_cast<const void*>(persistent), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    struct Mode { const char* name; int grid; size_t lds; int probe; };
    const Mode modes[] = {
        {"B idle, probe 1 workgroup", 0, 0, 1},
        {"B: 256 x 1024 threads persistent (every CU)", ncu, 0, 1},
        {"B: 128 x 1024 threads persistent", ncu / 2, 0, 1},
        {"B: 128 x 1024 threads + 100 KB LDS each", ncu / 2, 100 * 1024, 1},
        {"B: 120 x 1024 threads, probe 6 workgroups", 120, 0, 6},
        {"B: 120 x 1024 threads, probe 128 workgroups", 120, 0, 128},
        {"B: 120 x 1024 threads + 100 KB, probe 128", 120, 100 * 1024, 128},
        {"B: 120 x 1024 threads, probe 256 workgroups", 120, 0, 256},
        {"B: 64 x 1024 threads, probe 128 workgroups", 64, 0, 128},
    };
    for (const Mode& m : modes) {
        std::vector<double> lat;
        hipEvent_t ev; hipEventCreate(&ev);
        bool pending = false; int floods = 0;
        double flood_ms = 0;
        const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(600);
        auto tf = std::chrono::steady_clock::now();
        while (std::chrono::steady_clock::now() < t_end) {
            if (m.grid && (!pending || hipEventQuery(ev) == hipSuccess)) {
                if (pending) flood_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tf).count();
                hipLaunchKernelGGL(persistent, dim3(m.grid), dim3(1024), m.lds, b, out, 2000, 1000000u);
                hipEventRecord(ev, b); pending = true; ++floods; tf = std::chrono::steady_clock::now();
            }
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(whole_cu, dim3(m.probe), dim3(1024), 0, a, p, p);
            hipStreamSynchronize(a);
            lat.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        }
        hipDeviceSynchronize();
        printf("%-48s: %6zu round trips, median %9.1f us  p99 %9.1f us  max %9.1f us  (B kernels %d, ~%.1f ms each)\n", m.name, lat.size(), pct(lat, 0.5),
               pct(lat, 0.99), pct(lat, 1.0), floods, floods > 1 ? flood_ms / (floods - 1) : 0.0);
    }
    return 0;
}

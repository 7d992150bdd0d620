// micro-benchmark: issue cost (cycles per wave-instruction) of the compare-exchange candidates on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 2048
template <int MODE>
__global__ void k(double* out, long long* clk, double seed) {
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed * (threadIdx.x + 1) * (i + 3);
    uint32_t u[8];
    for (int i = 0; i < 8; ++i) u[i] = (uint32_t)(threadIdx.x * 2654435761u) ^ (i * 40503u);
    __syncthreads();
    long long t0 = clock64();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            if (MODE == 0) {  // f64 min/max pair
                double lo, hi;
                asm volatile("v_min_f64 %0, %2, %3\n v_max_f64 %1, %2, %3" : "=&v"(lo), "=&v"(hi) : "v"(a[i]), "v"(a[i + 1]));
                a[i] = hi; a[i + 1] = lo;
            } else if (MODE == 1) {  // u32 min/max pair
                uint32_t lo, hi;
                asm volatile("v_min_u32 %0, %2, %3\n v_max_u32 %1, %2, %3" : "=&v"(lo), "=&v"(hi) : "v"(u[i]), "v"(u[i + 1]));
                u[i] = hi; u[i + 1] = lo;
            } else if (MODE == 2) {  // u64 compare + 4 cndmask
                uint64_t x = (uint64_t)__double_as_longlong(a[i]), y = (uint64_t)__double_as_longlong(a[i + 1]);
                bool c = x < y;
                uint64_t lo = c ? x : y, hi = c ? y : x;
                asm volatile("" : "+v"(lo), "+v"(hi));
                a[i] = __longlong_as_double((long long)hi); a[i + 1] = __longlong_as_double((long long)lo);
            } else if (MODE == 3) {  // f64 fma
                asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(a[i + 1]));
                asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i + 1]) : "v"(a[i]));
            } else if (MODE == 5) {  // two DPP moves (quad_perm) of 32-bit words
                int x = (int)u[i], y = (int)u[i + 1];
                x = __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, true);
                y = __builtin_amdgcn_update_dpp(y, y, 0x141, 0xF, 0xF, true);
                asm volatile("" : "+v"(x), "+v"(y));
                u[i] = (uint32_t)x; u[i + 1] = (uint32_t)y;
            } else if (MODE == 6) {  // two packed f32 multiplies
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[i + 1]));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i + 1]) : "v"(a[i]));
            } else if (MODE == 7) {  // two cndmask
                uint32_t x = u[i], y = u[i + 1];
                asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(y) : "vcc");
                u[i] = x;
            } else if (MODE == 4) {  // f32 fma
                float x = (float)u[i], y = (float)u[i+1];
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y) : "v"(x));
                u[i] = (uint32_t)x; u[i+1] = (uint32_t)y;
            }
        }
    }
    long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char* name, int threads) {
    double* out; long long* clk;
    hipMalloc(&out, 1024 * 1024 * 8); hipMalloc(&clk, 4096 * 8);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, clk, 1.000001);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, clk, 1.000001);
    hipDeviceSynchronize();
    long long h[256]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 256; ++i) m += h[i]; m /= 256;
    // 8 instructions per iteration for modes 0,1,3,4 (4 pairs); per wave
    printf("%-28s threads %4d: %.2f clk64 ticks per pair-iteration (4 pairs) -> %.2f per pair\n", name, threads, m / ITERS, m / ITERS / 4);
    hipFree(out); hipFree(clk);
}
int main() {
    for (int th : {64, 256, 1024}) {
        run<0>("f64 min+max", th); run<1>("u32 min+max", th); run<2>("u64 cmp + 4 cndmask", th); run<3>("2x f64 fma", th); run<4>("2x f32 fma (+cvt)", th); run<5>("2x dpp mov", th); run<6>("2x v_pk_mul_f32", th); run<7>("cmp + cndmask", th);
    }
    // clock64 rate vs wall clock
    return 0;
}

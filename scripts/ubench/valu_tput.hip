// micro-benchmark: THROUGHPUT cost of VALU instruction classes on gfx950 — cycles of one SIMD per wave64 instruction with four
// wavefronts per SIMD and eight independent chains per wavefront (latency hidden), s_memtime cycles of wavefront 0 of the last
// workgroup to finish divided by the instructions its SIMD issued.  scripts/ubench/valu_rate.hip measures dependent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 4096
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int MODE>
__global__ __launch_bounds__(1024) void k(double* out, long long* clk, double seed) {
    double a[8];
    uint32_t u[8], w[8];
    float f[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed * (threadIdx.x + 1) * (i + 3); u[i] = (uint32_t)(threadIdx.x * 2654435761u) ^ (i * 40503u); w[i] = u[i] * 7u + 1u; f[i] = (float)a[i]; }
    const double c = seed * 1.0000001;
    const float cf = (float)c;
    const uint32_t cu = (uint32_t)(seed * 12345.0);
    unsigned long long sel = __ballot(threadIdx.x & 1), selv[4] = {0, 0, 0, 0};
    asm volatile("s_mov_b64 vcc, %0" : : "s"(sel) : "vcc");
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < ITERS; ++it) {
#define M0(i) asm volatile("v_min_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define M1(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
#define M2(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define M3(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(cf));
#define M4(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(cf));
#define M5(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define M6(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(cu) : );
#define M7(i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(u[i]) : "v"(w[i]));
#define M8(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(u[i]), "v"(cu) : "vcc");
#define M9(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(u[i]), "v"(cu) : "vcc");
#define M10(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(cu));
#define M11(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define M12(i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define M13(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[i]) : "v"(cu));
#define M14(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(cu));
#define M15(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
#define M16(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[i]));
#define M18(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[i]) : "v"(cu), "s"(sel));
#define M19(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(w[i]) : "vcc");
#define M20(i) asm volatile("v_cmp_lt_u32_e64 %2, %0, %1\n v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[i]), "+v"(w[i]), "=s"(selv[i & 3]) : );
#define M21(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(u[i]) : "v"(cu), "v"(w[i]));
#define M22(i) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(cu), "v"(w[i]));
#define M23(i) asm volatile("v_max_u32 %0, %0, %1" : "+v"(u[i]) : "v"(cu));
#define M17(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[i]) : "v"(c));
        if (MODE == 0) { REP8(M0) } else if (MODE == 1) { REP8(M1) } else if (MODE == 2) { REP8(M2) } else if (MODE == 3) { REP8(M3) }
        else if (MODE == 4) { REP8(M4) } else if (MODE == 5) { REP8(M5) } else if (MODE == 6) { REP8(M6) } else if (MODE == 7) { REP8(M7) }
        else if (MODE == 8) { REP8(M8) } else if (MODE == 9) { REP8(M9) } else if (MODE == 10) { REP8(M10) } else if (MODE == 11) { REP8(M11) }
        else if (MODE == 12) { REP8(M12) } else if (MODE == 13) { REP8(M13) } else if (MODE == 14) { REP8(M14) } else if (MODE == 15) { REP8(M15) }
        else if (MODE == 16) { REP8(M16) } else if (MODE == 17) { REP8(M17) }
        else if (MODE == 18) { REP8(M18) } else if (MODE == 19) { REP8(M19) } else if (MODE == 20) { REP8(M20) } else if (MODE == 21) { REP8(M21) }
        else if (MODE == 22) { REP8(M22) } else if (MODE == 23) { REP8(M23) }
    }
    const long long t1 = clock64();
    __syncthreads();
    const long long t2 = clock64();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + u[i] + f[i] + w[i] + (double)selv[i & 3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = t2 - t0; }
}
template <int MODE>
void run(const char* name) {
    double* out; long long* clk;
    hipMalloc(&out, 1024 * 1024 * 8); hipMalloc(&clk, 4096 * 8);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, out, clk, 1.000001);
    hipDeviceSynchronize();
    long long h[512]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 256; ++i) m += h[2 * i + 1]; m /= 256;   // all 16 wavefronts of the workgroup done
    // a SIMD issued 4 wavefronts x ITERS x 8 instructions in that time
    printf("%-34s %.2f cycles per wave64 instruction per SIMD (4 wavefronts / SIMD)\n", name, m / (4.0 * ITERS * 8));
    hipFree(out); hipFree(clk);
}
int main() {
    run<3>("v_mul_f32"); run<4>("v_fma_f32"); run<10>("v_add_u32"); run<13>("v_min_u32"); run<6>("v_cndmask_b32 (vcc)"); run<9>("v_cmp_lt_u32 -> vcc");
    run<7>("v_mov_b32_dpp quad_perm"); run<5>("v_pk_mul_f32"); run<0>("v_min_f64"); run<12>("v_max_f64"); run<2>("v_add_f64"); run<11>("v_mul_f64"); run<1>("v_fma_f64");
    run<8>("v_mad_u64_u32"); run<14>("v_mul_lo_u32"); run<17>("v_lshl_add_u64"); run<15>("v_rcp_f32"); run<16>("v_sqrt_f32");
    run<18>("v_cndmask_b32_e64 (sgpr pair)"); run<19>("v_cmp + v_cndmask (vcc), per PAIR"); run<20>("v_cmp_e64 + v_cndmask_e64, per PAIR"); run<21>("v_bfi_b32");
    run<22>("v_min3_u32"); run<23>("v_max_u32");
    return 0;
}

#include <hip/hip_runtime.h>
__global__ void k(const uint4* __restrict__ g, uint4* out) {
    __shared__ uint4 s[256];
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + threadIdx.x),
        (__attribute__((address_space(3))) void*)(s + (threadIdx.x & ~63u)), 16, 0, 0);
    __syncthreads();
    out[threadIdx.x] = s[255 - threadIdx.x];
}
int main() {
    uint4 *g, *o; hipMalloc(&g, 4096); hipMalloc(&o, 4096);
    uint4 h[256]; for (int i = 0; i < 256; ++i) h[i] = make_uint4(i, i + 1, i + 2, i + 3);
    hipMemcpy(g, h, 4096, hipMemcpyHostToDevice);
    k<<<1, 256>>>(g, o); hipMemcpy(h, o, 4096, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 256; ++i) bad += h[i].x != (unsigned)(255 - i) || h[i].w != (unsigned)(255 - i + 3);
    printf("bad %d\n", bad); return bad != 0;
}

// Which SIMD do the waves of a 512-thread workgroup land on?  (development aid for the unit mapping of k_mlp_fused)
//   hipcc --offload-arch=gfx950 -O2 -o tools/_bin/simdmap tools/simdmap.hip && tools/_bin/simdmap
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out)
{
    extern __shared__ char lds[];
    if ((threadIdx.x & 63) == 0)
        out[blockIdx.x * 8 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID, 32 bits
    if (threadIdx.x == 9999) lds[0] = 1;
}
int main()
{
    unsigned *d, h[8 * 512];
    hipMalloc(&d, sizeof(h));
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 119 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k, dim3(512), dim3(512), 119 * 1024, 0, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    }
    int hist[8][4] = {};
    for (int b = 0; b < 512; ++b)
        for (int w = 0; w < 8; ++w) hist[w][(h[b * 8 + w] >> 4) & 3]++;
    for (int w = 0; w < 8; ++w) printf("wave %d: simd histogram %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    for (int b = 0; b < 3; ++b) {
        for (int w = 0; w < 8; ++w) printf("%08x ", h[b * 8 + w]);
        printf("\n");
    }
    return 0;
}

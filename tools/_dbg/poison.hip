#include <hip/hip_runtime.h>
// fill every CU's LDS with a pattern (development aid: makes reads of never-written LDS visible)
__global__ __launch_bounds__(1024) void k_poison(unsigned pat)
{
    extern __shared__ unsigned sm[];
    for (int i = threadIdx.x; i < 40960; i += 1024) sm[i] = pat;
    __syncthreads();
    if (sm[threadIdx.x] == 0x12345u) sm[0] = 1;
}
extern "C" int poison_lds(unsigned pat)
{
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_poison), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    hipLaunchKernelGGL(k_poison, dim3(256 * 4), dim3(1024), 163840, 0, pat);
    return (int)hipDeviceSynchronize();
}

// Do MFMA and VALU work of two different waves on the same SIMD overlap on gfx950?
//   hipcc --offload-arch=gfx950 -O3 tools/overlapbench.hip -o overlapbench && ./overlapbench
// One workgroup of 512 threads per CU (8 waves = 2 per SIMD).  Waves 0-3 run an MFMA stream (4 independent 32x32x16 f16
// accumulators), waves 4-7 a dependent-free fp32 FMA stream.  Timed: MFMA alone, VALU alone, both together.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k(int mode, int iters, float *out)
{
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        if (mode & 1) {
            h8 a, b;
            for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f - i); }
            f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
            for (int it = 0; it < iters; ++it) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
            }
            r = c0[0] + c1[1] + c2[2] + c3[3];
        }
    } else {
        if (mode & 2) {
            float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f, x4 = 4.f, x5 = 5.f, x6 = 6.f, x7 = 7.f;
            const float m = 1.0000001f, c = 1e-9f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {   // 32 independent FMAs per iteration = 128 cycles, same as 4 MFMAs
                    x0 = __builtin_fmaf(x0, m, c); x1 = __builtin_fmaf(x1, m, c); x2 = __builtin_fmaf(x2, m, c);
                    x3 = __builtin_fmaf(x3, m, c); x4 = __builtin_fmaf(x4, m, c); x5 = __builtin_fmaf(x5, m, c);
                    x6 = __builtin_fmaf(x6, m, c); x7 = __builtin_fmaf(x7, m, c);
                }
            }
            r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
        }
    }
    if (r == 12345.678f) out[0] = r;
}

int main()
{
    float *out;
    hipMalloc(&out, 4);
    const int iters = 20000;
    for (int mode = 1; mode <= 3; ++mode) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, 100, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, iters, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d (%s): %.3f ms  (%.1f cycles per iteration at 2.4 GHz; 4 MFMA = 128 pipe cycles, 32 FMA = 128 issue cycles)\n",
               mode, mode == 1 ? "MFMA only" : mode == 2 ? "VALU only" : "MFMA + VALU", ms, ms * 1e-3 * 2.4e9 / iters);
    }
    return 0;
}

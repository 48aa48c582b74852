// Does a wave's own VALU work run in the shadow of its MFMAs on gfx950?  One stream per wave:
//   repeat { v_mfma_f32_32x32x16_f16 (two dependent chains, alternating) ; N independent v_fma_f32 }
// for N = 0 .. 10, with 1 and 2 waves per SIMD.  Prints wave cycles per MFMA: 32 = the matrix pipe's own pace;
// 32 + 4 N would be "no overlap at all".
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/mfma_valu tools/mfma_valu.hip && tools/_bin/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int N, int TRANS>
__global__ __launch_bounds__(512) void k(int iters, float *out, long long *cyc)
{
    const int lane = threadIdx.x & 63;
    h8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(0.001f * (lane + i)); y[i] = (_Float16)(0.002f * (lane - i)); }
    f16v c0 = {}, c1 = {};
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = 0.01f * (lane + i);
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c0, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                if (TRANS && i < TRANS) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
                else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(v[11]));
            }
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, c1, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                if (TRANS && i < TRANS) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
                else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(v[11]));
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float r = c0[0] + c1[3];
    for (int i = 0; i < 12; ++i) r += v[i];
    if (r == 12345.678f) out[0] = r;
    if (lane == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int N, int TRANS>
static void run(int waves, float *out, long long *cyc)
{
    const int iters = 2000;
    hipLaunchKernelGGL((k<N, TRANS>), dim3(256), dim3(64 * waves), 0, 0, 10, out, cyc);
    hipLaunchKernelGGL((k<N, TRANS>), dim3(256), dim3(64 * waves), 0, 0, iters, out, cyc);
    hipDeviceSynchronize();
    long long h[2048];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) s += (double)h[b * 8 + w];
    printf("  N=%2d trans=%d waves/SIMD=%d: %6.1f wave cycles per MFMA  (pipe per SIMD: %5.1f)\n", N, TRANS, waves / 4,
           s / (256.0 * waves) / (iters * 8.0), s / (256.0 * waves) / (iters * 8.0) / (waves / 4));
}

int main()
{
    float *out; long long *cyc;
    hipMalloc(&out, 4); hipMalloc(&cyc, 256 * 8 * 8);
    for (int waves = 4; waves <= 8; waves += 4) {
        run<0, 0>(waves, out, cyc); run<2, 0>(waves, out, cyc); run<4, 0>(waves, out, cyc); run<6, 0>(waves, out, cyc);
        run<8, 0>(waves, out, cyc); run<10, 0>(waves, out, cyc); run<4, 1>(waves, out, cyc); run<6, 2>(waves, out, cyc);
    }
    return 0;
}

// Third look at matrix-pipe / vector-ALU concurrency on gfx950 (tools/overlap2.hip: two DIFFERENT waves of a SIMD, one
// saturating the matrix pipe, one doing epilogue arithmetic, take the SUM of their times).  Here every wave carries
// both streams, interleaved finely: after every third MFMA of a GEMM-like stream (LDS fragment reads, two dependent
// chains) one element pair of an epilogue-like stream (fma, exp2, min, med3, fp16 split, LDS write).  512-thread
// workgroups, one per CU (2 waves per SIMD).  Times: GEMM stream alone, epilogue stream alone, both interleaved.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/overlap3 tools/overlap3.hip && tools/_bin/overlap3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE, int VAR>   // MODE: 1 = MFMAs, 2 = epilogue pieces, 3 = both; VAR bits: 1 = no exp, 2 = no LDS writes, 4 = no LDS reads
__global__ __launch_bounds__(512) void k(int iters, float *out, long long *cyc)
{
    __shared__ __attribute__((aligned(16))) _Float16 lds[4096 + 2 * 16384];
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = (_Float16)(0.001f * (i & 255));
    __syncthreads();
    const int lane = threadIdx.x & 63, tid = threadIdx.x;
    const h8 *a = reinterpret_cast<const h8 *>(lds) + lane;
    f16v c0 = {}, c1 = {};
    h8 xr = a[0], yr = a[128];
    f16v e0, e1;   // the "other tile's" accumulators being post-processed
    for (int r = 0; r < 16; ++r) { e0[r] = 0.01f * (r + 1) + 1e-4f * tid; e1[r] = 0.02f * (r + 1) - 1e-4f * tid; }
    float s = 0.f;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {   // 8 groups: 2 "k steps" of 6 MFMAs, 4 elements (2 pairs) of epilogue
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if (MODE & 1) {
                    const h8 x = (VAR & 4) ? xr : a[(u & 1) * 64], y = (VAR & 4) ? yr : a[128 + (u & 1) * 64];
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, c1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, y, c0, 0, 0, 0);
                }
                if (MODE & 2) {
                    const int r = 2 * (2 * (u & 3) + half);
                    f16v &e = (u & 4) ? e1 : e0;
                    const v2f x = v2f{e[r], e[r + 1]} * 0.37f + v2f{0.01f, 0.02f};
                    const v2f t = x * 14.4f;
                    const v2f ex = (VAR & 1) ? t * t + 0.5f : v2f{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
                    const v2f y = ex * 0.1f - 0.1f;
                    s += fminf(ex.x, 1.0f) + fminf(ex.y, 1.0f);
                    const float y0 = __builtin_amdgcn_fmed3f(x.x, y.x, 0.f), y1 = __builtin_amdgcn_fmed3f(x.y, y.y, 0.f);
                    h2 hi, lo;
                    hi[0] = (_Float16)(y0 * 64.f); hi[1] = (_Float16)(y1 * 64.f);
                    lo[0] = (_Float16)__builtin_fmaf(y0, 64.f, -(float)hi[0]);
                    lo[1] = (_Float16)__builtin_fmaf(y1, 64.f, -(float)hi[1]);
                    if (!(VAR & 2)) {
                        *reinterpret_cast<h2 *>(lds + 4096 + tid * 32 + r) = hi;
                        *reinterpret_cast<h2 *>(lds + 4096 + 16384 + tid * 32 + r) = lo;
                    } else {
                        s += (float)hi[0] + (float)lo[1] + (float)hi[1] + (float)lo[0];
                    }
                    e[r] = y0 + 1e-3f; e[r + 1] = y1 + 1e-3f;
                }
                if (MODE & 1) {
                    const h8 x = (VAR & 4) ? xr : a[(u & 1) * 64], y = (VAR & 4) ? yr : a[128 + (u & 1) * 64];
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, x, c1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, c1, 0, 0, 0);
                }
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float r = c0[0] + c1[5] + s + e0[3] + e1[7];
    if (r == 12345.678f) out[0] = r;
    if (lane == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE, int VAR>
static void run(const char *name, float *out, long long *cyc)
{
    const int iters = 1000;
    hipLaunchKernelGGL((k<MODE, VAR>), dim3(256), dim3(512), 0, 0, 10, out, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, VAR>), dim3(256), dim3(512), 0, 0, iters, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2048]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double g = 0;
    for (int i = 0; i < 2048; ++i) g += (double)h[i];
    printf("%-28s %8.3f ms   wave ticks per iteration (96 MFMA, 32 elements): %8.1f\n", name, ms, g / 2048 / iters);
}

int main()
{
    float *out; long long *cyc;
    hipMalloc(&out, 4); hipMalloc(&cyc, 256 * 8 * 8);
    printf("-- as in the fused kernel\n");
    run<1, 0>("GEMM stream alone", out, cyc); run<2, 0>("epilogue stream alone", out, cyc); run<3, 0>("interleaved in every wave", out, cyc);
    printf("-- exp2 replaced by an fma\n");
    run<2, 1>("epilogue stream alone", out, cyc); run<3, 1>("interleaved in every wave", out, cyc);
    printf("-- no LDS writes in the epilogue\n");
    run<2, 2>("epilogue stream alone", out, cyc); run<3, 2>("interleaved in every wave", out, cyc);
    printf("-- MFMA operands from registers (no LDS reads)\n");
    run<1, 4>("GEMM stream alone", out, cyc); run<3, 4>("interleaved in every wave", out, cyc);
    printf("-- all three\n");
    run<1, 7>("GEMM stream alone", out, cyc); run<2, 7>("epilogue stream alone", out, cyc); run<3, 7>("interleaved in every wave", out, cyc);
    return 0;
}

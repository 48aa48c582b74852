// What does a 32x32x16 MFMA cost in POWER on the MI355X, by operand type and operand values?  The network kernel runs the
// package at its power limit (tools/pipe_model.hip: 1.33-1.37 kW by rocm-smi, shader clock 1.95 GHz instead of 2.4), so its
// throughput is set by energy per item, not by cycles per item.  Every kernel here is a pure matrix-pipe stream (four
// independent accumulators per wave, operands in registers, 2 waves per SIMD on every CU) that runs for >= 100 ms; printed:
// sustained TFLOP/s (dense, as issued) and the effective shader clock (ticks of s_memtime / wall time).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/_bin/mfma_power tools/mfma_power.hip && tools/_bin/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef int i16v __attribute__((ext_vector_type(16)));
typedef int i4v __attribute__((ext_vector_type(4)));

// KIND: 0 f16 32x32x16, 1 bf16 32x32x16, 2 f16 16x16x32, 3 fp8 32x32x16, 4 i8 32x32x32
template <int KIND>
__global__ __launch_bounds__(512) void k(const uint4 *opa, const uint4 *opb, int iters, float *out, long long *cyc)
{
    const int tid = threadIdx.x, lane = tid & 63;
    // four operand pairs per lane, 16 bytes each (8 halves / 8 bf16 / 16 bytes of fp8 or i8)
    uint4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = opa[(i * 512 + tid) & 4095];
        b[i] = opb[(i * 512 + tid) & 4095];
    }
    const long long t0 = __builtin_readcyclecounter();
    float r = 0.f;
    if constexpr (KIND == 0 || KIND == 1) {
        f16v c[4] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if constexpr (KIND == 0)
                        c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(h8 *)&a[(i + u) & 3], *(h8 *)&b[i], c[i], 0, 0, 0);
                    else
                        c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(b8 *)&a[(i + u) & 3], *(b8 *)&b[i], c[i], 0, 0, 0);
                }
        }
        r = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    } else if constexpr (KIND == 2) {
        f4v c[4] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)   // (two 16x16x32 = the flops of one 32x32x16)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*(h8 *)&a[(i + u) & 3], *(h8 *)&b[i], c[i], 0, 0, 0);
        }
        r = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    } else if constexpr (KIND == 3) {
        f16v c[4] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const long x = ((long)a[(i + u) & 3].y << 32) | a[(i + u) & 3].x, y = ((long)b[i].y << 32) | b[i].x;
                    c[i] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(x, y, c[i], 0, 0, 0);
                }
        }
        r = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    } else {
        i16v c[4] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    c[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(*(i4v *)&a[(i + u) & 3], *(i4v *)&b[i], c[i], 0, 0, 0);
        }
        r = (float)(c[0][0] + c[1][1] + c[2][2] + c[3][3]);
    }
    const long long t1 = __builtin_readcyclecounter();
    if (r == 12345.678f) out[0] = r;
    if (lane == 0) cyc[blockIdx.x * 8 + (tid >> 6)] = t1 - t0;
}

enum Fill { ZEROS, F16_RANDOM, F16_LO, BF16_RANDOM, FP8_RANDOM, I8_RANDOM, F16_ONEBIT };

static unsigned short f16_bits(float f)
{
    _Float16 h = (_Float16)f;
    unsigned short u;
    memcpy(&u, &h, 2);
    return u;
}

static void fill(std::vector<unsigned short> &v, Fill how, unsigned seed)
{
    unsigned s = seed;
    for (size_t i = 0; i < v.size(); ++i) {
        s = s * 1664525u + 1013904223u;
        const float u = ((s >> 9) & 0x7fff) * (1.0f / 16384.0f) - 1.0f;   // [-1, 1)
        switch (how) {
        case ZEROS: v[i] = 0; break;
        case F16_RANDOM: v[i] = f16_bits(u * 8192.0f); break;
        case F16_LO: v[i] = f16_bits(u * 4.0f); break;                    // the residual plane: eleven binades below
        case BF16_RANDOM: { float f = u * 8192.0f; unsigned w; memcpy(&w, &f, 4); v[i] = (unsigned short)(w >> 16); break; }
        case FP8_RANDOM: v[i] = (unsigned short)(((s >> 8) & 0x7777) | ((s >> 3) & 0x8080)); break;   // (no NaN patterns)
        case I8_RANDOM: v[i] = (unsigned short)(s >> 12); break;
        case F16_ONEBIT: v[i] = f16_bits(u < 0 ? -1024.0f : 1024.0f); break;  // random signs, trivial mantissas
        }
    }
}

template <int KIND>
static void run(const char *name, Fill fa, Fill fb, double flop_per_iter_wave, int iters)
{
    std::vector<unsigned short> ha(4096 * 8), hb(4096 * 8);
    fill(ha, fa, 1u);
    fill(hb, fb, 77u);
    uint4 *da, *db; float *out; long long *cyc;
    (void)hipMalloc(&da, 4096 * 16); (void)hipMalloc(&db, 4096 * 16); (void)hipMalloc(&out, 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
    (void)hipMemcpy(da, ha.data(), 4096 * 16, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, hb.data(), 4096 * 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(512), 0, 0, da, db, 100, out, cyc);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(512), 0, 0, da, db, iters, out, cyc);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[2048];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0;
    for (int i = 0; i < 2048; ++i) c += (double)h[i];
    c /= 2048;
    const double flops = flop_per_iter_wave * iters * 2048.0;
    printf("%-46s %8.2f ms  %7.1f TFLOP/s (TOP/s)   clock %5.2f GHz   %5.1f ticks per 32768-flop MFMA per SIMD\n", name, ms,
           flops / (ms * 1e-3) * 1e-12, c / (ms * 1e6), c / (iters * 16.0 * 2));
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(out); (void)hipFree(cyc);
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 150000;
    const double F = 16.0 * 32768.0;   // flops per wave per iteration of the 32x32x16 kernels (16 MFMAs)
    run<0>("f16 32x32x16, zeros", ZEROS, ZEROS, F, iters);
    run<0>("f16 32x32x16, random hi x random hi", F16_RANDOM, F16_RANDOM, F, iters);
    run<0>("f16 32x32x16, random hi x residual-plane lo", F16_RANDOM, F16_LO, F, iters);
    run<0>("f16 32x32x16, +-1024 (signs only)", F16_ONEBIT, F16_ONEBIT, F, iters);
    run<1>("bf16 32x32x16, random", BF16_RANDOM, BF16_RANDOM, F, iters);
    run<2>("f16 16x16x32, random", F16_RANDOM, F16_RANDOM, F, iters);
    run<3>("fp8 32x32x16 (unscaled), random", FP8_RANDOM, FP8_RANDOM, F, iters);
    run<4>("i8 32x32x32, random", I8_RANDOM, I8_RANDOM, 2 * F, iters);
    run<0>("f16 32x32x16, random hi x random hi (again)", F16_RANDOM, F16_RANDOM, F, iters);
    return 0;
}

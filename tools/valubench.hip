// Development microbenchmark: issue cost (cycles per wave64 instruction and per SIMD) of the VALU / DPP / LDS
// instructions the AEV kernels are made of, at 1, 2, 4 and 8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valubench tools/valubench.hip && /tmp/valubench
// Every test runs a loop whose body is 32 independent copies of one instruction (inline asm, so the compiler
// cannot fold or reorder them); the wave reads the shader clock around the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)

#define KERNEL_VV(name, INSTR)                                                                       \
    __global__ __launch_bounds__(256) void name(int iters, long long *cyc, float *sink)               \
    {                                                                                                  \
        float a0 = threadIdx.x * 1e-3f + 0.5f, a1 = a0 + .1f, a2 = a0 + .2f, a3 = a0 + .3f;            \
        float a4 = a0 + .4f, a5 = a0 + .5f, a6 = a0 + .6f, a7 = a0 + .7f;                              \
        float b = 0.999f, c = 1e-6f;                                                                   \
        int s0 = __builtin_amdgcn_readfirstlane(iters), s1 = s0 + 1;                                   \
        unsigned long long m64 = __builtin_amdgcn_ballot_w64(threadIdx.x & 1);                         \
        __shared__ float lds[4096];                                                                    \
        lds[threadIdx.x] = a0; lds[threadIdx.x + 256] = a1;                                            \
        int la = (threadIdx.x & 63) * 16, lb = (threadIdx.x & 7) * 4, lc = (threadIdx.x & 63) * 4;     \
        __syncthreads();                                                                               \
        long long t0 = __builtin_readcyclecounter();                                                   \
        for (int it = 0; it < iters; ++it) {                                                           \
            REP4(asm volatile(INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)  \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6),  \
                                "+v"(a7), "+s"(s0), "+s"(s1)                                           \
                              : "v"(b), "v"(c), "v"(la), "v"(lb), "v"(lc), "s"(m64)                    \
                              : "memory", "vcc", "scc");)                                                            \
        }                                                                                              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                             \
        long long t1 = __builtin_readcyclecounter();                                                   \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 1234.5f) sink[0] = a0 + s0 + s1;                            \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;               \
    }

// operand numbers: %0..%7 = a0..a7, %8 = b, %9 = c, %10 = la (distinct 64-B addresses), %11 = lb (8 distinct), %12 = lc
#define I_FMA(n) "v_fma_f32 %" #n ", %" #n ", %10, %11\n"
#define I_MUL(n) "v_mul_f32 %" #n ", %" #n ", %10\n"
#define I_EXP(n) "v_exp_f32 %" #n ", %" #n "\n"
#define I_LOG(n) "v_log_f32 %" #n ", %" #n "\n"
#define I_RCP(n) "v_rcp_f32 %" #n ", %" #n "\n"
#define I_SQRT(n) "v_sqrt_f32 %" #n ", %" #n "\n"
#define I_RSQ(n) "v_rsq_f32 %" #n ", %" #n "\n"
#define I_COS(n) "v_cos_f32 %" #n ", %" #n "\n"
#define I_CND(n) "v_cndmask_b32 %" #n ", %" #n ", %10, vcc\n"
#define I_CNDS(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %10, %15\n"
#define I_CMPCND(n) "v_cmp_lt_f32 vcc, %" #n ", %10\nv_cndmask_b32 %" #n ", %" #n ", %11, vcc\n"
#define I_CMP(n) "v_cmp_lt_f32 vcc, %" #n ", %10\n"
#define I_MED3(n) "v_med3_f32 %" #n ", %" #n ", %10, %11\n"
#define I_MIN(n) "v_min_i32 %" #n ", %" #n ", %10\n"
#define I_SUBCO(n) "v_sub_u32 %" #n ", %" #n ", %10\n"
#define I_RDLANE(n) "v_readlane_b32 %8, %" #n ", 3\n"
#define I_CVTI(n) "v_cvt_i32_f32 %" #n ", %" #n "\n"
#define I_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %10\n"
#define I_MAD24(n) "v_mad_u32_u24 %" #n ", %" #n ", %10, %11\n"
#define I_MAX(n) "v_max_f32 %" #n ", %" #n ", %10\n"
#define I_DPPQ(n) "v_add_f32_dpp %" #n ", %" #n ", %10 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_DPPR(n) "v_add_f32_dpp %" #n ", %" #n ", %10 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define I_DPPM(n) "v_add_f32_dpp %" #n ", %" #n ", %10 row_mirror row_mask:0xf bank_mask:0xf\n"
#define I_MOVDPP(n) "v_mov_b32_dpp %" #n ", %10 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n"
#define I_FMAEXP(n) "v_fma_f32 %" #n ", %" #n ", %10, %11\nv_exp_f32 %" #n ", %" #n "\n"
#define I_FMA3EXP(n) "v_fma_f32 %" #n ", %" #n ", %10, %11\nv_fma_f32 %" #n ", %" #n ", %10, %11\nv_fma_f32 %" #n ", %" #n ", %10, %11\nv_exp_f32 %" #n ", %" #n "\n"
#define I_FMASALU(n) "v_fma_f32 %" #n ", %" #n ", %10, %11\ns_add_u32 %8, %8, 1\n"
#define I_SALU(n) "s_add_u32 %8, %8, 1\ns_lshr_b32 %9, %8, 3\n"
#define I_DSR32(n) "ds_read_b32 %" #n ", %14\n"
#define I_DSR128B(n) "ds_read_b32 %" #n ", %13\n"
#define I_DSADD(n) "ds_add_f32 %14, %" #n "\n"
#define I_DSADD8(n) "ds_add_f32 %13, %" #n "\n"
#define SW_0 "%0, %1"
#define SW_1 "%2, %3"
#define SW_2 "%4, %5"
#define SW_3 "%6, %7"
#define SW_4 "%1, %2"
#define SW_5 "%3, %4"
#define SW_6 "%5, %6"
#define SW_7 "%7, %0"
#define I_SWAP32(n) "v_permlane32_swap_b32 " SW_##n "\n"
#define I_SWAP16(n) "v_permlane16_swap_b32 " SW_##n "\n"
#define I_MFMA4(n) ""

KERNEL_VV(k_fma, I_FMA)
KERNEL_VV(k_mul, I_MUL)
KERNEL_VV(k_exp, I_EXP)
KERNEL_VV(k_log, I_LOG)
KERNEL_VV(k_rcp, I_RCP)
KERNEL_VV(k_sqrt, I_SQRT)
KERNEL_VV(k_rsq, I_RSQ)
KERNEL_VV(k_cos, I_COS)
KERNEL_VV(k_cnd, I_CND)
KERNEL_VV(k_max, I_MAX)
KERNEL_VV(k_cnds, I_CNDS)
KERNEL_VV(k_cmpcnd, I_CMPCND)
KERNEL_VV(k_cmp, I_CMP)
KERNEL_VV(k_med3, I_MED3)
KERNEL_VV(k_min, I_MIN)
KERNEL_VV(k_subco, I_SUBCO)
KERNEL_VV(k_cvti, I_CVTI)
KERNEL_VV(k_mullo, I_MULLO)
KERNEL_VV(k_mad24, I_MAD24)
KERNEL_VV(k_dppq, I_DPPQ)
KERNEL_VV(k_dppr, I_DPPR)
KERNEL_VV(k_dppm, I_DPPM)
KERNEL_VV(k_movdpp, I_MOVDPP)
KERNEL_VV(k_fmaexp, I_FMAEXP)
KERNEL_VV(k_fma3exp, I_FMA3EXP)
KERNEL_VV(k_fmasalu, I_FMASALU)
KERNEL_VV(k_salu, I_SALU)
KERNEL_VV(k_dsr32, I_DSR32)
KERNEL_VV(k_dsr32b, I_DSR128B)
KERNEL_VV(k_dsadd, I_DSADD)
KERNEL_VV(k_dsadd8, I_DSADD8)
KERNEL_VV(k_swap32, I_SWAP32)
KERNEL_VV(k_swap16, I_SWAP16)

// 64-bit / 128-bit operand forms need their own register classes
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));

#define KERNEL_PK(name, OP)                                                                            \
    __global__ __launch_bounds__(256) void name(int iters, long long *cyc, float *sink)                 \
    {                                                                                                  \
        float2v a0 = {threadIdx.x * 1e-3f + .5f, .25f}, a1 = a0 + .1f, a2 = a0 + .2f, a3 = a0 + .3f;    \
        float2v a4 = a0 + .4f, a5 = a0 + .5f, a6 = a0 + .6f, a7 = a0 + .7f;                            \
        float2v b = {0.999f, 0.998f}, c = {1e-6f, 2e-6f};                                              \
        long long t0 = __builtin_readcyclecounter();                                                   \
        for (int it = 0; it < iters; ++it) {                                                           \
            REP4(asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                          \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6),  \
                                "+v"(a7)                                                               \
                              : "v"(b), "v"(c));)                                                      \
        }                                                                                              \
        long long t1 = __builtin_readcyclecounter();                                                   \
        float2v s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                             \
        if (s.x + s.y == 1234.5f) sink[0] = s.x;                                                       \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;               \
    }
#define P_FMA(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define P_MUL(n) "v_pk_mul_f32 %" #n ", %" #n ", %8\n"
#define P_ADD(n) "v_pk_add_f32 %" #n ", %" #n ", %8\n"
KERNEL_PK(k_pkfma, P_FMA)
KERNEL_PK(k_pkmul, P_MUL)
KERNEL_PK(k_pkadd, P_ADD)

__global__ __launch_bounds__(256) void k_dsr128(int iters, long long *cyc, float *sink, int mode)
{
    __shared__ float4v lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = float4v{(float)i, 1.f, 2.f, 3.f};
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // mode 0: 64 distinct consecutive float4; 1: 4 distinct addresses 128 B apart (block-broadcast pattern); 2: all same
    int addr = mode == 0 ? lane * 16 : mode == 1 ? (lane & 3) * 128 : 0;
    float4v a0, a1, a2, a3, a4, a5, a6, a7;
    float4v acc = {0.f, 0.f, 0.f, 0.f};
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        REP4(asm volatile("ds_read_b128 %0, %8\nds_read_b128 %1, %8 offset:16\nds_read_b128 %2, %8 offset:32\n"
                          "ds_read_b128 %3, %8 offset:48\nds_read_b128 %4, %8 offset:64\nds_read_b128 %5, %8 offset:80\n"
                          "ds_read_b128 %6, %8 offset:96\nds_read_b128 %7, %8 offset:112\ns_waitcnt lgkmcnt(0)\n"
                          : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7)
                          : "v"(addr)
                          : "memory");
             acc += a0 + a7;)
    }
    long long t1 = __builtin_readcyclecounter();
    if (acc.x == 1234.5f) sink[0] = acc.y;
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_mfma4(int iters, long long *cyc, float *sink)
{
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float a = threadIdx.x * 1e-3f, b = 0.5f;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    if (c0[0] + c1[1] + c2[2] + c3[3] == 1234.5f) sink[0] = c0[0];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

struct Test {
    const char *name;
    void (*fn)(int, long long *, float *);
    int per_iter;   // instructions per loop iteration
};

int main(int argc, char **argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    long long *cyc;
    float *sink;
    hipMalloc(&cyc, 256 * 8 * 4 * sizeof(long long));
    hipMalloc(&sink, 64);
    const int iters = 2000;
    std::vector<Test> tests = {
        {"v_fma_f32", k_fma, 32}, {"v_mul_f32", k_mul, 32}, {"v_exp_f32", k_exp, 32}, {"v_log_f32", k_log, 32},
        {"v_rcp_f32", k_rcp, 32}, {"v_sqrt_f32", k_sqrt, 32}, {"v_rsq_f32", k_rsq, 32}, {"v_cos_f32", k_cos, 32},
        {"v_cndmask_b32 vcc", k_cnd, 32}, {"v_cndmask_b32 sgpr mask", k_cnds, 32}, {"v_cmp+v_cndmask (per pair)", k_cmpcnd, 32},
        {"v_cmp_lt_f32", k_cmp, 32}, {"v_med3_f32", k_med3, 32}, {"v_min_i32", k_min, 32}, {"v_sub_u32", k_subco, 32},
        {"v_cvt_i32_f32", k_cvti, 32}, {"v_mul_lo_u32", k_mullo, 32}, {"v_mad_u32_u24", k_mad24, 32}, {"v_max_f32", k_max, 32},
        {"v_add dpp quad_perm", k_dppq, 32}, {"v_add dpp row_shr:4", k_dppr, 32}, {"v_add dpp row_mirror", k_dppm, 32},
        {"v_mov dpp quad bcast", k_movdpp, 32},
        {"fma+exp pairs (per pair)", k_fmaexp, 32}, {"3fma+exp (per group of 4)", k_fma3exp, 32},
        {"ds_read_b32 distinct", k_dsr32, 32}, {"ds_read_b32 8 addrs", k_dsr32b, 32},
        {"ds_add_f32 distinct", k_dsadd, 32}, {"ds_add_f32 8-way same addr", k_dsadd8, 32},
        {"v_permlane32_swap", k_swap32, 32}, {"v_permlane16_swap", k_swap16, 32},
        {"v_pk_fma_f32", k_pkfma, 32}, {"v_pk_mul_f32", k_pkmul, 32}, {"v_pk_add_f32", k_pkadd, 32},
        {"v_mfma_f32_4x4x1_16B_f32", k_mfma4, 32},
        {"fma + s_add (per pair)", k_fmasalu, 32}, {"2 SALU (per pair)", k_salu, 32},
    };
    printf("%-30s %s\n", "instruction", "cycles per wave-instruction seen by ONE wave | per SIMD (= /waves per SIMD), for 1 2 4 8 waves/SIMD");
    for (auto &t : tests) {
        printf("%-30s", t.name);
        for (int wps : {1, 2, 4, 8}) {
            // 256-thread blocks = one wave per SIMD each; wps blocks per CU
            hipLaunchKernelGGL(t.fn, dim3(256 * wps), dim3(256), 0, 0, 10, cyc, sink);
            hipLaunchKernelGGL(t.fn, dim3(256 * wps), dim3(256), 0, 0, iters, cyc, sink);
            hipDeviceSynchronize();
            std::vector<long long> h(256 * wps * 4);
            hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
            double s = 0;
            for (auto v : h) s += (double)v;
            const double per = s / h.size() / ((double)iters * t.per_iter);
            printf("  %6.2f|%5.2f", per, per / wps);
        }
        printf("\n");
    }
    for (int mode = 0; mode < 3; ++mode) {
        printf("ds_read_b128 mode %d            ", mode);
        for (int wps : {1, 2, 4, 8}) {
            hipLaunchKernelGGL(k_dsr128, dim3(256 * wps), dim3(256), 0, 0, 10, cyc, sink, mode);
            hipLaunchKernelGGL(k_dsr128, dim3(256 * wps), dim3(256), 0, 0, iters, cyc, sink, mode);
            hipDeviceSynchronize();
            std::vector<long long> h(256 * wps * 4);
            hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
            double s = 0;
            for (auto v : h) s += (double)v;
            const double per = s / h.size() / ((double)iters * 32);
            printf("  %6.2f|%5.2f", per, per / wps);
        }
        printf("\n");
    }
    return 0;
}

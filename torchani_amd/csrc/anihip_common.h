// Shared device/host declarations of libanihip (gfx950 only; wave64 is hard-coded).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/anihip.h"

namespace anihip {

constexpr int WAVE = 64;
constexpr int MAX_S = ANIHIP_MAX_SPECIES;
constexpr int MAXA = ANIHIP_MAX_ANG;
constexpr int MAXR = ANIHIP_MAX_RAD;
constexpr int META_W = ANIHIP_META_WORDS;
constexpr uint32_t IDX_MASK = 0x0FFFFFFFu;  // low 28 bits of ent.w = atom index, high bits = species
constexpr uint32_t SP_PAD = 7u;             // species code of a padding atom in packed positions

// offsets into the constant table (anihip_aev_table_pack)
constexpr int TAB_SHFR = 0, TAB_SHFA = 32, TAB_COSZ = 48, TAB_SINZ = 64;
constexpr int TAB_SHFRQ = 80, TAB_SHFAQ = 96, TAB_COSZH = 112, TAB_SINZH = 128;   // pre-scaled copies
// Gaussian recurrences of the backward kernel (ANIHIP_AEV_REC_BWD; 16 / 8 x 4 / 4 x 8 grids only, whose ShfR block uses
// sixteen of its 32 slots and whose q_A ShfA and sin / 2 blocks use at most eight of their sixteen): D = spacing of
// the pre-scaled shifts, K_m = exp2(-(m D)^2), all evaluated in double from the fp32 shift arrays (every wave-uniform product
// is made on the host: gfx950 has no scalar float multiply, uniform products made in the kernel would sit in vector registers)
constexpr int TAB_RECR = 16;    // 2 D_R | K_1 | D_R K_1 | K_2 | 2 D_R K_2 | exp2(8 D_R^2) | exp2(16 D_R^2) | exp2(-8 D_R^2) | exp2(-16 D_R^2) | 120 - 16 D_R^2
constexpr int TAB_RECA = 104;   // 2 D_A
constexpr int TAB_RECAK = 136;  // (K_m, m D_A K_m), m = 1 .. 4

void set_error(const char *fmt, ...);

#define ANIHIP_CHECK_HIP(expr)                                                            \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            anihip::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return 2;                                                                     \
        }                                                                                 \
    } while (0)

#define ANIHIP_REQUIRE(cond, ...)             \
    do {                                      \
        if (!(cond)) {                        \
            anihip::set_error(__VA_ARGS__);   \
            return 1;                         \
        }                                     \
    } while (0)

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// number of set bits of `m` below this lane
__device__ __forceinline__ int mbcnt(uint64_t m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
}

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// sum over the 64 lanes, the same value in every lane.  Scan inside the 16-lane DPP rows (four shifted adds on the VALU), then
// the four row totals through scalar registers: a dozen instructions.  (Six __shfl_xor steps are six ds_bpermute round
// trips through the LDS crossbar, ~100 clocks each and dependent: the AEV backward spends three of these sums per atom.)
__device__ __forceinline__ float wave_sum(float v)
{
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, true));   // row_shr:1
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xF, 0xF, true));   // row_shr:2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xF, 0xF, true));   // row_shr:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xF, 0xF, true));   // row_shr:8
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 15));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 47));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
    return (r0 + r1) + (r2 + r3);
}

// per-wave LDS hand-off between lanes: keep the compiler from reordering LDS traffic around it
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


// Stream-ordered zero fill by a kernel instead of hipMemsetAsync: memset nodes of a captured HIP graph
// faulted on replay (ROCm 7.0), a plain kernel node replays fine.
static __global__ void k_zero_words(uint32_t *p, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
static inline void zero_words_async(hipStream_t stream, void *p, size_t bytes)
{
    const size_t n = bytes / 4;   // (all callers pass multiples of 4)
    if (n == 0) return;
    size_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_zero_words, dim3((unsigned)blocks), dim3(256), 0, stream, (uint32_t *)p, n);
}

}  // namespace anihip

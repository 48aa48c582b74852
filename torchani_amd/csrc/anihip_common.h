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

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// per-wave LDS hand-off between lanes: keep the compiler from reordering LDS traffic around it
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace anihip

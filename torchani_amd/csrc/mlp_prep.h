// Preparation and finishing kernels of the network stage (csrc/mlp_prep.hip), shared with the host entry points and GEMM kernels of
// csrc/mlp.hip: the species buckets, the tile table of the fused kernel (and its cost-sorted copy), the one-launch preparation
// of small inputs, the member-energy finish, the padding rows and the molecular energy sums.
#pragma once
#include "anihip_common.h"
#include "train.h"

namespace anihip {

constexpr int BM = 128;   // rows per tile of the grouped GEMM kernels (csrc/mlp.hip): ctl[CTL_TILE + s] counts these
constexpr int SP_CHUNK = 1024;  // atoms per wave in the bucketing kernels
constexpr int SMALL_PREP_MAX = 16384;
constexpr int SMALL_PREP_WAVES = 16;
constexpr int TO_CHUNK = 16;   // tiles a workgroup of k_tile_order places (4 waves x 4 tiles)
#ifndef ANIHIP_TILE_QUEUE
#define ANIHIP_TILE_QUEUE 1   // 0: tiles b, b + grid, ... at every size (development A/B)
#endif

struct FinishArgs {   // k_fused_finish, or the extra blocks of k_gemm_l0s
    const int *ctl;
    const int *perm;
    const float *member_part;
    float *atomic_e, *member_e;
    int64_t n_atoms;
    int S, M;
    int first_block;   // k_gemm_l0s: blocks from here on do this instead of a tile (0: none)
};

// sum the per-member energies of the fused kernel: atomic_e = mean_m, optional [M][n_atoms] copy
__device__ __forceinline__ void fused_finish(const FinishArgs &f, int64_t first, int64_t stride)
{
    const int64_t n = f.ctl[CTL_OFF + f.S];
    for (int64_t p = first; p < n; p += stride) {
        const int atom = f.perm[p];
        float e = 0.f;
        for (int m = 0; m < f.M; ++m) {
            const float v = f.member_part[p * f.M + m];
            e += v;
            if (f.member_e) f.member_e[(int64_t)m * f.n_atoms + atom] = v;
        }
        f.atomic_e[atom] = e / (float)f.M;
    }
}

struct TileOrderArgs {
    const int4 *tile_tab;
    const int *tile_rows;
    int4 *tile_tab2;
    int *tile_rows2;
    int tiles_total;
    int H1[MAX_S];
};

// ---- launchers (stream-ordered, no host synchronisation) ------------------------------------------------------------------
// species buckets of the atoms lo..hi: control block zeroed, count -> scan -> scatter (stable: index order inside a species);
// the outputs of padding atoms (atomic_e, grad_aev row, member_e; each optional) are zeroed on the way.  chunk_cnt: scratch of
// ceil(n / SP_CHUNK) * MAX_S ints
void launch_bucketing(hipStream_t stream, int64_t lo, int64_t hi, const int32_t *species, int S, int *ctl, int words_to_zero,
                      int *chunk_cnt, int *perm, float *atomic_e, float *grad_aev, int L, float *member_e, int M, int64_t n_atoms);
// the same, the tile table and the padding rows in ONE launch (n <= SMALL_PREP_MAX atoms); tile_tab may be NULL
int launch_small_prep(hipStream_t stream, int64_t lo, int64_t hi, const int32_t *species, int S, int *ctl, int *perm,
                      const uint32_t *slab_mask, uint32_t all_slabs, int tiles_total, int rows_per_tile, int4 *tile_tab,
                      int *tile_rows, float *atomic_e, float *grad_aev, int L, float *member_e, int M, int64_t n_atoms);
void launch_tile_table(hipStream_t stream, const int *ctl, int S, const int *perm, const uint32_t *slab_mask, uint32_t all_slabs,
                       int tiles_total, int rows_per_tile, int4 *tile_tab, int *tile_rows, int ani_species = 0);
void launch_tile_order(hipStream_t stream, const TileOrderArgs &a);
void launch_fused_finish(hipStream_t stream, const FinishArgs &f, int64_t n);
void launch_zero_padding(hipStream_t stream, int64_t lo, int64_t hi, const int32_t *species, float *atomic_e, float *grad_aev,
                         int L, float *member_e, int M, int64_t n_atoms);

}  // namespace anihip

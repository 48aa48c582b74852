// Shared between csrc/mlp.hip (host entry points, layer-by-layer kernels, preparation kernels) and csrc/mlp_fused.hip (the fused
// network kernel k_mlp_fused): the kernel's argument block, its LDS geometry and the launcher the host side calls.
#pragma once
#include "anihip_common.h"
#include "train.h"

namespace anihip {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const v4f gf4;  // global-memory float4 (forces global_load_dwordx4)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) const h8 gh8;


// d E / d act0 in tile-major MFMA-fragment order: halves / floats from the start of a 64-row x H1 member block to
// the 2-KB unit (column block cb, row block rb, 16-column half ks)
__device__ __forceinline__ int tm_unit(int cb, int rb, int ks) { return ((cb * 2 + rb) * 2 + ks) * 512; }

// K' order helpers: slab j -> first AEV column and number of valid columns
__device__ __forceinline__ int kp_col(int kp_rad, int j)
{
    const int rs = (kp_rad + 31) >> 5;
    return j < rs ? 32 * j : kp_rad + 32 * (j - rs);
}
__device__ __forceinline__ int kp_valid(int kp_rad, int j)
{
    const int rs = (kp_rad + 31) >> 5;
    return (j == rs - 1) ? kp_rad - 32 * (rs - 1) : 32;
}

constexpr int FR_MAXH = 256;      // largest padded hidden width (8 column blocks)
// layer-0 backward inside the fused kernel from this many atoms on.  Round 6 (16-column units in phase 5: no hand-over, no mid-phase
// barrier): measured crossover on water boxes 17 496 atoms 0.387 vs 0.344 ms (loses), 24 000 0.405 vs 0.429, 31 944 0.466 vs
// 0.535, 59 049 0.812 vs 0.860; solvated 1hz5 (46 357 atoms, five elements) 1.09 vs 1.18 ms per step, 1C17 (16 649) 0.95 vs 0.79
// (loses).  (65 536 in rounds 4-5.)
constexpr int64_t FUSED_L0B_MIN_ATOMS = 24000;
#ifndef ANIHIP_OWNER_GROUP
#define ANIHIP_OWNER_GROUP 1
#endif
constexpr int FUSED_OWNER_GROUP = ANIHIP_OWNER_GROUP;   // tiles a workgroup takes through the members together (owner order)
constexpr int FRAG = 512;         // halves per fragment plane: 64 lanes x 8
constexpr int FR_SLAB_LD = 48;    // halves per staged slab row (32 + 16: conflict-free ds_read_b128 of 16-row fragments)
constexpr int FR_GROUP = 3;       // slabs per staging slot

template <int RB, int NB>
struct FusedCfg {
    static constexpr int NW = 8 / NB;               // waves; wave w owns column blocks w, w + NW, ...
    static constexpr int THREADS = 64 * NW;
    static constexpr int ROWS = 32 * RB;            // atoms per workgroup (= THREADS / 8: one staging piece each)
    static constexpr int TPR = THREADS / ROWS;      // = 8
    static constexpr int DEPTH = 2;                 // k2 steps (32 reduction indices) of weight fragments in flight per wave (even)
    // two workgroups per CU hide each other's latencies: per-column parameters are then fetched right where
    // they are used instead of ahead of the GEMM (32 registers less per array during the MFMA loops)
    static constexpr int SLAB = 2 * ROWS * FR_SLAB_LD;            // halves per staged slab {hi plane, lo plane}
    // fixed part of the dynamic LDS: [0] tile max | energy partials [NW][ROWS] | staging slot 0
    // fixed part of the dynamic LDS: [0] tile max | per-species {tile-major base of d0, first sorted position} |
    // energy partials [NW][ROWS] | staging slot 0
    static constexpr int FIXED_BYTES = 16 + 128 + NW * ROWS * 4 + 2 * ROWS * 4;   // ... | atoms of the tile rows [2][ROWS]
    // slot 0 doubles as the home of a tile's KEPT layer-0 operand (owner order, tiles with <= 4 flagged slabs): four slabs,
    // {hi, lo} planes, unpadded 64-byte rows whose 16-byte pieces are XOR-swizzled with (-(row >> 2)) & 3 (conflict-free reads
    // of the 16-row MFMA fragments and of the staging writes)
    static constexpr int KEEP_SLABS = 4, SLABU = 2 * ROWS * 32;
    static constexpr int SLOT0 = FR_GROUP * SLAB > KEEP_SLABS * SLABU ? FR_GROUP * SLAB : KEEP_SLABS * SLABU;
    static constexpr int FIXED_HALVES = FIXED_BYTES / 2 + SLOT0;
    static_assert(THREADS == ROWS * 8, "one 16-B staging piece per thread");
};

struct FusedSpecies {
    int H1, H2, H3;                          // padded widths
    // fragment-ordered planes, per member: [N/32][K/16][2][64][8]
    const _Float16 *w0, *w1, *w2, *w2t, *w1t;  // (N,K) = (H1,K0p), (H2,H1), (H3,H2), (H2,H3), (H1,H2)
    const _Float16 *w0t;                       // (N,K) = (K0p, H1): layer 0 transposed (l0b), or NULL
    float is0, is1, is2;                     // 1 / weight scales of layers 0, 1 and 2
    const float *b0, *b1, *b2;               // [M*H1], [M][H2], [M][H3]
    const float *w3, *b3;                    // output layer [M][H3], [M]
    const float *bounds;                     // [M][8] operand bounds (include/anihip.h, fused_bounds)
};

struct FusedArgs {
    FusedSpecies sp[MAX_S];
    const int *ctl;
    unsigned *amax;
    const float *aev;          // [n_atoms][L]
    int64_t L;
    int kp_rad, n_slabs;       // slab order of the layer-0 planes (kp_rad = 0: plain order, 32-column slabs)
    const uint32_t *slab_mask; // per atom, or NULL (all slabs)
    float *d0;                 // [n][ld0]: out: d E / d act0 (member m at columns m*H1..)
    int64_t ld0;
    int d0_tm;                 // d0 in the tile-major layout (tm_species_base) instead of [n][ld0]
    const int *perm;           // sorted position -> atom
    const int4 *tile_tab;      // [tiles_total] {species (-1: empty), first sorted position, rows, slab mask}
    const int *tile_rows;      // [tiles_total][rows per tile] atom of each row (short tiles: last atom repeated)
    // owner order over single tiles: workgroup b starts with tile t_lo + b of its launch's range and draws its next tile from
    // *queue (a counter that starts at ZERO: position q = tile t_lo + grid + q) -- or NULL: tiles b, b + grid, ...  Mid-size
    // systems: the table is sorted by falling tile cost first (k_tile_order); per-species launches: one counter per species, the
    // launches of two species overlap on two streams and a workgroup that starts late draws fewer tiles
    int *queue;
    float *member_part;        // [n][M] per-member atomic energies (summed by k_fused_finish)
    int S, M;
    int tiles_total;           // upper bound of the number of tiles (work items = tiles_total * M)
    float alpha, inv_alpha;
    int want_grad;
    int l0b;                   // layer-0 backward inside the kernel (needs owner = 1): d E / d AEV -> grad_aev, no d0
    float *grad_aev;           // [n_atoms][L] (l0b): the tile's flagged slabs of its atoms' rows, summed over the members
    int only_species;          // owner order: the launch takes the tiles of this species only (-1: of every species)
    int owner;                 // item order: 0 = member-major sweep over the tiles; G > 0: a workgroup OWNS its tiles, taken in groups of G
    unsigned long long *trace;   // development builds (-DANIHIP_DEV_TRACE, tools/fused_trace.py): [item][wave][32] stamps
    // TRAIN instantiation (anihip_mlp_train_forward of a split-fp16 pack): everything the weight gradients need leaves the
    // kernel as fp32 rows in sorted order, member m at columns m * H: the activations of the three hidden layers and
    // d e / d (pre-activation) of layers 2 and 1 for a unit upstream gradient (layer 0's is d0)
    float *tr_act[3];
    float *tr_dlt[3];          // ([0] unused: d0)
    int64_t tr_ld[3];
};
constexpr int FR_XPAD = 16;       // halves of padding per activation-plane row
// Tiles are handed out by falling cost up to this many of them (a solvated protein: 22 % of the tiles flag more than four slabs --
// up to 17: five passes of phase 5 -- and with tiles b, b + grid, ... the slowest workgroup carries 1.5 x the mean load; beyond
// a few dozen rounds of tiles the static order evens out by itself and keeps neighbouring tiles on neighbouring CUs)
constexpr int64_t FUSED_TILE_QUEUE_MAX = 8192;

// instantiations of k_mlp_fused (csrc/mlp_fused.hip)
enum FusedVariant { FUSED_CELU = 0, FUSED_CELU_L0B = 1, FUSED_CELU_L0B_B2 = 2, FUSED_GELU = 3, FUSED_TRAIN = 4,
                    // compile-time widths of the ANI-2x networks: H 256 / 192 / 160, N O 192 / 160 / 128, C 224 / 192 / 160, S F Cl 160 / 128 / 96
                    FUSED_CELU_L0B_256 = 5, FUSED_CELU_L0B_192 = 6, FUSED_CELU_L0B_224 = 7, FUSED_CELU_L0B_160 = 8 };
const void *fused_kernel(int variant);   // (for hipFuncSetAttribute)
void launch_fused(int variant, unsigned grid, size_t lds_bytes, hipStream_t stream, const FusedArgs &f);

}  // namespace anihip

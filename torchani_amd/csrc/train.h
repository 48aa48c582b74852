// Internal declarations shared by mlp.hip and train.hip (training passes of the ensemble; not part of the C ABI).
#pragma once
#include "anihip_common.h"

namespace anihip {

// control block layout (ints) at the head of the network workspace (written by the species bucketing kernels)
constexpr int CTL_CNT = 0;      // [8]  atoms per species
constexpr int CTL_CURSOR = 8;   // [8]  scatter cursors
constexpr int CTL_OFF = 16;     // [9]  first sorted position of each species
constexpr int CTL_TILE = 32;    // [9]  first row tile of each species
constexpr int CTL_QUEUE = 41;    // [1]  tile queue of the fused kernel (owner order, tiles handed out by falling cost)
constexpr int CTL_WORDS = 48;
// behind the control block: running |max| of the intermediate tensors of the layer-by-layer split-fp16 kernels, per stage and
// species, spread over slots to keep the atomics off a single address (float bits)
constexpr int AMAX_STAGES = 8, AMAX_SLOTS = 32;
constexpr int AMAX_WORDS = AMAX_STAGES * MAX_S * AMAX_SLOTS;
// stages 0..5: the layer-by-layer kernels and the fused kernel's d0 scale; of the training step's weight gradients:
constexpr int AMAX_STAGE_ACT0 = 6;    // [species] max |act0| over the tiles and members of the fused training kernel
constexpr int AMAX_STAGE_GATOM = 7;   // [0] max |d Loss / d atomic_e| over the atoms of the call

// dW = D^T X over the rows (atoms) of one species on v_mfma_f32_32x32x16_bf16 with three-way bf16 splits (train.hip)
struct WgradB3Problem {
    const float *X;   // input of the layer: [rows][ldx], batch b at columns b * x_boff
    int64_t ldx;
    int x_boff;
    int k_valid;      // input width (multiple of 4): columns >= k_valid are neither read nor written
    const float *D;   // d E / d (pre-activation) of the layer for a unit upstream gradient: [rows][ldd], batch b at b * d_boff
    int64_t ldd;
    int d_boff;
    int N;            // D columns of one batch entry (layer 0: all members side by side, N = M * n_per)
    int n_per;        // output units per member
    float *dW;        // member 0's [n_per][ldw] array (torch.nn.Linear layout); member m at + m * w_mstride
    int64_t ldw, w_mstride;
};
struct WgradB3Args {
    WgradB3Problem prob[MAX_S];
    const int *ctl;
    const int *perm;       // sorted position -> atom (the rows of D, and of X unless x_gather, are in sorted order)
    const int *x_gather;   // sorted position -> source row of X (layer 0: the AEV rows lie in atom order) or NULL
    const float *g_atom;   // upstream d Loss / d atomic_e per ATOM: row p of D is scaled by g_atom[perm[p]]
    int S, batch, ki_max, nj_max;
    int rows_per_chunk;    // atoms per workgroup (multiple of 32): partial tiles of the chunks meet in dW through float atomics
    // layer 0 of a whole system in the ANI layout of the AEV row (x_slab_rad = its radial length = 16 per species, then one
    // 32-column block per species pair): the columns of species (pairs) that do not occur in the system are zero for every
    // atom, so their gradient columns are zero -- the X tiles run over the COMPACTED list of the slabs that can be non-zero
    // (H C N O under ANI-2x: 12 of 32 slabs, 3 column tiles instead of 8).  0: plain columns
    int x_slab_rad, ani_species;
    // fp16 x 3 arithmetic (round 6; amax and bounds[0] non-NULL): scales from bounds instead of a pass over the data --
    // bounds[s] = anihip_species_net.fused_bounds ([M][8]), amax = the workspace's running-max table holding max |g_atom|
    // (AMAX_STAGE_GATOM) and max |act0| per species (AMAX_STAGE_ACT0, written by the fused training kernel); layer = 0, 1, 2
    const float *bounds[MAX_S];
    const unsigned *amax;
    int layer, M;
    // bias gradients on the way (workgroups of the first X tile only): gbias[s][member][j] += sum_a g_a D[a][j]
    float *gbias[MAX_S];
    int64_t b_mstride[MAX_S];   // floats between the members' bias gradients
};
// rows_total: atoms of all species together (bounds the number of row chunks)
void launch_wgrad_b3(hipStream_t stream, const WgradB3Args &a, int64_t rows_total);
void launch_absmax(hipStream_t stream, const float *x, int64_t n, unsigned *amax, int stage);

// anihip_mlp_repack of an ANIHIP_MLP_F16X3 descriptor (train.hip)
int repack_f16(hipStream_t stream, const anihip_mlp_desc *d, const void *const *src, const int32_t *out_in, int32_t *status,
               int32_t flags);

}  // namespace anihip

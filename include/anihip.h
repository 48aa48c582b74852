/*
 * anihip.h -- C ABI of libanihip.so, the MI355X (gfx950) ANI hot-path engine.
 *
 * One shared object replaces, for the ANI energy+forces path, the three native plug-ins of the
 * reference (paths relative to /root/reference/torchani/):
 *   cuaev.so      csrc/cuaev.cpp:246-294   (cuaev::run, run_with_half_nbrlist, run_with_full_nbrlist,
 *                                           CuaevComputer::{forward,backward}, csrc/aev.cu:1687-2067)
 *   cell_list.so  csrc/cell_list.cpp:342-363 (cell_list::cell_list)
 *   mnp.so        csrc/mnp.cpp:238-280     (mnp::run, MultiNetFunction fwd / input-grad bwd)
 *
 * Conventions:
 *   - plain C: raw device pointers, explicit sizes, no torch types, no exceptions across the ABI;
 *   - every function returns 0 on success, non-zero on error; anihip_last_error() gives the text
 *     (thread-local), replacing the reference's TORCH_CHECK -> RuntimeError (csrc/aev.cu:1693-1710);
 *   - the CALLER owns all device memory (inputs, outputs, workspaces); the library never allocates,
 *     frees or retains device pointers across calls;
 *   - every call is asynchronous on the given hipStream_t (passed as void*), never synchronises the
 *     host (the reference syncs >= 3 times per forward, csrc/aev.cu:1292,1763-1767);
 *   - capacity overflows are reported through the device-side `status` words (never assert/trap like
 *     csrc/aev.cu:229); the host wrapper reads them when convenient;
 *   - fp32 compute, int32 indices, fp64 energy accumulation.
 *
 * Atoms are the flattened [C*A] array of the reference's (species[C,A], coords[C,A,3]) pair
 * (aev/_computer.py:193-199); species are element indices 0..S-1, padding = -1 (utils.py:67-74).
 */
#ifndef ANIHIP_H
#define ANIHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ANIHIP_ABI_VERSION 12

/* status word bits written by the kernels into status[0] */
#define ANIHIP_ST_ENTRY_OVERFLOW 1u   /* neighbor entries exceeded ent_capacity */
#define ANIHIP_ST_ROW_OVERFLOW 2u     /* one atom has > ANIHIP_MAX_ANG / ANIHIP_MAX_RAD neighbors or > 255 of one species */
#define ANIHIP_ST_GRID_OVERFLOW 4u    /* internal: grid coarsened to fit max_cells (not an error) */
#define ANIHIP_STATUS_WORDS 8         /* status[1] = total neighbor entries, status[2] = #cells */

#define ANIHIP_MAX_SPECIES 8
#define ANIHIP_MAX_ANG 128 /* angular neighbors per atom (cuAEV's analogous bound: csrc/aev.cu:11) */
#define ANIHIP_MAX_RAD 256 /* radial neighbors per atom */
#define ANIHIP_META_WORDS 6 /* uint32 words of per-atom neighbor metadata */

/* Cutoff envelopes: cutoffs.py:74-81 (CutoffCosine, 0.5 cos(pi r / Rc) + 0.5) and cutoffs.py:84-101 (CutoffSmooth,
 * order 2, eps 1e-10: exp(1 - 1 / max(eps, 1 - (r/Rc)^2))); the reference's kernels select them with the
 * use_cos_cutoff template flag (csrc/aev.cu:150-178). */
#define ANIHIP_CUTOFF_COSINE 0
#define ANIHIP_CUTOFF_SMOOTH 1

/* Scalar AEV hyper-parameters; replaces the CuaevComputer constructor arguments
 * (csrc/cuaev.cpp:248: Rcr, Rca, EtaR, ShfR, EtaA, Zeta, ShfA, ShfZ, num_species, use_cos_cutoff). */
typedef struct {
    int32_t num_species; /* S <= 7 */
    int32_t n_shf_r;     /* <= 32.  16 with an 8 x 4 (ANI-2x) or 4 x 8 (ANI-1x) angular grid: the tuned kernels, with slab */
    int32_t n_shf_a;     /* <= 16   masks and the forward-mode derivative; any other grid: the general kernels           */
    int32_t n_shf_z;     /* <= 16   (csrc/aev_generic.hip; cuAEV is templated on these lengths, csrc/aev.cu:1687-1777)    */
    float Rcr, Rca;
    float EtaR, EtaA, Zeta;
    int32_t cutoff_kind; /* ANIHIP_CUTOFF_COSINE | ANIHIP_CUTOFF_SMOOTH (use_cos_cutoff = false) */
    int32_t flags;       /* ANIHIP_AEV_*: written by anihip_aev_table_pack from the shift arrays, zero before that */
} anihip_aev_params;

/* flags: the angular radial shifts ShfA are equally spaced and narrow enough for fp32 exponents -- the forward kernel
 * then evaluates its NA Gaussians per neighbor pair by a three-exponential recurrence instead of NA exponentials. */
#define ANIHIP_AEV_UNIFORM_SHFA 1
/* ShfR AND ShfA are equally spaced and every exponent of the chained Gaussians stays inside fp32's range for distances up to
 * the cutoffs: the backward kernel evaluates its 16 radial Gaussians per neighbor from four of them and its NA angular ones per
 * neighbor pair from one (products with exp2(+-2 D x) instead of exponentials; constants in free slots of the table). */
#define ANIHIP_AEV_REC_BWD 2

/* Length in floats of the device constant table consumed by the AEV kernels, and a host-side packer:
 * table = ShfR[32] | ShfA[16] | cos(ShfZ)[16] | sin(ShfZ)[16]  (trig evaluated in double on the
 * fp32-rounded ShfZ) | q_R ShfR[16] | q_A ShfA[16] | cos/2 [16] | sin/2 [16] with
 * q = sqrt(Eta log2 e)  (exp(-Eta x^2) = exp2(-(q x)^2): the kernels keep distances pre-scaled).  The caller uploads
 * it to the device.  For the 16 / 8 x 4 / 4 x 8 grids the upper halves of the q_A ShfA and cos/2 blocks (slots 104-111,
 * 120-127) carry the constants of the backward kernel's Gaussian recurrences (ANIHIP_AEV_REC_BWD). */
#define ANIHIP_AEV_TABLE_FLOATS 144
int anihip_aev_table_pack(anihip_aev_params *p /* flags are set */, const float *ShfR, const float *ShfA,
                          const float *ShfZ, float *table_out /* host, ANIHIP_AEV_TABLE_FLOATS */);

const char *anihip_last_error(void);
int anihip_abi_version(void);

/* ---------------------------------------------------------------------------------------------
 * Neighbor lists.  Output format (shared by all builders and consumed by the AEV kernels):
 *   meta[i*6 + 0]   = first entry of atom i's row
 *   meta[i*6 + 1]   = nA | (nF << 16): #neighbors with r <= Rca, # with Rca < r <= Rcr
 *   meta[i*6 + 2,3] = 8 x uint8: per-species counts of the nA angular-range neighbors
 *   meta[i*6 + 4,5] = 8 x uint8: per-species counts of the nF far neighbors
 *   ent[row..]      = float4 {dx, dy, dz, bits(j | species_j << 28)}, d = r_j (+ image shift) - r_i,
 *                     angular-range neighbors first, each group ordered by species then by discovery.
 * It is a FULL list (both directions), like cuAEV's internal lists (csrc/aev.cu:975-1039).
 * Only central atoms lo <= i < hi get rows (the data-parallel shard of this rank); all atoms are
 * neighbor candidates.
 */

/* Workspace bytes for anihip_nbr_build_batch / anihip_nbr_build_cell (two-call pattern like
 * csrc/cuaev_cub.cuh:10-17). */
size_t anihip_nbr_workspace_bytes(int64_t n_atoms, int64_t max_cells);

/* Batched molecules, all pairs inside each molecule: replaces pairwiseDistance + postProcessNbrList1
 * (csrc/aev.cu:180-249,975-1039) and neighbors.py:187-275 (all_pairs incl. PBC images).
 * cell: device float[9] (rows = lattice vectors) or NULL; pbc_mask bit k = periodic along vector k.
 * anihip_nbr_build_batch and anihip_nbr_build_cell reset the ANIHIP_STATUS_WORDS status words themselves (their first
 * kernel does); the conversions below (from_half / from_full / refresh) and the AEV calls OR into what they are given. */
int anihip_nbr_build_batch(void *stream, const anihip_aev_params *p, int32_t n_mol, int32_t n_atoms_per_mol,
                           const int32_t *species, const float *coords, const float *cell,
                           int32_t pbc_mask, int64_t lo, int64_t hi, void *workspace,
                           size_t workspace_bytes, uint32_t *meta, float *ent, int64_t ent_capacity,
                           uint32_t *status);

/* One large system through a cell grid (O(N)): replaces cell_list::cell_list
 * (csrc/cell_list.cpp:342-354, neighbors.py:366-507) + postProcessExternalHalfNbrList
 * (csrc/aev.cu:1128-1208).  Without pbc the grid spans the bounding box (neighbors.py:389-394). */
int anihip_nbr_build_cell(void *stream, const anihip_aev_params *p, int64_t n_atoms,
                          const int32_t *species, const float *coords, const float *cell,
                          int32_t pbc_mask, int64_t lo, int64_t hi, int64_t max_cells, void *workspace,
                          size_t workspace_bytes, uint32_t *meta, float *ent, int64_t ent_capacity,
                          uint32_t *status);

/* Rows from an externally supplied HALF neighbor list (LAMMPS / Amber style drivers, the reference's
 * AEVComputer.compute_from_neighbors, aev/_computer.py:251-272, and cuaev::run_with_half_nbrlist ->
 * postProcessExternalHalfNbrList, csrc/cuaev.cpp:205-224, csrc/aev.cu:1128-1208).  Pair p joins the
 * flattened atom indices idx[p] and idx[n_pairs + p] (int64 like Neighbors.indices, neighbors.py:22-29) with
 * diff[p] = r_idx0 - r_idx1 (+ image shift) as produced by narrow_down (neighbors.py:105-112).  Pairs longer
 * than Rcr or touching a padding atom are dropped; every other pair is entered in both rows.  Rows come out
 * in the same format (and a canonical order) as those of the builders above. */
size_t anihip_nbr_half_workspace_bytes(int64_t n_central);
int anihip_nbr_from_half(void *stream, const anihip_aev_params *p, int64_t n_atoms, const int32_t *species,
                         int64_t n_pairs, const int64_t *idx, const float *diff, int64_t lo, int64_t hi,
                         void *workspace, size_t workspace_bytes, uint32_t *meta, float *ent,
                         int64_t ent_capacity, uint32_t *status);

/* The reverse direction: the reference's half-list format from neighbor rows -- what cell_list::cell_list returns
 * (csrc/cell_list.cpp:342-354: idx [2, P] i64, dist [P], diff [P, 3], every pair once; call site neighbors.py:285-294), so
 * that anihip_nbr_build_cell + this call replace torch.ops.cell_list.cell_list for FastCellList / model.neighborlist
 * consumers.  Rows lo..hi emit their pairs with j > i (and one of each +-image pair of an atom with itself), in row order:
 * deterministic.  diff = r_i - r_j (+ image shift), the reference's sign (neighbors.py:105-112).  lo = 0, hi = n_atoms
 * gives the complete list.  idx is [2][capacity]; *n_pairs (device) receives the number of pairs whatever the capacity
 * (call with capacity 0 to size the outputs).  Workspace: anihip_nbr_rows_to_half_workspace_bytes(hi - lo). */
size_t anihip_nbr_rows_to_half_workspace_bytes(int64_t n_central);
int anihip_nbr_rows_to_half(void *stream, int64_t n_atoms, int64_t lo, int64_t hi, const uint32_t *meta, const float *ent,
                            void *workspace, size_t workspace_bytes, int64_t capacity, int64_t *idx, float *dist,
                            float *diff, int64_t *n_pairs /* device */);

/* Rows from an externally supplied FULL neighbor list in the LAMMPS convention (local + ghost atoms, every ghost
 * with its own coordinates; cuaev::run_with_full_nbrlist -> postProcessNbrList2, csrc/cuaev.cpp:226-244,
 * csrc/aev.cu:1048-1126, call site aev/_computer.py:420-438): listed atom ilist[g] has the numneigh[g] neighbors
 * jlist[start[g] .. start[g] + numneigh[g]) (start = exclusive prefix sum of numneigh, int64).  Displacements are
 * r_j - r_i from coords [n_atoms, 3]; neighbors beyond Rcr, padding atoms and j == i are dropped.  Atoms that are
 * not listed get empty rows (zero AEVs).  ent_capacity = n_atoms * row capacity. */
int anihip_nbr_from_full(void *stream, const anihip_aev_params *p, int64_t n_atoms, const int32_t *species,
                         const float *coords, int64_t n_listed, const int32_t *ilist, const int32_t *numneigh,
                         const int64_t *start, const int32_t *jlist, uint32_t *meta, float *ent,
                         int64_t ent_capacity, uint32_t *status);

/* Verlet-skin reuse (VerletCellList, neighbors.py:759-884): verlet_meta / verlet_ent are rows produced by one of the
 * builders above from coords_build with an ENLARGED radial cutoff (Rcr + skin; the caller passes a copy of the
 * parameters with Rcr raised -- and Rca lowered to ~0 so that all of a row counts against the 256-entry radial
 * limit).  While no atom has moved more than skin / 2 since then, the rows for the current coordinates follow
 * without a pair search: every stored displacement is updated with the motion of its two atoms, screened against
 * the real Rcr of p and re-sorted (the narrow_down step, neighbors.py:64-113).  coords and coords_build must be the
 * same unwrapped trajectory (an atom re-wrapped by a lattice vector in between looks like a large move: rebuild).
 * Rows come out for lo <= i < hi in the standard format; ent_capacity = (hi - lo) * row capacity. */
int anihip_nbr_refresh(void *stream, const anihip_aev_params *p, int64_t n_atoms, int64_t lo, int64_t hi,
                       const int32_t *species, const float *coords, const float *coords_build,
                       const uint32_t *verlet_meta, const float *verlet_ent, uint32_t *meta, float *ent,
                       int64_t ent_capacity, uint32_t *status);

/* ---------------------------------------------------------------------------------------------
 * AEV forward / backward: replace cuRadialAEVs + cuAngularAEVs (csrc/aev.cu:768-834,323-472) and their
 * backward kernels (csrc/aev.cu:837-967,474-766).  aev / grad_aev are [n_atoms, L] row-major with
 * L = S*16 + S(S+1)/2*32, layout [radial | angular] (aev/_computer.py:298).  Rows of atoms outside
 * [lo,hi) are not touched; padding atoms inside the range get zero rows.
 * grad_coords [n_atoms,3] is ACCUMULATED into (caller zeroes it), i.e.
 * grad_coords += d(sum grad_aev * aev)/d coords  (csrc/aev.cu:1958-1984).  The radial part of a pair is finished by
 * each of its two atoms for itself -- atom i gathers the 16-float block grad_aev[j][species(i)*16 ..] of every
 * neighbor row j that this call may read, i.e. lo <= j < hi (and, with slab_mask, flagged there) -- so only the
 * angular part and pairs whose partner row belongs to another shard travel through float atomics.  That needs
 * SYMMETRIC rows (j in row i <=> i in row j), which every builder of this library produces except
 * anihip_nbr_from_full (a LAMMPS list names ghost atoms that have no row of their own): leave ANIHIP_BWD_SYMMETRIC out
 * of `flags` for those rows and every pair term is pushed to its neighbor instead.
 * ANIHIP_BWD_FIXED_POINT: grad_coords is then int64_t[n_atoms][3] (caller zeroes it) in units of 2^-32; contributions are
 * added with 64-bit integer atomics, so the result is bit-identical from run to run whatever the order of the waves
 * (the reference's cuAEV backward is not: float atomics, csrc/aev.cu:700-704).  The caller converts: g = acc * 2^-32.
 *
 * slab_mask (optional, may be NULL; needs ceil(S/2) + S(S+1)/2 <= 32): slab_mask[i] flags the 32-wide
 * "slabs" of row i that can be non-zero -- bit j < ceil(S/2): radial blocks of species 2j, 2j+1; bit
 * ceil(S/2) + P: angular block of species pair P.  An AEV block is identically zero when atom i has no
 * neighbor (pair) of that species inside the cutoff; anihip_mlp_forward_backward skips those slabs.
 * anihip_aev_backward(slab_mask != NULL) reads grad_aev only inside flagged slabs (of the rows lo..hi); with
 * slab_mask == NULL every entry of the rows lo..hi must be valid.
 * Any OTHER grid (n_shf_r <= 32, n_shf_a, n_shf_z <= 16: the general kernels; rows of at most 1024 columns): the flags
 * are those of the PLAIN 32-column slabs of the row -- bit j <=> columns 32 j .. 32 j + 31 can be non-zero (a block of a
 * present species or species pair flags every slab it overlaps) -- which is the order anihip_mlp_pack gives the layer-0
 * planes of such networks (aev_radial_len = 0).  The general backward reads only blocks of present species and takes
 * no flags.  anihip_aev_jvp serves every grid as well (ABI 9). */
#define ANIHIP_BWD_SYMMETRIC 1   /* rows are symmetric: gather the radial partner blocks instead of pushing */
#define ANIHIP_BWD_FIXED_POINT 2 /* grad_coords is an int64 fixed-point accumulator (2^-32): reproducible sums */
int anihip_aev_forward(void *stream, const anihip_aev_params *p, const float *table, int64_t n_atoms,
                       int64_t lo, int64_t hi, const int32_t *species, const uint32_t *meta,
                       const float *ent, float *aev, uint32_t *slab_mask, uint32_t *status);
/* The same rows UPDATED IN PLACE (ABI 9; the 16 / 8x4 / 4x8 grids): aev is the buffer of the previous call (any system
 * of the same n_atoms -- which atom a row described does not matter), prev_mask[i] = the slabs of row i that may hold
 * non-zero data (the slab_mask that call wrote; the caller alternates between two flag buffers: prev_mask is only read,
 * slab_mask only written, they must not be the same buffer).  Only those slabs and the ones flagged now are written: an atom of a water box flags
 * 4-5 of its 32 slabs step after step, so an MD loop writes 0.6 KB of each 4 KB row instead of all of it (the rest is
 * zeros nobody has to write again).  The rows are complete afterwards, exactly as anihip_aev_forward leaves them.
 * First use: aev zero-filled, prev_mask zero-filled (or: aev anything, prev_mask all ones).  No reference counterpart
 * (csrc/aev.cu:1732 allocates and zero-fills a fresh tensor per call). */
int anihip_aev_forward_update(void *stream, const anihip_aev_params *p, const float *table, int64_t n_atoms,
                              int64_t lo, int64_t hi, const int32_t *species, const uint32_t *meta,
                              const float *ent, float *aev, const uint32_t *prev_mask, uint32_t *slab_mask,
                              uint32_t *status);
int anihip_aev_backward(void *stream, const anihip_aev_params *p, const float *table, int64_t n_atoms,
                        int64_t lo, int64_t hi, const int32_t *species, const uint32_t *meta,
                        const float *ent, const float *grad_aev, const uint32_t *slab_mask, int32_t flags,
                        float *grad_coords, uint32_t *status);

/* Forward-mode derivative of the AEV rows along a coordinate-space direction: daev[i] = sum_k (d aev[i] / d r_k) .
 * tangent[k]  (tangent: [n_atoms][3]).  This is the reference's cuaev double backward (csrc/aev.cu:1986-2015,
 * cuaev_double_backward, with the is_double_backward kernel variants :474-766,837-967): training on forces
 * differentiates grad_coords = J^T grad_aev with respect to grad_aev, and the gradient arriving at the forces is the
 * direction.  Rows lo <= i < hi of daev are written (rows of padding atoms zeroed). */
int anihip_aev_jvp(void *stream, const anihip_aev_params *p, const float *table, int64_t n_atoms, int64_t lo,
                   int64_t hi, const int32_t *species, const uint32_t *meta, const float *ent,
                   const float *tangent, float *daev, uint32_t *status);

/* anihip_aev_backward plus the virial of the back-propagated scalar,
 *   virial[3a + b] = sum over central atoms lo <= i < hi and their neighbors j of (d E_i / d d_ij)[a] * d_ij[b]
 * (fp64 [9], OVERWRITTEN; d_ij = the displacement stored in the row): the reference's "fdotr" virial
 * (ase.py:164-168, dE/d(diff_vectors)^T @ diff_vectors), which under periodic boundary conditions equals the
 * derivative of the energy with respect to a strain of coordinates and cell (ase.py:170-173); stress = virial /
 * volume.  It needs neither the cell nor whole molecules, so shards / domains simply add their partial virials. */
int anihip_aev_backward_virial(void *stream, const anihip_aev_params *p, const float *table, int64_t n_atoms,
                               int64_t lo, int64_t hi, const int32_t *species, const uint32_t *meta,
                               const float *ent, const float *grad_aev, const uint32_t *slab_mask,
                               int32_t flags, float *grad_coords, double *virial, uint32_t *status);

/* ---------------------------------------------------------------------------------------------
 * Per-species MLP ensemble: replaces mnp::run (csrc/mnp.cpp:238-265; forward :32-136, input-gradient
 * backward :138-232) and BmmEnsemble (nn/_infer.py:61-216).
 *
 * Packed parameter layout (host code packs once per model, cf. BmmAtomicNetwork nn/_infer.py:141-161).
 * M = ensemble members, widths padded up to multiples of 32 with zeros (Hp):
 *   layer 0 : w [K0][M*H1p]  (column m*H1p+o = member m, unit o)   wt [M*H1p][K0p]  bias [M*H1p]
 *   layer l : w [M][Hlp][H(l+1)p]   wt [M][H(l+1)p][Hlp]   bias [M][H(l+1)p]      (hidden layers)
 *   final   : w [M][Hlastp]  bias [M]
 * K0 = AEV length (multiple of 16), K0p = K0 rounded up to a multiple of 32.
 *
 * Slab order of the layer-0 fp16 planes (desc.aev_radial_len = R > 0, needs (K0 - R) % 32 == 0): the AEV
 * index of wh[0] / wth[0] runs over [radial part zero-padded to a multiple of 32 | angular part], so that
 * every 32-deep slab holds whole species blocks of the AEV (two radial blocks or one angular block) and
 * K0p = 32 * (ceil(R/32) + (K0 - R)/32).  With R = 0 the planes are in plain AEV order.
 */
#define ANIHIP_MAX_LAYERS 4 /* Linear layers per network incl. the final one */
typedef struct {
    int32_t n_layers;                      /* Linear layers incl. final (ANI: 4) */
    int32_t dims[ANIHIP_MAX_LAYERS + 1];   /* padded widths: K0, H1p, H2p, H3p, 1 */
    const float *w[ANIHIP_MAX_LAYERS];     /* device pointers, layouts above */
    const float *wt[ANIHIP_MAX_LAYERS];    /* transposed copies (unused for the final layer) */
    const float *bias[ANIHIP_MAX_LAYERS];
    /* precision == ANIHIP_MLP_F16X3 only: the same hidden-layer matrices as two fp16 planes {hi, lo}
     * with hi + lo = w * wh_scale (power of two), stored [2][rows = output index][cols = reduction index]:
     * wh[l] has the shape of wt[l] (forward GEMMs), wth[l] the shape of w[l] (backward GEMMs; layer 0
     * padded to K0p rows).  Layer 0 uses the slab order described above when aev_radial_len > 0. */
    const void *wh[ANIHIP_MAX_LAYERS];
    const void *wth[ANIHIP_MAX_LAYERS];
    float wh_scale[ANIHIP_MAX_LAYERS];
    /* optional: the same planes re-ordered into MFMA fragment order
     * [member][N/32][K/16][2 planes][64 lanes][8 halves] (lane l <-> output n = 32 cb + (l & 31),
     * k = 16 ks + 8 (l >> 5) + j) for the fused network kernel, which streams them straight into
     * registers: whf[l] for layers 0..n_layers-2 (layer 0 per member: N = H1p, K = K0p in the order of
     * wh[0]), wthf[l] for layers 1..n_layers-2, and (4-layer networks) wthf[0] = layer 0 transposed, per member: N = K0p
     * in the order of wh[0] (a column block = one AEV slab), K = H1p -- the layer-0 backward inside the fused kernel.
     * whf / wthf[1..] NULL disables the fused kernel, wthf[0] NULL its layer-0 backward phase. */
    const void *whf[ANIHIP_MAX_LAYERS];
    const void *wthf[ANIHIP_MAX_LAYERS];
    /* optional, fused kernel of 4-layer networks: per member 8 floats ([5..7] = 0) that bound the operands
     * of the inner GEMMs so their fp16 split scales need no reduction over the tile:
     *   [0] max_j sum_k |W1[j][k]|   [1] max_j |b1[j]|   [2] max_j |w3[j]| / M
     *   [3] [2] * max_k sum_j |W2[j][k]|   [4] [3] * max_k sum_j |W1[j][k]|
     * (|act1| <= max|act0| * [0] + [1], |d act2| <= [2], |d act1| <= [3], |d act0| <= [4]).
     * NULL disables the fused kernel. */
    const float *fused_bounds;
} anihip_species_net;

/* GEMM arithmetic of the hidden layers.
 *   FP32  : v_mfma_f32_32x32x2_f32, exact fp32 products.
 *   F16X3 : every fp32 operand x is split on the fly into two fp16 numbers hi + lo = x * 2^e (e chosen per
 *           tensor from a device-side running max so nothing overflows); a product is evaluated as
 *           hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation: relative error
 *           ~4e-7 per product (fp32: 6e-8) at 16/3 of the fp32-MFMA rate.  Layer-0 inputs use the static
 *           scale 4, valid for |x| < 16376 -- above the largest value an AEV element can take with the
 *           row limits of this library (2 * C(128,2) = 16256). */
#define ANIHIP_ACT_CELU 0
#define ANIHIP_ACT_GELU 1
#define ANIHIP_MLP_FP32 0
#define ANIHIP_MLP_F16X3 1
/* anihip_mlp_desc.flags: algorithm choices of anihip_mlp_forward_backward that the library otherwise makes from the
 * problem size (tests and benchmarks pin them; results agree to rounding whatever is chosen) */
#define ANIHIP_MLP_FLAG_NO_FUSED 1u       /* layer-by-layer GEMMs instead of the fused network kernel */
#define ANIHIP_MLP_FLAG_BIG_TILES 2u      /* 256 x 256 layer-0 tiles (default from 16384 atoms) even for few atoms */
#define ANIHIP_MLP_FLAG_SMALL_TILES 4u    /* 128 x 128 layer-0 tiles whatever the size */
#define ANIHIP_MLP_FLAG_NO_SLAB_MASK 8u   /* ignore slab_mask: multiply every AEV slab */
/* (16, 64, 128, 256: development switches of rounds 1-4 -- 32-atom tiles, separate preparation launches, the 4-wave
 * layer-0 backward, tile-owner order without phase 5 -- retired in ABI 11; setting them changes nothing) */
#define ANIHIP_MLP_FLAG_D0_ROWS 32u       /* d E/d act0 handed to the layer-0 backward row-major, not tile-major */
#define ANIHIP_MLP_FLAG_FUSED_L0B 512u    /* layer-0 backward inside the fused kernel whatever the size (default: from 24000 atoms) */
#define ANIHIP_MLP_FLAG_NO_FUSED_L0B 1024u /* ... never: d E/d act0 through HBM + a layer-0 backward GEMM launch */
#define ANIHIP_MLP_FLAG_SHAPED 4096u       /* with the layer-0 backward inside the fused kernel: ONE LAUNCH PER SPECIES, restricted to its tiles, with the
                                             * network widths as compile-time constants where an instantiation exists (every ANI-2x network) -- 5-6 % faster
                                             * per tile.  Below four rounds of tiles these are plain launches on the caller's stream, each ending with a
                                             * partly filled last round of the CUs (a species of a handful of atoms still costs a tile through all members);
                                             * from four rounds on the launches draw their tiles from a queue per species and alternate between the
                                             * caller's stream and a SECOND STREAM OF THE LIBRARY'S OWN (one per device, created on first use; forked from and
                                             * joined to the caller's stream with events, so the call stays stream-ordered for the caller; not inside a
                                             * stream capture).  Whether it pays depends on the composition: the Python host prices both schemes
                                             * (models.ANI._per_species_launches_pay) -- 2.3 M-atom water box: yes; 46 k-atom solvated protein: no */
#define ANIHIP_MLP_FLAG_BWD_TWO_PRODUCTS 2048u /* OFF by default.  With the layer-0 backward inside the fused kernel (>= 24000 atoms,
                                                 * CELU): its backward GEMMs leave out (weight lo) x (gradient hi), i.e. use the weights
                                                 * rounded to fp16 -- energies unchanged, d E/d AEV and the forces differ from the default's
                                                 * by ~1e-6 Ha/A (inside the 1e-4 Ha/A gate of the parity tests, OUTSIDE their 5e-6 regression
                                                 * gate), a sixth fewer MFMAs */
typedef struct {
    int32_t num_species;
    int32_t n_members;
    int32_t aev_len;
    float celu_alpha;
    int32_t precision; /* ANIHIP_MLP_FP32 or ANIHIP_MLP_F16X3 */
    int32_t aev_radial_len; /* R of the slab order of wh[0] / wth[0] (0 = plain order) */
    int32_t flags;          /* ANIHIP_MLP_FLAG_* (0 = let the library choose); the library reads no environment */
    int32_t activation;     /* ANIHIP_ACT_CELU (celu_alpha) or ANIHIP_ACT_GELU (torch.nn.GELU(), exact; energies and
                             * input gradients through the fused network kernel only: 3 hidden layers <= 256 wide) */
    anihip_species_net net[ANIHIP_MAX_SPECIES];
} anihip_mlp_desc;

/* Packing a model behind the ABI (replaces what BmmEnsemble / MNPNetworks do at construction, nn/_infer.py:141-161,
 * 263-372: stacking the members' Linear parameters for the batched kernels; mnp::run takes them as Tensor lists,
 * csrc/mnp.cpp:238-248).  The caller describes the networks, asks for the buffer size, and hands over the parameters of
 * every (member m, species s, layer l) as torch.nn.Linear lays them out: weights[(m * S + s) * n_layers + l] -> float
 * [out][in] (in = aev_len for l = 0, else out_dims[s][l - 1]), biases[...] -> float [out]; device or host pointers.
 * anihip_mlp_pack fills out_buffer (device memory, or host memory with dst_on_device = 0 -- then nothing touches a GPU)
 * with every layout above -- fp32 arrays, transposed copies, {hi, lo} fp16 planes in slab order, MFMA fragment order,
 * fused_bounds -- and writes the descriptor whose pointers refer to out_buffer.  The buffer must stay alive and
 * unchanged while the descriptor is in use; the call synchronises the stream once (model load, not the hot path). */
typedef struct {
    int32_t n_members, num_species, n_layers;   /* n_layers = Linear layers per network incl. the final one (2..4) */
    int32_t aev_len;
    int32_t aev_radial_len; /* R of the slab order; -1: 16 S when aev_len has the ANI form 16 S + 32 S (S + 1) / 2, else 0 */
    int32_t precision;      /* ANIHIP_MLP_FP32 | ANIHIP_MLP_F16X3 */
    int32_t activation;     /* ANIHIP_ACT_CELU | ANIHIP_ACT_GELU */
    float celu_alpha;
    int32_t out_dims[ANIHIP_MAX_SPECIES][ANIHIP_MAX_LAYERS]; /* unpadded output width of every layer; the last is 1 */
} anihip_mlp_shape;
size_t anihip_mlp_pack_bytes(const anihip_mlp_shape *shape);   /* 0 + anihip_last_error() for a shape the kernels refuse */
int anihip_mlp_pack(void *stream, const anihip_mlp_shape *shape, const float *const *weights, const float *const *biases,
                    int32_t src_on_device, void *out_buffer, size_t out_bytes, int32_t dst_on_device,
                    anihip_mlp_desc *out_desc);

/* Workspace for n central atoms (activations of every hidden layer for all members, species-sorted
 * index lists): enough for every entry point below that takes a workspace. */
size_t anihip_mlp_workspace_bytes(const anihip_mlp_desc *d, int64_t n_central);
/* What ONE anihip_mlp_forward_backward call over n_central atoms with this descriptor (its flags included) touches
 * (ABI 10): the fused network kernel keeps the activations in LDS, so its calls need the index lists, the tile table and
 * the per-member energies (~100 B per atom) plus -- below 65536 atoms, where the layer-0 backward is a GEMM of its own --
 * d E / d act0 (8 KB per atom for ANI-2x); want_grad = (grad_aev != NULL).  Never more than anihip_mlp_workspace_bytes. */
size_t anihip_mlp_forward_backward_workspace_bytes(const anihip_mlp_desc *d, int64_t n_central, int32_t want_grad);

/* atomic_e[i] = mean over members of net_{m,species(i)}(aev[i]) for lo <= i < hi (0 for padding);
 * if grad_aev != NULL also grad_aev[i] = d atomic_e[i] / d aev[i] (rows of padding atoms zeroed).
 * member_e (optional) = [M, n_atoms] per-member energies (ensemble_values, nn/_containers.py:638-651).
 * slab_mask (optional; F16X3 with aev_radial_len > 0 only, otherwise ignored): the per-atom slab flags
 * written by anihip_aev_forward.  Slabs that no atom of a row tile flags are skipped in the layer-0
 * GEMMs (their AEV entries are exactly zero, so energies are unchanged); the matching entries of
 * grad_aev are then NOT written -- the caller must only consume grad_aev inside the flagged slabs
 * (anihip_aev_backward does) or pre-zero it. */
int anihip_mlp_forward_backward(void *stream, const anihip_mlp_desc *d, int64_t n_atoms, int64_t lo,
                                int64_t hi, const int32_t *species, const float *aev,
                                const uint32_t *slab_mask, void *workspace, size_t workspace_bytes,
                                float *atomic_e, float *grad_aev, float *member_e);

/* ---------------------------------------------------------------------------------------------
 * Training pass: gradients with respect to every weight and bias.  The reference gets them from torch autograd
 * through nn/_core.py:146-149 + nn/_containers.py:377-421,608-636 (its native MNP path has no weight gradients,
 * csrc/mnp.cpp:138-232); the loop this serves is tools/training-aev-benchmark.py:120-135 (BASELINE config 5).
 *
 * For  Loss = sum_i grad_atomic_e[i] * atomic_e[i]  (the caller folds d Loss / d E_molecule into per-atom factors;
 * atomic_e = ensemble mean as above):
 *   grads[s].gw[l]    = d Loss / d weight of layer l of species s as [M][out_p][in_p] (in_p = K0 for layer 0, [M][Hp]
 *                       for the output layer): per member torch.nn.Linear's [out][in] layout at the padded widths, so
 *                       unpadded networks (ANI-2x) read their gradients in place
 *   grads[s].gbias[l] = d Loss / d bias[l] of species s, same shape as bias[l]
 * (padded rows/columns come out zero).  The gradient arrays are OVERWRITTEN, or -- accumulate != 0 -- ADDED to (the caller
 * zeroes them: gradient accumulation over several calls, optimizers that own one flat gradient buffer).  atomic_e is written
 * as in anihip_mlp_forward_backward; grad_aev (optional) = d Loss / d aev rows, i.e. already scaled by grad_atomic_e.
 * Arithmetic, by descriptor (sums over atoms use float atomics either way: the last bits depend on the execution order):
 *   ANIHIP_MLP_FP32 (any shape, CELU or GELU): exact fp32 on v_mfma_f32_32x32x2_f32, layer by layer; only w / wt / bias are read.
 *   ANIHIP_MLP_F16X3, CELU, three hidden layers <= 256 wide (ANI-1x / ANI-2x) -- the FAST training path (round 5): the forward
 *     and the backward down to d e / d z0 run as ONE launch of the fused network kernel (split-fp16 MFMA, the arithmetic of
 *     inference: per-atom energies within 1e-7 Ha of fp64) for a unit upstream gradient, leaving activations and d e / d z in
 *     the workspace; the weight gradients dW_l = sum_atoms g_a (d e / d z_l)^T x_{l-1} are one launch per layer on
 *     v_mfma_f32_32x32x16_f16 with both operands split into two fp16 planes on the fly (three products: 2^-22 relative; the
 *     power-of-two operand scales come from fused_bounds, max |g_a| and the largest |act0| the forward recorded -- round 6), or,
 *     for a pack without fused_bounds, on v_mfma_f32_32x32x16_bf16 with three-way bf16 splits (six products, no scales); the
 *     bias gradients one column reduction per layer.  A call that asks for grad_aev takes the exact-fp32 backward instead (on
 *     the activations of whichever forward ran).
 * member_stride (fast path only; 0 = the packed [M][...] arrays above): gw[l] / gbias[l] point at MEMBER 0's arrays and
 * member m's lie member_stride floats further on each -- the layout of a flat parameter buffer in torch's parameter order
 * (member -> species -> layer -> weight, bias), so an optimizer that owns ONE flat gradient buffer receives the gradients
 * where its update kernel reads them (needs accumulate != 0 and unpadded widths).
 */
typedef struct {
    float *gw[ANIHIP_MAX_LAYERS];
    float *gbias[ANIHIP_MAX_LAYERS];
    int64_t member_stride;
    int32_t accumulate;
} anihip_species_grads;

size_t anihip_mlp_train_workspace_bytes(const anihip_mlp_desc *d, int64_t n_central);
int anihip_mlp_weight_grads(void *stream, const anihip_mlp_desc *d, int64_t n_atoms, int64_t lo, int64_t hi,
                            const int32_t *species, const float *aev, const float *grad_atomic_e,
                            void *workspace, size_t workspace_bytes, const anihip_species_grads *grads /* [num_species] */,
                            float *atomic_e, float *grad_aev, int32_t forward_done);

/* The two halves of a training step as autograd needs them (forward now, backward later): the forward that leaves the
 * species buckets and every hidden activation (fast path: also d e / d z of every layer) in the workspace
 * (anihip_mlp_train_workspace_bytes), writing atomic_e -- exact fp32 layer by layer, or the fused split-fp16 kernel, by
 * descriptor as above.  A later anihip_mlp_weight_grads call with forward_done != 0 and the SAME descriptor, range,
 * species, aev and (untouched) workspace skips its own forward. */
int anihip_mlp_train_forward(void *stream, const anihip_mlp_desc *d, int64_t n_atoms, int64_t lo, int64_t hi,
                             const int32_t *species, const float *aev, void *workspace, size_t workspace_bytes,
                             float *atomic_e);

/* Second-order pass for training on forces.  For per-atom tangents (tangent: [n_atoms][aev_len]) let
 *     S = sum_i tangent[i] . d atomic_e[i] / d aev[i]
 * (with tangent = -J t, J t from anihip_aev_jvp, S = sum_k t_k . F_k for the forces F = -dE/dr: the part of a force
 * loss that the reference back-propagates through torch.autograd.grad(E, coords, create_graph=True),
 * tools/training-aev-benchmark.py:136-150).  grads = d S / d (weights, biases) in the layouts of
 * anihip_mlp_weight_grads (OVERWRITTEN); datomic_e[i] = tangent[i] . d atomic_e[i] / d aev[i] (the directional
 * derivative of the energies; entries of padding atoms zeroed).  Exact fp32: forward-over-reverse with the
 * activations a_l, their tangents and both adjoint streams kept in the workspace. */
size_t anihip_mlp_tangent_workspace_bytes(const anihip_mlp_desc *d, int64_t n_central);
int anihip_mlp_tangent_weight_grads(void *stream, const anihip_mlp_desc *d, int64_t n_atoms, int64_t lo, int64_t hi,
                                    const int32_t *species, const float *aev, const float *tangent,
                                    void *workspace, size_t workspace_bytes, const anihip_species_grads *grads,
                                    float *datomic_e);

/* Refresh the packed parameter arrays of a descriptor IN PLACE (padding stays zero) from the torch.nn.Linear tensors after
 * an optimizer step -- one or two launches instead of re-packing on the host (cf. BmmAtomicNetwork packing once per model,
 * nn/_infer.py:141-161).  ANIHIP_MLP_FP32: w / wt / bias.  ANIHIP_MLP_F16X3 (round 5): every layout anihip_mlp_pack derives
 * -- w / wt / bias, the {hi, lo} fp16 planes wh / wth, their fragment orders whf / wthf, fused_bounds -- with the
 * power-of-two weight scales the pack was built with (wh_scale is not changed): should a weight have outgrown the fp16
 * range of its layer's scale, bit 0 of *status (device int32, optional, never cleared here) is set and the caller packs
 * again on the host.
 * src: DEVICE array of 2 * M * S * n_layers pointers ordered member, species, layer, {weight [out][in] row-major,
 * bias [out]}; out_in: HOST array [S][n_layers][2] with the (out, in) widths of the source tensors. */
#define ANIHIP_REPACK_FUSED_ONLY 1   /* F16X3: only what the fused network kernel and the fast training pass read -- bias, output
                                        layer, whf, wthf of the hidden layers, fused_bounds; w / wt / wh / wth / wthf[0] go stale */
int anihip_mlp_repack(void *stream, const anihip_mlp_desc *d, const void *const *src, const int32_t *out_in,
                      int32_t *status, int32_t flags);

/* One Adam update over flat buffers (torch.optim.Adam with amsgrad = False, maximize = False: the optimizer of the
 * reference's training recipe, tools/training-aev-benchmark.py:88; weight_decay is torch's L2 form, grad += wd * param):
 *   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  param -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
 * with t = *step + 1; *step (DEVICE int32: updates done so far) is incremented behind the update, so a captured HIP graph
 * of a training step advances it on every replay.  One launch over n parameters (28 bytes of traffic each) instead of a
 * dozen foreach launches over the model's 448 tensors; buffers 16-byte aligned.  zero_grads != 0: the gradients are zeroed
 * behind the update (the weight-gradient kernels of the next step ADD into them: no memset launch per step).
 * The hyper-parameters are doubles (ABI 12), as torch.optim.Adam holds them: 1 - beta2 taken from the fp32-rounded 0.999 is
 * 1.3e-5 off, and exp_avg_sq then differs from torch's by that factor. */
int anihip_adam_step(void *stream, float *params, float *grads, float *exp_avg, float *exp_avg_sq, int64_t n,
                     double lr, double beta1, double beta2, double eps, double weight_decay, int32_t *step, int32_t zero_grads);

/* ---------------------------------------------------------------------------------------------
 * Pair potentials on the neighbor rows: the xTB repulsion term of the reference's ANI-2xr / ANI-2dr models
 * (potentials/xtb.py:17-77 RepulsionXTB, envelope and per-atom halves of potentials/core.py:155-207):
 *   e_ij = y_ab / d * exp(-sqrt(alpha_ab) d^k_ab) * fc(r_ij, cutoff),  d = r_ij in Bohr (r clamped to >= 1e-7 A);
 * pair_table: device float[8][8][4] = {y_ab, sqrt(alpha_ab), k_ab, 0} indexed [species_i][species_j].
 * atomic_e[i] += sum_j e_ij / 2 for lo <= i < hi (may be NULL); grad_coords [n_atoms][3] += d(sum of those) / d r (may be
 * NULL); virial [9] fp64 += sum (dE_i / d d_ij) (x) d_ij (may be NULL).  cutoff must not exceed the radial cutoff the rows
 * were built with.  Symmetric rows (every builder except anihip_nbr_from_full): no atomics, each atom completes its own
 * entries, deterministic; pass ANIHIP_PAIR_PUSH for asymmetric rows. */
#define ANIHIP_PAIR_PUSH 1
int anihip_pair_xtb_repulsion(void *stream, int64_t n_atoms, int64_t lo, int64_t hi, const int32_t *species,
                              const uint32_t *meta, const float *ent, const float *pair_table, float cutoff,
                              int32_t cutoff_kind, int32_t flags, float *atomic_e, float *grad_coords, double *virial);

/* The other closed-form pair potentials of torchani.potentials on the same rows, same outputs and flags; pair_table:
 * device float[8][8][4] per [species_i][species_j]; distances d in Bohr, r in Angstrom (r clamped to >= 1e-7 A unless
 * ANIHIP_PAIR_NO_CLAMP, core.py:138-139):
 *   ANIHIP_PAIR_XTB      {y_ab, sqrt(alpha_ab), k_ab, -}        y / d exp(-sqrt(alpha) d^k)              (xtb.py:17-77)
 *   ANIHIP_PAIR_ZBL      {Za Zb, (Za^kz + Zb^kz) / k, -, -}     Za Zb / d sum_i c_i exp(-b_i d s)        (zbl.py:10-81)
 *                        extra: float[8] = c_0..3, b_0..3 (host memory)
 *   ANIHIP_PAIR_LJ       {4 eps_ab, sigma_ab, c12, c6}          4 eps (c12 x^12 + c6 x^6), x = sigma / r (lj.py:42-108)
 *   ANIHIP_PAIR_COULOMB  {q_a q_b / dielectric, 1 / eta_ab, -, -}   qq / sqrt(d^2 + 1 / eta^2)   (fixed_coulomb.py:8-75;
 *                        1 / eta = 0: FixedCoulomb, clamped; FixedMNOK: 2 / (eta_a + eta_b), ANIHIP_PAIR_NO_CLAMP) */
#define ANIHIP_PAIR_XTB 0
#define ANIHIP_PAIR_ZBL 1
#define ANIHIP_PAIR_LJ 2
#define ANIHIP_PAIR_COULOMB 3
#define ANIHIP_PAIR_NO_CLAMP 2
int anihip_pair_analytic(void *stream, int32_t kind, int64_t n_atoms, int64_t lo, int64_t hi, const int32_t *species,
                         const uint32_t *meta, const float *ent, const float *pair_table, const float *extra,
                         float cutoff, int32_t cutoff_kind, int32_t flags, float *atomic_e, float *grad_coords,
                         double *virial);

/* DFT-D3(BJ) two-body dispersion (potentials/dftd3.py:113-330 TwoBodyDispersionD3, damping :44-110 BeckeJohnsonDamp;
 * envelope and per-atom halves as above), distances in Bohr:
 *   CN_i   = sum_j 1 / (1 + exp(-16 (4/3 (Rcov_a + Rcov_b) / d_ij - 1)))                     (all neighbors of the row)
 *   C6_ij  = sum_ref c6ref L / sum_ref L,  L = exp(-4 ((CN_i - cn_a)^2 + (CN_j - cn_b)^2)) over the references with c6ref > 0
 *   e_ij   = -(s6 C6 / (d^6 + R^6) + s8 3 C6 q_a q_b / (d^8 + R^8)) fc(r_ij),  R = a1 sqrt(3 q_a q_b) + a2
 * c6_table: device float[8][8][25][4] = {c6ref, cn_a, cn_b, -} per [species_i][species_j]: the reference pairs with
 * c6ref > 0 first (any order), their number in the 4th float of entry 0 (missing references are -1 in Grimme's table).
 * The rows must hold ALL atoms 0 .. n_atoms (coordination numbers of every neighbor are needed) and be symmetric;
 * lo / hi select the central atoms whose energies / gradient rows are accumulated (atomic_e[i] += sum_j e_ij / 2,
 * grad_coords[i] += d E / d r_i, both complete for i in lo .. hi: nothing is pushed to other atoms, no atomics,
 * deterministic).  The gradient includes the dependence of C6 on the coordination numbers (the reference gets it from
 * autograd).  cn, gcn: caller-provided scratch, n_atoms floats each.  virial as above (may be NULL). */
typedef struct anihip_d3_params {
    float s6, s8, a1, a2;
    float cov_radius_bohr[8];   /* per species index */
    float sqrt_q[8];            /* sqrt of the empirical charge, per species index */
} anihip_d3_params;
int anihip_pair_d3(void *stream, int64_t n_atoms, int64_t lo, int64_t hi, const int32_t *species,
                   const uint32_t *meta, const float *ent, const float *c6_table, const anihip_d3_params *params,
                   float cutoff, int32_t cutoff_kind, float *cn, float *gcn, float *atomic_e, float *grad_coords,
                   double *virial);

/* mol_e[c] (fp64) = sum_a atomic_e[c,a] + sae[species[c,a]] over the atoms lo <= c*A+a < hi; padding
 * contributes nothing (sae.py:54-64).  sae may be NULL.  mol_e is overwritten. */
int anihip_energy_reduce(void *stream, int32_t n_mol, int32_t n_atoms_per_mol, int64_t lo, int64_t hi,
                         const int32_t *species, const float *atomic_e, const double *sae, double *mol_e);

/* The last launch of an energies-and-forces step: anihip_energy_reduce, and grad_coords[0 .. n_grad) (the d E / d coords
 * that anihip_aev_backward and the pair terms accumulated) negated in place into forces (grad.py:283-290); one launch for
 * batches of small molecules, where a step is bound by the number of launches. */
int anihip_energy_forces_finish(void *stream, int32_t n_mol, int32_t n_atoms_per_mol, int64_t lo, int64_t hi,
                                const int32_t *species, const float *atomic_e, const double *sae, double *mol_e,
                                float *grad_coords, int64_t n_grad);

#ifdef __cplusplus
}
#endif
#endif /* ANIHIP_H */

// Preparation and finishing kernels of the network stage (declared in mlp_prep.h; the host entry points that launch them and
// the GEMM kernels: csrc/mlp.hip; the fused network kernel: csrc/mlp_fused.hip).  Replaces the nonzero() / index_select
// bucketing of nn/_containers.py:406-416 and the energy sums of nn/_containers.py:417-421, sae.py:54-64.
#include "mlp_prep.h"

#include <cstdio>
#include <stdlib.h>

namespace anihip {

// ---- species bucketing --------------------------------------------------------------------------


// Stable counting sort of the atoms lo..hi by species: count per 1024-atom chunk (one wave each) -> exclusive scan
// over the chunks -> scatter.  No atomics: the sorted order (index order inside a species) and with it the row tiles,
// their maxima and every rounding downstream are the same from run to run.  chunk_cnt: [n_chunks][MAX_S] ints of
// scratch (the launcher lends the not yet written member_part buffer).
__global__ void k_sp_count(int64_t lo, int64_t hi, const int32_t *species, int S, int *chunk_cnt)
{
    const int64_t wave = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t c0 = lo + wave * SP_CHUNK;
    if (c0 >= hi) return;
    int cnt[MAX_S];
#pragma unroll
    for (int t = 0; t < MAX_S; ++t) cnt[t] = 0;
    for (int it = 0; it < SP_CHUNK / WAVE; ++it) {
        const int64_t i = c0 + it * WAVE + lane_id();
        const int sp = (i < hi) ? species[i] : -1;
#pragma unroll
        for (int t = 0; t < MAX_S; ++t)
            if (t < S) cnt[t] += __popcll(__ballot(sp == t));
    }
    if (lane_id() < MAX_S) {
        int v = 0;
#pragma unroll
        for (int t = 0; t < MAX_S; ++t) v = lane_id() == t ? cnt[t] : v;
        chunk_cnt[wave * MAX_S + lane_id()] = v;
    }
}

// one workgroup: chunk_cnt[c][t] -> number of atoms of species t in the chunks before c; totals / offsets -> ctl
__global__ __launch_bounds__(256) void k_sp_offsets(int S, int n_chunks, int *chunk_cnt, int *ctl)
{
    __shared__ int s_sum[256][MAX_S];
    const int tid = threadIdx.x;
    const int per = (n_chunks + 255) / 256;
    const int c0 = tid * per, c1 = min(n_chunks, c0 + per);
    int loc[MAX_S];
#pragma unroll
    for (int t = 0; t < MAX_S; ++t) loc[t] = 0;
    for (int c = c0; c < c1; ++c)
#pragma unroll
        for (int t = 0; t < MAX_S; ++t) loc[t] += chunk_cnt[c * MAX_S + t];
#pragma unroll
    for (int t = 0; t < MAX_S; ++t) s_sum[tid][t] = loc[t];
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {   // inclusive scan over the threads
        int add[MAX_S];
#pragma unroll
        for (int t = 0; t < MAX_S; ++t) add[t] = tid >= o ? s_sum[tid - o][t] : 0;
        __syncthreads();
#pragma unroll
        for (int t = 0; t < MAX_S; ++t) s_sum[tid][t] += add[t];
        __syncthreads();
    }
    int run[MAX_S];
#pragma unroll
    for (int t = 0; t < MAX_S; ++t) run[t] = s_sum[tid][t] - loc[t];   // exclusive
    for (int c = c0; c < c1; ++c)
#pragma unroll
        for (int t = 0; t < MAX_S; ++t) {
            const int v = chunk_cnt[c * MAX_S + t];
            chunk_cnt[c * MAX_S + t] = run[t];
            run[t] += v;
        }
    if (tid == 0) {
        int tot = 0, trun = 0;
        for (int t = 0; t < S; ++t) {
            const int cnt = s_sum[255][t];
            ctl[CTL_CNT + t] = cnt;
            ctl[CTL_OFF + t] = tot;
            ctl[CTL_TILE + t] = trun;
            tot += cnt;
            trun += (cnt + BM - 1) / BM;
        }
        ctl[CTL_OFF + S] = tot;
        ctl[CTL_TILE + S] = trun;
    }
}

// (the outputs of PADDING atoms -- per-atom energy, gradient row, member energies -- are zeroed here as well: a wave that
// meets one zeroes it with all its lanes; a system without padding pays one ballot per 64 atoms instead of the separate
// k_zero_padding launch, 50 us at 2.3 M atoms)
__global__ void k_sp_scatter(int64_t lo, int64_t hi, const int32_t *species, int S, const int *ctl,
                             const int *chunk_cnt, int *perm, float *atomic_e, float *grad_aev, int L, float *member_e,
                             int M, int64_t n_atoms)
{
    const int64_t wave = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t c0 = lo + wave * SP_CHUNK;
    if (c0 >= hi) return;
    int base[MAX_S];
#pragma unroll
    for (int t = 0; t < MAX_S; ++t) base[t] = t < S ? ctl[CTL_OFF + t] + chunk_cnt[wave * MAX_S + t] : 0;
    for (int it = 0; it < SP_CHUNK / WAVE; ++it) {
        const int64_t i = c0 + it * WAVE + lane_id();
        const int sp = (i < hi) ? species[i] : -1;
#pragma unroll
        for (int t = 0; t < MAX_S; ++t)
            if (t < S) {
                const uint64_t m = __ballot(sp == t);
                if (sp == t) perm[base[t] + mbcnt(m)] = (int)i;
                base[t] += __popcll(m);
            }
        // (the per-atom scalars by the padding atom's own lane; only the gradient rows take the whole wave, one row at a time --
        // a training batch is half padding, 512 serial trips per wave here when every output went that way: 55 -> 15 us)
        const bool is_pad = i < hi && sp < 0;
        if (is_pad && atomic_e) atomic_e[i] = 0.f;
        if (is_pad && member_e)
            for (int m = 0; m < M; ++m) member_e[(int64_t)m * n_atoms + i] = 0.f;
        if (grad_aev)
            for (uint64_t pad = __ballot(is_pad); pad; pad &= pad - 1) {
                const int64_t ip = c0 + it * WAVE + (int)__builtin_ctzll(pad);
                float4 *row = reinterpret_cast<float4 *>(grad_aev + (size_t)ip * L);
                for (int f = lane_id(); f < (L >> 2); f += WAVE) row[f] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
    }
}

// (the fused network kernel k_mlp_fused lives in csrc/mlp_fused.hip; its argument block and launcher: mlp_fused.h)
// Tile table of the fused kernel: one wave per tile resolves (species, rows, atoms, OR of the atoms' slab
// masks) once, so that the member workgroups of a tile start from two independent loads instead of a chain of
// five dependent ones.
// ani_species > 0 (no per-atom flags, the whole system in one call, ANI layout of the AEV row: 16 radial columns per species,
// then one 32-column block per species pair): the slabs of species (pairs) that do not occur in the system at all are zero for
// every atom -- a superset of each atom's flags that costs nothing (the training batches: H C N O flag 12 of 32 slabs).
__global__ __launch_bounds__(256) void k_tile_table(const int *ctl, int S, const int *perm,
                                                    const uint32_t *slab_mask, uint32_t all_slabs,
                                                    int tiles_total, int rows_per_tile, int4 *tile_tab,
                                                    int *tile_rows, int ani_species = 0)
{
    if (!slab_mask && ani_species > 0) {
        const int nrs = (16 * ani_species + 31) / 32;
        uint32_t mk = 0u;
        for (int a = 0; a < ani_species; ++a) {
            if (ctl[CTL_CNT + a] <= 0) continue;
            mk |= 1u << (a >> 1);
            for (int b = a; b < ani_species; ++b)
                if (ctl[CTL_CNT + b] > 0) mk |= 1u << (nrs + a * ani_species - a * (a - 1) / 2 + (b - a));
        }
        all_slabs &= mk;
    }
    const int tile0 = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (tile0 >= tiles_total) return;
    int tile = tile0, s = 0, cnt = 0;
    for (; s < S; ++s) {
        cnt = ctl[CTL_CNT + s];
        const int nt = (cnt + rows_per_tile - 1) / rows_per_tile;
        if (tile < nt) break;
        tile -= nt;
    }
    if (s >= S) {   // (the fused kernel prefetches the rows of the next item before it looks at its entry: atom 0)
        if (lane == 0) tile_tab[tile0] = make_int4(-1, 0, 0, 0);
        if (lane < rows_per_tile) tile_rows[(size_t)tile0 * rows_per_tile + lane] = 0;
        return;
    }
    const int n_rows = min(rows_per_tile, cnt - tile * rows_per_tile);
    const int p0 = ctl[CTL_OFF + s] + tile * rows_per_tile;
    const int atom = perm[p0 + min(lane, n_rows - 1)];
    if (lane < rows_per_tile) tile_rows[(size_t)tile0 * rows_per_tile + lane] = atom;
    uint32_t mk = all_slabs;
    if (slab_mask) {
        mk = slab_mask[atom];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mk |= (uint32_t)__shfl_xor((int)mk, o);
    }
    if (lane == 0) tile_tab[tile0] = make_int4(s, p0, n_rows, (int)mk);
}

// The tile table sorted by falling tile cost, for the fused kernel's tile queue (owner order, mid-size systems): a
// counting sort over 64 cost classes, entries and row lists copied to their places in a second table (empty entries last, as
// the kernel expects).  Cost of a tile through one member ~ its species' first hidden width x (1 + 0.17 per pass of four
// flagged slabs behind the first: the layer-0 k loop and phase 5 grow with the slabs, the phases between them do not).  The order
// inside a class is whatever the LDS atomics make it: every tile's result is independent of which workgroup computes it and when.
__global__ __launch_bounds__(256) void k_tile_order(TileOrderArgs g)
{
    // every workgroup counts the whole table (16 bytes per tile, L2 hits) -- the classes' totals and what lies ahead of its
    // own chunk -- and places its chunk: no second launch, no global histogram
    __shared__ int s_tot[65], s_pos[65];   // [64]: the empty entries
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c0 = blockIdx.x * TO_CHUNK;
    if (tid < 65) { s_tot[tid] = 0; s_pos[tid] = 0; }
    __syncthreads();
    auto cls = [&](const int4 &t) {
        if (t.x < 0) return 64;
        const int passes = max(1, (__popc((uint32_t)t.w) + 3) >> 2);
        const float c = (float)g.H1[t.x] * (1.0f / 256.0f) * (1.0f + 0.17f * (float)(passes - 1));
        return 63 - min(63, (int)(c * 24.0f));   // class 0 = the most expensive
    };
    for (int t = tid; t < g.tiles_total; t += 256) {
        const int c = cls(g.tile_tab[t]);
        atomicAdd(&s_tot[c], 1);
        if (t < c0) atomicAdd(&s_pos[c], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int c = 0; c < 65; ++c) { const int n = s_tot[c]; s_pos[c] += run; run += n; }
    }
    __syncthreads();
    // one wave per tile: lane 0 draws the place, every lane copies one row index
#pragma unroll
    for (int k = 0; k < TO_CHUNK / 4; ++k) {
        const int t = c0 + wave * (TO_CHUNK / 4) + k;
        if (t >= g.tiles_total) break;
        const int4 e = g.tile_tab[t];
        int dst = 0;
        if (lane == 0) dst = atomicAdd(&s_pos[cls(e)], 1);
        dst = __builtin_amdgcn_readfirstlane(dst);
        if (lane == 0) g.tile_tab2[dst] = e;
        g.tile_rows2[(size_t)dst * 64 + lane] = g.tile_rows[(size_t)t * 64 + lane];
    }
}

// ---- small inputs: the whole preparation in ONE launch ------------------------------------------------------------
// Below SMALL_PREP_MAX atoms a step is bound by the number of dependent launches, not by work.  Block 0 (16 waves) does what
// zero_words + k_sp_count + k_sp_offsets + k_sp_scatter + k_tile_table do in five launches: every wave loads its
// contiguous chunk of species (and slab flags) in one go and counts, the counts are scanned through LDS, and every atom
// goes from its register straight to its sorted position (same stable order as the chunked kernels: index order inside
// a species): permutation, row of its tile, the tile's slab flags (an LDS OR); one thread per tile then writes the table
// entry.  Three barriers, ~10 us.  Blocks 1.. zero the rows of the padding atoms (k_zero_padding).
constexpr int SMALL_PREP_ITERS = SMALL_PREP_MAX / (SMALL_PREP_WAVES * WAVE);   // 16

#ifdef ANIHIP_DEV_TRACE   // development builds: 100-MHz clock stamps of block 0's phases
__device__ unsigned long long g_prep_trace[16];
#define PREP_STAMP(k) if (threadIdx.x == 0) g_prep_trace[k] = wall_clock64();
#else
#define PREP_STAMP(k)
#endif

__global__ __launch_bounds__(SMALL_PREP_WAVES * WAVE) void k_small_prep(
    int64_t lo, int64_t hi, const int32_t *species, int S, int *ctl, int *perm, const uint32_t *slab_mask,
    uint32_t all_slabs, int tiles_total, int rows_per_tile, int4 *tile_tab, int *tile_rows, float *atomic_e,
    float *grad_aev, int L, float *member_e, int M, int64_t n_atoms)
{
    extern __shared__ int s_dyn[];   // [tiles] OR of the slab flags per tile
    __shared__ int s_cnt[SMALL_PREP_WAVES][MAX_S];
    __shared__ int s_ctl[CTL_WORDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (blockIdx.x > 0) {   // padding atoms: zero energy / zero gradient rows, one wave per atom
        const int64_t nw = (int64_t)(gridDim.x - 1) * SMALL_PREP_WAVES;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int64_t i = lo + (int64_t)(blockIdx.x - 1) * SMALL_PREP_WAVES + wave; i < hi; i += nw) {
            if (species[i] >= 0) continue;
            if (lane == 0) atomic_e[i] = 0.f;
            if (member_e && lane < M) member_e[(int64_t)lane * n_atoms + i] = 0.f;
            if (grad_aev) {
                float4 *row = reinterpret_cast<float4 *>(grad_aev + (size_t)i * L);
                for (int f = lane; f < (L >> 2); f += WAVE) row[f] = z4;
            }
        }
        return;
    }
    PREP_STAMP(0)
    const int n = (int)(hi - lo);
    int *s_tmask = s_dyn;   // [tiles] OR of the slab flags of a tile's atoms
    const int chunk = (((n + SMALL_PREP_WAVES - 1) / SMALL_PREP_WAVES) + WAVE - 1) & ~(WAVE - 1);
    const int c0 = wave * chunk;
    int sp[SMALL_PREP_ITERS];
    uint32_t mk[SMALL_PREP_ITERS];
#pragma unroll
    for (int it = 0; it < SMALL_PREP_ITERS; ++it) {
        const int r = c0 + it * WAVE + lane;
        const bool ok = it * WAVE < chunk && r < n;
        sp[it] = ok ? species[lo + r] : -1;
        mk[it] = (ok && slab_mask) ? slab_mask[lo + r] : all_slabs;
    }
    // running maxima behind the control block start from zero; the control words themselves are written below
    for (int q = CTL_WORDS + tid; q < CTL_WORDS + AMAX_WORDS; q += SMALL_PREP_WAVES * WAVE) ctl[q] = 0;
    if (tid < CTL_WORDS) s_ctl[tid] = 0;
    if (tile_tab)
        for (int q = tid; q < tiles_total; q += SMALL_PREP_WAVES * WAVE) s_tmask[q] = 0;
    int cnt[MAX_S];
#pragma unroll
    for (int t = 0; t < MAX_S; ++t) cnt[t] = 0;
#pragma unroll
    for (int it = 0; it < SMALL_PREP_ITERS; ++it) {
        if (it * WAVE >= chunk) break;
#pragma unroll
        for (int t = 0; t < MAX_S; ++t)
            if (t < S) cnt[t] += __popcll(__ballot(sp[it] == t));
    }
    if (lane < MAX_S) {
        int v = 0;
#pragma unroll
        for (int t = 0; t < MAX_S; ++t) v = lane == t ? cnt[t] : v;
        s_cnt[wave][lane] = v;
    }
    PREP_STAMP(1)
    __syncthreads();
    // thread t < S: exclusive scan of the waves' counts of species t (16 independent LDS reads), totals -> s_tot
    __shared__ int s_tot[MAX_S];
    if (tid < MAX_S) {
        int c[SMALL_PREP_WAVES], run = 0;
#pragma unroll
        for (int w = 0; w < SMALL_PREP_WAVES; ++w) c[w] = s_cnt[w][tid];
#pragma unroll
        for (int w = 0; w < SMALL_PREP_WAVES; ++w) {
            s_cnt[w][tid] = run;
            run += c[w];
        }
        s_tot[tid] = tid < S ? run : 0;
    }
    PREP_STAMP(2)
    __syncthreads();
    PREP_STAMP(3)
    // every thread: species offsets / first tiles (registers), its wave's scatter bases
    int base[MAX_S], off[MAX_S + 1], tfirst[MAX_S + 1];
    {
        int tot = 0, trun = 0, ftrun = 0;
#pragma unroll
        for (int t = 0; t < MAX_S; ++t) {
            const int all = s_tot[t];
            off[t] = tot;
            tfirst[t] = ftrun;
            base[t] = tot + s_cnt[wave][t];
            if (tid == 0 && t < S) {
                s_ctl[CTL_CNT + t] = all;
                s_ctl[CTL_OFF + t] = tot;
                s_ctl[CTL_TILE + t] = trun;
            }
            tot += all;
            trun += (all + BM - 1) / BM;
            ftrun += (all + rows_per_tile - 1) / rows_per_tile;
        }
        off[MAX_S] = tot;
        tfirst[MAX_S] = ftrun;
        if (tid == 0) {
            s_ctl[CTL_OFF + S] = tot;
            s_ctl[CTL_TILE + S] = trun;
        }
    }
    // scatter: sorted position of every atom (index order inside a species), and with it straight to memory: the
    // permutation, the atom's row of its tile, the tile's slab flags (LDS OR); the atom that closes a species also fills
    // the rows its last tile leaves open with itself (what k_tile_table's clamped read does)
    const int shift = rows_per_tile == 64 ? 6 : 5;
#pragma unroll
    for (int it = 0; it < SMALL_PREP_ITERS; ++it) {
        if (it * WAVE >= chunk) break;
        int pos = 0, my_off = 0, my_tf = 0, my_end = 0;
#pragma unroll
        for (int t = 0; t < MAX_S; ++t)
            if (t < S) {
                const uint64_t m = __ballot(sp[it] == t);
                if (sp[it] == t) {
                    pos = base[t] + mbcnt(m);
                    my_off = off[t];
                    my_tf = tfirst[t];
                    my_end = off[t + 1];
                }
                base[t] += __popcll(m);
            }
        if (sp[it] >= 0) {
            const int atom = (int)lo + c0 + it * WAVE + lane;
            perm[pos] = atom;
            if (tile_tab) {
                const int rel = pos - my_off, tile = my_tf + (rel >> shift);
                int *rows = tile_rows + (size_t)tile * rows_per_tile;
                rows[rel & (rows_per_tile - 1)] = atom;
                atomicOr(&s_tmask[tile], (int)mk[it]);
                if (pos == my_end - 1)
                    for (int r = (rel & (rows_per_tile - 1)) + 1; r < rows_per_tile; ++r) rows[r] = atom;
            }
        }
    }
    PREP_STAMP(4)
    __syncthreads();
    PREP_STAMP(5)
    if (tid < CTL_WORDS) ctl[tid] = s_ctl[tid];
    if (!tile_tab) return;
    // one thread per tile: its entry (and atom 0 for the rows of the tiles past the last species: the fused kernel
    // prefetches the rows of the next item before it looks at its entry)
    for (int tile0 = tid; tile0 < tiles_total; tile0 += SMALL_PREP_WAVES * WAVE) {
        int s = -1;
#pragma unroll
        for (int t = 0; t < MAX_S; ++t)
            if (t < S && tile0 >= tfirst[t] && tile0 < tfirst[t + 1]) s = t;
        if (s < 0) {
            int *rows = tile_rows + (size_t)tile0 * rows_per_tile;
            tile_tab[tile0] = make_int4(-1, 0, 0, 0);
            for (int r = 0; r < rows_per_tile; ++r) rows[r] = 0;
            continue;
        }
        int o = 0, tf = 0, c = 0;
#pragma unroll
        for (int t = 0; t < MAX_S; ++t) {
            o = s == t ? off[t] : o;
            tf = s == t ? tfirst[t] : tf;
            c = s == t ? off[t + 1] - off[t] : c;
        }
        const int tile = tile0 - tf;
        tile_tab[tile0] = make_int4(s, o + tile * rows_per_tile, min(rows_per_tile, c - tile * rows_per_tile), s_tmask[tile0]);
    }
    PREP_STAMP(8)
}



__global__ void k_fused_finish(FinishArgs f)
{
    fused_finish(f, blockIdx.x * (int64_t)blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
}

// padding atoms inside the shard: zero energy / zero gradient rows
static inline unsigned zero_pad_blocks(int64_t n)   // one wave per atom, four per block
{
    const int64_t b = (n + 3) / 4;
    return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

__global__ void k_zero_padding(int64_t lo, int64_t hi, const int32_t *species, float *atomic_e,
                               float *grad_aev, int L, float *member_e, int M, int64_t n_atoms)
{
    // one wave per atom (grid-stride), 16-B stores (L is a multiple of 4: aev_len % 16 == 0)
    const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t i = lo + blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6); i < hi; i += nw) {
        if (species[i] >= 0) continue;
        if (lane_id() == 0) atomic_e[i] = 0.f;
        if (member_e && lane_id() < M) member_e[(int64_t)lane_id() * n_atoms + i] = 0.f;
        if (grad_aev) {
            float4 *row = reinterpret_cast<float4 *>(grad_aev + (size_t)i * L);
            for (int f = lane_id(); f < (L >> 2); f += WAVE) row[f] = z4;
        }
    }
}

__global__ __launch_bounds__(256) void k_energy_reduce(int n_mol, int A, int64_t lo, int64_t hi,
                                                       const int32_t *species, const float *atomic_e,
                                                       const double *sae, double *mol_e)
{
    const int mol = blockIdx.x;
    double acc = 0.0;
    for (int a = blockIdx.y * blockDim.x + threadIdx.x; a < A; a += gridDim.y * blockDim.x) {
        const int64_t i = (int64_t)mol * A + a;
        if (i < lo || i >= hi) continue;
        const int sp = species[i];
        if (sp < 0) continue;
        acc += (double)atomic_e[i] + (sae ? sae[sp] : 0.0);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    __shared__ double part[4];
    if (lane_id() == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double v = part[0] + part[1] + part[2] + part[3];
        if (gridDim.y == 1) mol_e[mol] = v;   // (one block per molecule: plain store, mol_e needs no zero fill)
        else atomicAdd(&mol_e[mol], v);
    }
}

// energies + forces = -gradient in one launch (few atoms per molecule): blocks 0 .. n_mol - 1 reduce one molecule each,
// the others negate 1024 floats of the gradient each
__global__ __launch_bounds__(256) void k_energy_forces_finish(int n_mol, int A, int64_t lo, int64_t hi,
                                                              const int32_t *species, const float *atomic_e,
                                                              const double *sae, double *mol_e, float *grad, int64_t n_grad)
{
    if ((int)blockIdx.x >= n_mol) {
        const int64_t i0 = ((int64_t)blockIdx.x - n_mol) * 1024 + threadIdx.x * 4;
        if (i0 + 4 <= n_grad && ((uintptr_t)grad & 15) == 0) {
            float4 *p = reinterpret_cast<float4 *>(grad + i0);
            const float4 v = *p;
            *p = make_float4(-v.x, -v.y, -v.z, -v.w);
        } else {
            for (int64_t i = i0; i < i0 + 4 && i < n_grad; ++i) grad[i] = -grad[i];
        }
        return;
    }
    const int mol = blockIdx.x;
    double acc = 0.0;
    for (int a = threadIdx.x; a < A; a += blockDim.x) {
        const int64_t i = (int64_t)mol * A + a;
        if (i < lo || i >= hi) continue;
        const int sp = species[i];
        if (sp < 0) continue;
        acc += (double)atomic_e[i] + (sae ? sae[sp] : 0.0);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    __shared__ double part[4];
    if (lane_id() == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) mol_e[mol] = part[0] + part[1] + part[2] + part[3];
}

__global__ void k_negate(float *x, int64_t n)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] = -x[i];
}

// ---- launchers ----------------------------------------------------------------------------------------------------------------
void launch_bucketing(hipStream_t stream, int64_t lo, int64_t hi, const int32_t *species, int S, int *ctl, int words_to_zero,
                      int *chunk_cnt, int *perm, float *atomic_e, float *grad_aev, int L, float *member_e, int M, int64_t n_atoms)
{
    const int64_t n = hi - lo;
    zero_words_async(stream, ctl, sizeof(int) * (size_t)words_to_zero);
    const unsigned cblk = (unsigned)((n + 4 * SP_CHUNK - 1) / (4 * SP_CHUNK));
    const int n_chunks = (int)((n + SP_CHUNK - 1) / SP_CHUNK);
    hipLaunchKernelGGL(k_sp_count, dim3(cblk), dim3(256), 0, stream, lo, hi, species, S, chunk_cnt);
    hipLaunchKernelGGL(k_sp_offsets, dim3(1), dim3(256), 0, stream, S, n_chunks, chunk_cnt, ctl);
    hipLaunchKernelGGL(k_sp_scatter, dim3(cblk), dim3(256), 0, stream, lo, hi, species, S, ctl, chunk_cnt, perm, atomic_e,
                       grad_aev, L, member_e, M, n_atoms);
}

int launch_small_prep(hipStream_t stream, int64_t lo, int64_t hi, const int32_t *species, int S, int *ctl, int *perm,
                      const uint32_t *slab_mask, uint32_t all_slabs, int tiles_total, int rows_per_tile, int4 *tile_tab,
                      int *tile_rows, float *atomic_e, float *grad_aev, int L, float *member_e, int M, int64_t n_atoms)
{
    const int64_t n = hi - lo;
    const size_t lds = sizeof(int) * (size_t)tiles_total;
    ANIHIP_CHECK_HIP(hipFuncSetAttribute((const void *)k_small_prep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int64_t pad_blocks = (n + SMALL_PREP_WAVES - 1) / SMALL_PREP_WAVES;   // one atom per wave
    if (pad_blocks < 1) pad_blocks = 1;
    hipLaunchKernelGGL(k_small_prep, dim3((unsigned)(1 + pad_blocks)), dim3(SMALL_PREP_WAVES * WAVE), lds, stream, lo, hi,
                       species, S, ctl, perm, slab_mask, all_slabs, tiles_total, rows_per_tile, tile_tab, tile_rows, atomic_e,
                       grad_aev, L, member_e, M, n_atoms);
#ifdef ANIHIP_DEV_TRACE
    if (getenv("ANIHIP_PREP_TRACE")) {
        unsigned long long h[16];
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_prep_trace), sizeof(h));
        fprintf(stderr, "k_small_prep block 0, ns since start: load+count %llu, barrier+scan %llu, barrier %llu, bases+scatter %llu,"
                " barrier %llu, tile table %llu\n", (h[1] - h[0]) * 10, (h[2] - h[0]) * 10, (h[3] - h[0]) * 10,
                (h[4] - h[0]) * 10, (h[5] - h[0]) * 10, (h[8] - h[0]) * 10);
    }
#endif
    return 0;
}

void launch_tile_table(hipStream_t stream, const int *ctl, int S, const int *perm, const uint32_t *slab_mask, uint32_t all_slabs,
                       int tiles_total, int rows_per_tile, int4 *tile_tab, int *tile_rows, int ani_species)
{
    hipLaunchKernelGGL(k_tile_table, dim3((unsigned)((tiles_total + 3) / 4)), dim3(256), 0, stream, ctl, S, perm, slab_mask,
                       all_slabs, tiles_total, rows_per_tile, tile_tab, tile_rows, ani_species);
}

void launch_tile_order(hipStream_t stream, const TileOrderArgs &a)
{
    hipLaunchKernelGGL(k_tile_order, dim3((unsigned)((a.tiles_total + TO_CHUNK - 1) / TO_CHUNK)), dim3(256), 0, stream, a);
}

void launch_fused_finish(hipStream_t stream, const FinishArgs &f, int64_t n)
{
    int64_t fb = (n + 255) / 256;
    if (fb > 2048) fb = 2048;
    if (fb < 1) fb = 1;
    hipLaunchKernelGGL(k_fused_finish, dim3((unsigned)fb), dim3(256), 0, stream, f);
}

void launch_zero_padding(hipStream_t stream, int64_t lo, int64_t hi, const int32_t *species, float *atomic_e, float *grad_aev,
                         int L, float *member_e, int M, int64_t n_atoms)
{
    hipLaunchKernelGGL(k_zero_padding, dim3(zero_pad_blocks(hi - lo)), dim3(256), 0, stream, lo, hi, species, atomic_e, grad_aev,
                       L, member_e, M, n_atoms);
}

}  // namespace anihip

using namespace anihip;

extern "C" int anihip_energy_reduce(void *stream_, int32_t n_mol, int32_t A, int64_t lo, int64_t hi,
                                    const int32_t *species, const float *atomic_e, const double *sae,
                                    double *mol_e)
{
    hipStream_t stream = (hipStream_t)stream_;
    ANIHIP_REQUIRE(species && atomic_e && mol_e, "null pointer argument");
    ANIHIP_REQUIRE(n_mol >= 1 && A >= 1, "bad shape");
    int ny = (A + 256 * 16 - 1) / (256 * 16);
    if (ny < 1) ny = 1;
    if (ny > 1024) ny = 1024;
    if (ny > 1) zero_words_async(stream, mol_e, sizeof(double) * (size_t)n_mol);
    hipLaunchKernelGGL(k_energy_reduce, dim3((unsigned)n_mol, (unsigned)ny), dim3(256), 0, stream, (int)n_mol,
                       (int)A, lo, hi, species, atomic_e, sae, mol_e);
    ANIHIP_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int anihip_energy_forces_finish(void *stream_, int32_t n_mol, int32_t A, int64_t lo, int64_t hi,
                                           const int32_t *species, const float *atomic_e, const double *sae,
                                           double *mol_e, float *grad_coords, int64_t n_grad)
{
    hipStream_t stream = (hipStream_t)stream_;
    ANIHIP_REQUIRE(species && atomic_e && mol_e, "null pointer argument");
    ANIHIP_REQUIRE(n_mol >= 1 && A >= 1 && n_grad >= 0 && (grad_coords || n_grad == 0), "bad shape");
    if (A <= 256 * 16) {   // (one block per molecule is enough: everything in one launch)
        const int64_t nblk = (n_grad + 1023) / 1024;
        hipLaunchKernelGGL(k_energy_forces_finish, dim3((unsigned)(n_mol + nblk)), dim3(256), 0, stream, (int)n_mol, (int)A,
                           lo, hi, species, atomic_e, sae, mol_e, grad_coords, n_grad);
    } else {
        if (int rc = anihip_energy_reduce(stream_, n_mol, A, lo, hi, species, atomic_e, sae, mol_e)) return rc;
        if (n_grad > 0) {
            int64_t nblk = (n_grad + 1023) / 1024;
            if (nblk > 4096) nblk = 4096;
            hipLaunchKernelGGL(k_negate, dim3((unsigned)nblk), dim3(256), 0, stream, grad_coords, n_grad);
        }
    }
    ANIHIP_CHECK_HIP(hipGetLastError());
    return 0;
}

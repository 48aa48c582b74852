// Neighbor-list builders of libanihip (gfx950, wave64).
//
// One wave owns one central atom: lanes sweep the candidate atoms (the atom's own molecule, or the
// 27+ grid bins around it), hits are compacted with ballot + mbcnt prefix ranks into a per-wave LDS
// list (no LDS/global atomics), classified into {r <= Rca, r > Rca} x species with ballots and written
// as one species-sorted fixed-capacity row.  Single pass, no host sync, deterministic.
//
// What this replaces in the reference (paths relative to /root/reference/torchani/):
//   neighbors.py:187-275 all_pairs (+PBC images), :366-507 cell_list, :64-113 narrow_down,
//   csrc/aev.cu:180-321 pairwiseDistance*, :975-1039 postProcessNbrList1, csrc/cell_list.cpp.
#include "anihip_common.h"

namespace anihip {

// Device-resident description of the periodic cell / bin grid (written by k_setup, read by the rest).
struct GridDesc {
    double cell[9];   // rows = lattice vectors (identity without a cell)
    double inv[9];    // inverse: frac = r @ inv
    double f0[3];     // non-periodic axes: lower fractional bound of the atoms
    double sc[3];     // (frac - f0) * sc = continuous bin coordinate
    float step[9];    // step[k] = cell_k / sc_k : displacement of one bin along axis k
    int nb[3];        // bins per axis
    int range[3];     // stencil half-width per axis
    int rep[3];       // batch mode: periodic image repeats per axis (neighbors.py:250-275)
    int pbc[3];
    int ncell;
    unsigned lo_enc[3], hi_enc[3];  // order-preserving encodings of the fractional bounding box
};

struct NbrWorkspace {
    GridDesc *desc;
    float4 *pos4;      // [N] batch: wrapped xyz + packed (i | species << 28); cell: bin-local offsets
    int *cellid;       // [N]
    int *sorted_idx;   // [N]
    float4 *pos4s;     // [N] pos4 in bin-sorted order
    int *cell_start;   // [max_cells + 1]
    int *cell_fill;    // [max_cells]
    int *scan_tmp;     // [max_cells / 1024 + 2]
};

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t carve(NbrWorkspace *w, char *base, int64_t n, int64_t max_cells)
{
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char *p = base ? base + off : nullptr;
        off += align256(bytes);
        return p;
    };
    GridDesc *desc = (GridDesc *)take(sizeof(GridDesc));
    float4 *pos4 = (float4 *)take(sizeof(float4) * (size_t)n);
    int *cellid = (int *)take(sizeof(int) * (size_t)n);
    int *sorted_idx = (int *)take(sizeof(int) * (size_t)n);
    float4 *pos4s = (float4 *)take(sizeof(float4) * (size_t)n);
    int *cell_start = (int *)take(sizeof(int) * (size_t)(max_cells + 1));
    int *cell_fill = (int *)take(sizeof(int) * (size_t)(max_cells + 1));
    int *scan_tmp = (int *)take(sizeof(int) * (size_t)(max_cells / 1024 + 4));
    if (w) *w = NbrWorkspace{desc, pos4, cellid, sorted_idx, pos4s, cell_start, cell_fill, scan_tmp};
    return off;
}

__device__ __forceinline__ unsigned enc_f(float f)
{
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec_f(unsigned e)
{
    unsigned u = (e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e;
    return __uint_as_float(u);
}

__device__ void inv3(const double *m, double *o)
{
    double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
    double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    double id = 1.0 / det;
    o[0] = (e * i - f * h) * id; o[1] = (c * h - b * i) * id; o[2] = (b * f - c * e) * id;
    o[3] = (f * g - d * i) * id; o[4] = (a * i - c * g) * id; o[5] = (c * d - a * f) * id;
    o[6] = (d * h - e * g) * id; o[7] = (b * g - a * h) * id; o[8] = (a * e - b * d) * id;
}

// ---- setup -------------------------------------------------------------------------------------

__device__ void desc_fill(GridDesc *g, const float *cell, int pbc_mask, float cutoff)
{
    for (int q = 0; q < 9; ++q) g->cell[q] = cell ? (double)cell[q] : ((q % 4 == 0) ? 1.0 : 0.0);
    inv3(g->cell, g->inv);
    for (int k = 0; k < 3; ++k) {
        g->pbc[k] = (cell && ((pbc_mask >> k) & 1)) ? 1 : 0;
        double nrm = sqrt(g->inv[0 + k] * g->inv[0 + k] + g->inv[3 + k] * g->inv[3 + k] +
                          g->inv[6 + k] * g->inv[6 + k]);
        g->rep[k] = g->pbc[k] ? (int)ceil((double)cutoff * nrm) : 0;  // neighbors.py:253-256
        g->lo_enc[k] = 0xFFFFFFFFu;
        g->hi_enc[k] = 0u;
        g->f0[k] = 0.0;
        g->sc[k] = 1.0;
        g->nb[k] = 1;
        g->range[k] = 0;
    }
    g->ncell = 1;
}

// first kernel of a build: the descriptor, and the status words start from zero
__global__ void k_desc_init(GridDesc *g, const float *cell, int pbc_mask, float cutoff, uint32_t *status)
{
    if (blockIdx.x != 0) return;
    if (threadIdx.x < ANIHIP_STATUS_WORDS) status[threadIdx.x] = 0u;
    if (threadIdx.x == 0) desc_fill(g, cell, pbc_mask, cutoff);
}

// fractional bounding box of the real atoms (needed along non-periodic axes only)
__global__ void k_bbox(GridDesc *g, int64_t n, const int32_t *species, const float *coords)
{
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        if (species[i] < 0) continue;
        double x = coords[3 * i], y = coords[3 * i + 1], z = coords[3 * i + 2];
        for (int k = 0; k < 3; ++k) {
            float f = (float)(x * g->inv[0 + k] + y * g->inv[3 + k] + z * g->inv[6 + k]);
            lo[k] = fminf(lo[k], f);
            hi[k] = fmaxf(hi[k], f);
        }
    }
    for (int k = 0; k < 3; ++k) {
        for (int o = 32; o > 0; o >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], o));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], o));
        }
        if (lane_id() == 0) {
            atomicMin(&g->lo_enc[k], enc_f(lo[k]));
            atomicMax(&g->hi_enc[k], enc_f(hi[k]));
        }
    }
}

__global__ void k_grid_setup(GridDesc *g, float cutoff, int64_t max_cells, uint32_t *status)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double h[3], span[3];
    for (int k = 0; k < 3; ++k) {
        double nrm = sqrt(g->inv[0 + k] * g->inv[0 + k] + g->inv[3 + k] * g->inv[3 + k] +
                          g->inv[6 + k] * g->inv[6 + k]);
        h[k] = 1.0 / nrm;  // distance between the faces perpendicular to axis k
        if (g->pbc[k]) {
            span[k] = 1.0;
            g->f0[k] = 0.0;
        } else {
            // widen by one float ulp-ish margin so every atom falls strictly inside
            double lo = (double)dec_f(g->lo_enc[k]), hi = (double)dec_f(g->hi_enc[k]);
            if (!(hi >= lo)) { lo = 0.0; hi = 0.0; }  // no real atoms
            double pad = 1e-4 * nrm + 1e-6 * (fabs(lo) + fabs(hi));
            lo -= pad; hi += pad;
            span[k] = hi - lo;
            g->f0[k] = lo;
        }
        int nb = (int)floor(span[k] * h[k] / (double)cutoff);
        g->nb[k] = nb < 1 ? 1 : nb;
    }
    // coarsen until the grid fits the workspace (bins only get wider => still correct)
    bool coarsened = false;
    while ((int64_t)g->nb[0] * g->nb[1] * g->nb[2] > max_cells) {
        int k = 0;
        if (g->nb[1] > g->nb[k]) k = 1;
        if (g->nb[2] > g->nb[k]) k = 2;
        g->nb[k] = (g->nb[k] + 1) / 2;
        coarsened = true;
    }
    if (coarsened) atomicOr(&status[0], ANIHIP_ST_GRID_OVERFLOW);
    for (int k = 0; k < 3; ++k) {
        g->sc[k] = (double)g->nb[k] / span[k];
        double width = span[k] * h[k] / g->nb[k];
        g->range[k] = g->pbc[k] ? (int)ceil((double)cutoff / width) : 1;
        for (int q = 0; q < 3; ++q) g->step[3 * k + q] = (float)(g->cell[3 * k + q] / g->sc[k]);
    }
    g->ncell = g->nb[0] * g->nb[1] * g->nb[2];
    status[2] = (uint32_t)g->ncell;
}

// ---- batch mode: packed (wrapped) positions -------------------------------------------------------

// (also the first kernel of a batch build: block 0 writes the descriptor k_nbr_batch reads and resets the status words;
// the wrap itself works from the caller's cell so that it does not have to wait for a descriptor kernel)
__global__ void k_prep_batch(GridDesc *g, int64_t n, const int32_t *species, const float *coords, float4 *pos4,
                             const float *cell, int pbc_mask, float cutoff, uint32_t *status)
{
    if (blockIdx.x == 0) {
        if (threadIdx.x < ANIHIP_STATUS_WORDS) status[threadIdx.x] = 0u;
        if (threadIdx.x == 0) desc_fill(g, cell, pbc_mask, cutoff);
    }
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = coords[3 * i], y = coords[3 * i + 1], z = coords[3 * i + 2];
    if (cell && (pbc_mask & 7)) {
        // utils.py:237-255 map_to_central, in double
        double c[9], inv[9], f[3];
        for (int q = 0; q < 9; ++q) c[q] = (double)cell[q];
        inv3(c, inv);
        for (int k = 0; k < 3; ++k) {
            f[k] = (double)x * inv[0 + k] + (double)y * inv[3 + k] + (double)z * inv[6 + k];
            if ((pbc_mask >> k) & 1) f[k] -= floor(f[k]);
        }
        x = (float)(f[0] * c[0] + f[1] * c[3] + f[2] * c[6]);
        y = (float)(f[0] * c[1] + f[1] * c[4] + f[2] * c[7]);
        z = (float)(f[0] * c[2] + f[1] * c[5] + f[2] * c[8]);
    }
    int sp = species[i];
    uint32_t w = ((uint32_t)i & IDX_MASK) | ((sp < 0 ? SP_PAD : (uint32_t)sp) << 28);
    pos4[i] = make_float4(x, y, z, __uint_as_float(w));
}

// ---- cell mode: binning ----------------------------------------------------------------------------

// Runs of equal keys among the consecutive lanes of a wave (atoms that follow each other in memory mostly share their bin:
// lattice order, cell-sorted order): the first lane of a run does ONE atomic for the whole run -- the 2.3 M same-address
// conflicts of a lane-per-atom atomic went through the L2 one by one (k_bin_fill: 101 us).  key < 0: the lane takes no part.
// Returns the run's length in its first lane (0 elsewhere); rank = this lane's place in its run, head = the run's first lane.
__device__ __forceinline__ int wave_runs(int key, int &rank, int &head)
{
    const int lane = lane_id();
    const int k = key < 0 ? -1 - lane : key;   // (unique: a run of its own)
    const int prev = __shfl_up(k, 1);
    const bool is_head = lane == 0 || k != prev;
    const uint64_t heads = __ballot(is_head);
    const uint64_t below = heads & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
    head = 63 - (int)__builtin_clzll(below);
    rank = lane - head;
    const uint64_t above = lane == 63 ? 0ull : (heads >> (lane + 1));
    const int next = above ? lane + 1 + (int)__builtin_ctzll(above) : 64;
    return (is_head && key >= 0) ? next - lane : 0;
}

__global__ void k_bin_count(const GridDesc *g, int64_t n, const int32_t *species, const float *coords,
                            float4 *pos4, int *cellid, int *cell_cnt)
{
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    int c = -1;
    if (i < n) {
        int sp = species[i];
        if (sp >= 0) {
            double x = coords[3 * i], y = coords[3 * i + 1], z = coords[3 * i + 2];
            int b[3];
            double loc[3];
            for (int k = 0; k < 3; ++k) {
                double f = x * g->inv[0 + k] + y * g->inv[3 + k] + z * g->inv[6 + k];
                if (g->pbc[k]) f -= floor(f);
                double q = (f - g->f0[k]) * g->sc[k];
                int bk = (int)floor(q);
                bk = bk < 0 ? 0 : (bk >= g->nb[k] ? g->nb[k] - 1 : bk);
                b[k] = bk;
                loc[k] = (q - bk) / g->sc[k];  // fractional offset from the bin corner
            }
            // bin-local cartesian offset: small magnitude => fp32 keeps ~1e-7 A resolution in any box size
            float ox = (float)(loc[0] * g->cell[0] + loc[1] * g->cell[3] + loc[2] * g->cell[6]);
            float oy = (float)(loc[0] * g->cell[1] + loc[1] * g->cell[4] + loc[2] * g->cell[7]);
            float oz = (float)(loc[0] * g->cell[2] + loc[1] * g->cell[5] + loc[2] * g->cell[8]);
            uint32_t w = ((uint32_t)i & IDX_MASK) | ((uint32_t)sp << 28);
            pos4[i] = make_float4(ox, oy, oz, __uint_as_float(w));
            c = (b[0] * g->nb[1] + b[1]) * g->nb[2] + b[2];
        }
        cellid[i] = c;
    }
    int rank, head;
    const int run = wave_runs(c, rank, head);
    if (run > 0) atomicAdd(&cell_cnt[c], run);
}

// exclusive scan of `in[0..n)` (n read from device) into out[0..n], three small kernels
__global__ void k_scan_local(const GridDesc *g, const int *in, int *out, int *block_sums)
{
    __shared__ int s[1024];
    const int n = g->ncell;
    int i = blockIdx.x * 1024 + threadIdx.x;
    if (blockIdx.x * 1024 >= n + 1) return;
    int v = (i < n) ? in[i] : 0;
    s[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        int t = (threadIdx.x >= o) ? s[threadIdx.x - o] : 0;
        __syncthreads();
        s[threadIdx.x] += t;
        __syncthreads();
    }
    if (i <= n) out[i] = s[threadIdx.x] - v;  // exclusive
    if (threadIdx.x == 1023) block_sums[blockIdx.x] = s[1023];
}

// (one workgroup scans the block sums 1024 at a time: a single thread walking them is a chain of dependent loads -- 20 us for the
// 172 blocks of a 2.3 M-atom box)
__global__ __launch_bounds__(1024) void k_scan_top(const GridDesc *g, int *block_sums)
{
    __shared__ int s[1024];
    __shared__ int carry;
    if (blockIdx.x != 0) return;
    const int nblk = (g->ncell + 1 + 1023) / 1024;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nblk; base += 1024) {
        const int i = base + (int)threadIdx.x;
        const int v = i < nblk ? block_sums[i] : 0;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int t = (threadIdx.x >= (unsigned)o) ? s[threadIdx.x - o] : 0;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nblk) block_sums[i] = carry + s[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += s[1023];
        __syncthreads();
    }
}

__global__ void k_scan_add(const GridDesc *g, int *out, const int *block_sums)
{
    int i = blockIdx.x * 1024 + threadIdx.x;
    if (i <= g->ncell) out[i] += block_sums[blockIdx.x];
}

__global__ void k_bin_fill(int64_t n, const int *cellid, const int *cell_start, int *cell_fill,
                           int *sorted_idx)
{
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int c = i < n ? cellid[i] : -1;
    int rank, head;
    const int run = wave_runs(c, rank, head);
    int base = 0;
    if (run > 0) base = atomicAdd(&cell_fill[c], run);   // (one returning atomic per run of atoms of one bin)
    base = __shfl(base, head);
    if (c >= 0) sorted_idx[cell_start[c] + base + rank] = (int)i;
}

// make the order inside every bin deterministic (ascending atom index) and gather the positions: one thread per ATOM
// counts the atoms of its bin with a smaller index (the bin's list in the arbitrary order the slot atomics left it: ~13
// contiguous integers, shared by the bin's atoms through the caches) and puts its position at that rank -- no serial
// insertion sort per bin in global memory (one thread per bin: 100 us at 2.3 M atoms)
__global__ void k_bin_rank_gather(int64_t n, const int *cellid, const int *cell_start, const int *sorted_idx,
                                  const float4 *pos4, float4 *pos4s)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = cellid[i];
    if (c < 0) return;
    const int b = cell_start[c], e = cell_start[c + 1];
    int rank = 0;
    for (int p = b; p < e; ++p) rank += sorted_idx[p] < (int)i ? 1 : 0;
    pos4s[b + rank] = pos4[i];
}

// ---- the per-atom search -----------------------------------------------------------------------------

constexpr int NBR_WPB = 4;  // waves per block

// periodic wrap of a bin index that is at most a few bins outside [0, nb): no integer division on the
// common path (the stencil half-width exceeds the grid only for cells thinner than the cutoff)
__device__ __forceinline__ int wrap_bin(int c, int nb)
{
    c = c < 0 ? c + nb : c;
    c = c >= nb ? c - nb : c;
    if (c < 0 || c >= nb) c = ((c % nb) + nb) % nb;
    return c;
}

struct HitList {
    float4 *buf;  // per-wave LDS, MAXR entries
    int n;        // wave-uniform
    bool overflow;
};

// Append the lanes with `hit` set (ballot-compacted, keeps candidate order => deterministic rows).
__device__ __forceinline__ void push_hits(HitList &h, bool hit, float dx, float dy, float dz, float w)
{
    uint64_t m = __ballot(hit);
    if (m == 0) return;
    int pos = h.n + mbcnt(m);
    if (hit && pos < MAXR) h.buf[pos] = make_float4(dx, dy, dz, w);
    h.n += __popcll(m);
    if (h.n > MAXR) {
        h.overflow = true;
        h.n = MAXR;
    }
}

// Classify the compacted hits into {r<=Rca, r>Rca} x species, write the row + metadata.
__device__ void emit_row(HitList &h, int S, float rca2, int row_cap, uint32_t *meta_i, float4 *row,
                         uint32_t *status)
{
    wave_sync();
    const int lane = lane_id();
    int cnt[2][MAX_S];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int t = 0; t < MAX_S; ++t) cnt[g][t] = 0;
    const int n = h.n;
    for (int b0 = 0; b0 < n; b0 += WAVE) {
        int e = b0 + lane;
        bool v = e < n;
        float4 q = h.buf[v ? e : 0];
        int grp = (q.x * q.x + q.y * q.y + q.z * q.z <= rca2) ? 0 : 1;
        int sp = (int)(__float_as_uint(q.w) >> 28);
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int t = 0; t < MAX_S; ++t)
                if (t < S) cnt[g][t] += __popcll(__ballot(v && grp == g && sp == t));
    }
    int nA = 0, nF = 0, mx = 0;
#pragma unroll
    for (int t = 0; t < MAX_S; ++t) {
        nA += cnt[0][t];
        nF += cnt[1][t];
        mx = max(mx, max(cnt[0][t], cnt[1][t]));
    }
    bool bad = h.overflow || (nA + nF > row_cap) || nA > MAXA || mx > 255;
    if (bad) {
        if (lane == 0) {
            atomicOr(&status[0], ANIHIP_ST_ROW_OVERFLOW);
            meta_i[1] = 0; meta_i[2] = 0; meta_i[3] = 0; meta_i[4] = 0; meta_i[5] = 0;
        }
        return;
    }
    int base[2][MAX_S];
    int run = 0;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int t = 0; t < MAX_S; ++t) {
            base[g][t] = run;
            run += cnt[g][t];
        }
    for (int b0 = 0; b0 < n; b0 += WAVE) {
        int e = b0 + lane;
        bool v = e < n;
        float4 q = h.buf[v ? e : 0];
        int grp = (q.x * q.x + q.y * q.y + q.z * q.z <= rca2) ? 0 : 1;
        int sp = (int)(__float_as_uint(q.w) >> 28);
        int pos = -1;
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int t = 0; t < MAX_S; ++t)
                if (t < S) {
                    bool mine = v && grp == g && sp == t;
                    uint64_t m = __ballot(mine);
                    if (mine) pos = base[g][t] + mbcnt(m);
                    base[g][t] += __popcll(m);
                }
        if (pos >= 0) row[pos] = q;
    }
    if (lane == 0) {
        uint32_t pk[4] = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < MAX_S; ++t) {
            pk[t >> 2] |= (uint32_t)cnt[0][t] << (8 * (t & 3));
            pk[2 + (t >> 2)] |= (uint32_t)cnt[1][t] << (8 * (t & 3));
        }
        meta_i[1] = (uint32_t)nA | ((uint32_t)nF << 16);
        meta_i[2] = pk[0]; meta_i[3] = pk[1]; meta_i[4] = pk[2]; meta_i[5] = pk[3];
    }
}

__global__ __launch_bounds__(NBR_WPB * WAVE) void k_nbr_batch(
    const GridDesc *g, int S, float rcr2, float rca2, int A, int64_t lo, int64_t hi,
    const float4 *pos4, int row_cap, uint32_t *meta, float4 *ent, uint32_t *status)
{
    __shared__ float4 s_hits[NBR_WPB][MAXR];
    const int wib = threadIdx.x >> 6, lane = lane_id();
    const int64_t nw = (int64_t)gridDim.x * NBR_WPB;
    const int r0 = g->rep[0], r1 = g->rep[1], r2 = g->rep[2];
    for (int64_t i = lo + blockIdx.x * (int64_t)NBR_WPB + wib; i < hi; i += nw) {
        uint32_t *meta_i = meta + (size_t)i * META_W;
        const size_t row0 = (size_t)(i - lo) * row_cap;
        if (lane == 0) meta_i[0] = (uint32_t)row0;
        const float4 pi = pos4[i];
        if ((__float_as_uint(pi.w) >> 28) == SP_PAD) {
            if (lane == 0) { meta_i[1] = 0; meta_i[2] = 0; meta_i[3] = 0; meta_i[4] = 0; meta_i[5] = 0; }
            continue;
        }
        HitList h{s_hits[wib], 0, false};
        const int64_t mol0 = (i / A) * A;
        for (int n0 = -r0; n0 <= r0; ++n0)
            for (int n1 = -r1; n1 <= r1; ++n1)
                for (int n2 = -r2; n2 <= r2; ++n2) {
                    const float sx = (float)(n0 * g->cell[0] + n1 * g->cell[3] + n2 * g->cell[6]);
                    const float sy = (float)(n0 * g->cell[1] + n1 * g->cell[4] + n2 * g->cell[7]);
                    const float sz = (float)(n0 * g->cell[2] + n1 * g->cell[5] + n2 * g->cell[8]);
                    const bool central_img = (n0 == 0 && n1 == 0 && n2 == 0);
                    for (int a0 = 0; a0 < A; a0 += WAVE) {
                        int a = a0 + lane;
                        bool v = a < A;
                        float4 c = pos4[mol0 + (v ? a : 0)];
                        float dx = (c.x + sx) - pi.x, dy = (c.y + sy) - pi.y, dz = (c.z + sz) - pi.z;
                        float d2 = dx * dx + dy * dy + dz * dz;
                        bool hit = v && ((__float_as_uint(c.w) >> 28) != SP_PAD) && d2 <= rcr2 &&
                                   !(central_img && (mol0 + a) == i);
                        push_hits(h, hit, dx, dy, dz, c.w);
                    }
                }
        emit_row(h, S, rca2, row_cap, meta_i, ent + row0, status);
        wave_sync();
    }
}

// Cell mode: one wave per central atom.  The stencil bins (27 for bins >= cutoff) are resolved by the LANES --
// lane c looks up bin c's [start, end) in the sorted array and its image shift -- and a wave prefix sum turns
// them into one flat candidate range that the 64 lanes then sweep together (each lane finds its bin by a
// binary search in a per-wave LDS table).  No wave-uniform loop over bins, no scalar loads on the path: the
// earlier bin-by-bin sweep issued 1.6 scalar instructions per vector instruction and filled only 60 % of
// the lanes of an iteration.
__global__ __launch_bounds__(NBR_WPB * WAVE) void k_nbr_cell(
    const GridDesc *g, int S, float rcr2, float rca2, int64_t lo, int64_t hi, const float4 *pos4,
    const int *cellid, const int *cell_start, const float4 *pos4s, int row_cap, uint32_t *meta,
    float4 *ent, uint32_t *status, const int *handled, const int *n_unhandled)
{
    // second pass behind k_nbr_cell2: only the atoms of the cells that kernel left (more candidates than it stages, or a
    // stencil of more than 64 bins); nothing left -> nothing to do
    if (n_unhandled && *n_unhandled == 0) return;
    __shared__ float4 s_hits[NBR_WPB][MAXR];
    __shared__ int s_pend[NBR_WPB][WAVE];      // inclusive prefix of the bin populations
    __shared__ int s_k0[NBR_WPB][WAVE];        // sorted position of flat candidate x in bin c: x + k0[c]
    __shared__ float4 s_shift[NBR_WPB][WAVE];  // image shift of bin c; w = 1 for the central bin itself
    const int wib = threadIdx.x >> 6, lane = lane_id();
    int *pend = s_pend[wib], *k0t = s_k0[wib];
    float4 *shift = s_shift[wib];
    const int64_t nw = (int64_t)gridDim.x * NBR_WPB;
    const int nb0 = g->nb[0], nb1 = g->nb[1], nb2 = g->nb[2];
    const int R0 = g->range[0], R1 = g->range[1], R2 = g->range[2];
    const int p0 = g->pbc[0], p1 = g->pbc[1], p2 = g->pbc[2];
    float st[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) st[q] = g->step[q];
    const int n1 = 2 * R1 + 1, n2 = 2 * R2 + 1, ncells = (2 * R0 + 1) * n1 * n2;
    const float inv_n2 = 1.0f / (float)n2, inv_n1 = 1.0f / (float)n1;
    for (int64_t i = lo + blockIdx.x * (int64_t)NBR_WPB + wib; i < hi; i += nw) {
        uint32_t *meta_i = meta + (size_t)i * META_W;
        const size_t row0 = (size_t)(i - lo) * row_cap;
        if (lane == 0) meta_i[0] = (uint32_t)row0;
        const int c = cellid[i];
        if (c < 0) {
            if (lane == 0) { meta_i[1] = 0; meta_i[2] = 0; meta_i[3] = 0; meta_i[4] = 0; meta_i[5] = 0; }
            continue;
        }
        if (handled && handled[c]) continue;
        const float4 pi = pos4[i];
        // bin coordinates, wave-uniform (scalar registers): one division chain per atom
        const int cu = uniform(c);
        const int b2 = cu % nb2, b1 = (cu / nb2) % nb1, b0 = cu / (nb2 * nb1);
        HitList h{s_hits[wib], 0, false};
        for (int cb = 0; cb < ncells; cb += WAVE) {
            // ---- lane = stencil bin (lexicographic in (o0, o1, o2): the row order of the sweep) ----
            const int ci = cb + lane;
            const int q2 = (int)(((float)ci + 0.5f) * inv_n2);        // ci / n2
            const int q1 = (int)(((float)q2 + 0.5f) * inv_n1);        // q2 / n1
            const int o2 = ci - q2 * n2 - R2, o1 = q2 - q1 * n1 - R1, o0 = q1 - R0;
            int c0 = b0 + o0, c1 = b1 + o1, c2 = b2 + o2;
            bool ok = ci < ncells;
            if (p0) c0 = wrap_bin(c0, nb0); else ok = ok && c0 >= 0 && c0 < nb0;
            if (p1) c1 = wrap_bin(c1, nb1); else ok = ok && c1 >= 0 && c1 < nb1;
            if (p2) c2 = wrap_bin(c2, nb2); else ok = ok && c2 >= 0 && c2 < nb2;
            const int cw = ok ? (c0 * nb1 + c1) * nb2 + c2 : 0;
            const int kbeg = cell_start[cw], kend = cell_start[cw + 1];
            const int cnt = ok ? kend - kbeg : 0;
            int incl = cnt;   // inclusive wave prefix sum
#pragma unroll
            for (int d = 1; d < WAVE; d <<= 1) {
                const int up = __shfl_up(incl, d);
                incl += lane >= d ? up : 0;
            }
            const int total = __builtin_amdgcn_readlane(incl, WAVE - 1);
            pend[lane] = incl;
            k0t[lane] = kbeg - (incl - cnt);
            // same evaluation order as the distance below needs: (o0 s0 + o1 s3) first, o2 s6 added last
            shift[lane] = make_float4(o0 * st[0] + o1 * st[3], o0 * st[1] + o1 * st[4], o0 * st[2] + o1 * st[5],
                                      __int_as_float((o2 + 512) | ((o0 == 0 && o1 == 0 && o2 == 0) ? 1024 : 0)));
            wave_sync();
            // ---- lanes sweep the flat candidate range ----
            for (int x0 = 0; x0 < total; x0 += WAVE) {
                const int x = x0 + lane;
                const bool v = x < total;
                int seg = 0;   // first bin whose inclusive prefix exceeds x
#pragma unroll
                for (int stp = WAVE / 2; stp > 0; stp >>= 1) seg += pend[seg + stp - 1] <= x ? stp : 0;
                seg = seg < WAVE ? seg : WAVE - 1;
                const int k = v ? x + k0t[seg] : 0;
                const float4 cnd = pos4s[k];
                const float4 sh = shift[seg];
                const int code = __float_as_int(sh.w);   // (o2 + 512) | (central bin ? 1024 : 0)
                const int o2s = (code & 1023) - 512;
                const float dx = cnd.x + (sh.x - pi.x) + o2s * st[6];
                const float dy = cnd.y + (sh.y - pi.y) + o2s * st[7];
                const float dz = cnd.z + (sh.z - pi.z) + o2s * st[8];
                const float d2 = dx * dx + dy * dy + dz * dz;
                const bool self = (code & 1024) && ((__float_as_uint(cnd.w) & IDX_MASK) == ((uint32_t)i & IDX_MASK));
                const bool hit = v && d2 <= rcr2 && !self;
                push_hits(h, hit, dx, dy, dz, cnd.w);
            }
            wave_sync();
        }
        emit_row(h, S, rca2, row_cap, meta_i, ent + row0, status);
        wave_sync();
    }
}

// rows of the (up to) two central atoms of a sweep from their hits in the staged candidate list (see k_nbr_cell2); the two
// atoms go through every step together, so that the dependent chain hit index -> candidate -> class -> ballot -> rank of
// one atom fills the gaps of the other's
template <int NAT, int NCH>
__device__ __forceinline__ void emit_from_stage(const float4 *cand, const uint16_t *hidx, const int (&nh_)[NAT],
                                                const bool (&act)[NAT], const float4 (&pa)[NAT], const int64_t (&ia)[NAT],
                                                int64_t lo, int row_cap, float rca2, uint32_t cls_mask, uint32_t *meta,
                                                float4 *ent, uint32_t *status)
{
    const int lane = lane_id();
    constexpr int CAP = NCH * WAVE;   // hits this instantiation handles per atom (<= MAXR)
    // ---- classify {r <= Rca, r > Rca} x species over the species that occur in the stencil.  Keys of the NCH chunks
    // of hits stay in registers ----
    int nh[NAT];
    bool over[NAT];
    int key[NAT][NCH];
    float4 dv[NAT][NCH];
#pragma unroll
    for (int t = 0; t < NAT; ++t) {
        over[t] = nh_[t] > CAP;
        nh[t] = act[t] ? (over[t] ? CAP : nh_[t]) : 0;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            key[t][ch] = 16;
            dv[t][ch] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ch * WAVE < nh[t]) {   // (wave-uniform)
                const int e = ch * WAVE + lane;
                const bool v = e < nh[t];
                const float4 q = cand[v ? hidx[t * MAXR + e] : 0];
                const float dx = q.x - pa[t].x, dy = q.y - pa[t].y, dz = q.z - pa[t].z;
                dv[t][ch] = make_float4(dx, dy, dz, q.w);
                key[t][ch] = v ? (((dx * dx + dy * dy + dz * dz <= rca2) ? 0 : 8) + (int)(__float_as_uint(q.w) >> 28)) : 16;
            }
        }
    }
    // ---- ONE pass over the classes in row order (angular group by species, then the far group by species): the position
    // of a hit is the number of hits of earlier classes + its rank inside its class; the class sizes go to packed byte
    // counters in scalar registers ----
    uint64_t pk[NAT][2];
    int run[NAT], pos[NAT][NCH];
    bool big[NAT];   // a class with more than 255 entries
#pragma unroll
    for (int t = 0; t < NAT; ++t) {
        pk[t][0] = 0ull; pk[t][1] = 0ull;
        run[t] = 0;
        big[t] = false;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) pos[t][ch] = -1;
    }
    for (uint32_t pm = cls_mask; pm; pm &= pm - 1) {
        const int k = __builtin_ctz(pm);
#pragma unroll
        for (int t = 0; t < NAT; ++t) {
            int pop = 0;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
                if (ch * WAVE < nh[t]) {
                    const uint64_t m = __ballot(key[t][ch] == k);
                    if (key[t][ch] == k) pos[t][ch] = run[t] + pop + mbcnt(m);
                    pop += __popcll(m);
                }
            run[t] += pop;
            big[t] = big[t] || pop > 255;
            pk[t][k >> 3] += (uint64_t)(pop & 255) << (8 * (k & 7));
        }
    }
#pragma unroll
    for (int t = 0; t < NAT; ++t) {
        if (!act[t]) continue;
        uint32_t *meta_i = meta + (size_t)ia[t] * META_W;
        const size_t row0 = (size_t)(ia[t] - lo) * row_cap;
        float4 *row = ent + row0;
        const int nA = (int)((pk[t][0] * 0x0101010101010101ull) >> 56), nF = (int)((pk[t][1] * 0x0101010101010101ull) >> 56);
        const bool bad = over[t] || big[t] || nA + nF > row_cap || nA > MAXA;
        if (bad && lane == 0) atomicOr(&status[0], ANIHIP_ST_ROW_OVERFLOW);
        if (!bad) {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
                if (pos[t][ch] >= 0) row[pos[t][ch]] = dv[t][ch];
        }
        if (lane == 0) {   // (24-byte rows: three 8-byte stores)
            uint2 *m2 = reinterpret_cast<uint2 *>(meta_i);
            m2[0] = make_uint2((uint32_t)row0, bad ? 0u : ((uint32_t)nA | ((uint32_t)nF << 16)));
            m2[1] = bad ? make_uint2(0u, 0u) : make_uint2((uint32_t)pk[t][0], (uint32_t)(pk[t][0] >> 32));
            m2[2] = bad ? make_uint2(0u, 0u) : make_uint2((uint32_t)pk[t][1], (uint32_t)(pk[t][1] >> 32));
        }
    }
}

// ---- cell mode, cell-centric: one wave per BIN ---------------------------------------------------------------------
// All atoms of a bin see the same 27-bin stencil: the wave resolves it once, stages the candidates -- position already
// shifted by the bin's image -- in LDS (CAND_CAP entries), and every central atom of the bin then sweeps that list from
// LDS: no per-candidate binary search, one global read of a candidate per BIN instead of per central atom.  Hits are
// kept as 16-bit positions in the staged list; the row is classified {angular, far} x species over the classes that
// actually occur (packed byte counters in scalar registers) and written in the same order as k_nbr_cell writes it.
// Bins with more candidates than the stage holds, or stencils of more than 64 bins (cells thinner than the cutoff),
// are left to k_nbr_cell (handled[c] = 0, counted in n_unhandled).
// ONE workgroup of 16 waves per CU whose waves take the workgroup's bins (b, b + blocks, ... in groups of 16) from an LDS
// counter: with a fixed share per wave the waves of a SIMD finish far apart (the issue arbiter favours the oldest) and
// the tail of the kernel runs on a few waves per SIMD (as in the AEV kernels, aev.hip AtomQueue).
constexpr int NBR2_WPB = 16;
constexpr int CAND_CAP = 448;   // (7 KB + 1 KB of hit positions, which the staging tables share: 8 KB of LDS per wave)
#ifdef ANIHIP_TRACE
// development: per-phase shader-clock sums of wave 0 of every block (s_memtime stamps), read by anihip_dev_nbr_trace_read
__device__ unsigned long long g_nbr_trace[1024][10];
#define NTR_STAMP(k_)                                                      \
    {                                                                      \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();      \
        if (wib == 0) tr_sum[k_] += now_ - tr_last;                        \
        tr_last = now_;                                                    \
    }
#else
#define NTR_STAMP(k_)
#endif

__device__ __forceinline__ uint32_t wave_or(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v |= (uint32_t)__shfl_xor((int)v, o);
    return v;
}

__global__ __launch_bounds__(NBR2_WPB * WAVE, 4) void k_nbr_cell2(
    const GridDesc *g, int S, float rcr2, float rca2, int64_t lo, int64_t hi, const int *cellid,
    const int *cell_start, const float4 *pos4s, int row_cap, uint32_t *meta, float4 *ent, uint32_t *status,
    int *handled, int *n_unhandled)
{
    __shared__ float4 s_cand[NBR2_WPB][CAND_CAP];
    __shared__ uint16_t s_hidx[NBR2_WPB][2 * MAXR];   // hit positions of the two central atoms of a sweep; while a bin is
                                                      // being staged: the prefix sums and first atoms of its stencil bins
    static_assert(2 * MAXR * sizeof(uint16_t) >= 2 * WAVE * sizeof(int), "staging tables share the hit list's LDS");
    const int wib = threadIdx.x >> 6, lane = lane_id();
    float4 *cand = s_cand[wib];
    uint16_t *hidx = s_hidx[wib];
    int *pend = reinterpret_cast<int *>(s_hidx[wib]), *k0t = pend + WAVE;
    // padding atoms are in no bin: empty rows
    for (int64_t i = lo + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < hi; i += (int64_t)gridDim.x * blockDim.x)
        if (cellid[i] < 0) {
            uint32_t *m = meta + (size_t)i * META_W;
            m[0] = (uint32_t)((size_t)(i - lo) * row_cap);
            m[1] = 0; m[2] = 0; m[3] = 0; m[4] = 0; m[5] = 0;
        }
    const int nb0 = g->nb[0], nb1 = g->nb[1], nb2 = g->nb[2];
    const int R0 = g->range[0], R1 = g->range[1], R2 = g->range[2];
    const int p0 = g->pbc[0], p1 = g->pbc[1], p2 = g->pbc[2];
    float st[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) st[q] = g->step[q];
    const int n1 = 2 * R1 + 1, n2 = 2 * R2 + 1, ncells = (2 * R0 + 1) * n1 * n2;
    const float inv_n2 = 1.0f / (float)n2, inv_n1 = 1.0f / (float)n1;
    const int nbins = nb0 * nb1 * nb2;
    const int nw = gridDim.x * NBR2_WPB;
    // position p of the workgroup's bin list = bin (p / 16) * all waves + 16 * workgroup + p % 16; the first one is the wave's own
    __shared__ uint32_t s_queue;
    if (threadIdx.x == 0) s_queue = NBR2_WPB;
    __syncthreads();
    auto bin_at = [&](uint32_t p) { return (int)(p / NBR2_WPB) * nw + (int)blockIdx.x * NBR2_WPB + (int)(p % NBR2_WPB); };
    uint32_t qv = 0u;   // lane 0: the position the request in flight returns
#ifdef ANIHIP_TRACE
    unsigned long long tr_sum[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tr_last = __builtin_amdgcn_s_memtime();
#endif
    for (int c = bin_at(wib); c < nbins; c = bin_at((uint32_t)__builtin_amdgcn_readfirstlane((int)qv))) {
        if (lane == 0) qv = atomicAdd(&s_queue, 1u);   // (the bin after this one: granted long before the bin is done)
        NTR_STAMP(0)   // loop tail
        const int ab = cell_start[c], ae = cell_start[c + 1];
        if (ae == ab) continue;
#ifdef ANIHIP_TRACE
        if (wib == 0) tr_sum[8] += 1;
#endif
        // does this rank own any atom of the bin?
        bool mine = false;
        for (int a0 = ab; a0 < ae; a0 += WAVE) {
            const int a = a0 + lane;
            const int64_t ia = a < ae ? (int64_t)(__float_as_uint(pos4s[a].w) & IDX_MASK) : -1;
            mine = mine || __ballot(ia >= lo && ia < hi) != 0ull;
        }
        if (!mine) continue;
        // ---- lane = stencil bin (as in k_nbr_cell) ----
        const int b2 = c % nb2, b1 = (c / nb2) % nb1, b0 = c / (nb2 * nb1);
        int total = 0, self_base = 0;
        if (ncells <= WAVE) {
            const int ci = lane;
            const int q2 = (int)(((float)ci + 0.5f) * inv_n2);
            const int q1 = (int)(((float)q2 + 0.5f) * inv_n1);
            const int o2 = ci - q2 * n2 - R2, o1 = q2 - q1 * n1 - R1, o0 = q1 - R0;
            int c0 = b0 + o0, c1 = b1 + o1, c2 = b2 + o2;
            bool ok = ci < ncells;
            if (p0) c0 = wrap_bin(c0, nb0); else ok = ok && c0 >= 0 && c0 < nb0;
            if (p1) c1 = wrap_bin(c1, nb1); else ok = ok && c1 >= 0 && c1 < nb1;
            if (p2) c2 = wrap_bin(c2, nb2); else ok = ok && c2 >= 0 && c2 < nb2;
            const int cw = ok ? (c0 * nb1 + c1) * nb2 + c2 : 0;
            const int kbeg = cell_start[cw], kend = cell_start[cw + 1];
            const int cnt = ok ? kend - kbeg : 0;
            int incl = cnt;
#pragma unroll
            for (int d = 1; d < WAVE; d <<= 1) {
                const int up = __shfl_up(incl, d);
                incl += lane >= d ? up : 0;
            }
            total = __builtin_amdgcn_readlane(incl, WAVE - 1);
            pend[lane] = incl;
            k0t[lane] = kbeg - (incl - cnt);
            // flat position of the central bin's own atoms (the (0,0,0) entry of the stencil)
            const bool central = ok && o0 == 0 && o1 == 0 && o2 == 0;
            const uint64_t cm = __ballot(central);
            self_base = __builtin_amdgcn_readlane(incl - cnt, (int)__builtin_ctzll(cm | (1ull << 63)));
        }
        const bool fits = ncells <= WAVE && total <= CAND_CAP;
        if (lane == 0) {
            handled[c] = fits ? 1 : 0;
            if (!fits) atomicAdd(n_unhandled, 1);
        }
        if (!fits) continue;
        wave_sync();
        NTR_STAMP(1)   // ownership test, stencil, prefix sum
        // ---- stage the candidates: position + image shift, w = packed (index | species) ----
        {
            // all global reads of the stage first (one per 64 candidates, CAND_CAP / 64 at most), then the LDS writes: the
            // reads of a bin are in flight together instead of one round trip per 64 candidates
            constexpr int NST = CAND_CAP / WAVE;
            float4 qs[NST];
#pragma unroll
            for (int it = 0; it < NST; ++it) {
                const int x = it * WAVE + lane;
                qs[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (it * WAVE < total) {   // (wave-uniform)
                    const int xc = x < total ? x : total - 1;
                    int seg = 0;
#pragma unroll
                    for (int stp = WAVE / 2; stp > 0; stp >>= 1) seg += pend[seg + stp - 1] <= xc ? stp : 0;
                    seg = seg < WAVE ? seg : WAVE - 1;
                    const float4 q = pos4s[xc + k0t[seg]];
                    // image shift of stencil bin `seg` (same evaluation order as k_nbr_cell: (o0 s0 + o1 s3) first, o2 s6
                    // added to the sum)
                    const int q2 = (int)(((float)seg + 0.5f) * inv_n2);
                    const int q1 = (int)(((float)q2 + 0.5f) * inv_n1);
                    const int o2 = seg - q2 * n2 - R2, o1 = q2 - q1 * n1 - R1, o0 = q1 - R0;
                    qs[it] = make_float4(q.x + (o0 * st[0] + o1 * st[3] + o2 * st[6]), q.y + (o0 * st[1] + o1 * st[4] + o2 * st[7]),
                                         q.z + (o0 * st[2] + o1 * st[5] + o2 * st[8]), q.w);
                }
            }
#pragma unroll
            for (int it = 0; it < NST; ++it)
                if (it * WAVE + lane < total) cand[it * WAVE + lane] = qs[it];
        }
        // classes a row of this bin can hold: {angular, far} x the species present in the stencil (bits g * 8 + species)
        uint32_t cls_mask = 0u;
        wave_sync();   // (also: the staging tables are dead, the sweeps may write hit positions over them)
        for (int x0 = 0; x0 < total; x0 += WAVE)
            cls_mask |= (x0 + lane < total) ? 1u << (__float_as_uint(cand[x0 + lane].w) >> 28) : 0u;
        cls_mask = wave_or(cls_mask);
        cls_mask = __builtin_amdgcn_readfirstlane(cls_mask | (cls_mask << 8));
        NTR_STAMP(2)   // staging + class mask
        // ---- every owned atom of the bin sweeps the staged list ----
        for (int a = ab; a < ae; a += 2) {   // two central atoms per sweep: one LDS read of a candidate serves both
            const int xa = self_base + (a - ab), xb = xa + 1;
            const float4 pa = cand[xa];                              // the atoms themselves (zero shift)
            const float4 pb = cand[a + 1 < ae ? xb : xa];
            // (the same LDS word in every lane: through a scalar register, so that everything derived from the atom index --
            // the ownership tests, the row addresses, the class counters of the rows -- is scalar work)
            const int64_t ia = (int64_t)((uint32_t)uniform((int)__float_as_uint(pa.w)) & IDX_MASK);
            const int64_t ib = (int64_t)((uint32_t)uniform((int)__float_as_uint(pb.w)) & IDX_MASK);
            const bool da = ia >= lo && ia < hi, db = a + 1 < ae && ib >= lo && ib < hi;
            if (!da && !db) continue;
            int nha = 0, nhb = 0;
            // (branch-free: the predicates are combined as masks, the next 64 candidates are read ahead of this step's work)
            float4 q = cand[lane < total ? lane : 0];
            for (int x0 = 0; x0 < total; x0 += WAVE) {
                const int x = x0 + lane, xn = x + WAVE;
                const float4 qn = cand[xn < total ? xn : 0];
                const float ax = q.x - pa.x, ay = q.y - pa.y, az = q.z - pa.z;
                const float bx = q.x - pb.x, by = q.y - pb.y, bz = q.z - pb.z;
                const float d2a = ax * ax + ay * ay + az * az, d2b = bx * bx + by * by + bz * bz;
                const bool inr = x < total;
                const bool hita = (int)da & (int)inr & (int)(d2a <= rcr2) & (int)(x != xa);
                const bool hitb = (int)db & (int)inr & (int)(d2b <= rcr2) & (int)(x != xb);
                const uint64_t ma = __ballot(hita), mb = __ballot(hitb);
                const int posa = nha + mbcnt(ma), posb = nhb + mbcnt(mb);
                if ((int)hita & (int)(posa < MAXR)) hidx[posa] = (uint16_t)x;
                if ((int)hitb & (int)(posb < MAXR)) hidx[MAXR + posb] = (uint16_t)x;
                nha += __popcll(ma);
                nhb += __popcll(mb);
                q = qn;
            }
            wave_sync();
            NTR_STAMP(3)   // sweeps
            {
                const int nh2[2] = {nha, nhb};
                const bool act2[2] = {da, db};
                const float4 p2[2] = {pa, pb};
                const int64_t i2[2] = {ia, ib};
                if (max(nha, nhb) <= 2 * WAVE) {   // (nearly always: both atoms together, two chunks of hits each)
                    emit_from_stage<2, 2>(cand, hidx, nh2, act2, p2, i2, lo, row_cap, rca2, cls_mask, meta, ent, status);
                } else {
                    {
                        const int nh1[1] = {nha};
                        const bool act1[1] = {da};
                        const float4 p1[1] = {pa};
                        const int64_t i1[1] = {ia};
                        emit_from_stage<1, MAXR / WAVE>(cand, hidx, nh1, act1, p1, i1, lo, row_cap, rca2, cls_mask, meta, ent,
                                                        status);
                    }
                    {
                        const int nh1[1] = {nhb};
                        const bool act1[1] = {db};
                        const float4 p1[1] = {pb};
                        const int64_t i1[1] = {ib};
                        emit_from_stage<1, MAXR / WAVE>(cand, hidx + MAXR, nh1, act1, p1, i1, lo, row_cap, rca2, cls_mask, meta,
                                                        ent, status);
                    }
                }
            }
            wave_sync();
            NTR_STAMP(4)   // rows
        }
        wave_sync();
    }
#ifdef ANIHIP_TRACE
    if (wib == 0 && lane == 0 && blockIdx.x < 1024)
        for (int q_ = 0; q_ < 10; ++q_) g_nbr_trace[blockIdx.x][q_] = tr_sum[q_];
#endif
}

// ---- rows from an external FULL neighbor list (LAMMPS style: local + ghost atoms with explicit coordinates) ----
// one wave per listed atom gi: i = ilist[gi], neighbors jlist[start[gi] .. start[gi] + numneigh[gi]),
// displacement r_j - r_i straight from the coordinates, cutoff screen, sorted row (postProcessNbrList2,
// csrc/aev.cu:1048-1126)
__global__ __launch_bounds__(NBR_WPB * WAVE) void k_nbr_full(int S, float rcr2, float rca2, int64_t n_atoms,
                                                            const int32_t *species, const float *coords,
                                                            int64_t n_i, const int32_t *ilist,
                                                            const int32_t *numneigh, const int64_t *start,
                                                            const int32_t *jlist, int row_cap, uint32_t *meta,
                                                            float4 *ent, uint32_t *status)
{
    __shared__ float4 s_hits[NBR_WPB][MAXR];
    const int wib = threadIdx.x >> 6, lane = lane_id();
    const int64_t nw = (int64_t)gridDim.x * NBR_WPB;
    for (int64_t gi = blockIdx.x * (int64_t)NBR_WPB + wib; gi < n_i; gi += nw) {
        const int64_t i = ilist[gi];
        if (i < 0 || i >= n_atoms) {
            if (lane == 0) atomicOr(&status[0], ANIHIP_ST_ENTRY_OVERFLOW);
            continue;
        }
        if (species[i] < 0) continue;   // (meta rows were zeroed)
        uint32_t *meta_i = meta + (size_t)i * META_W;
        const size_t row0 = (size_t)i * row_cap;
        if (lane == 0) meta_i[0] = (uint32_t)row0;
        const float xi = coords[3 * i], yi = coords[3 * i + 1], zi = coords[3 * i + 2];
        const int jnum = numneigh[gi];
        const int64_t st = start[gi];
        HitList h{s_hits[wib], 0, false};
        for (int j0 = 0; j0 < jnum; j0 += WAVE) {
            const int jj = j0 + lane;
            const bool v = jj < jnum;
            const int64_t j = v ? (int64_t)jlist[st + jj] : i;
            const bool okj = v && j >= 0 && j < n_atoms && j != i;
            const int sj = okj ? species[j] : -1;
            const int64_t jc = okj ? j : i;
            const float dx = coords[3 * jc] - xi, dy = coords[3 * jc + 1] - yi, dz = coords[3 * jc + 2] - zi;
            const bool hit = okj && sj >= 0 && dx * dx + dy * dy + dz * dz <= rcr2;
            push_hits(h, hit, dx, dy, dz, __uint_as_float(((uint32_t)jc & IDX_MASK) | ((uint32_t)(sj < 0 ? 0 : sj) << 28)));
        }
        emit_row(h, S, rca2, row_cap, meta_i, ent + row0, status);
        wave_sync();
    }
}

// ---- Verlet-skin reuse -----------------------------------------------------------------------------------
// Rows built once with an enlarged radial cutoff (Rcr + skin) stay a superset of every atom's neighborhood while
// no atom has moved more than skin / 2 (VerletCellList, neighbors.py:759-884).  One wave per central atom walks its
// stored row, updates each displacement with the motion since the build,
//     d(t) = d(t0) + (x_j(t) - x_j(t0)) - (x_i(t) - x_i(t0))
// (image shifts are constants of the stored row, so unwrapped coordinates need no cell here), screens it against the
// real cutoff and emits a row in the standard order -- the narrow_down step of the reference (neighbors.py:64-113)
// without the pair search.
__global__ __launch_bounds__(NBR_WPB * WAVE) void k_nbr_refresh(int S, float rcr2, float rca2, int64_t lo, int64_t hi,
                                                               const int32_t *species, const float *coords,
                                                               const float *coords0, const uint32_t *vmeta,
                                                               const float4 *vent, int row_cap, uint32_t *meta,
                                                               float4 *ent, uint32_t *status)
{
    __shared__ float4 s_hits[NBR_WPB][MAXR];
    const int wib = threadIdx.x >> 6, lane = lane_id();
    const int64_t nw = (int64_t)gridDim.x * NBR_WPB;
    for (int64_t i = lo + blockIdx.x * (int64_t)NBR_WPB + wib; i < hi; i += nw) {
        uint32_t *meta_i = meta + (size_t)i * META_W;
        const size_t row0 = (size_t)(i - lo) * row_cap;
        if (lane == 0) meta_i[0] = (uint32_t)row0;
        const uint32_t *vm = vmeta + (size_t)i * META_W;
        const int cnt = (int)(vm[1] & 0xFFFFu) + (int)(vm[1] >> 16);
        if (species[i] < 0 || cnt == 0) {
            if (lane == 0) { meta_i[1] = 0; meta_i[2] = 0; meta_i[3] = 0; meta_i[4] = 0; meta_i[5] = 0; }
            continue;
        }
        const float mx = coords[3 * i] - coords0[3 * i], my = coords[3 * i + 1] - coords0[3 * i + 1],
                    mz = coords[3 * i + 2] - coords0[3 * i + 2];
        const float4 *row = vent + vm[0];
        HitList h{s_hits[wib], 0, false};
        for (int e0 = 0; e0 < cnt; e0 += WAVE) {
            const int e = e0 + lane;
            const bool v = e < cnt;
            const float4 d0 = v ? row[e] : make_float4(0.f, 0.f, 0.f, 0.f);
            const int64_t j = v ? (int64_t)(__float_as_uint(d0.w) & IDX_MASK) : i;
            const float dx = d0.x + (coords[3 * j] - coords0[3 * j]) - mx;
            const float dy = d0.y + (coords[3 * j + 1] - coords0[3 * j + 1]) - my;
            const float dz = d0.z + (coords[3 * j + 2] - coords0[3 * j + 2]) - mz;
            push_hits(h, v && dx * dx + dy * dy + dz * dz <= rcr2, dx, dy, dz, d0.w);
        }
        emit_row(h, S, rca2, row_cap, meta_i, ent + row0, status);
        wave_sync();
    }
}

// ---- rows from an external half neighbor list ---------------------------------------------------------
// pass 1: every pair (a, b) with diff = r_a - r_b (+ image shift) appends {r_b - r_a, b} to row a and
// {r_a - r_b, a} to row b (slot = atomic counter of the row; rows outside [lo, hi) are skipped)
__global__ __launch_bounds__(256) void k_half_scatter(int64_t n_pairs, const int64_t *idx, const float *diff,
                                                     const int32_t *species, int64_t n_atoms, float rcr2,
                                                     int64_t lo, int64_t hi, int row_cap, int *count,
                                                     float4 *ent, uint32_t *status)
{
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n_pairs;
         p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t a = idx[p], b = idx[n_pairs + p];
        if (a < 0 || b < 0 || a >= n_atoms || b >= n_atoms) {
            atomicOr(&status[0], ANIHIP_ST_ENTRY_OVERFLOW);
            continue;
        }
        const int sa = species[a], sb = species[b];
        if (sa < 0 || sb < 0) continue;   // dummy atoms (neighbors.py:72-83)
        const float dx = diff[3 * p], dy = diff[3 * p + 1], dz = diff[3 * p + 2];
        if (dx * dx + dy * dy + dz * dz > rcr2) continue;
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const int64_t c = side ? b : a, o = side ? a : b;
            const int so = side ? sa : sb;
            const float sg = side ? 1.0f : -1.0f;
            if (c < lo || c >= hi) continue;
            const int slot = atomicAdd(&count[c - lo], 1);
            if (slot < row_cap)
                ent[(size_t)(c - lo) * row_cap + slot] =
                    make_float4(sg * dx, sg * dy, sg * dz, __uint_as_float(((uint32_t)o & IDX_MASK) | ((uint32_t)so << 28)));
        }
    }
}

// pass 2: one wave per central atom loads its unsorted row, puts it into a canonical order (by neighbor
// index, then displacement: the atomic slots of pass 1 are not reproducible) and emits the sorted row
__global__ __launch_bounds__(NBR_WPB * WAVE) void k_half_finish(int S, float rca2, int64_t lo, int64_t hi,
                                                               const int32_t *species, int row_cap,
                                                               const int *count, uint32_t *meta, float4 *ent,
                                                               uint32_t *status)
{
    __shared__ float4 s_raw[NBR_WPB][MAXR];
    __shared__ float4 s_hits[NBR_WPB][MAXR];
    const int wib = threadIdx.x >> 6, lane = lane_id();
    const int64_t nw = (int64_t)gridDim.x * NBR_WPB;
    for (int64_t i = lo + blockIdx.x * (int64_t)NBR_WPB + wib; i < hi; i += nw) {
        uint32_t *meta_i = meta + (size_t)i * META_W;
        const size_t row0 = (size_t)(i - lo) * row_cap;
        if (lane == 0) meta_i[0] = (uint32_t)row0;
        const int cnt = count[i - lo];
        if (species[i] < 0 || cnt == 0) {
            if (lane == 0) { meta_i[1] = 0; meta_i[2] = 0; meta_i[3] = 0; meta_i[4] = 0; meta_i[5] = 0; }
            continue;
        }
        HitList h{s_hits[wib], min(cnt, min(row_cap, MAXR)), cnt > row_cap || cnt > MAXR};
        float4 *raw = s_raw[wib];
        for (int e = lane; e < h.n; e += WAVE) raw[e] = ent[row0 + e];
        wave_sync();
        auto less = [](const float4 &x, const float4 &y) {
            const uint32_t jx = __float_as_uint(x.w) & IDX_MASK, jy = __float_as_uint(y.w) & IDX_MASK;
            if (jx != jy) return jx < jy;
            if (x.x != y.x) return x.x < y.x;
            if (x.y != y.y) return x.y < y.y;
            return x.z < y.z;
        };
        for (int e = lane; e < h.n; e += WAVE) {   // rank sort (rows are short; this path is not hot)
            const float4 me = raw[e];
            int rank = 0;
            for (int o = 0; o < h.n; ++o) {
                const float4 ot = raw[o];
                rank += (less(ot, me) || (!less(me, ot) && o < e)) ? 1 : 0;
            }
            h.buf[rank] = me;
        }
        emit_row(h, S, rca2, row_cap, meta_i, ent + row0, status);
        wave_sync();
    }
}

}  // namespace anihip

using namespace anihip;

extern "C" size_t anihip_nbr_workspace_bytes(int64_t n_atoms, int64_t max_cells)
{
    return carve(nullptr, nullptr, n_atoms, max_cells < 1 ? 1 : max_cells);
}

static int nbr_grid_blocks(int64_t n_central)
{
    int64_t b = (n_central + NBR_WPB - 1) / NBR_WPB;
    if (b < 1) b = 1;
    if (b > 256 * 16) b = 256 * 16;  // persistent: each wave strides over central atoms
    return (int)b;
}

extern "C" int anihip_nbr_build_batch(void *stream_, const anihip_aev_params *p, int32_t n_mol,
                                      int32_t A, const int32_t *species, const float *coords,
                                      const float *cell, int32_t pbc_mask, int64_t lo, int64_t hi,
                                      void *workspace, size_t workspace_bytes, uint32_t *meta, float *ent,
                                      int64_t ent_capacity, uint32_t *status)
{
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t n = (int64_t)n_mol * A;
    ANIHIP_REQUIRE(p && species && coords && workspace && meta && ent && status, "null pointer argument");
    ANIHIP_REQUIRE(p->num_species >= 1 && p->num_species <= MAX_S - 1, "num_species must be 1..7");
    ANIHIP_REQUIRE(0 <= lo && lo <= hi && hi <= n, "central range [%lld,%lld) outside 0..%lld",
                   (long long)lo, (long long)hi, (long long)n);
    ANIHIP_REQUIRE(n < (int64_t)IDX_MASK, "too many atoms for 28-bit neighbor indices");
    ANIHIP_REQUIRE(workspace_bytes >= anihip_nbr_workspace_bytes(n, 1), "workspace too small");
    if (hi == lo) return 0;
    const int64_t row_cap = ent_capacity / (hi - lo);
    ANIHIP_REQUIRE(row_cap >= 1 && (hi - lo) * row_cap < ((int64_t)1 << 32), "bad ent_capacity");
    NbrWorkspace w;
    carve(&w, (char *)workspace, n, 1);
    hipLaunchKernelGGL(k_prep_batch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, w.desc, n,
                       species, coords, w.pos4, cell, pbc_mask, p->Rcr, status);
    hipLaunchKernelGGL(k_nbr_batch, dim3(nbr_grid_blocks(hi - lo)), dim3(NBR_WPB * WAVE), 0, stream,
                       w.desc, p->num_species, p->Rcr * p->Rcr, p->Rca * p->Rca, (int)A, lo, hi, w.pos4,
                       (int)(row_cap > MAXR ? MAXR : row_cap), meta, (float4 *)ent, status);
    ANIHIP_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int anihip_nbr_build_cell(void *stream_, const anihip_aev_params *p, int64_t n,
                                     const int32_t *species, const float *coords, const float *cell,
                                     int32_t pbc_mask, int64_t lo, int64_t hi, int64_t max_cells,
                                     void *workspace, size_t workspace_bytes, uint32_t *meta, float *ent,
                                     int64_t ent_capacity, uint32_t *status)
{
    hipStream_t stream = (hipStream_t)stream_;
    ANIHIP_REQUIRE(p && species && coords && workspace && meta && ent && status, "null pointer argument");
    ANIHIP_REQUIRE(p->num_species >= 1 && p->num_species <= MAX_S - 1, "num_species must be 1..7");
    ANIHIP_REQUIRE(0 <= lo && lo <= hi && hi <= n, "central range outside 0..n");
    ANIHIP_REQUIRE(n < (int64_t)IDX_MASK, "too many atoms for 28-bit neighbor indices");
    ANIHIP_REQUIRE(max_cells >= 1 && max_cells < ((int64_t)1 << 30), "bad max_cells");
    ANIHIP_REQUIRE(workspace_bytes >= anihip_nbr_workspace_bytes(n, max_cells), "workspace too small");
    if (hi == lo) return 0;
    const int64_t row_cap = ent_capacity / (hi - lo);
    ANIHIP_REQUIRE(row_cap >= 1 && (hi - lo) * row_cap < ((int64_t)1 << 32), "bad ent_capacity");
    NbrWorkspace w;
    carve(&w, (char *)workspace, n, max_cells);
    const unsigned nblk = (unsigned)((n + 255) / 256);
    const unsigned cblk = (unsigned)((max_cells + 1 + 1023) / 1024);
    hipLaunchKernelGGL(k_desc_init, dim3(1), dim3(64), 0, stream, w.desc, cell, pbc_mask, p->Rcr, status);
    const bool all_pbc = cell && ((pbc_mask & 7) == 7);
    if (!all_pbc)
        hipLaunchKernelGGL(k_bbox, dim3(nblk > 1024 ? 1024 : nblk), dim3(256), 0, stream, w.desc, n,
                           species, coords);
    hipLaunchKernelGGL(k_grid_setup, dim3(1), dim3(64), 0, stream, w.desc, p->Rcr, max_cells, status);
    zero_words_async(stream, w.cell_fill, sizeof(int) * (size_t)(max_cells + 1));
    hipLaunchKernelGGL(k_bin_count, dim3(nblk), dim3(256), 0, stream, w.desc, n, species, coords, w.pos4,
                       w.cellid, w.cell_fill);
    hipLaunchKernelGGL(k_scan_local, dim3(cblk), dim3(1024), 0, stream, w.desc, w.cell_fill, w.cell_start,
                       w.scan_tmp);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, stream, w.desc, w.scan_tmp);
    hipLaunchKernelGGL(k_scan_add, dim3(cblk), dim3(1024), 0, stream, w.desc, w.cell_start, w.scan_tmp);
    zero_words_async(stream, w.cell_fill, sizeof(int) * (size_t)(max_cells + 1));
    hipLaunchKernelGGL(k_bin_fill, dim3(nblk), dim3(256), 0, stream, n, w.cellid, w.cell_start, w.cell_fill,
                       w.sorted_idx);
    hipLaunchKernelGGL(k_bin_rank_gather, dim3(nblk), dim3(256), 0, stream, n, w.cellid, w.cell_start, w.sorted_idx,
                       w.pos4, w.pos4s);
    // one wave per bin stages the stencil's candidates in LDS for all atoms of the bin; what it leaves (rare: very dense
    // bins, cells thinner than the cutoff) goes through the per-atom kernel.  cell_fill / scan_tmp are free by now.
    int *handled = w.cell_fill, *n_unhandled = w.scan_tmp;
    zero_words_async(stream, n_unhandled, sizeof(int));
    const int rc = (int)(row_cap > MAXR ? MAXR : row_cap);
    int64_t bins_blocks = (max_cells + NBR2_WPB - 1) / NBR2_WPB;
    if (bins_blocks > 256) bins_blocks = 256;   // persistent: one workgroup of 16 waves per CU
    hipLaunchKernelGGL(k_nbr_cell2, dim3((unsigned)bins_blocks), dim3(NBR2_WPB * WAVE), 0, stream, w.desc,
                       p->num_species, p->Rcr * p->Rcr, p->Rca * p->Rca, lo, hi, w.cellid, w.cell_start, w.pos4s, rc,
                       meta, (float4 *)ent, status, handled, n_unhandled);
    hipLaunchKernelGGL(k_nbr_cell, dim3(nbr_grid_blocks(hi - lo)), dim3(NBR_WPB * WAVE), 0, stream, w.desc,
                       p->num_species, p->Rcr * p->Rcr, p->Rca * p->Rca, lo, hi, w.pos4, w.cellid,
                       w.cell_start, w.pos4s, rc, meta, (float4 *)ent, status, handled, n_unhandled);
    ANIHIP_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" size_t anihip_nbr_half_workspace_bytes(int64_t n_central)
{
    return align256(sizeof(int) * (size_t)(n_central > 0 ? n_central : 1));
}

extern "C" int anihip_nbr_from_half(void *stream_, const anihip_aev_params *p, int64_t n, const int32_t *species,
                                    int64_t n_pairs, const int64_t *idx, const float *diff, int64_t lo,
                                    int64_t hi, void *workspace, size_t workspace_bytes, uint32_t *meta,
                                    float *ent, int64_t ent_capacity, uint32_t *status)
{
    hipStream_t stream = (hipStream_t)stream_;
    ANIHIP_REQUIRE(p && species && workspace && meta && ent && status, "null pointer argument");
    ANIHIP_REQUIRE(n_pairs == 0 || (idx && diff), "null neighbor list");
    ANIHIP_REQUIRE(p->num_species >= 1 && p->num_species <= MAX_S - 1, "num_species must be 1..7");
    ANIHIP_REQUIRE(0 <= lo && lo <= hi && hi <= n, "central range outside 0..n");
    ANIHIP_REQUIRE(n < (int64_t)IDX_MASK, "too many atoms for 28-bit neighbor indices");
    ANIHIP_REQUIRE(n_pairs >= 0, "negative pair count");
    ANIHIP_REQUIRE(workspace_bytes >= anihip_nbr_half_workspace_bytes(hi - lo), "workspace too small");
    if (hi == lo) return 0;
    const int64_t row_cap = ent_capacity / (hi - lo);
    ANIHIP_REQUIRE(row_cap >= 1 && (hi - lo) * row_cap < ((int64_t)1 << 32), "bad ent_capacity");
    const int cap = (int)(row_cap > MAXR ? MAXR : row_cap);
    int *count = (int *)workspace;
    zero_words_async(stream, count, sizeof(int) * (size_t)(hi - lo));
    if (n_pairs > 0) {
        int64_t blocks = (n_pairs + 255) / 256;
        if (blocks > 256 * 32) blocks = 256 * 32;
        hipLaunchKernelGGL(k_half_scatter, dim3((unsigned)blocks), dim3(256), 0, stream, n_pairs, idx, diff, species,
                           n, p->Rcr * p->Rcr, lo, hi, cap, count, (float4 *)ent, status);
    }
    hipLaunchKernelGGL(k_half_finish, dim3(nbr_grid_blocks(hi - lo)), dim3(NBR_WPB * WAVE), 0, stream,
                       p->num_species, p->Rca * p->Rca, lo, hi, species, cap, count, meta, (float4 *)ent, status);
    ANIHIP_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---- rows -> the reference's half list ------------------------------------------------------------------
// cell_list::cell_list returns (idx [2, P] i64, dist [P], diff [P, 3]) with every pair once (csrc/cell_list.cpp:342-354,
// consumed by neighbors.py:285-294).  A full row holds a pair twice -- {r_j - r_i, j} in row i and {r_i - r_j, i} in row
// j -- so row i emits its entries with j > i, and of an atom's own periodic images (j == i, both signs in the same row)
// the displacement whose first non-zero component is positive; diff = r_i - r_j (+ image shift) = minus the entry, the
// reference's sign (neighbors.py:105-112).  Pass 1 counts per atom, a scan turns counts into offsets (per-atom order =
// row order: deterministic), pass 2 writes.
__device__ __forceinline__ bool half_keep(int64_t i, const float4 e)
{
    const int64_t j = (int64_t)(__float_as_uint(e.w) & IDX_MASK);
    if (j != i) return j > i;
    return e.x > 0.f || (e.x == 0.f && (e.y > 0.f || (e.y == 0.f && e.z > 0.f)));
}

__global__ __launch_bounds__(256) void k_half_count(int64_t lo, int64_t hi, const uint32_t *meta, const float4 *ent,
                                                   int64_t *counts)
{
    for (int64_t i = lo + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < hi; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t *m = meta + (size_t)i * META_W;
        const int n = (int)(m[1] & 0xFFFFu) + (int)(m[1] >> 16);
        int c = 0;
        for (int k = 0; k < n; ++k) c += half_keep(i, ent[m[0] + k]) ? 1 : 0;
        counts[i - lo] = c;
    }
}

// exclusive scan of n int64 values in place, n_pairs = total: blocks of 1024 (sums -> one block scans the sums -> offsets)
__global__ __launch_bounds__(256) void k_scan_blocks(int64_t n, int64_t *v, int64_t *block_sums)
{
    __shared__ int64_t s_part[256];
    const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4;
    int64_t x[4], run = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        x[k] = base + k < n ? v[base + k] : 0;
        run += x[k];
    }
    s_part[threadIdx.x] = run;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const int64_t t = threadIdx.x >= o ? s_part[threadIdx.x - o] : 0;
        __syncthreads();
        s_part[threadIdx.x] += t;
        __syncthreads();
    }
    int64_t excl = s_part[threadIdx.x] - run;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (base + k < n) v[base + k] = excl;
        excl += x[k];
    }
    if (threadIdx.x == 255) block_sums[blockIdx.x] = s_part[255];
}
__global__ __launch_bounds__(256) void k_scan_sums(int64_t nb, int64_t *block_sums, int64_t *total)
{
    __shared__ int64_t s_part[256];
    __shared__ int64_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int64_t b0 = 0; b0 < nb; b0 += 256) {
        const int64_t i = b0 + threadIdx.x;
        const int64_t x = i < nb ? block_sums[i] : 0;
        s_part[threadIdx.x] = x;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const int64_t t = threadIdx.x >= o ? s_part[threadIdx.x - o] : 0;
            __syncthreads();
            s_part[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nb) block_sums[i] = s_carry + s_part[threadIdx.x] - x;
        __syncthreads();
        if (threadIdx.x == 255) s_carry += s_part[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = s_carry;
}
__global__ __launch_bounds__(256) void k_half_write(int64_t lo, int64_t hi, const uint32_t *meta, const float4 *ent,
                                                   const int64_t *offs, const int64_t *block_sums, int64_t capacity,
                                                   int64_t *idx, float *dist, float *diff)
{
    for (int64_t i = lo + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < hi; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t *m = meta + (size_t)i * META_W;
        const int n = (int)(m[1] & 0xFFFFu) + (int)(m[1] >> 16);
        int64_t p = offs[i - lo] + block_sums[(i - lo) >> 10];
        for (int k = 0; k < n; ++k) {
            const float4 e = ent[m[0] + k];
            if (!half_keep(i, e)) continue;
            if (p < capacity) {
                idx[p] = i;
                idx[capacity + p] = (int64_t)(__float_as_uint(e.w) & IDX_MASK);
                diff[3 * p] = -e.x;
                diff[3 * p + 1] = -e.y;
                diff[3 * p + 2] = -e.z;
                dist[p] = sqrtf(e.x * e.x + e.y * e.y + e.z * e.z);
            }
            ++p;
        }
    }
}

extern "C" size_t anihip_nbr_rows_to_half_workspace_bytes(int64_t n_central)
{
    const int64_t n = n_central > 0 ? n_central : 1;
    return (size_t)(8 * (n + (n + 1023) / 1024 + 8));
}

extern "C" int anihip_nbr_rows_to_half(void *stream_, int64_t n_atoms, int64_t lo, int64_t hi, const uint32_t *meta,
                                       const float *ent, void *workspace, size_t workspace_bytes, int64_t capacity,
                                       int64_t *idx, float *dist, float *diff, int64_t *n_pairs)
{
    hipStream_t stream = (hipStream_t)stream_;
    ANIHIP_REQUIRE(meta && ent && workspace && n_pairs, "null pointer argument");
    ANIHIP_REQUIRE(0 <= lo && lo <= hi && hi <= n_atoms, "central range outside 0..n_atoms");
    ANIHIP_REQUIRE(capacity >= 0 && (capacity == 0 || (idx && dist && diff)), "null output with capacity > 0");
    const int64_t n = hi - lo;
    ANIHIP_REQUIRE(workspace_bytes >= anihip_nbr_rows_to_half_workspace_bytes(n), "workspace too small");
    int64_t *counts = (int64_t *)workspace, *sums = counts + (n > 0 ? n : 1);
    if (n == 0) {
        zero_words_async(stream, n_pairs, 8);
        return 0;
    }
    const int64_t nb = (n + 1023) / 1024;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_half_count, dim3((unsigned)blocks), dim3(256), 0, stream, lo, hi, meta, (const float4 *)ent, counts);
    hipLaunchKernelGGL(k_scan_blocks, dim3((unsigned)nb), dim3(256), 0, stream, n, counts, sums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, stream, nb, sums, n_pairs);
    if (capacity > 0)
        hipLaunchKernelGGL(k_half_write, dim3((unsigned)blocks), dim3(256), 0, stream, lo, hi, meta, (const float4 *)ent,
                           counts, sums, capacity, idx, dist, diff);
    ANIHIP_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int anihip_nbr_refresh(void *stream_, const anihip_aev_params *p, int64_t n, int64_t lo, int64_t hi,
                                  const int32_t *species, const float *coords, const float *coords_build,
                                  const uint32_t *verlet_meta, const float *verlet_ent, uint32_t *meta, float *ent,
                                  int64_t ent_capacity, uint32_t *status)
{
    hipStream_t stream = (hipStream_t)stream_;
    ANIHIP_REQUIRE(p && species && coords && coords_build && verlet_meta && verlet_ent && meta && ent && status,
                   "null pointer argument");
    ANIHIP_REQUIRE(p->num_species >= 1 && p->num_species <= MAX_S - 1, "num_species must be 1..7");
    ANIHIP_REQUIRE(0 <= lo && lo <= hi && hi <= n, "central range outside 0..n_atoms");
    if (hi == lo) return 0;
    const int64_t row_cap = ent_capacity / (hi - lo);
    ANIHIP_REQUIRE(row_cap >= 1 && (hi - lo) * row_cap < ((int64_t)1 << 32), "bad ent_capacity");
    hipLaunchKernelGGL(k_nbr_refresh, dim3(nbr_grid_blocks(hi - lo)), dim3(NBR_WPB * WAVE), 0, stream,
                       p->num_species, p->Rcr * p->Rcr, p->Rca * p->Rca, lo, hi, species, coords, coords_build,
                       verlet_meta, (const float4 *)verlet_ent, (int)(row_cap > MAXR ? MAXR : row_cap), meta,
                       (float4 *)ent, status);
    ANIHIP_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int anihip_nbr_from_full(void *stream_, const anihip_aev_params *p, int64_t n, const int32_t *species,
                                    const float *coords, int64_t n_i, const int32_t *ilist,
                                    const int32_t *numneigh, const int64_t *start, const int32_t *jlist,
                                    uint32_t *meta, float *ent, int64_t ent_capacity, uint32_t *status)
{
    hipStream_t stream = (hipStream_t)stream_;
    ANIHIP_REQUIRE(p && species && coords && meta && ent && status, "null pointer argument");
    ANIHIP_REQUIRE(n_i == 0 || (ilist && numneigh && start && jlist), "null neighbor list");
    ANIHIP_REQUIRE(p->num_species >= 1 && p->num_species <= MAX_S - 1, "num_species must be 1..7");
    ANIHIP_REQUIRE(n >= 1 && n < (int64_t)IDX_MASK, "bad atom count");
    ANIHIP_REQUIRE(n_i >= 0 && n_i <= n, "more listed atoms than atoms");
    const int64_t row_cap = ent_capacity / n;
    ANIHIP_REQUIRE(row_cap >= 1 && n * row_cap < ((int64_t)1 << 32), "bad ent_capacity");
    const int cap = (int)(row_cap > MAXR ? MAXR : row_cap);
    zero_words_async(stream, meta, sizeof(uint32_t) * META_W * (size_t)n);   // atoms that are not listed: empty rows
    if (n_i > 0)
        hipLaunchKernelGGL(k_nbr_full, dim3(nbr_grid_blocks(n_i)), dim3(NBR_WPB * WAVE), 0, stream, p->num_species,
                           p->Rcr * p->Rcr, p->Rca * p->Rca, n, species, coords, n_i, ilist, numneigh, start, jlist,
                           cap, meta, (float4 *)ent, status);
    ANIHIP_CHECK_HIP(hipGetLastError());
    return 0;
}

#ifdef ANIHIP_TRACE
extern "C" int anihip_dev_nbr_trace_read(unsigned long long *dst /* host, 1024 x 10 */)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(anihip::g_nbr_trace), sizeof(unsigned long long) * 1024 * 10);
}
#endif

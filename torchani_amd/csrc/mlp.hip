// Per-species MLP ensemble forward + input-gradient backward on the gfx950 matrix cores.
//
// Replaces mnp::run (csrc/mnp.cpp:32-232) and BmmEnsemble (nn/_infer.py:61-216).  Common to all paths:
// the atoms of the shard are bucketed by species on the device (ballot ranks, no host sync, no
// nonzero()/index_select like nn/_containers.py:406-416).  Then, by network shape and precision:
//
//   A. split-fp16 ("f16x3", default), three hidden layers of width <= 256 (ANI-1x / ANI-2x):
//        >= 16384 atoms: k_tile_table -> k_mlp_fused<RB,NB> -> k_fused_finish -> k_gemm_l0b + k_gemm_h2<EPI_SCATTER>
//        >= 65536 atoms: k_tile_table -> k_mlp_fused<2,1,ACT,L0B = true> (layer-0 backward inside: phase 5) -> k_fused_finish
//        fewer:          k_small_prep (bucketing + tile table + padding rows) -> k_mlp_fused -> k_gemm_l0s (+ finish)
//      one fused kernel from the AEV rows to d E / d act0 (layer 0 only over the AEV slabs flagged non-zero,
//      activations in LDS, weights streamed from L2 in MFMA fragment order), then the layer-0 backward
//      GEMM over the flagged slabs.  See the comment block above k_mlp_fused.
//   B. split-fp16, other shapes: every layer ONE grouped-GEMM launch over all species and members
//      (k_gemm_h, 128 x 128 x 32 tiles; layer 0 = [n_s, K0] x [K0, M*H1] with rows gathered through the bucket
//      list, hidden layers = M independent GEMMs, backward against pre-transposed weights, the last one
//      scatters d E / d AEV rows back to atom order), bias + CELU / CELU' fused in the epilogues, k_head for
//      the output layer and the backward seed.
//   C. exact fp32 (precision = ANIHIP_MLP_FP32): the same grouped GEMMs on v_mfma_f32_32x32x2_f32 (k_gemm,
//      128 x 128 x 16 tiles, A transposed on the way into LDS so both fragment reads are conflict-free
//      ds_read_b32, register-prefetched double buffering).
//   D. training pass (anihip_mlp_train_forward / anihip_mlp_weight_grads): path C's GEMMs with every activation kept,
//      the backward into separate gradient buffers, k_wgrad (dW^T = D^T X straight from row-major operands on
//      v_mfma_f32_32x32x2_f32) and k_col_reduce for the weight / bias gradients, k_repack to refresh the packed
//      parameters after an optimizer step.
//   E. second-order pass of force training (anihip_mlp_tangent_weight_grads): tangent forward (EPI_TANGENT),
//      k_head_tangent, two adjoint GEMMs per layer (EPI_ADJ_P / EPI_ADJ_Q), two k_wgrad launches per layer.
// All GEMM kernels use an XCD-aware bijective tile order (tiles sharing an A stripe land on one XCD's L2).
#include <stdlib.h>

#include "anihip_common.h"
#include "train.h"

#include <type_traits>
#include <vector>
#include <cstdio>

namespace anihip {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const v4f gf4;  // global-memory float4 (forces global_load_dwordx4)

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDS_LD = BM + 4;
constexpr int GEMM_THREADS = 256;

enum Epilogue {
    EPI_BIAS_CELU = 0, EPI_DCELU = 1, EPI_SCATTER = 2,
    // second-order (tangent) pass of force training, exact-fp32 k_gemm only:
    EPI_TANGENT = 3,   // v = W adot_prev:        C = zdot = v,  C2 = adot = c'(Y) v            (Y = activations)
    EPI_ADJ_P = 4,     // v = mu = W^T p_next:    C = p = mu c'(Y),  C2 = mu c''(Y) Z           (Z = zdot)
    EPI_ADJ_Q = 5      // v = nu = W^T q_next:    C += nu c'(Y)                                 (C holds mu c'' zdot)
};

struct GemmProblem {
    const float *B;     // fp32 path: [batch][K][ldb]
    const float *bias;  // [batch][N] (EPI_BIAS_CELU)
    int K, N, ldb;
    int a_boff, c_boff;        // column offset of batch b in A / C rows = b * off
    int64_t b_stride;          // elements between consecutive batches of B
    int bias_stride;
    // f16x3 path: B as two fp16 planes {hi, lo}, each [batch][N][ldbh] (reduction index contiguous)
    const _Float16 *Bh;
    int64_t bh_plane;          // elements between the hi and the lo plane
    int64_t bh_stride;         // elements between consecutive batches inside a plane
    int ldbh;
    int k_valid;               // A columns >= k_valid are read as zero (K padded up to a multiple of 32)
    float w_inv_scale;         // 1 / (power-of-two scale baked into Bh)
};

struct GemmArgs {
    GemmProblem prob[MAX_S];
    const int *ctl;        // device control block (see MlpCtl)
    const float *A;
    int64_t lda;
    const int *a_gather;   // sorted position -> source row (layer 0) or NULL
    float *C;
    int64_t ldc;
    const float *Y;        // EPI_DCELU of the training pass (k_gemm): activations read from here, C only written
    int64_t ldy;           //   (NULL: C holds the activations and is overwritten in place)
    int skinny;            // k_gemm_h2<EPI_SCATTER>: row tiles with <= L0B_MAXNB flagged column blocks belong to k_gemm_l0b
    int a_tm_members;      // > 0: A is the tile-major d E / d act0 buffer (see tm_species_base), this many members
    int a_tm_h[MAX_S];     //      and per-species row width H_s
    const float *Z;        // tangent pass: zdot (same leading dimension as Y)
    float *C2;             // tangent pass: second output (same leading dimension as C)
    // training passes of GELU networks (act = ANIHIP_ACT_GELU): GELU' cannot be recovered from the stored activation
    // (x Phi(x) is not monotonic), so the forward keeps the PRE-activations too: Xout (EPI_BIAS_CELU, laid out like C) and
    // the passes that need activation derivatives read them back: X (laid out like Y)
    int act;
    const float *X;
    float *Xout;
    const int *c_scatter;  // sorted position -> destination row (last backward GEMM) or NULL
    int n_store;           // EPI_SCATTER: only columns < n_store are stored
    int S, batch, ncol_max, nrow_tiles_ub;
    float alpha, inv_alpha;
    // f16x3 layer-0 GEMMs: the reduction (fwd) / output (bwd) index runs over the AEV in "K' order": the
    // radial part padded up to a multiple of 32, then one 32-wide slab per species pair, so every 32-deep
    // stage / 32-column block is one (pair of) species block(s) of the AEV.  kp_rad = radial length
    // (0 = plain order).  stage_mask[atom] (optional) flags the slabs that can be non-zero for that atom
    // (written by the AEV forward kernel); tiles skip the slabs no row needs.
    int kp_rad;
    const uint32_t *stage_mask;
    // f16x3 path: operand scaling.  amax_in/out = stage index into the running-max table (or -1)
    int amax_in, amax_out;
    float a_static_scale;
    unsigned *amax;            // device [AMAX_STAGES][MAX_S][AMAX_SLOTS] float bits
};

struct FinishArgs {   // k_fused_finish, or the extra blocks of k_gemm_l0s
    const int *ctl;
    const int *perm;
    const float *member_part;
    float *atomic_e, *member_e;
    int64_t n_atoms;
    int S, M;
    int first_block;   // k_gemm_l0s: blocks from here on do this instead of a tile (0: none)
};

// (control block layout of the workspace: CTL_* in train.h)
// running |max| of every intermediate tensor (per stage, per species), spread over slots to keep the
// atomics off a single address; lives right behind the control block
constexpr int AMAX_STAGES = 8, AMAX_SLOTS = 32;
constexpr int AMAX_WORDS = AMAX_STAGES * MAX_S * AMAX_SLOTS;

// Tile-major layout of d E / d act0 between the fused network kernel and the layer-0 backward GEMM: the rows of a
// species are cut into blocks of 64, and block b of species s stores, member after member, 64 rows x H_s columns
// contiguously:   offset(s, rel, m, c) = base_s + (((rel >> 6) * M + m) * 64 + (rel & 63)) * H_s + c,
// base_s = sum_{s' < s} ceil(cnt_s' / 64) * 64 * M * H_s'.  Both kernels then stream whole 16..64-KB blocks instead of
// 128-B .. 1-KB pieces strided by the 8-KB row of the plain [n][M * H] layout.
// d E / d act0 from the fused kernel to the layer-0 backward GEMMs, "tile-major": per species, per 64-atom tile, member
// after member a 64 x H block; INSIDE a block the floats lie in MFMA A-fragment order:
//   [column block cb][row block rb][k step ks][lane = c8 * 32 + row][8 floats]   (column = 32 cb + 16 ks + 8 c8 + j)
// i.e. 2-KB units of 32 rows x 16 columns.  The fused kernel's store instruction (32 rows x 2 runs of 4 columns)
// fills 1 KB of a unit contiguously, a wave of the skinny GEMM reads its whole fragment of a k step as one
// coalesced 2-KB load, the 256 x 256 GEMM's staging threads read 32-B pieces 32 B apart.
__device__ __forceinline__ int tm_unit(int cb, int rb, int ks) { return ((cb * 2 + rb) * 2 + ks) * 512; }

__device__ __forceinline__ int64_t tm_species_base(const int *ctl, const int *H, int M, int s)
{
    int64_t base = 0;
    for (int t = 0; t < s; ++t) base += (int64_t)((ctl[8 * 0 + t] + 63) >> 6) * 64 * M * H[t];   // ctl[CTL_CNT + t]
    return base;
}

// ---- species bucketing --------------------------------------------------------------------------

constexpr int SP_CHUNK = 1024;  // atoms per wave in the bucketing kernels

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
        for (uint64_t pad = __ballot(i < hi && sp < 0); pad; pad &= pad - 1) {
            const int64_t ip = c0 + it * WAVE + (int)__builtin_ctzll(pad);
            if (lane_id() == 0 && atomic_e) atomic_e[ip] = 0.f;
            if (member_e && lane_id() < M) member_e[(int64_t)lane_id() * n_atoms + ip] = 0.f;
            if (grad_aev) {
                float4 *row = reinterpret_cast<float4 *>(grad_aev + (size_t)ip * L);
                for (int f = lane_id(); f < (L >> 2); f += WAVE) row[f] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
}

// ---- grouped GEMM -----------------------------------------------------------------------------------

__device__ __forceinline__ float celu(float x, float alpha, float inv_alpha)
{
    // nn/_core.py:163-167 : celu(x, 0.1) = max(0,x) + min(0, alpha (exp(x/alpha) - 1))
    return x > 0.f ? x : alpha * (__expf(x * inv_alpha) - 1.0f);
}
// torch.nn.GELU() (approximate = 'none'): x Phi(x); derivatives Phi + x phi and phi (2 - x^2)
__device__ __forceinline__ float gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678f)); }
// first and second derivative of the activation: CELU from the stored activation y (x > 0: (1, 0); else (y / alpha + 1,
// that / alpha)), GELU from the stored pre-activation x
__device__ __forceinline__ void act_derivs(int act, float y, float x, float inv_alpha, float &c1, float &c2)
{
    if (act == ANIHIP_ACT_GELU) {
        const float ph = 0.5f * (1.0f + erff(x * 0.70710678f));
        const float pd = 0.39894228f * __expf(-0.5f * x * x);
        c1 = ph + x * pd;
        c2 = pd * (2.0f - x * x);
    } else {
        c1 = y > 0.f ? 1.0f : y * inv_alpha + 1.0f;
        c2 = y > 0.f ? 0.f : c1 * inv_alpha;
    }
}

// K loop of one 128 x (32 NB) tile: register-prefetched global loads, double-buffered LDS, one
// barrier per K step; NB is compile-time so the MFMA stream has no branches.
constexpr int LDS_BUF = BK * LDS_LD;  // floats per buffer
template <int NB>
__device__ __forceinline__ void gemm_kloop(f32x16 (&acc)[4], const gf4 *a_src, const float *b_src,
                                           int64_t b_step, int nk, float *a_dst, float *b_dst,
                                           const float *a_frag, const float *b_frag)
{
    v4f ra0, ra1, rb0, rb1;
    auto gload = [&](int kt) {
        ra0 = a_src[kt * (BK / 4)];
        ra1 = a_src[kt * (BK / 4) + 1];
        const gf4 *bp = (const gf4 *)(b_src + kt * b_step);
        rb0 = bp[0];
        rb1 = bp[1];
    };
    auto lstore = [&](int buf) {
        float *ap = a_dst + buf * LDS_BUF;
        ap[0 * LDS_LD] = ra0.x; ap[1 * LDS_LD] = ra0.y; ap[2 * LDS_LD] = ra0.z; ap[3 * LDS_LD] = ra0.w;
        ap[4 * LDS_LD] = ra1.x; ap[5 * LDS_LD] = ra1.y; ap[6 * LDS_LD] = ra1.z; ap[7 * LDS_LD] = ra1.w;
        *reinterpret_cast<v4f *>(b_dst + buf * LDS_BUF) = rb0;
        *reinterpret_cast<v4f *>(b_dst + buf * LDS_BUF + 4) = rb1;
    };
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        const float *af = a_frag + buf * LDS_BUF, *bf = b_frag + buf * LDS_BUF;
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const float av = af[2 * kk * LDS_LD];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const float bv = bf[2 * kk * LDS_LD + nb * 32];
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[nb], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }
}

template <int EPI>
__global__ __launch_bounds__(GEMM_THREADS) void k_gemm(GemmArgs g)
{
    __shared__ float As[2][BK][LDS_LD];
    __shared__ float Bs[2][BK][LDS_LD];

    // XCD-aware bijective remap: consecutive logical tiles (same A stripe) share one XCD / L2
    const int nwg = gridDim.x;
    int id = blockIdx.x;
    {
        const int qd = nwg >> 3, rm = nwg & 7, xcd = id & 7;
        id = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (id >> 3);
    }
    const int col_t = id % g.ncol_max;
    const int bb = (id / g.ncol_max) % g.batch;
    const int row_t = id / (g.ncol_max * g.batch);

    const int *ctl = g.ctl;
    if (row_t >= ctl[CTL_TILE + g.S]) return;
    int s = 0;
    while (s + 1 < g.S && row_t >= ctl[CTL_TILE + s + 1]) ++s;
    const GemmProblem &pr = g.prob[s];
    const int n0 = col_t * BN;
    if (n0 >= pr.N) return;
    const int m0 = (row_t - ctl[CTL_TILE + s]) * BM;          // first row inside the species
    const int n_rows = ctl[CTL_CNT + s] - m0;                 // valid rows in this tile (may be > BM)
    const int p0 = ctl[CTL_OFF + s] + m0;                     // sorted position of tile row 0
    const int nb_act = min(4, (pr.N - n0) >> 5);              // active 32-column blocks

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // A loader: row = tid>>1, 8 consecutive k.  Rows past the species end re-read row 0 of the tile
    // (always valid memory); their accumulators are simply never stored.
    const int a_row = tid >> 1, a_k = (tid & 1) * 8;
    const gf4 *a_src;
    {
        const int rr = a_row < n_rows ? a_row : 0;
        const int64_t src_row = g.a_gather ? (int64_t)g.a_gather[p0 + rr] : (int64_t)(p0 + rr);
        a_src = (const gf4 *)(g.A + src_row * g.lda + (int64_t)bb * pr.a_boff + a_k);
    }
    // B loader: k = tid>>4, 8 consecutive n.  Column chunks past N (inactive blocks) re-read chunk 0.
    const int b_k = tid >> 4, b_n = (tid & 15) * 8;
    const int b_col = (n0 + b_n < pr.N) ? n0 + b_n : n0;
    const float *b_src = pr.B + (int64_t)bb * pr.b_stride + (int64_t)b_k * pr.ldb + b_col;
    const int64_t b_step = (int64_t)BK * pr.ldb;

    f32x16 acc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

    const int nk = pr.K / BK;
    const int fr = lane & 31, fk = lane >> 5;
    float *a_dst = &As[0][a_k][a_row];
    float *b_dst = &Bs[0][b_k][b_n];
    const float *a_frag = &As[0][fk][wave * 32 + fr];
    const float *b_frag = &Bs[0][fk][fr];
    switch (nb_act) {
        case 4: gemm_kloop<4>(acc, a_src, b_src, b_step, nk, a_dst, b_dst, a_frag, b_frag); break;
        case 3: gemm_kloop<3>(acc, a_src, b_src, b_step, nk, a_dst, b_dst, a_frag, b_frag); break;
        case 2: gemm_kloop<2>(acc, a_src, b_src, b_step, nk, a_dst, b_dst, a_frag, b_frag); break;
        default: gemm_kloop<1>(acc, a_src, b_src, b_step, nk, a_dst, b_dst, a_frag, b_frag); break;
    }

    // epilogue.  C/D layout of 32x32: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        if (nb >= nb_act) continue;
        const int col = n0 + nb * 32 + fr;
        float bias = 0.f;
        if (EPI == EPI_BIAS_CELU) bias = pr.bias[(int64_t)bb * pr.bias_stride + col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
            if (row >= n_rows) continue;
            float v = acc[nb][r];
            if (EPI == EPI_BIAS_CELU) {
                const int64_t ic = (int64_t)(p0 + row) * g.ldc + (int64_t)bb * pr.c_boff + col;
                const float x = v + bias;
                g.C[ic] = g.act == ANIHIP_ACT_GELU ? gelu(x) : celu(x, g.alpha, g.inv_alpha);
                if (g.Xout) g.Xout[ic] = x;
            } else if (EPI == EPI_DCELU) {
                // stored activation y = celu(x):  celu'(x) = 1 (y > 0)  or  exp(x/alpha) = y/alpha + 1   (GELU: from x)
                float *cp = g.C + (int64_t)(p0 + row) * g.ldc + (int64_t)bb * pr.c_boff + col;
                const int64_t iy = (int64_t)(p0 + row) * g.ldy + (int64_t)bb * pr.c_boff + col;
                const float y = g.Y ? g.Y[iy] : __builtin_nontemporal_load(cp);
                float c1, c2;
                act_derivs(g.act, y, g.X ? g.X[iy] : 0.f, g.inv_alpha, c1, c2);
                *cp = v * c1;
            } else if (EPI == EPI_TANGENT || EPI == EPI_ADJ_P || EPI == EPI_ADJ_Q) {
                // act'(x) and act''(x): CELU from the stored activation y, GELU from the stored pre-activation
                const int64_t ic = (int64_t)(p0 + row) * g.ldc + (int64_t)bb * pr.c_boff + col;
                const int64_t iy = (int64_t)(p0 + row) * g.ldy + (int64_t)bb * pr.c_boff + col;
                float c1, c2;
                act_derivs(g.act, g.Y[iy], g.X ? g.X[iy] : 0.f, g.inv_alpha, c1, c2);
                if (EPI == EPI_TANGENT) {
                    g.C[ic] = v;
                    g.C2[ic] = c1 * v;
                } else if (EPI == EPI_ADJ_P) {
                    g.C[ic] = v * c1;
                    g.C2[ic] = v * c2 * g.Z[iy];
                } else {
                    g.C[ic] += v * c1;
                }
            } else {
                if (col < g.n_store)
                    g.C[(int64_t)g.c_scatter[p0 + row] * g.ldc + col] = v;
            }
        }
    }
}

// ---- f16x3 grouped GEMM ---------------------------------------------------------------------------
// Same tiling / epilogues as k_gemm, but every fp32 operand is split into two fp16 numbers
// (hi + lo = x * 2^e) and a product is hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 (fp32
// accumulate): ~2^-21.5 relative error per product at 16/3 of the fp32-MFMA rate.  A (AEV rows /
// activations / gradients, fp32 in HBM) is split in the loader; its power-of-two scale comes from the
// running max the producing kernel left in the amax table, so the scaled values sit in [2^13, 2^14)
// at most and can neither overflow nor lose their low part.  B (weights) is pre-split at pack time.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const h8 gh8;

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

constexpr int HBK = 32;            // reduction depth per LDS stage
// LDS stage: 4 planes (A_hi, A_lo, B_hi, B_lo) of [128 rows][32 halves] = 64-B rows, no padding; the four
// 16-B pieces of a row are XOR-swizzled with (row >> 2) & 3, which makes both the staging stores
// (8-lane groups: two whole rows) and the MFMA fragment loads (ds_read_b128, 16-lane groups) conflict-free.
constexpr int H_PLANE = BM * HBK;  // halves per operand plane per stage
__device__ __forceinline__ int h_off(int row, int piece) { return row * HBK + ((piece ^ ((row >> 2) & 3)) << 3); }

__device__ __forceinline__ float amax_scale(const unsigned *amax, int stage, int s)
{
    // every lane reads one slot; wave max; scale = 2^(13 - floor(log2(amax)))
    const unsigned *p = amax + (stage * MAX_S + s) * AMAX_SLOTS;
    unsigned v = p[lane_id() & (AMAX_SLOTS - 1)];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o));
    if (v == 0u) return 1.0f;
    const int e = (int)(v >> 23) - 127;  // floor(log2(amax)) for normal floats
    return __uint_as_float((unsigned)(127 + 13 - e) << 23);
}

__device__ __forceinline__ void amax_update(unsigned *amax, int stage, int s, float m)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane_id() == 0)
        atomicMax(amax + (stage * MAX_S + s) * AMAX_SLOTS + (blockIdx.x & (AMAX_SLOTS - 1)), __float_as_uint(m));
}

// K loop of one 128 x (32 NB) tile.  Thread t stages rows t>>2 and 64 + (t>>2), 16-B piece t&3 of all four
// planes; global loads run two stages ahead of the MFMAs (register ring), LDS is double buffered, one
// barrier per stage.
struct HStage {
    v4f a[2][2];   // [row pass][2 x float4 = 8 k values]
    h8 bh[2], bl[2];
};

template <int NB>
__device__ __forceinline__ void gemm_h_kloop(f32x16 (&acc)[4], const gf4 *a_src0, const gf4 *a_src1,
                                             int k_valid, int kp_rad, float sa, const _Float16 *b_src0,
                                             const _Float16 *b_src1, int64_t bh_plane, int nk, _Float16 *sm,
                                             int wave)
{
    constexpr int STAGE = 4 * H_PLANE;  // halves per LDS stage: A_hi, A_lo, B_hi, B_lo
    const int tid = threadIdx.x, lane = tid & 63;
    const int srow = tid >> 2, piece = tid & 3;
    const int fr = lane & 31, fk = lane >> 5;
    const v4f z4 = v4f{0.f, 0.f, 0.f, 0.f};
    auto gload = [&](HStage &st, int kt) {
        // clamped, branch-free: invalid columns re-read chunk 0 and are zeroed by a select
        const int acol = kp_rad ? kp_col(kp_rad, kt) : kt * HBK;
        const bool ok = kp_rad ? piece * 8 < kp_valid(kp_rad, kt) : kt * HBK + piece * 8 < k_valid;
        const int o = ok ? acol / 4 : 0;
        st.a[0][0] = a_src0[o]; st.a[0][1] = a_src0[o + 1];
        st.a[1][0] = a_src1[o]; st.a[1][1] = a_src1[o + 1];
        if (!ok) { st.a[0][0] = z4; st.a[0][1] = z4; st.a[1][0] = z4; st.a[1][1] = z4; }
        st.bh[0] = *(const gh8 *)(b_src0 + kt * HBK);
        st.bl[0] = *(const gh8 *)(b_src0 + bh_plane + kt * HBK);
        st.bh[1] = *(const gh8 *)(b_src1 + kt * HBK);
        st.bl[1] = *(const gh8 *)(b_src1 + bh_plane + kt * HBK);
    };
    auto lstore = [&](const HStage &st, int buf) {
        _Float16 *base = sm + buf * STAGE;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            h8 hi, lo;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float x = (c < 4 ? st.a[ps][0][c & 3] : st.a[ps][1][c & 3]) * sa;
                const _Float16 h = (_Float16)x;
                hi[c] = h;
#ifndef ANIHIP_ABLATE_NOCONV
                lo[c] = (_Float16)(x - (float)h);
#else
                lo[c] = h;
#endif
            }
            const int off = h_off(srow + 64 * ps, piece);
            *reinterpret_cast<h8 *>(base + off) = hi;
            *reinterpret_cast<h8 *>(base + H_PLANE + off) = lo;
            *reinterpret_cast<h8 *>(base + 2 * H_PLANE + off) = st.bh[ps];
            *reinterpret_cast<h8 *>(base + 3 * H_PLANE + off) = st.bl[ps];
        }
    };
    auto compute = [&](int buf) {
        const _Float16 *base = sm + buf * STAGE;
#pragma unroll
        for (int ks = 0; ks < HBK / 16; ++ks) {
            const int pc = ks * 2 + fk;
            const int ao = h_off(wave * 32 + fr, pc);
            const h8 ahi = *reinterpret_cast<const h8 *>(base + ao);
            const h8 alo = *reinterpret_cast<const h8 *>(base + H_PLANE + ao);
            h8 bhi[NB], blo[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int bo = h_off(nb * 32 + fr, pc);
                bhi[nb] = *reinterpret_cast<const h8 *>(base + 2 * H_PLANE + bo);
                blo[nb] = *reinterpret_cast<const h8 *>(base + 3 * H_PLANE + bo);
            }
            // small terms first, independent accumulators between dependent MFMAs
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi[nb], acc[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo[nb], acc[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi[nb], acc[nb], 0, 0, 0);
        }
    };
    // ring of two register stages: s0 holds stage kt+1 (even kt) / kt+2 ..., see the unrolled-by-2 loop
    HStage s0, s1;
    gload(s0, 0);
    gload(s1, min(1, nk - 1));
    lstore(s0, 0);
    gload(s0, min(2, nk - 1));
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        // stage kt is in LDS buffer 0; registers: s1 = stage kt+1, s0 = stage kt+2
        compute(0);
#ifndef ANIHIP_ABLATE_NOSTORE
        if (kt + 1 < nk) lstore(s1, 1);
#endif
#ifndef ANIHIP_ABLATE_NOLOAD
        gload(s1, min(kt + 3, nk - 1));
#endif
        __syncthreads();
        if (kt + 1 < nk) {
            compute(1);
#ifndef ANIHIP_ABLATE_NOSTORE
            if (kt + 2 < nk) lstore(s0, 0);
#endif
#ifndef ANIHIP_ABLATE_NOLOAD
            gload(s0, min(kt + 4, nk - 1));
#endif
            __syncthreads();
        }
    }
}

#ifdef ANIHIP_DEV_TRACE   // development builds: [workgroup][8] = {shader clock, 100-MHz clock} x {start, loop, epilogue, end}
__device__ unsigned long long g_gemm_trace[8 * 4096];
#define GEMM_STAMP(k)                                                                       \
    if (EPI == EPI_SCATTER && threadIdx.x == 0 && blockIdx.x < 4096) {                      \
        g_gemm_trace[blockIdx.x * 8 + 2 * (k)] = __builtin_readcyclecounter();              \
        g_gemm_trace[blockIdx.x * 8 + 2 * (k) + 1] = wall_clock64();                        \
    }
#else
#define GEMM_STAMP(k)
#endif

template <int EPI>
__global__ __launch_bounds__(GEMM_THREADS) void k_gemm_h(GemmArgs g)
{
    __shared__ __attribute__((aligned(16))) _Float16 sm[2 * 4 * H_PLANE];
    GEMM_STAMP(0)

    const int nwg = gridDim.x;
    int id = blockIdx.x;
    {
        const int qd = nwg >> 3, rm = nwg & 7, xcd = id & 7;
        id = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (id >> 3);
    }
    const int col_t = id % g.ncol_max;
    const int bb = (id / g.ncol_max) % g.batch;
    const int row_t = id / (g.ncol_max * g.batch);

    const int *ctl = g.ctl;
    if (row_t >= ctl[CTL_TILE + g.S]) return;
    int s = 0;
    while (s + 1 < g.S && row_t >= ctl[CTL_TILE + s + 1]) ++s;
    const GemmProblem &pr = g.prob[s];
    const int n0 = col_t * BN;
    const bool compact = (EPI == EPI_SCATTER) && g.stage_mask;
    if (!compact && n0 >= pr.N) return;
    const int m0 = (row_t - ctl[CTL_TILE + s]) * BM;
    const int n_rows = ctl[CTL_CNT + s] - m0;
    const int p0 = ctl[CTL_OFF + s] + m0;
    int nb_act = min(4, (pr.N - n0) >> 5);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float sa = g.amax_in >= 0 ? amax_scale(g.amax, g.amax_in, s) : g.a_static_scale;
    const float out_scale = pr.w_inv_scale / sa;

    // loaders: thread t -> rows t>>2 and 64 + (t>>2), 8 consecutive k (piece t&3) of every plane
    const int srow = tid >> 2, piece = tid & 3;

    // ---- layer-0 backward with slab masks: the tile's 4 column blocks are the (4 col_t .. 4 col_t + 3)-th blocks
    // flagged in the OR of its atoms' masks (as in k_gemm_h2); column tiles past the last flagged block exit ----
    int cbw[4] = {(n0 >> 5), (n0 >> 5) + 1, (n0 >> 5) + 2, (n0 >> 5) + 3};   // this tile's column blocks
    if (compact) {
        __shared__ int s_tab[5];   // [0] = tile mask, [1..4] = column blocks
        const int *rows = g.a_gather ? g.a_gather : g.c_scatter;   // sorted position -> atom
        const int q0 = srow < n_rows ? srow : 0, q1 = srow + 64 < n_rows ? srow + 64 : 0;
        uint32_t mk = g.stage_mask[rows[p0 + q0]] | g.stage_mask[rows[p0 + q1]];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mk |= (uint32_t)__shfl_xor((int)mk, o);
        if (tid == 0) s_tab[0] = 0;
        __syncthreads();
        if (lane == 0) atomicOr(reinterpret_cast<unsigned *>(&s_tab[0]), mk);
        __syncthreads();
        const uint32_t tmask = (uint32_t)s_tab[0];
        if (4 * col_t >= __popc(tmask)) return;   // (also: atoms without neighbors, nothing to differentiate)
        if (tid < 4) {
            uint32_t m = tmask;
            for (int k = 0; k < 4 * col_t + tid; ++k) m &= m - 1;
            s_tab[1 + tid] = m ? (int)__builtin_ctz(m) : -1;
        }
        __syncthreads();
        nb_act = 0;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            cbw[nb] = s_tab[1 + nb];
            if (cbw[nb] >= 0) nb_act = nb + 1;
        }
    }
    const gf4 *a_src0, *a_src1;
    {
        const int r0 = srow < n_rows ? srow : 0, r1 = srow + 64 < n_rows ? srow + 64 : 0;
        const int64_t s0r = g.a_gather ? (int64_t)g.a_gather[p0 + r0] : (int64_t)(p0 + r0);
        const int64_t s1r = g.a_gather ? (int64_t)g.a_gather[p0 + r1] : (int64_t)(p0 + r1);
        a_src0 = (const gf4 *)(g.A + s0r * g.lda + (int64_t)bb * pr.a_boff + piece * 8);
        a_src1 = (const gf4 *)(g.A + s1r * g.lda + (int64_t)bb * pr.a_boff + piece * 8);
    }
    int bn0 = (n0 + srow < pr.N) ? n0 + srow : n0, bn1 = (n0 + srow + 64 < pr.N) ? n0 + srow + 64 : n0;
    if (compact) {   // B rows staged by this thread: tile columns srow and srow + 64 of the compacted blocks
        const int c0 = cbw[srow >> 5], c1 = cbw[2 + (srow >> 5)];
        bn0 = (c0 >= 0 ? c0 : cbw[0]) * 32 + (srow & 31);
        bn1 = (c1 >= 0 ? c1 : cbw[0]) * 32 + (srow & 31);
    }
    const _Float16 *b_src0 = pr.Bh + (int64_t)bb * pr.bh_stride + (int64_t)bn0 * pr.ldbh + piece * 8;
    const _Float16 *b_src1 = pr.Bh + (int64_t)bb * pr.bh_stride + (int64_t)bn1 * pr.ldbh + piece * 8;

    f32x16 acc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

    const int nk = pr.K / HBK;
    const int fr = lane & 31, fk = lane >> 5;
    GEMM_STAMP(1)
    switch (nb_act) {
        case 4: gemm_h_kloop<4>(acc, a_src0, a_src1, pr.k_valid, EPI == EPI_BIAS_CELU ? g.kp_rad : 0, sa, b_src0, b_src1, pr.bh_plane, nk, sm, wave); break;
        case 3: gemm_h_kloop<3>(acc, a_src0, a_src1, pr.k_valid, EPI == EPI_BIAS_CELU ? g.kp_rad : 0, sa, b_src0, b_src1, pr.bh_plane, nk, sm, wave); break;
        case 2: gemm_h_kloop<2>(acc, a_src0, a_src1, pr.k_valid, EPI == EPI_BIAS_CELU ? g.kp_rad : 0, sa, b_src0, b_src1, pr.bh_plane, nk, sm, wave); break;
        default: gemm_h_kloop<1>(acc, a_src0, a_src1, pr.k_valid, EPI == EPI_BIAS_CELU ? g.kp_rad : 0, sa, b_src0, b_src1, pr.bh_plane, nk, sm, wave); break;
    }

    GEMM_STAMP(2)
    float vmax = 0.f;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        if (nb >= nb_act) continue;
        const int col = cbw[nb] * 32 + fr;
        float bias = 0.f;
        if (EPI == EPI_BIAS_CELU) bias = pr.bias[(int64_t)bb * pr.bias_stride + col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
            if (row >= n_rows) continue;
            float v = acc[nb][r] * out_scale;
            if (EPI == EPI_BIAS_CELU) {
                v = celu(v + bias, g.alpha, g.inv_alpha);
                g.C[(int64_t)(p0 + row) * g.ldc + (int64_t)bb * pr.c_boff + col] = v;
            } else if (EPI == EPI_DCELU) {
                float *cp = g.C + (int64_t)(p0 + row) * g.ldc + (int64_t)bb * pr.c_boff + col;
                const float y = *cp;
                v = v * (y > 0.f ? 1.0f : y * g.inv_alpha + 1.0f);
                *cp = v;
            } else {
                // column of the K'-ordered output -> AEV feature
                const int cb = col >> 5;
                const int feat = g.kp_rad ? kp_col(g.kp_rad, cb) + fr : col;
                const bool okc = g.kp_rad ? fr < kp_valid(g.kp_rad, cb) : true;
                if (okc && feat < g.n_store) g.C[(int64_t)g.c_scatter[p0 + row] * g.ldc + feat] = v;
            }
            vmax = fmaxf(vmax, fabsf(v));
        }
    }
    if (g.amax_out >= 0) amax_update(g.amax, g.amax_out, s, vmax);
    GEMM_STAMP(3)
}

// ---- layer-0 backward for few atoms: 128 x 128 x 32 tiles, EIGHT waves -------------------------------------------
// Below the 256 x 256 tiling's threshold the layer-0 backward has only a hundred-odd tiles, one workgroup per CU, and
// k_gemm_h's four waves (one per SIMD) run a pure latency chain: barrier -> fragment reads -> 24 MFMAs -> barrier,
// 2.5 k clocks per 32-deep stage of which 0.8 k are MFMA issue (measured with the phase stamps of the development
// build).  Here the same tile is owned by 4 (rows) x 2 (column halves) waves, two per SIMD, so one wave's MFMAs cover
// the other's LDS latency, the staging work per thread halves (one row of A and of B per stage), and the destination rows
// of the scatter are loaded once before the stores instead of between them.  d E / d AEV only, compacted to the flagged
// column blocks (stage_mask), A = d E / d act0 row-major as the fused kernel leaves it for this path.
constexpr int L0S_THREADS = 512;
__device__ __forceinline__ void fused_finish(const FinishArgs &f, int64_t first, int64_t stride);

struct L0sStage {
    v4f a[2];   // 8 k values of this thread's A row
    h8 bh, bl;  // 8 k values of this thread's B row, hi / lo plane
};

constexpr int L0S_BUFS = 3;   // LDS stages: the fragments of stage k + 1 are read while stage k multiplies, stage k + 2 is written

template <int NBW>
struct L0sFrag {
    h8 ahi[HBK / 16], alo[HBK / 16], bhi[HBK / 16][NBW], blo[HBK / 16][NBW];
};

template <int NBW>
__device__ __forceinline__ void l0s_kloop(f32x16 (&acc)[2], const gf4 *a_src, float sa, const _Float16 *b_src,
                                          int64_t bh_plane, int nk, _Float16 *sm, int wr, int wc)
{
    constexpr int STAGE = 4 * H_PLANE;  // halves per LDS stage: A_hi, A_lo, B_hi, B_lo
    const int tid = threadIdx.x, lane = tid & 63;
    const int srow = tid >> 2, piece = tid & 3;
    const int fr = lane & 31, fk = lane >> 5;
    auto gload = [&](L0sStage &st, int kt) {
        st.a[0] = a_src[kt * (HBK / 4)];
        st.a[1] = a_src[kt * (HBK / 4) + 1];
        st.bh = *(const gh8 *)(b_src + kt * HBK);
        st.bl = *(const gh8 *)(b_src + bh_plane + kt * HBK);
    };
    auto lstore = [&](const L0sStage &st, int buf) {
        _Float16 *base = sm + buf * STAGE;
        h8 hi, lo;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float x = st.a[c >> 2][c & 3] * sa;
            const _Float16 h = (_Float16)x;
            hi[c] = h;
            lo[c] = (_Float16)(x - (float)h);
        }
        const int off = h_off(srow, piece);
        *reinterpret_cast<h8 *>(base + off) = hi;
        *reinterpret_cast<h8 *>(base + H_PLANE + off) = lo;
        *reinterpret_cast<h8 *>(base + 2 * H_PLANE + off) = st.bh;
        *reinterpret_cast<h8 *>(base + 3 * H_PLANE + off) = st.bl;
    };
    auto fload = [&](L0sFrag<NBW> &f, int buf) {
        const _Float16 *base = sm + buf * STAGE;
#pragma unroll
        for (int ks = 0; ks < HBK / 16; ++ks) {
            const int pc = ks * 2 + fk;
            const int ao = h_off(wr * 32 + fr, pc);
            f.ahi[ks] = *reinterpret_cast<const h8 *>(base + ao);
            f.alo[ks] = *reinterpret_cast<const h8 *>(base + H_PLANE + ao);
#pragma unroll
            for (int j = 0; j < NBW; ++j) {
                const int bo = h_off((wc * 2 + j) * 32 + fr, pc);
                f.bhi[ks][j] = *reinterpret_cast<const h8 *>(base + 2 * H_PLANE + bo);
                f.blo[ks][j] = *reinterpret_cast<const h8 *>(base + 3 * H_PLANE + bo);
            }
        }
    };
    auto mma = [&](const L0sFrag<NBW> &f) {   // small terms first, same order per accumulator as k_gemm_h
#pragma unroll
        for (int ks = 0; ks < HBK / 16; ++ks) {
#pragma unroll
            for (int j = 0; j < NBW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.alo[ks], f.bhi[ks][j], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NBW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ahi[ks], f.blo[ks][j], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NBW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ahi[ks], f.bhi[ks][j], acc[j], 0, 0, 0);
        }
    };
    // Stage k lives in LDS buffer k % 3.  Iteration k: read the fragments of stage k + 1 (stored in iteration k - 1, visible
    // since that iteration's barrier), multiply stage k from registers, convert and store stage k + 2 (its buffer was last
    // read for stage k - 1, whose fragments are in registers since iteration k - 2), load stage k + 4 from memory: the
    // only thing a wave waits for inside an iteration is its own MFMA queue.
    L0sStage s0, s1;
    L0sFrag<NBW> f0, f1;
    gload(s0, 0);
    gload(s1, min(1, nk - 1));
    lstore(s0, 0);
    gload(s0, min(2, nk - 1));
    lstore(s1, 1);
    gload(s1, min(3, nk - 1));
    __syncthreads();
    fload(f0, 0);
    int b1 = 1, b2 = 2;   // buffers of stages kt + 1 and kt + 2
    for (int kt = 0; kt < nk; kt += 2) {
        if (kt + 1 < nk) fload(f1, b1);
        mma(f0);
        if (kt + 2 < nk) lstore(s0, b2);
        gload(s0, min(kt + 4, nk - 1));
        __syncthreads();
        b1 = b1 == 2 ? 0 : b1 + 1; b2 = b2 == 2 ? 0 : b2 + 1;
        if (kt + 1 < nk) {
            if (kt + 2 < nk) fload(f0, b1);
            mma(f1);
            if (kt + 3 < nk) lstore(s1, b2);
            gload(s1, min(kt + 5, nk - 1));
            __syncthreads();
            b1 = b1 == 2 ? 0 : b1 + 1; b2 = b2 == 2 ? 0 : b2 + 1;
        }
    }
}

__global__ __launch_bounds__(L0S_THREADS) void k_gemm_l0s(GemmArgs g, FinishArgs fin)
{
    extern __shared__ __attribute__((aligned(16))) _Float16 sm[];   // L0S_BUFS x 4 planes x H_PLANE halves (96 KB)
    __shared__ int s_tab[5];   // [0] = tile mask, [1..4] = column blocks

    // the per-atom energies of the fused kernel are independent of this GEMM: a few extra workgroups (the tiles leave half
    // of the CUs idle) finish them here instead of in a launch of their own in front of this one
    if (fin.first_block > 0 && (int)blockIdx.x >= fin.first_block) {
        fused_finish(fin, (int64_t)(blockIdx.x - fin.first_block) * L0S_THREADS + threadIdx.x,
                     (int64_t)(gridDim.x - fin.first_block) * L0S_THREADS);
        return;
    }
    const int nwg = fin.first_block > 0 ? fin.first_block : (int)gridDim.x;
    int id = blockIdx.x;
    {
        const int qd = nwg >> 3, rm = nwg & 7, xcd = id & 7;
        id = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (id >> 3);
    }
    const int col_t = id % g.ncol_max;
    const int row_t = id / g.ncol_max;
    const int *ctl = g.ctl;
    if (row_t >= ctl[CTL_TILE + g.S]) return;
    int s = 0;
    while (s + 1 < g.S && row_t >= ctl[CTL_TILE + s + 1]) ++s;
    const GemmProblem &pr = g.prob[s];
    const int m0 = (row_t - ctl[CTL_TILE + s]) * BM;
    const int n_rows = ctl[CTL_CNT + s] - m0;
    const int p0 = ctl[CTL_OFF + s] + m0;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave & 3, wc = wave >> 2;
    const int srow = tid >> 2, piece = tid & 3;
    const int arow = srow < n_rows ? srow : 0;

    // the tile's column blocks: the (4 col_t .. 4 col_t + 3)-th blocks flagged in the OR of its atoms' masks
    {
        uint32_t mk = g.stage_mask[g.c_scatter[p0 + arow]];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mk |= (uint32_t)__shfl_xor((int)mk, o);
        if (tid == 0) s_tab[0] = 0;
        __syncthreads();
        if (lane == 0) atomicOr(reinterpret_cast<unsigned *>(&s_tab[0]), mk);
        __syncthreads();
        const uint32_t tmask = (uint32_t)s_tab[0];
        if (4 * col_t >= __popc(tmask)) return;   // (also: atoms without neighbors, nothing to differentiate)
        if (tid < 4) {
            uint32_t m = tmask;
            for (int k = 0; k < 4 * col_t + tid; ++k) m &= m - 1;
            s_tab[1 + tid] = m ? (int)__builtin_ctz(m) : -1;
        }
        __syncthreads();
    }
    const int cb0 = __builtin_amdgcn_readfirstlane(s_tab[1]);
    const int cmine = s_tab[1 + (srow >> 5)];                     // block of the B row this thread stages
    // this wave's two column blocks (-1: none); wave-uniform, and the branches below must look uniform to the compiler
    const int cw[2] = {__builtin_amdgcn_readfirstlane(s_tab[1 + 2 * wc]), __builtin_amdgcn_readfirstlane(s_tab[2 + 2 * wc])};
    const int nbw = cw[0] < 0 ? 0 : (cw[1] < 0 ? 1 : 2);

    const float sa = amax_scale(g.amax, g.amax_in, s);
    const float out_scale = pr.w_inv_scale / sa;
    const gf4 *a_src = (const gf4 *)(g.A + (int64_t)(p0 + arow) * g.lda + piece * 8);
    const _Float16 *b_src = pr.Bh + (int64_t)((cmine >= 0 ? cmine : cb0) * 32 + (srow & 31)) * pr.ldbh + piece * 8;

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int nk = pr.K / HBK;
    if (nbw == 2) l0s_kloop<2>(acc, a_src, sa, b_src, pr.bh_plane, nk, sm, wr, wc);
    else l0s_kloop<1>(acc, a_src, sa, b_src, pr.bh_plane, nk, sm, wr, wc);   // (a wave without a block stages and idles along)
    if (nbw == 0) return;

    const int fr = lane & 31, fk = lane >> 5;
    int dst[16];   // destination rows first, the stores do not wait on them one by one
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
        dst[r] = row < n_rows ? g.c_scatter[p0 + row] : -1;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (j >= nbw) continue;
        const int cb = cw[j];
        const int feat = g.kp_rad ? kp_col(g.kp_rad, cb) + fr : cb * 32 + fr;
        const bool okc = (g.kp_rad ? fr < kp_valid(g.kp_rad, cb) : true) && feat < g.n_store;
        if (!okc) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (dst[r] >= 0) g.C[(int64_t)dst[r] * g.ldc + feat] = acc[j][r] * out_scale;
    }
}

// ---- f16x3 grouped GEMM, 256 x 256 x 32 tiles (layer-0 GEMMs) -----------------------------------------
// The 128 x 128 kernel above is bound by LDS traffic (staging stores + fragment reads ~ the MFMA time).  For the
// two big layer-0 GEMMs (N >= 1024) this variant runs 8 waves as 4 (M) x 2 (N), each wave owning a
// 64 x 128 sub-tile: every A fragment feeds 4 column blocks and every B fragment 2 row blocks, so LDS reads,
// staging stores and L2 traffic per MFMA are halved.  One workgroup per CU (128 KB of LDS, double
// buffered), two waves per SIMD.
constexpr int BM2 = 256, BN2 = 256;
constexpr int L0B_MAXNB = 6;               // most column blocks k_gemm_l0b (below) takes
constexpr int H2_PLANE = BM2 * HBK;        // halves per plane per stage
constexpr int H2_STAGE = 4 * H2_PLANE;     // A_hi, A_lo, B_hi, B_lo
constexpr int GEMM2_THREADS = 512;

template <int NB>
__device__ __forceinline__ void gemm_h2_kloop(f32x16 (&acc)[8], const gf4 *a_src0, const gf4 *a_src1, int tm_h,
                                              int k_valid, int kp_rad, uint32_t smask, int nk, float sa,
                                              const _Float16 *b_src0, const _Float16 *b_src1, int64_t bh_plane,
                                              _Float16 *sm, int wm, int wn)
{
    // smask != 0: iterate only the flagged 32-deep reduction slabs (layer-0 forward, K' order);
    // smask == 0: all nk slabs
    const int tid = threadIdx.x, lane = tid & 63;
    const int srow = tid >> 2, piece = tid & 3;   // staging: rows srow and 128 + srow, 16-B piece
    const int fr = lane & 31, fk = lane >> 5;
    const v4f z4 = v4f{0.f, 0.f, 0.f, 0.f};
    const float tm_inv_h = tm_h ? 1.0f / (float)tm_h : 0.f;
    HStage st;
    auto gload = [&](int kt) {
        int acol = kp_rad ? kp_col(kp_rad, kt) : kt * HBK;
        if (tm_h) {   // tile-major: column kt * 32 = member mm, column block cb; members are 64 * H apart
            const int mm = (int)(((float)(kt * HBK) + 0.5f) * tm_inv_h);
            acol = mm * 64 * tm_h + tm_unit((kt * HBK - mm * tm_h) >> 5, 0, 0);
        }
        const bool ok = kp_rad ? piece * 8 < kp_valid(kp_rad, kt) : kt * HBK + piece * 8 < k_valid;
        const int o = ok ? acol / 4 : 0;
        st.a[0][0] = a_src0[o]; st.a[0][1] = a_src0[o + 1];
        st.a[1][0] = a_src1[o]; st.a[1][1] = a_src1[o + 1];
        if (!ok) { st.a[0][0] = z4; st.a[0][1] = z4; st.a[1][0] = z4; st.a[1][1] = z4; }
        st.bh[0] = *(const gh8 *)(b_src0 + kt * HBK);
        st.bl[0] = *(const gh8 *)(b_src0 + bh_plane + kt * HBK);
        st.bh[1] = *(const gh8 *)(b_src1 + kt * HBK);
        st.bl[1] = *(const gh8 *)(b_src1 + bh_plane + kt * HBK);
    };
    auto lstore = [&](int buf) {
        _Float16 *base = sm + buf * H2_STAGE;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            h8 hi, lo;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float x = (c < 4 ? st.a[ps][0][c & 3] : st.a[ps][1][c & 3]) * sa;
                const _Float16 h = (_Float16)x;
                hi[c] = h;
                lo[c] = (_Float16)(x - (float)h);
            }
            const int off = h_off(srow + 128 * ps, piece);
            *reinterpret_cast<h8 *>(base + off) = hi;
            *reinterpret_cast<h8 *>(base + H2_PLANE + off) = lo;
            *reinterpret_cast<h8 *>(base + 2 * H2_PLANE + off) = st.bh[ps];
            *reinterpret_cast<h8 *>(base + 3 * H2_PLANE + off) = st.bl[ps];
        }
    };
    auto compute = [&](int buf) {
        const _Float16 *base = sm + buf * H2_STAGE;
#pragma unroll
        for (int ks = 0; ks < HBK / 16; ++ks) {
            const int pc = ks * 2 + fk;
            h8 ah[2], al[2];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const int ao = h_off(wm * 64 + rb * 32 + fr, pc);
                ah[rb] = *reinterpret_cast<const h8 *>(base + ao);
                al[rb] = *reinterpret_cast<const h8 *>(base + H2_PLANE + ao);
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int bo = h_off(wn * 128 + nb * 32 + fr, pc);
                const h8 bhi = *reinterpret_cast<const h8 *>(base + 2 * H2_PLANE + bo);
                const h8 blo = *reinterpret_cast<const h8 *>(base + 3 * H2_PLANE + bo);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
                    acc[rb * 4 + nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[rb], bhi, acc[rb * 4 + nb], 0, 0, 0);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
                    acc[rb * 4 + nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[rb], blo, acc[rb * 4 + nb], 0, 0, 0);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
                    acc[rb * 4 + nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[rb], bhi, acc[rb * 4 + nb], 0, 0, 0);
            }
        }
    };
    // stage sequence: set bits of smask (ascending) or 0..nk-1
    const bool masked = smask != 0u;
    const int nact = masked ? __popc(smask) : nk;
    if (nact == 0) return;
    uint32_t rem = smask;
    int seq = 0;
    auto next_stage = [&]() {
        int j;
        if (masked) {
            j = rem ? (int)__builtin_ctz(rem) : 0;
            rem &= rem - 1;
        } else {
            j = seq < nk ? seq : nk - 1;
            ++seq;
        }
        return j;
    };
    gload(next_stage());
    lstore(0);
    gload(nact > 1 ? next_stage() : 0);
    __syncthreads();
    // main loop without branches in the body: the compiler interleaves the conversion VALU / ds_write of
    // the next stage and the global loads of the one after into the MFMA stream of the current stage
    for (int it = 0; it < nact - 1; ++it) {
        const int buf = it & 1;
        if (NB > 0) compute(buf);
        lstore(buf ^ 1);                      // next stage (its loads were issued one iteration ago)
        gload(next_stage());                  // (past the end: harmless re-load of a valid slab)
        __syncthreads();
    }
    if (NB > 0) compute((nact - 1) & 1);
}

template <int EPI>
__global__ __launch_bounds__(GEMM2_THREADS, 2) void k_gemm_h2(GemmArgs g)
{
    extern __shared__ __attribute__((aligned(16))) _Float16 sm2[];

    const int nwg = gridDim.x;
    int id = blockIdx.x;
    {
        const int qd = nwg >> 3, rm = nwg & 7, xcd = id & 7;
        id = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (id >> 3);
    }
    const int col_t0 = id % g.ncol_max;
    const int bb = (id / g.ncol_max) % g.batch;
    int row_t = id / (g.ncol_max * g.batch);

    // row tile -> species (256-row tiles are counted here, the control block holds 128-row tiles)
    const int *ctl = g.ctl;
    int s = 0, cnt = 0;
    for (; s < g.S; ++s) {
        cnt = ctl[CTL_CNT + s];
        const int nt = (cnt + BM2 - 1) / BM2;
        if (row_t < nt) break;
        row_t -= nt;
    }
    if (s >= g.S) return;
    const GemmProblem &pr = g.prob[s];
    const int m0 = row_t * BM2;
    const int n_rows = cnt - m0;
    const int p0 = ctl[CTL_OFF + s] + m0;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // waves w and w + 4 share a SIMD (tools/simdmap.hip): they take DIFFERENT column halves, so that a partly
    // filled tile (5 active column blocks = 3 + 2) loads the four SIMDs equally
    const int wm = wave >> 1, wn = (wave & 1) ^ (wave >> 2);
    const int srow = tid >> 2, piece = tid & 3;
    const int r0 = srow < n_rows ? srow : 0, r1 = srow + 128 < n_rows ? srow + 128 : 0;
    const int64_t s0r = g.a_gather ? (int64_t)g.a_gather[p0 + r0] : (int64_t)(p0 + r0);
    const int64_t s1r = g.a_gather ? (int64_t)g.a_gather[p0 + r1] : (int64_t)(p0 + r1);

    // ---- slab mask of this row tile: OR over its atoms (layer 0 only) ----
    uint32_t tmask = 0u;
    int *s_tab = reinterpret_cast<int *>(sm2 + 2 * H2_STAGE);   // [0] = tile mask, [1..8] = column blocks
    if (g.stage_mask) {
        const int *rows = g.a_gather ? g.a_gather : g.c_scatter;   // sorted position -> atom
        uint32_t mk = g.stage_mask[rows[p0 + r0]] | g.stage_mask[rows[p0 + r1]];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mk |= (uint32_t)__shfl_xor((int)mk, o);
        if (tid == 0) s_tab[0] = 0;
        __syncthreads();
        if (lane == 0) atomicOr(reinterpret_cast<unsigned *>(&s_tab[0]), mk);
        __syncthreads();
        tmask = (uint32_t)s_tab[0];
        if (tmask == 0u && EPI == EPI_SCATTER) return;   // atoms without neighbors: nothing to differentiate
        if (EPI == EPI_SCATTER && g.skinny && __popc(tmask) <= L0B_MAXNB) return;   // k_gemm_l0b's tile
    }

    // ---- columns of this tile ----
    // EPI_SCATTER with a mask: the tile's 8 column blocks are the (8 col_t .. 8 col_t + 7)-th ACTIVE blocks;
    // otherwise block j of the tile is 8 col_t + j
    const bool compact = (EPI == EPI_SCATTER) && g.stage_mask;
    const int ncg = compact ? (__popc(tmask) + 7) >> 3 : 1;
    float vmax = 0.f;
    for (int cg = 0; cg < ncg; ++cg) {
    const int col_t = compact ? cg : col_t0;
    int n0 = col_t * BN2;
    int cbw[4];                 // this wave's column blocks (K' block index), -1 = inactive
    int nb_act;
    if (compact) {
        if (cg > 0) __syncthreads();   // the previous group is done with s_tab and the staging buffers
        if (tid < 8) {
            // slot nb of wave half wn takes the (8 col_t + 2 nb + wn)-th set bit of tmask (alternating, so
            // a partially filled tile is balanced over the two halves)
            const int want = col_t * 8 + 2 * (tid & 3) + (tid >> 2);
            uint32_t m = tmask;
            for (int k = 0; k < want; ++k) m &= m - 1;
            s_tab[1 + tid] = m ? (int)__builtin_ctz(m) : -1;
        }
        __syncthreads();
        nb_act = 0;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            cbw[nb] = s_tab[1 + wn * 4 + nb];
            if (cbw[nb] >= 0) nb_act = nb + 1;
        }
    } else {
        if (n0 >= pr.N) return;
        nb_act = max(0, min(4, (pr.N - n0 - wn * 128) >> 5));
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) cbw[nb] = (n0 >> 5) + wn * 4 + nb;
    }
    const float sa = g.amax_in >= 0 ? amax_scale(g.amax, g.amax_in, s) : g.a_static_scale;
    const float out_scale = pr.w_inv_scale / sa;

    const gf4 *a_src0 = (const gf4 *)(g.A + s0r * g.lda + (int64_t)bb * pr.a_boff + piece * 8);
    const gf4 *a_src1 = (const gf4 *)(g.A + s1r * g.lda + (int64_t)bb * pr.a_boff + piece * 8);
    int tm_h = 0;   // tile-major A (d E / d act0 of the fused kernel): row width of this species
    if (g.a_tm_members > 0) {
        tm_h = g.a_tm_h[s];
        const int64_t base = tm_species_base(ctl, g.a_tm_h, g.a_tm_members, s);
        const int rel0 = m0 + r0, rel1 = m0 + r1;
        // (row, 8 k values of piece p) = unit (cb, row block, k step p >> 1), lane slot (p & 1) * 32 + row
        a_src0 = (const gf4 *)(g.A + base + (int64_t)(rel0 >> 6) * g.a_tm_members * 64 * tm_h +
                               tm_unit(0, (rel0 >> 5) & 1, piece >> 1) + ((piece & 1) * 32 + (rel0 & 31)) * 8);
        a_src1 = (const gf4 *)(g.A + base + (int64_t)(rel1 >> 6) * g.a_tm_members * 64 * tm_h +
                               tm_unit(0, (rel1 >> 5) & 1, piece >> 1) + ((piece & 1) * 32 + (rel1 & 31)) * 8);
    }
    // B rows staged by this thread: tile columns srow and srow + 128
    int bn0, bn1;
    if (compact) {
        const int c0 = s_tab[1 + (srow >> 5)], c1 = s_tab[1 + 4 + (srow >> 5)];
        bn0 = (c0 >= 0 ? c0 : 0) * 32 + (srow & 31);
        bn1 = (c1 >= 0 ? c1 : 0) * 32 + (srow & 31);
    } else {
        bn0 = (n0 + srow < pr.N) ? n0 + srow : n0;
        bn1 = (n0 + srow + 128 < pr.N) ? n0 + srow + 128 : n0;
    }
    const _Float16 *b_src0 = pr.Bh + (int64_t)bb * pr.bh_stride + (int64_t)bn0 * pr.ldbh + piece * 8;
    const _Float16 *b_src1 = pr.Bh + (int64_t)bb * pr.bh_stride + (int64_t)bn1 * pr.ldbh + piece * 8;

    f32x16 acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    const int nk = pr.K / HBK;
    const int fr = lane & 31, fk = lane >> 5;
    const int kp_a = EPI == EPI_BIAS_CELU ? g.kp_rad : 0;               // K' order on the reduction side
    const uint32_t smask = (EPI == EPI_BIAS_CELU && g.stage_mask) ? tmask : 0u;
    if (!(EPI == EPI_BIAS_CELU && g.stage_mask && tmask == 0u)) {
        switch (nb_act) {
            case 4: gemm_h2_kloop<4>(acc, a_src0, a_src1, tm_h, pr.k_valid, kp_a, smask, nk, sa, b_src0, b_src1, pr.bh_plane, sm2, wm, wn); break;
            case 3: gemm_h2_kloop<3>(acc, a_src0, a_src1, tm_h, pr.k_valid, kp_a, smask, nk, sa, b_src0, b_src1, pr.bh_plane, sm2, wm, wn); break;
            case 2: gemm_h2_kloop<2>(acc, a_src0, a_src1, tm_h, pr.k_valid, kp_a, smask, nk, sa, b_src0, b_src1, pr.bh_plane, sm2, wm, wn); break;
            case 1: gemm_h2_kloop<1>(acc, a_src0, a_src1, tm_h, pr.k_valid, kp_a, smask, nk, sa, b_src0, b_src1, pr.bh_plane, sm2, wm, wn); break;
            default: gemm_h2_kloop<0>(acc, a_src0, a_src1, tm_h, pr.k_valid, kp_a, smask, nk, sa, b_src0, b_src1, pr.bh_plane, sm2, wm, wn); break;
        }
    }

#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        if (nb >= nb_act || cbw[nb] < 0) continue;
        const int col = cbw[nb] * 32 + fr;   // column in the (K'-ordered, for layer-0 bwd) output space
        float bias = 0.f;
        if (EPI == EPI_BIAS_CELU) bias = pr.bias[(int64_t)bb * pr.bias_stride + col];
        // layer-0 backward: K' column -> AEV feature
        const int feat = (EPI == EPI_SCATTER && g.kp_rad) ? kp_col(g.kp_rad, cbw[nb]) + fr : col;
        const bool okc = (EPI == EPI_SCATTER && g.kp_rad) ? fr < kp_valid(g.kp_rad, cbw[nb]) : true;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 64 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                if (row >= n_rows) continue;
                float v = acc[rb * 4 + nb][r] * out_scale;
                if (EPI == EPI_BIAS_CELU) {
                    v = celu(v + bias, g.alpha, g.inv_alpha);
                    g.C[(int64_t)(p0 + row) * g.ldc + (int64_t)bb * pr.c_boff + col] = v;
                } else if (EPI == EPI_SCATTER) {
                    if (okc && feat < g.n_store) g.C[(int64_t)g.c_scatter[p0 + row] * g.ldc + feat] = v;
                }
                vmax = fmaxf(vmax, fabsf(v));
            }
    }
    }   // column groups
    if (g.amax_out >= 0) amax_update(g.amax, g.amax_out, s, vmax);
}

// ---- layer-0 backward over FEW flagged slabs (f16x3): 256 rows x (32 NB) columns, NB <= 6 ------------------------
// With slab masks the output of the layer-0 backward is skinny (water: 5 column blocks against K = 8 H1 = 2048), and the
// 256 x 256 kernel above spends its stage on staging and re-reading an A tile that every wave uses for two or three
// column blocks only.  Here the eight waves of a workgroup each own 32 ROWS and ALL flagged column blocks: a wave's A
// operand (d E / d act0, fp32, tile-major) goes from global memory straight into registers -- lane (row, k half) reads
// its 8 consecutive k values, converts them to {hi, lo} fp16 fragments in place (v_cvt_pk_f16_f32 + mixed FMAs) -- and
// only the pre-split B planes (W0 of the flagged slabs, shared by all waves) are staged through LDS: 24 KB per
// 32-deep stage instead of 64 KB, no A stores, no A fragment reads, every wave issues the same 6 NB MFMAs per stage.
// Tiles with more than 6 flagged blocks are left to k_gemm_h2 (GemmArgs.skinny tells that kernel to skip the others).
constexpr int L0B_THREADS = 512;
constexpr int L0B_PLANE = L0B_MAXNB * 32 * HBK;   // halves per B plane per stage
constexpr int L0B_STAGE = 2 * L0B_PLANE;          // B_hi, B_lo

struct L0bB {
    h8 h[2], l[2];   // [row pass]
};

template <int NB>
__device__ __forceinline__ void l0b_tile(const GemmArgs &g, const int (&cbw)[L0B_MAXNB], int n_rows, int p0,
                                         float out_scale, const float *arow, int tm_h, int nk, float sa,
                                         const _Float16 *b_src0, const _Float16 *b_src1, int64_t bh_plane, bool st0,
                                         bool st1, _Float16 *sm)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int srow = tid >> 2, piece = tid & 3;
    const int fr = lane & 31, fk = lane >> 5;
    typedef float v2f_ __attribute__((ext_vector_type(2)));
    typedef _Float16 h2_ __attribute__((ext_vector_type(2)));
    f32x16 acc[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    // A of stage kt: member mm = 32 kt / tm_h, column block (32 kt - mm tm_h) / 32; members are 64 tm_h floats apart
    auto a_off = [&](int kt) {
        const int k = kt * HBK, mm = (int)(((float)k + 0.5f) / (float)tm_h);
        return (int64_t)mm * 64 * tm_h + tm_unit((k - mm * tm_h) >> 5, 0, 0);
    };
    auto aload = [&](v4f (&a)[4], int kt) {   // two coalesced 2-KB wave loads per k step
        const gf4 *p = (const gf4 *)(arow + a_off(kt));
        a[0] = p[0]; a[1] = p[1];        // k step 0: this lane's k = 8 fk .. 8 fk + 7
        a[2] = p[128]; a[3] = p[129];    // k step 1: the next unit (512 floats on)
    };
    auto bload = [&](L0bB &b, int kt) {
        b.h[0] = *(const gh8 *)(b_src0 + kt * HBK);
        b.l[0] = *(const gh8 *)(b_src0 + bh_plane + kt * HBK);
        b.h[1] = *(const gh8 *)(b_src1 + kt * HBK);
        b.l[1] = *(const gh8 *)(b_src1 + bh_plane + kt * HBK);
    };
    auto bstore = [&](const L0bB &b, int buf) {
        _Float16 *base = sm + buf * L0B_STAGE;
        if (st0) {
            const int off = h_off(srow, piece);
            *reinterpret_cast<h8 *>(base + off) = b.h[0];
            *reinterpret_cast<h8 *>(base + L0B_PLANE + off) = b.l[0];
        }
        if (st1) {
            const int off = h_off(srow + 128, piece);
            *reinterpret_cast<h8 *>(base + off) = b.h[1];
            *reinterpret_cast<h8 *>(base + L0B_PLANE + off) = b.l[1];
        }
    };
    // fp32 -> {hi, lo} fp16 fragment of one k step (8 values per lane)
    auto split = [&](const v4f &x0, const v4f &x1, h8 &ahi, h8 &alo) {
#pragma unroll
        for (int c = 0; c < 8; c += 2) {
            const v4f &src = c < 4 ? x0 : x1;
            const v2f_ x = v2f_{src[c & 3], src[(c & 3) + 1]};
            const h2_ h = __builtin_convertvector(x * sa, h2_);
            h2_ l;
            asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x[0]), "v"(sa), "v"(h));
            asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x[1]), "v"(sa), "v"(h));
            ahi[c] = h[0]; ahi[c + 1] = h[1];
            alo[c] = l[0]; alo[c + 1] = l[1];
        }
        // gfx950 needs two wait states between a VALU write of a VGPR and an MFMA that reads it; the compiler inserts them
        // for instructions it knows, not behind inline assembly -- and the first MFMA of a k step reads `alo` (with four
        // column blocks it was scheduled right behind the last v_fma_mixhi and multiplied a stale register: d E / d AEV of
        // the tile's FIRST flagged slab off by ~2 %).  The nop is tied to `alo`, so it stays between the two.
        asm volatile("s_nop 1" : "+v"(alo));
    };
    auto bfrags = [&](const _Float16 *base, int ks, h8 (&bhi)[NB], h8 (&blo)[NB]) {
        const int pc = ks * 2 + fk;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int bo = h_off(nb * 32 + fr, pc);
            bhi[nb] = *reinterpret_cast<const h8 *>(base + bo);
            blo[nb] = *reinterpret_cast<const h8 *>(base + L0B_PLANE + bo);
        }
    };
    auto mfmas = [&](const h8 &ahi, const h8 &alo, const h8 (&bhi)[NB], const h8 (&blo)[NB]) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi[nb], acc[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo[nb], acc[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi[nb], acc[nb], 0, 0, 0);
    };
    // one 32-deep stage: the B fragments of k step 1 are read from LDS before the MFMAs of k step 0
    auto compute = [&](int buf, const v4f (&a)[4]) {
        const _Float16 *base = sm + buf * L0B_STAGE;
        h8 b0h[NB], b0l[NB], b1h[NB], b1l[NB], ahi, alo;
        bfrags(base, 0, b0h, b0l);
        split(a[0], a[1], ahi, alo);
        bfrags(base, 1, b1h, b1l);
        mfmas(ahi, alo, b0h, b0l);
        split(a[2], a[3], ahi, alo);
        mfmas(ahi, alo, b1h, b1l);
    };
    // A: two register sets (the loads of stage kt + 1 travel while stage kt is computed: a stage is ~2 us);
    // B: one register set a stage ahead of the LDS double buffer
    v4f a0[4], a1[4];
    L0bB bs;
    aload(a0, 0);
    bload(bs, 0);
    bstore(bs, 0);
    bload(bs, min(1, nk - 1));
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        // stage kt (buffer 0): A in a0
        aload(a1, min(kt + 1, nk - 1));
        compute(0, a0);
        if (kt + 1 < nk) bstore(bs, 1);
        bload(bs, min(kt + 2, nk - 1));
        __syncthreads();
        if (kt + 1 < nk) {
            aload(a0, min(kt + 2, nk - 1));
            compute(1, a1);
            if (kt + 2 < nk) bstore(bs, 0);
            bload(bs, min(kt + 3, nk - 1));
            __syncthreads();
        }
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        // K' column -> AEV feature
        const int feat = g.kp_rad ? kp_col(g.kp_rad, cbw[nb]) + fr : cbw[nb] * 32 + fr;
        const bool okc = g.kp_rad ? fr < kp_valid(g.kp_rad, cbw[nb]) : true;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
            if (row >= n_rows) continue;
            if (okc && feat < g.n_store) g.C[(int64_t)g.c_scatter[p0 + row] * g.ldc + feat] = acc[nb][r] * out_scale;
        }
    }
}

__global__ __launch_bounds__(L0B_THREADS, 2) void k_gemm_l0b(GemmArgs g)
{
    extern __shared__ __attribute__((aligned(16))) _Float16 sm3[];
    int *s_tab = reinterpret_cast<int *>(sm3 + 2 * L0B_STAGE);   // [0] = tile mask
    const int nwg = gridDim.x;
    int row_t = blockIdx.x;
    {
        const int qd = nwg >> 3, rm = nwg & 7, xcd = row_t & 7;
        row_t = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (row_t >> 3);
    }
    const int *ctl = g.ctl;
    int s = 0, cnt = 0;
    for (; s < g.S; ++s) {
        cnt = ctl[CTL_CNT + s];
        const int nt = (cnt + BM2 - 1) / BM2;
        if (row_t < nt) break;
        row_t -= nt;
    }
    if (s >= g.S) return;
    const GemmProblem &pr = g.prob[s];
    const int m0 = row_t * BM2;
    const int n_rows = cnt - m0;
    const int p0 = ctl[CTL_OFF + s] + m0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int srow = tid >> 2, piece = tid & 3;
    const int fr = lane & 31, fk = lane >> 5;

    // slab mask of the row tile: OR over its atoms
    {
        const int *rows = g.c_scatter;   // sorted position -> atom
        const int q0 = srow < n_rows ? srow : 0, q1 = srow + 128 < n_rows ? srow + 128 : 0;
        uint32_t mk = g.stage_mask[rows[p0 + q0]] | g.stage_mask[rows[p0 + q1]];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mk |= (uint32_t)__shfl_xor((int)mk, o);
        if (tid == 0) s_tab[0] = 0;
        __syncthreads();
        if (lane == 0) atomicOr(reinterpret_cast<unsigned *>(&s_tab[0]), mk);
        __syncthreads();
    }
    const uint32_t tmask = (uint32_t)s_tab[0];
    const int nb_act = __popc(tmask);
    if (nb_act == 0 || nb_act > L0B_MAXNB) return;   // (nothing to differentiate / k_gemm_h2's tile)
    int cbw[L0B_MAXNB];   // K' block index of column block nb
    {
        uint32_t m = tmask;
#pragma unroll
        for (int nb = 0; nb < L0B_MAXNB; ++nb) {
            cbw[nb] = m ? (int)__builtin_ctz(m) : -1;
            m &= m - 1;
        }
    }
    const float sa = g.amax_in >= 0 ? amax_scale(g.amax, g.amax_in, s) : g.a_static_scale;
    const float out_scale = pr.w_inv_scale / sa;

    // this wave's 32 rows = one row block of a 64-atom tile (a wave past the species end re-reads the tile's first
    // row block; rows past the end inside a block are allocated, never stored)
    const int tm_h = g.a_tm_h[s];
    const int rel = m0 + (wave * 32 < n_rows ? wave * 32 : 0);
    const float *arow = g.A + tm_species_base(ctl, g.a_tm_h, g.a_tm_members, s) +
                        (int64_t)(rel >> 6) * g.a_tm_members * 64 * tm_h + tm_unit(0, (rel >> 5) & 1, 0) + lane * 8;
    // B rows staged by this thread: rows srow and 128 + srow of the compacted blocks
    const int blk0 = srow >> 5, blk1 = 4 + (srow >> 5);
    const bool st0 = blk0 < nb_act, st1 = srow < 64 && blk1 < nb_act;
    const int c0 = st0 ? cbw[blk0 < L0B_MAXNB ? blk0 : 0] : cbw[0];
    const int c1 = st1 ? cbw[blk1 < L0B_MAXNB ? blk1 : 0] : cbw[0];
    const _Float16 *b_src0 = pr.Bh + (int64_t)(c0 * 32 + (srow & 31)) * pr.ldbh + piece * 8;
    const _Float16 *b_src1 = pr.Bh + (int64_t)(c1 * 32 + (srow & 31)) * pr.ldbh + piece * 8;

    const int nk = pr.K / HBK;
    switch (nb_act) {
        case 6: l0b_tile<6>(g, cbw, n_rows, p0, out_scale, arow, tm_h, nk, sa, b_src0, b_src1, pr.bh_plane, st0, st1, sm3); break;
        case 5: l0b_tile<5>(g, cbw, n_rows, p0, out_scale, arow, tm_h, nk, sa, b_src0, b_src1, pr.bh_plane, st0, st1, sm3); break;
        case 4: l0b_tile<4>(g, cbw, n_rows, p0, out_scale, arow, tm_h, nk, sa, b_src0, b_src1, pr.bh_plane, st0, st1, sm3); break;
        case 3: l0b_tile<3>(g, cbw, n_rows, p0, out_scale, arow, tm_h, nk, sa, b_src0, b_src1, pr.bh_plane, st0, st1, sm3); break;
        case 2: l0b_tile<2>(g, cbw, n_rows, p0, out_scale, arow, tm_h, nk, sa, b_src0, b_src1, pr.bh_plane, st0, st1, sm3); break;
        default: l0b_tile<1>(g, cbw, n_rows, p0, out_scale, arow, tm_h, nk, sa, b_src0, b_src1, pr.bh_plane, st0, st1, sm3); break;
    }
}

// ---- fused network kernel (f16x3) ---------------------------------------------------------------------
// Run as separate GEMMs the network is bound by the HBM round trips of its activations (~73 KB per atom for
// the hidden layers, 16 KB more for the layer-0 output).  This kernel keeps them on chip: one workgroup
// takes 64 atoms of one species and ONE ensemble member through
//   AEV rows -> L0 -> act0 -> L1 -> act1 -> L2 -> act2 -> output layer (energy)
//            -> d act2 -> d act1 -> d act0                                   (-> layer-0 backward GEMM)
// Layer 0 reads only the AEV slabs the tile's slab mask flags (include/anihip.h): the fp32 slab tiles
// (64 rows x 32 columns) are split into fp16 {hi, lo} planes in LDS, three slabs per barrier, double
// buffered.  Activations / gradients live in LDS as split-fp16 planes with a per-tile power-of-two scale
// (tile max via an LDS atomic); the CELU derivatives of act0 / act1 stay in registers (the wave that
// produces a column block of a layer is the one that needs its derivative on the way back).  The weights
// are pre-packed on the host in MFMA FRAGMENT ORDER ([col block][k step][plane][lane][8 halves]) so every
// wave streams its B operands straight from L2 into registers with fully coalesced 1-KB loads through a
// register ring: no LDS staging of weights, no barriers inside a GEMM phase.  8 waves; wave w owns all 64
// rows (two 32-row MFMA blocks: every weight fragment feeds six MFMAs) and column block w of each phase.
// MFMAs are the three-product v_mfma_f32_32x32x16_f16 of k_gemm_h.
// Two tilings of the same kernel, template <RB, NB>: a wave owns RB 32-row blocks x NB 32-column blocks
// (32 accumulator elements per lane either way):
//   <2, 1>: 64 atoms, 8 waves, 119 KB LDS, one workgroup per CU -- every weight fragment feeds six MFMAs,
//           but MFMA loops and VALU epilogues of the whole CU alternate;
//   <1, 2>: 32 atoms, 4 waves, 60 KB LDS, TWO workgroups per CU that drift out of phase, so the epilogues
//           (VALU) of one overlap the GEMM phases (matrix pipe) of the other, at twice the L2 weight traffic.
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
    float *member_part;        // [n][M] per-member atomic energies (summed by k_fused_finish)
    int S, M;
    int tiles_total;           // upper bound of the number of tiles (work items = tiles_total * M)
    float alpha, inv_alpha;
    int want_grad;
    int l0b;                   // layer-0 backward inside the kernel (needs owner = 1): d E / d AEV -> grad_aev, no d0
    float *grad_aev;           // [n_atoms][L] (l0b): the tile's flagged slabs of its atoms' rows, summed over the members
    int owner;                 // item order: 0 = member-major sweep over the tiles; G > 0: a workgroup OWNS its tiles, taken in groups of G
    unsigned long long *trace;   // development builds (-DANIHIP_DEV_TRACE, tools/fused_trace.py): [item][wave][32] stamps
    // TRAIN instantiation (anihip_mlp_train_forward of a split-fp16 pack): everything the weight gradients need leaves the
    // kernel as fp32 rows in sorted order, member m at columns m * H: the activations of the three hidden layers and
    // d e / d (pre-activation) of layers 2 and 1 for a unit upstream gradient (layer 0's is d0)
    float *tr_act[3];
    float *tr_dlt[3];          // ([0] unused: d0)
    int64_t tr_ld[3];
};
// phase stamps of the fused kernel: compiled out of the shipped library
#ifdef ANIHIP_DEV_TRACE
#define ANIHIP_STAMP(ptr, slot)                                              \
    do {                                                                     \
        unsigned long long *p_ = (ptr);                                      \
        if (p_ && lane == 0) p_[slot] = __builtin_readcyclecounter();        \
    } while (0)
#else
#define ANIHIP_STAMP(ptr, slot) do { } while (0)
#endif

// max of a non-negative value over the wave, the same in every lane: four DPP steps inside the 16-lane rows, then the
// four row maxima through SGPRs (a __shfl_xor butterfly is six dependent ds_bpermute round trips)
__device__ __forceinline__ float wave_max_nonneg(float v)
{
#define ANIHIP_DPP_MAX(ctrl) v = fmaxf(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), ctrl, 0xF, 0xF, true)))
    ANIHIP_DPP_MAX(0xB1);    // quad_perm [1, 0, 3, 2]
    ANIHIP_DPP_MAX(0x4E);    // quad_perm [2, 3, 0, 1]
    ANIHIP_DPP_MAX(0x141);   // row_half_mirror
    ANIHIP_DPP_MAX(0x140);   // row_mirror
#undef ANIHIP_DPP_MAX
    const int x = __float_as_int(v);
    const int m01 = max(__builtin_amdgcn_readlane(x, 0), __builtin_amdgcn_readlane(x, 16));    // (>= 0: integer order)
    const int m23 = max(__builtin_amdgcn_readlane(x, 32), __builtin_amdgcn_readlane(x, 48));
    return __int_as_float(max(m01, m23));
}

__device__ __forceinline__ float pow2_scale_for(float mx)
{
    if (!(mx > 0.f)) return 1.0f;
    const int e = (int)(__float_as_uint(mx) >> 23) - 127;
    return __uint_as_float((unsigned)(127 + 13 - e) << 23);
}

// ---- GEMM machinery of the fused kernel: v_mfma_f32_16x16x32_f16 ----------------------------------------------------------
// Round 6: the fused kernel multiplies on 16 x 16 x 32 MFMAs instead of 32 x 32 x 16.  Same flops per instruction-cycle, same
// operand bytes, but the 32 x 32 x 16 form draws so much more power on real (non-zero mantissa) data that the kernel ran the
// package at its power limit (rocm-smi: 1.33-1.37 kW) with the shader clock pulled down to ~0.8 of its maximum; a pure stream
// of either instruction on random fp16 data sustains 1.29 PFLOP/s (32 x 32 x 16) against 1.87 PFLOP/s (16 x 16 x 32)
// (tools/mfma_power.hip, profiles/r06_mfma_power.txt), and a faithful model of this kernel's item loop runs 23.2 -> 18.3 us
// per item with nothing but the instruction exchanged (tools/pipe_model.hip, profiles/r06_pipeline_model.txt).
//
// Geometry.  A wave's unit is still a 32-column block x 32-row block; it is computed as four 16 x 16 tiles t = 2 ct + rt
// (ct = column half, rt = row half).  The MFMA computes the TRANSPOSED tile (weights are its first operand): lane
// (n16 = lane & 15, c4 = lane >> 4) holds tile row n16 and the four consecutive columns 4 c4 .. 4 c4 + 3, so accumulator
// element r = 4 t + e of a unit is (row 16 rt + n16, column 16 ct + 4 c4 + e) -- runs of four columns, 8-byte LDS stores.
// A k step covers 32 reduction indices ("k2 step" = two of the pack's 16-wide k steps); lane (x16, c4) of either operand
// holds the indices 32 s + 8 c4 .. + 7.  The WEIGHT fragments are read from the pack's unchanged 32 x 16 fragment order
// [N/32][K/16][plane][64 lanes][8 halves] with a different lane -> address map: lane (m, c4) of column half ct takes the
// 16 bytes of old lane 16 ct + m + 32 (c4 & 1) of old k step 2 s + (c4 >> 1) -- four contiguous 256-byte pieces per load
// instruction instead of one kilobyte, the same bytes in total.  The ACTIVATION planes are [row][k] with rows padded by 16
// halves and the 16-byte chunks of every 64-byte group XOR-swizzled with (row >> 2) & 1: conflict-free ds_read_b128 for the
// 16-row fragments of every hidden width (python brute force over the hardware's lane groups).
constexpr int FR_XPAD = 16;       // halves of padding per activation-plane row

typedef v4f Acc16[4];             // the four 16 x 16 tiles of a unit
#define ACC(a, r) (a)[(r) >> 2][(r) & 3]

// Register ring of the weight fragments {hi, lo} x {column half 0, 1} of one column block, D2 k2 steps deep.  Loads are
// unconditional (callers clamp the step): a branch around a load makes hipcc drain the whole ring with s_waitcnt vmcnt(0)
// at every join (CDNA guide, "load everything or hoist the condition").
#ifndef ANIHIP_FR_FENCE
#define ANIHIP_FR_FENCE 1
#endif
#if ANIHIP_FR_FENCE
#define FR_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define FR_FENCE() do { } while (0)
#endif
template <int D2>
struct WRing {
    h8 hi[D2][2], lo[D2][2];
    const _Float16 *base;   // fragment (cb, k2 step 0, plane 0) + this lane's offset (wring_lane_off)
    __device__ __forceinline__ void load(int slot, int s2)   // (slot: compile-time after unrolling)
    {
        const _Float16 *p = base + (int64_t)s2 * (4 * FRAG);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            hi[slot][ct] = *(const gh8 *)(p + ct * 128);
            lo[slot][ct] = *(const gh8 *)(p + ct * 128 + FRAG);
        }
    }
};
// halves from the start of a (column block, even k step) fragment pair to the 16 bytes lane (m, c4) of column half 0 needs
__device__ __forceinline__ int wring_lane_off(int lane)
{
    const int m = lane & 15, c4 = lane >> 4;
    return (c4 >> 1) * (2 * FRAG) + (m + 32 * (c4 & 1)) * 8;
}

// activation fragments {hi, lo} x {row half 0, 1} of one 32-row block for one k2 step
struct AFrag {
    h8 hi[2], lo[2];
    // a = this lane's address in the hi plane for row half 0; + a_plane = lo plane; + rt_stride = row half 1
    __device__ __forceinline__ void load(const _Float16 *a, int a_plane, int rt_stride)
    {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            hi[rt] = *reinterpret_cast<const h8 *>(a + rt * rt_stride);
            lo[rt] = *reinterpret_cast<const h8 *>(a + rt * rt_stride + a_plane);
        }
    }
};

// one k2 step of one row block, three products (twelve MFMAs; consecutive MFMAs write different tiles).
// TWO: the product (weight lo) x (activation hi) is left out -- the weights of this GEMM count as rounded to fp16 (2^-12
// relative): ANIHIP_MLP_FLAG_BWD_TWO_PRODUCTS, backward phases only, off by default
// (leaving out (weight hi) x (x lo) instead -- the gradients rounded, not the weights -- measures the same: max |dF| 6.9e-6
// against 5.0e-6 Ha/A on the headline sample)
// FIRST: the first k2 step of a GEMM -- the accumulators START from the MFMA's zero operand (no v_mov zero fill: 32-48 VALU
// instructions per phase and wave, an eighth of the kernel's vector instructions before round 6)
template <int D2, bool TWO = false, bool FIRST = false>
__device__ __forceinline__ void fr_mfma(Acc16 &acc, const WRing<D2> &rg, int slot, const AFrag &x)
{
    const v4f zero = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
            acc[2 * ct + rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(rg.hi[slot][ct], x.lo[rt], FIRST ? zero : acc[2 * ct + rt], 0, 0, 0);
    if constexpr (!TWO) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
                acc[2 * ct + rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(rg.lo[slot][ct], x.hi[rt], acc[2 * ct + rt], 0, 0, 0);
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
            acc[2 * ct + rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(rg.hi[slot][ct], x.hi[rt], acc[2 * ct + rt], 0, 0, 0);
}

// ring of column block cb of a [N/32][KS] fragment matrix of member m (KS = 16-wide k steps of the pack, even), the first D2
// k2 steps in flight.  EVERY ring register is loaded on every path (a wave without a block: block 0) -- inside the item
// loop of the fused kernel a ring that some path leaves undefined is carried around the loop and holds its registers through
// the whole item
template <int D2>
__device__ __forceinline__ void fr_ring(WRing<D2> &r, const _Float16 *w, int64_t member_halves, int m, int KS, int cb,
                                        int lane, int nblk)
{
    const int KS2 = KS >> 1;
    r.base = w + (int64_t)m * member_halves + (int64_t)(nblk > 0 ? cb : 0) * KS * (2 * FRAG) + wring_lane_off(lane);
#pragma unroll
    for (int sl = 0; sl < D2; ++sl) r.load(sl, min(sl, KS2 - 1));
}

// acc += X[rows, K] x W over all KS2 = K / 32 k2 steps: whole groups of D2 steps without a branch, the tail (<= D2 steps, all
// of them in the ring) issues no loads.  A group requests the D2 steps behind it, clamped to the last one; the group loop stops
// as soon as the ring holds everything that is left.  xa = hi plane of X at the wave's first row block, ldx = row stride
// (halves).  RBA = row blocks of the wave's unit (2, or 1).  The activation fragments of the next row block / k2 step are
// read from LDS before the MFMAs of the current one (two register sets).
template <int RBA, int D2, bool TWO = false>
__device__ __forceinline__ void fr_gemm(Acc16 (&acc)[2], const _Float16 *xa, int ldx, int x_plane, WRing<D2> &rg, int KS2,
                                        int lane)
{
    static_assert(D2 == 2, "the peeled first step and the alternating fragment sets are written for a ring of two k2 steps");
    const int n16 = lane & 15, c4 = lane >> 4;
    const _Float16 *af = xa + n16 * ldx + ((c4 ^ ((n16 >> 2) & 1)) << 3);
    const int rts = 16 * ldx, rbs = 32 * ldx;
    AFrag xe, xo;
    xe.load(af, x_plane, rts);
    // step 0, peeled: it WRITES the accumulators (fr_mfma<FIRST>), the callers do not zero them.  Behind it the ring slot of
    // step k is (k % D2): the loops below start at k0 = 1 and walk the slots 1, 0.
    // (the scheduler sinks the fragment reads to their first use and waits for each of them between the MFMAs: fences keep
    // the reads of the NEXT half step ahead of the twelve MFMAs of this one)
    int k0 = 1;
    if constexpr (RBA == 2) {
        xo.load(af + rbs, x_plane, rts);
        FR_FENCE();
        fr_mfma<D2, TWO, true>(acc[0], rg, 0, xe);
        FR_FENCE();
        xe.load(af + min(1, KS2 - 1) * 32, x_plane, rts);
        FR_FENCE();
        fr_mfma<D2, TWO, true>(acc[1], rg, 0, xo);
        rg.load(0, min(D2, KS2 - 1));
        FR_FENCE();
        for (; k0 + D2 < KS2; k0 += D2) {
#pragma unroll
            for (int sl = 0; sl < D2; ++sl) {
                const int slot = (sl + 1) % D2;
                xo.load(af + rbs + (k0 + sl) * 32, x_plane, rts);
                FR_FENCE();
                fr_mfma<D2, TWO>(acc[0], rg, slot, xe);
                FR_FENCE();
                xe.load(af + (k0 + sl + 1) * 32, x_plane, rts);
                FR_FENCE();
                fr_mfma<D2, TWO>(acc[1], rg, slot, xo);
                rg.load(slot, min(k0 + sl + D2, KS2 - 1));
                FR_FENCE();
            }
        }
        const int rem = KS2 - k0;   // (0 .. D2 steps, all of them in the ring)
#pragma unroll
        for (int sl = 0; sl < D2; ++sl) {
            if (rem > sl) {
                const int slot = (sl + 1) % D2;
                xo.load(af + rbs + (k0 + sl) * 32, x_plane, rts);
                FR_FENCE();
                fr_mfma<D2, TWO>(acc[0], rg, slot, xe);
                FR_FENCE();
                xe.load(af + min(k0 + sl + 1, KS2 - 1) * 32, x_plane, rts);
                FR_FENCE();
                fr_mfma<D2, TWO>(acc[1], rg, slot, xo);
                FR_FENCE();
            }
        }
    } else {
        // one row block: the fragment sets alternate with the steps (xe: even steps, xo: odd steps)
        xo.load(af + min(1, KS2 - 1) * 32, x_plane, rts);
        FR_FENCE();
        fr_mfma<D2, TWO, true>(acc[0], rg, 0, xe);
        rg.load(0, min(D2, KS2 - 1));
        FR_FENCE();
        for (; k0 + D2 < KS2; k0 += D2) {   // (steps k0 (odd: xo, slot 1) and k0 + 1 (even: xe, slot 0))
            xe.load(af + (k0 + 1) * 32, x_plane, rts);
            FR_FENCE();
            fr_mfma<D2, TWO>(acc[0], rg, 1, xo);
            rg.load(1, min(k0 + D2, KS2 - 1));
            FR_FENCE();
            xo.load(af + min(k0 + 2, KS2 - 1) * 32, x_plane, rts);
            FR_FENCE();
            fr_mfma<D2, TWO>(acc[0], rg, 0, xe);
            rg.load(0, min(k0 + 1 + D2, KS2 - 1));
            FR_FENCE();
        }
        const int rem = KS2 - k0;
        if (rem > 0) {
            xe.load(af + min(k0 + 1, KS2 - 1) * 32, x_plane, rts);
            FR_FENCE();
            fr_mfma<D2, TWO>(acc[0], rg, 1, xo);
            FR_FENCE();
            if (rem > 1) fr_mfma<D2, TWO>(acc[0], rg, 0, xe);
        }
    }
}

// The same for ONE 16-column half of a column block and all 64 rows (phase 5: eight halves of four slabs on eight waves): ring
// of {hi, lo} of the one column half, four tiles t = 2 rb + rt.  base already points at the column half (+ ct * 128).
template <int D2>
struct WRingHalf {
    h8 hi[D2], lo[D2];
    const _Float16 *base;
    __device__ __forceinline__ void load(int slot, int s2)
    {
        const _Float16 *p = base + (int64_t)s2 * (4 * FRAG);
        hi[slot] = *(const gh8 *)p;
        lo[slot] = *(const gh8 *)(p + FRAG);
    }
};
template <int D2, bool TWO = false>
__device__ __forceinline__ void fr_gemm_half(v4f (&acc)[4], const _Float16 *xa, int ldx, int x_plane, WRingHalf<D2> &rg, int KS2,
                                             int lane)
{
    const int n16 = lane & 15, c4 = lane >> 4;
    const _Float16 *af = xa + n16 * ldx + ((c4 ^ ((n16 >> 2) & 1)) << 3);
    const int rts = 16 * ldx, rbs = 32 * ldx;
    AFrag xe, xo;
    auto mm = [&](int rb, int slot, const AFrag &x, auto first_) {
        constexpr bool FIRST = decltype(first_)::value;
        const v4f zero = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
            acc[2 * rb + rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(rg.hi[slot], x.lo[rt], FIRST ? zero : acc[2 * rb + rt], 0, 0, 0);
        if constexpr (!TWO) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
                acc[2 * rb + rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(rg.lo[slot], x.hi[rt], acc[2 * rb + rt], 0, 0, 0);
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
            acc[2 * rb + rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(rg.hi[slot], x.hi[rt], acc[2 * rb + rt], 0, 0, 0);
    };
    xe.load(af, x_plane, rts);
    // step 0, peeled: it writes the accumulators (no zero fill); the loops below start at k0 = 1 with the slots 1, 0, ...
    xo.load(af + rbs, x_plane, rts);
    FR_FENCE();
    mm(0, 0, xe, std::true_type{});
    FR_FENCE();
    xe.load(af + min(1, KS2 - 1) * 32, x_plane, rts);
    FR_FENCE();
    mm(1, 0, xo, std::true_type{});
    rg.load(0, min(D2, KS2 - 1));
    FR_FENCE();
    int k0 = 1;
    for (; k0 + D2 < KS2; k0 += D2) {
#pragma unroll
        for (int sl = 0; sl < D2; ++sl) {
            const int slot = (sl + 1) % D2;
            xo.load(af + rbs + (k0 + sl) * 32, x_plane, rts);
            FR_FENCE();
            mm(0, slot, xe, std::false_type{});
            FR_FENCE();
            xe.load(af + (k0 + sl + 1) * 32, x_plane, rts);
            FR_FENCE();
            mm(1, slot, xo, std::false_type{});
            rg.load(slot, min(k0 + sl + D2, KS2 - 1));
            FR_FENCE();
        }
    }
    const int rem = KS2 - k0;   // (0 .. D2 steps, all of them in the ring)
#pragma unroll
    for (int sl = 0; sl < D2; ++sl) {
        if (rem > sl) {
            const int slot = (sl + 1) % D2;
            xo.load(af + rbs + (k0 + sl) * 32, x_plane, rts);
            FR_FENCE();
            mm(0, slot, xe, std::false_type{});
            FR_FENCE();
            xe.load(af + min(k0 + sl + 1, KS2 - 1) * 32, x_plane, rts);
            FR_FENCE();
            mm(1, slot, xo, std::false_type{});
            FR_FENCE();
        }
    }
}

// layer 0 of the fused kernel: the k2 steps of one pair of staging slots (2 slots x FR_GROUP slabs: a slab IS one k2 step; ring
// slot = step % D2).  s0 / s1 = this lane's fragment address (row half 0 of the wave's first row block) in the two slots; the
// steps of the slabs past the tile's last flagged one (n_live of the pair's 2 FR_GROUP are live) have zero operands: no
// MFMAs, the ring request stays unconditional.
// STEPS / REQS: the general form walks all 2 FR_GROUP steps and requests a fragment behind every one of them (clamped repeats
// once the tile's slabs are used up); a tile with at most FOUR flagged slabs -- every tile of a water box -- has 4 steps, D2 of
// them in the ring when the loop starts: the short form walks 4 steps and requests 4 - D2.
// KEPT: the operand is the tile's kept copy (FusedCfg::SLABU layout: unpadded rows, swizzled chunks) -- s0 = this lane's
// fragment address in slab 0
template <int RBA, int D2, int ROWS, int STEPS, int REQS, bool KEPT = false, bool FIRST = false, class NextS2>
__device__ __forceinline__ void fr_l0_pair(Acc16 (&acc)[2], WRing<D2> &rg, const _Float16 *s0, const _Float16 *s1, int n_live,
                                           NextS2 &&next_s2)
{
    static_assert(STEPS <= 2 * FR_GROUP && REQS <= STEPS && STEPS <= D2 + REQS, "ring coverage");
    constexpr int SLAB = KEPT ? 2 * ROWS * 32 : 2 * ROWS * FR_SLAB_LD, PL = KEPT ? ROWS * 32 : ROWS * FR_SLAB_LD,
                  RTS = KEPT ? 16 * 32 : 16 * FR_SLAB_LD, RBS = 2 * RTS;
    auto addr = [&](int st) {
        if (KEPT) return s0 + st * SLAB;
        return (st / FR_GROUP ? s1 : s0) + (st % FR_GROUP) * SLAB;
    };
    AFrag xe, xo;
    xe.load(addr(0), PL, RTS);
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
        // (FIRST: the tile's first flagged slab -- n_live >= 1 -- writes the accumulators, the caller does not zero them)
        const bool live = st < n_live;
        constexpr bool FST_ = FIRST;
        if constexpr (RBA == 2) {
            xo.load(addr(st) + RBS, PL, RTS);
            if (FST_ && st == 0) fr_mfma<D2, false, true>(acc[0], rg, 0, xe);
            else if (live) fr_mfma<D2>(acc[0], rg, st % D2, xe);
            if (st + 1 < STEPS) xe.load(addr(st + 1), PL, RTS);
            if (FST_ && st == 0) fr_mfma<D2, false, true>(acc[1], rg, 0, xo);
            else if (live) fr_mfma<D2>(acc[1], rg, st % D2, xo);
        } else {
            AFrag &xc = (st & 1) ? xo : xe, &xn = (st & 1) ? xe : xo;
            if (st + 1 < STEPS) xn.load(addr(st + 1), PL, RTS);
            if (FST_ && st == 0) fr_mfma<D2, false, true>(acc[0], rg, 0, xc);
            else if (live) fr_mfma<D2>(acc[0], rg, st % D2, xc);
        }
        if (st < REQS) rg.load(st % D2, next_s2());
    }
}

// What a wave of the fused kernel computes in a phase that produces H = 32 nb columns: column blocks cb, cb + NW, ...
// (nba of them) of the row blocks rb0 .. rb0 + nrb - 1 of the tile.  With 64-row tiles (RB = 2, NB = 1) and 5 to 7
// column blocks, the 2 nb (row block, column block) units are dealt so that the four SIMDs get the same number:
// waves w and w + 4 share a SIMD (tools/simdmap.hip), and "wave w takes column block w" leaves SIMDs 0 / 1 with four
// units and SIMDs 2 / 3 with two when nb = 6 -- every phase, MFMA loop and epilogue alike, then runs at the pace of
// the full SIMDs.  The first 2 nb - 8 waves keep a whole column block (both row blocks: every weight fragment feeds
// six MFMAs), the others take one row block of one of the remaining column blocks: 3 + 3 + 3 + 3 units for nb = 6,
// 3 + 3 + 2 + 2 for nb = 5, 4 + 4 + 3 + 3 for nb = 7.
struct FusedUnit {
    int cb, rb0, nrb, nba;
};
template <int RB, int NB>
__device__ __forceinline__ FusedUnit fused_unit(int H, int wave)
{
    const int nb = H >> 5;
    if constexpr (RB == 2 && NB == 1) {
        const bool deal = nb > 4 && nb < 8;
        const int whole = deal ? 2 * nb - 8 : nb;   // waves that keep both row blocks of a column block
        if (wave < whole) return FusedUnit{wave, 0, 2, 1};
        const int idx = wave - whole, cb = whole + (idx >> 1);
        if (deal && cb < nb) return FusedUnit{cb, idx & 1, 1, 1};
        return FusedUnit{0, 0, 0, 0};
    } else {
        constexpr int NW = 8 / NB;
        const int t = nb - wave, n = t <= 0 ? 0 : (t + NW - 1) / NW;
        return FusedUnit{wave, 0, n > 0 ? RB : 0, n};
    }
}

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

// ---- small inputs: the whole preparation in ONE launch ------------------------------------------------------------
// Below SMALL_PREP_MAX atoms a step is bound by the number of dependent launches, not by work.  Block 0 (16 waves) does what
// zero_words + k_sp_count + k_sp_offsets + k_sp_scatter + k_tile_table do in five launches: every wave loads its
// contiguous chunk of species (and slab flags) in one go and counts, the counts are scanned through LDS, and every atom
// goes from its register straight to its sorted position (same stable order as the chunked kernels: index order inside
// a species): permutation, row of its tile, the tile's slab flags (an LDS OR); one thread per tile then writes the table
// entry.  Three barriers, ~10 us.  Blocks 1.. zero the rows of the padding atoms (k_zero_padding).
constexpr int SMALL_PREP_MAX = 16384;
constexpr int SMALL_PREP_WAVES = 16;
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

// wave-uniform dispatch on the active part of a wave's accumulators in a phase (compile-time inside)
#define FR_UNIT(u, CALL)                                                                    \
    if ((u).nrb == RB && (u).nba >= NB) { constexpr int RBA = RB, NBA = NB; CALL; }         \
    else if ((u).nrb == RB && (u).nba == 1) { constexpr int RBA = RB, NBA = 1; CALL; }      \
    else if ((u).nrb == 1) { constexpr int RBA = 1, NBA = 1; CALL; }

// L0B: the layer-0 backward as phase 5 of the kernel (owner order, d E / d AEV accumulated in place; RB = 2, NB = 1 only)
// TRAIN: the forward half of a training step -- the hidden activations and the backward's per-layer gradients are also
// written to global memory (FusedArgs::tr_*), from the registers of the epilogues that produce them
// B2: the backward GEMMs (phases 3, 4, 5) with two products instead of three (ANIHIP_MLP_FLAG_BWD_TWO_PRODUCTS)
template <int RB, int NB, int ACT, bool L0B, bool TRAIN = false, bool B2 = false>   // ACT: 0 = CELU(alpha), 1 = GELU (exact, erf)
__global__ __launch_bounds__(64 * (8 / NB), 2) void k_mlp_fused(FusedArgs g)
{
    static_assert(!B2 || (L0B && !TRAIN && ACT == 0), "the two-product backward exists for the large-system CELU instantiation");
    static_assert(!L0B || (RB == 2 && NB == 1), "phase 5 is written for 64-row tiles on 8 waves");
    static_assert(!TRAIN || (!L0B && ACT == 0 && NB == 1), "the training instantiation: CELU, d act0 to global memory");
    using C = FusedCfg<RB, NB>;
    static_assert(NB == 1, "the 16 x 16 x 32 machinery is written for one column block per wave");
    constexpr int NW = C::NW, ROWS = C::ROWS, D = C::DEPTH, SLAB = C::SLAB, NE = RB * NB;
    typedef WRing<D> Ring;
    extern __shared__ __attribute__((aligned(16))) _Float16 fsm_all[];
    unsigned *s_tab = reinterpret_cast<unsigned *>(fsm_all);
    unsigned &s_max = s_tab[0];
    long long *s_tmb = reinterpret_cast<long long *>(s_tab + 4);          // [8] tile-major base of species s in d0
    int *s_off = reinterpret_cast<int *>(s_tab + 4 + 16);                 // [8] first sorted position of species s
    float *s_e = reinterpret_cast<float *>(s_tab + 4 + 32);               // [NW][ROWS]
    int *s_orow = reinterpret_cast<int *>(s_tab + 4 + 32 + NW * ROWS);    // [2][ROWS] atom of every row of this / the next item's tile
    _Float16 *slot0 = fsm_all + C::FIXED_BYTES / 2;                       // staging slot 0
    _Float16 *fsm = fsm_all + C::FIXED_HALVES;                            // X1 | XU; staging slots 1..3 overlay
    auto slot = [&](int k) { return k == 0 ? slot0 : fsm + (k - 1) * (FR_GROUP * SLAB); };

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (in an SGPR: the unit tests below are scalar branches)
    // (re-derived from an opaque copy of threadIdx.x at the head of every item: hoisted out of the item loop, the
    // per-lane addresses built from these cost more registers than the kernel has)
    int tid = threadIdx.x, lane = tid & 63;
    int n16 = lane & 15, c4 = lane >> 4;   // this lane in an MFMA tile: row n16, k chunk / column run c4
    // staging role of this thread: row srow, 16-B piece spc (4 of a slab's 32 columns)
    int srow = tid >> 3, spc = tid & 7;
    const int KS0 = g.n_slabs * 2;
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));

    // the d0 scale of the layer-0 backward GEMM comes from the weight-norm bounds too (amax stage 5)
    if (blockIdx.x == 0 && tid < g.S) {
        float b = 0.f;
        for (int mm = 0; mm < g.M; ++mm) b = fmaxf(b, g.sp[tid].bounds[8 * mm + 4]);
        g.amax[(5 * MAX_S + tid) * AMAX_SLOTS] = __float_as_uint(b);
    }

    // ---- AEV slab fetch / staging (layer-0 A operand) ----
    const float *arow = nullptr;   // this thread's AEV row (+ its 16-B piece)
    uint32_t rem_a = 0u;           // slabs not yet fetched
    auto fetch_group = [&](v4f (&v)[FR_GROUP]) {   // next FR_GROUP flagged slabs -> registers (zeros past the end)
#pragma unroll
        for (int j = 0; j < FR_GROUP; ++j) {
            const bool live = rem_a != 0u;
            const int slab = live ? (int)__builtin_ctz(rem_a) : 0;
            rem_a &= rem_a - 1u;
            const int c0 = g.kp_rad ? kp_col(g.kp_rad, slab) : 32 * slab;
            const int nv = g.kp_rad ? kp_valid(g.kp_rad, slab) : min(32, (int)g.L - 32 * slab);
            const bool ok = live && spc * 4 < nv;
            v[j] = *(const gf4 *)(arow + (ok ? c0 : 0));
            if (!ok) v[j] = v4f{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_group = [&](const v4f (&v)[FR_GROUP], _Float16 *buf) {   // registers -> split planes of the staged slabs
#pragma unroll
        for (int j = 0; j < FR_GROUP; ++j) {
            h4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // static scale 4 of the layer-0 input (include/anihip.h)
                const _Float16 h = (_Float16)(v[j][e] * 4.0f);
                hi[e] = h;
                lo[e] = (_Float16)__builtin_fmaf(v[j][e], 4.0f, -(float)h);
            }
            _Float16 *d = buf + j * SLAB + srow * FR_SLAB_LD + spc * 4;
            *reinterpret_cast<h4 *>(d) = hi;
            *reinterpret_cast<h4 *>(d + ROWS * FR_SLAB_LD) = lo;
        }
    };

    // the same into the tile's KEPT copy (C::SLABU layout), slabs base .. base + 2 of the tile's first KEEP_SLABS
    auto store_kept = [&](const v4f (&v)[FR_GROUP], int base) {
#pragma unroll
        for (int j = 0; j < FR_GROUP; ++j) {
            if (base + j >= C::KEEP_SLABS) continue;
            h4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const _Float16 h = (_Float16)(v[j][e] * 4.0f);
                hi[e] = h;
                lo[e] = (_Float16)__builtin_fmaf(v[j][e], 4.0f, -(float)h);
            }
            _Float16 *d = slot0 + (base + j) * C::SLABU + srow * 32 + ((((spc >> 1) ^ (0 - (srow >> 2))) & 3) << 3) + (spc & 1) * 4;
            *reinterpret_cast<h4 *>(d) = hi;
            *reinterpret_cast<h4 *>(d + ROWS * 32) = lo;
        }
    };

    // ---- persistent workgroups: item = blockIdx.x, + gridDim.x, ... (member-major order: at any time the chip
    // works on one or two members, whose weights stay resident in every XCD's L2).  The dependent chain at the
    // head of an item (tile entry -> atom rows -> AEV slabs -> first weight fragments, about 9 k clocks of pure
    // latency when exposed) is issued for item i + 1 while the backward phases of item i run.
    int n_tiles = 0;   // the non-empty tiles come first in the table
    for (int t = 0; t < g.S; ++t) n_tiles += (g.ctl[CTL_CNT + t] + ROWS - 1) / ROWS;
    const int n_items = n_tiles * g.M;
    int item = blockIdx.x;
    if (item >= (g.owner ? n_tiles : n_items)) return;
#ifdef ANIHIP_YOUNG_PRIO
    // the second-dispatched half of the workgroup loses the issue arbitration of its SIMD on every segment: static priority
    if ((wave >= 4) == (ANIHIP_YOUNG_PRIO == 1)) __builtin_amdgcn_s_setprio(1);
#endif
    // per-species constants of the items: looked up from LDS at the head of an item instead of scalar-load chains
    if (threadIdx.x < MAX_S) {
        const int t_ = threadIdx.x;
        long long b = 0;
        for (int t = 0; t < t_ && t < g.S; ++t) b += (long long)((g.ctl[CTL_CNT + t] + 63) >> 6) * 64 * g.M * g.sp[t].H1;
        s_tmb[t_] = b;
        s_off[t_] = t_ < g.S ? g.ctl[CTL_OFF + t_] : 0;
    }
    if (threadIdx.x == 0) s_max = 0u;   // (tile maximum: reset again by every item once it has been read)
    __syncthreads();
    // (member, tile) of the item, advanced by gridDim.x tiles per step without divisions
    int mem = item / n_tiles, tile = item - mem * n_tiles;
    // owner order: the workgroup's tiles b, b + grid, ... can be taken in GROUPS of g.owner of them -- member after member
    // over the tiles of a group -- so that a member's weights enter the XCD's L2 once per group and member instead of once
    // per tile and member.  Measured at the headline size (groups of 1 / 2 / 4 / 8): the same time (30.2-30.6 ms), and with
    // groups of 4 MORE counted fetches, 14.0 instead of 12.2 GB per launch -- the tile's AEV slabs and d E/d AEV rows
    // come back after four items instead of one and find less of themselves in L2.  Default: groups of one tile.
    int gj = 0, gsz = 1;   // position inside the group, tiles of the group
    if (g.owner) {
        mem = 0; tile = item; item = tile * g.M;
        gsz = min(g.owner, (n_tiles - 1 - tile) / (int)gridDim.x + 1);
    }
    typedef Ring Ring0;
    Ring0 rg;                  // layer-0 weight ring of the item being started
    uint32_t rem_w = 0u;       // slabs (= k2 steps) of the layer-0 weight ring not yet requested
    uint32_t tmask = 0u;
    auto next_s2 = [&]() {    // slab (= k2 step in the slab order of W0) of the next ring request, clamped to the last flagged one
        const int slab = rem_w ? (int)__builtin_ctz(rem_w) : (31 - (int)__builtin_clz(tmask | 1u));
        rem_w &= rem_w - 1u;
        return slab;
    };
    v4f va[FR_GROUP], vb[FR_GROUP];
    // slabs 0..5 of an item -> registers (rem_a = the rest)
    auto prefetch_aev = [&](const int4 &t, int atom) {
        arow = g.aev + (int64_t)atom * g.L + spc * 4;
        rem_a = (uint32_t)t.w;
        fetch_group(va);
        fetch_group(vb);
    };
    // (an item that needs no slabs still DEFINES the registers: a path that leaves them undefined would carry the previous
    // item's values around the item loop and hold 24 VGPRs through every phase)
    auto no_aev = [&]() {
#pragma unroll
        for (int j = 0; j < FR_GROUP; ++j) {
            va[j] = v4f{0.f, 0.f, 0.f, 0.f};
            vb[j] = v4f{0.f, 0.f, 0.f, 0.f};
        }
    };
    // first D weight fragments of layer 0 of an item
    auto prefetch_w0 = [&](const int4 &t, int m) {
        const int s = t.x;
        const FusedSpecies &fs = g.sp[s];
        tmask = (uint32_t)t.w;
        rem_w = tmask;
        // every ring register is written on every path (a wave without a block: block 0), or the ring of the previous item
        // would stay live through the whole item
        const FusedUnit u = fused_unit<RB, NB>(fs.H1, wave);
        const _Float16 *wm = fs.w0 + (int64_t)m * (fs.H1 >> 5) * KS0 * (2 * FRAG) + wring_lane_off(lane);
        rg.base = wm + (int64_t)u.cb * KS0 * (2 * FRAG);
#pragma unroll
        for (int sl = 0; sl < D; ++sl) rg.load(sl, next_s2());
    };
    int4 te = g.tile_tab[tile];
    // phase 5, tiles whose flagged slabs fit ONE pass (every tile of a water box), owner order over single tiles: the members'
    // d E / d AEV of this wave's 16 columns x 64 rows stay in these sixteen registers from the first member to the last, which
    // stores them -- one store per tile instead of a read-add-write per member (7 x 512 B read and 7 x 512 B written less per
    // atom and tile; the same additions in the same order: bit-identical to the read-add-write)
    v4f gsum[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) gsum[t] = v4f{0.f, 0.f, 0.f, 0.f};
    int staged_tile = -1;   // the tile whose layer-0 operand slot 0 keeps (L0B, <= KEEP_SLABS flagged slabs), or -1
    int par = 0;   // which half of s_orow holds the current item's rows
    {
        const int atom0 = g.tile_rows[(size_t)tile * ROWS + srow];
        if (spc == 0) s_orow[srow] = atom0;   // (read behind the barriers of the item's phases)
        prefetch_aev(te, atom0);
        prefetch_w0(te, mem);
    }
    for (;;) {
        asm volatile("" : "+v"(tid));
        lane = tid & 63; n16 = lane & 15; c4 = lane >> 4; srow = tid >> 3; spc = tid & 7;
        float alpha = g.alpha, inv_alpha = g.inv_alpha;   // (same reason: their vector copies and products)
        int Mi = g.M;
        asm volatile("" : "+s"(alpha), "+s"(inv_alpha), "+s"(Mi));
        // entry and atom rows of the next item (the last item of a workgroup prefetches itself again: loads
        // stay unconditional)
        int mem_n = mem, tile_n = tile + (int)gridDim.x;
        bool has_next;
        int gj_n = gj, gsz_n = gsz;
        if (g.owner) {   // the group's next tile, then the group's first tile with the next member, then the next group
            mem_n = mem; tile_n = tile + (int)gridDim.x; gj_n = gj + 1;
            if (gj_n >= gsz) {
                gj_n = 0;
                if (mem + 1 < Mi) {
                    mem_n = mem + 1; tile_n = tile - (gsz - 1) * (int)gridDim.x;
                } else {
                    mem_n = 0;   // (tile_n is the first tile behind the group)
                    gsz_n = min(g.owner, (n_tiles - 1 - tile_n) / (int)gridDim.x + 1);
                }
            }
            has_next = tile_n < n_tiles;
        } else {
            while (tile_n >= n_tiles) { tile_n -= n_tiles; ++mem_n; }
            has_next = mem_n < Mi;
        }
        if (!has_next) { mem_n = mem; tile_n = tile; }
        const int4 te_n = g.tile_tab[tile_n];
        const int atom_n = g.tile_rows[(size_t)tile_n * ROWS + srow];
#ifdef ANIHIP_DEV_TRACE
        if (g.trace && lane == 0) {
            g.trace[((size_t)item * 8 + wave) * 32 + 0] = __builtin_readcyclecounter();
            // placement: HW_REG_HW_ID (cu / sh / se) and HW_REG_XCC_ID, for co-residency analysis
            g.trace[((size_t)item * 8 + wave) * 32 + 14] = 1 + (((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4)) |
                                                   ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32));
        }
#endif
        const int m = mem, s = te.x, n_rows = te.z, p0 = te.y;
        const FusedSpecies &fs = g.sp[s];
        const int64_t tm_base = s_tmb[s];
        const int rel_tile = p0 - s_off[s];
        const int H1 = fs.H1, H2 = fs.H2, H3 = fs.H3;
        // LDS carve (halves): X1 planes [2][ROWS][H2+16] | XU = max(X0 planes [2][ROWS][H1+16], X2 planes)
        const int ld0 = H1 + FR_XPAD, ld1 = H2 + FR_XPAD, ld2 = H3 + FR_XPAD;
        const int x0_plane = ROWS * ld0, x1_plane = ROWS * ld1, x2_plane = ROWS * ld2;
        _Float16 *X1 = fsm;
        _Float16 *XU = fsm + 2 * ROWS * ld1;
        _Float16 *X0 = XU, *X2 = XU;
        // this wave's part of the phases producing H1 / H2 / H3 columns (fused_unit)
        const FusedUnit u1 = fused_unit<RB, NB>(H1, wave), u2 = fused_unit<RB, NB>(H2, wave), u3 = fused_unit<RB, NB>(H3, wave);
        // accumulator element (rb, r = 4 q + e) of this lane, q = 2 ct + rt  <->  tile row urow(u, rb, q), column ucol(u, q) + e
        auto urow = [&](const FusedUnit &u, int rb, int q) { return (u.rb0 + rb) * 32 + 16 * (q & 1) + n16; };
        auto ucol = [&](const FusedUnit &u, int q) { return u.cb * 32 + 16 * (q >> 1) + 4 * c4; };
        // where the run of four columns ucol(u, q) .. + 3 lies in a row of the activation planes (halves): the 16-byte chunks of
        // every 64-byte group are XOR-swizzled with (row >> 2) & 1 (the same for both row halves: they are 16 rows apart)
        auto xcol = [&](const FusedUnit &u, int q) {
            return u.cb * 32 + (((2 * (q >> 1) + (c4 >> 1)) ^ ((n16 >> 2) & 1)) << 3) + (c4 & 1) * 4;
        };
#ifdef ANIHIP_DEV_TRACE
        unsigned long long *trace = g.trace ? g.trace + ((size_t)item * 8 + wave) * 32 : nullptr;
#endif
        ANIHIP_STAMP(trace, 1);

        Acc16 acc[NE];
        auto zero_acc = [&]() {
#pragma unroll
            for (int i = 0; i < NE; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) ACC(acc[i], r) = 0.f;
        };
        auto tile_max = [&](float vmax) {  // workgroup max of a non-negative value (s_max was reset at the head of the item)
            vmax = wave_max_nonneg(vmax);
            if (lane == 0) atomicMax(&s_max, __float_as_uint(vmax));
            __syncthreads();
            return __uint_as_float(s_max);
        };
        // acc * scale -> split planes of X (row stride ldx): this lane's runs of 4 columns of its unit
        auto put_acc = [&](_Float16 *X, int plane, int ldx, float scale, const FusedUnit &u) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                if (rb >= u.nrb || u.nba < 1) continue;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    h4 hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        // hi = fp16(x * scale) for two elements at once (v_pk_mul_f32, v_cvt_pk_f16_f32),
                        // lo = fp16(x * scale - hi) as one mixed-precision FMA each (the fp16 hi is an operand)
                        typedef float v2f_ __attribute__((ext_vector_type(2)));
                        typedef _Float16 h2_ __attribute__((ext_vector_type(2)));
                        const v2f_ x = v2f_{ACC(acc[rb], 4 * q + e), ACC(acc[rb], 4 * q + e + 1)};
                        const h2_ h = __builtin_convertvector(x * scale, h2_);
                        hi[e] = h[0];
                        hi[e + 1] = h[1];
                        // (written out: left to itself hipcc converts hi back to fp32 and packs again, 2 more
                        // instructions per pair)
                        h2_ l;
                        asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]"
                            : "=v"(l) : "v"(x[0]), "v"(scale), "v"(h));
                        asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                            : "+v"(l) : "v"(x[1]), "v"(scale), "v"(h));
                        lo[e] = l[0];
                        lo[e + 1] = l[1];
                    }
                    _Float16 *d = X + urow(u, rb, q) * ldx + xcol(u, q);
                    *reinterpret_cast<h4 *>(d) = hi;
                    *reinterpret_cast<h4 *>(d + plane) = lo;
                }
            }
        };
        // 16 per-column parameters per block of this lane (bias / output weights), as float4 loads
        // (v[nb][4 q + e] = the parameter of the column of accumulator element 4 q + e: the two row halves of a column half share
        // their four columns, so these are two float4 loads and eight registers)
        auto load_cols = [&](const float *base, float (&v)[NB][16], const FusedUnit &u) {   // (a wave without a block: block 0, unused)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const v4f t = *(const gf4 *)(base + (nb < u.nba ? u.cb * 32 : 0) + 16 * ct + 4 * c4);
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[nb][4 * (2 * ct + rt) + e] = t[e];
                }
        };

        // TRAIN: this lane's accumulator elements as fp32 rows [sorted position][member m's H columns] (runs of four columns)
        auto store_rows = [&](float *base, int64_t ld, int H, const FusedUnit &u) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    if (rb >= u.nrb || nb >= u.nba) continue;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = urow(u, rb, q);
                        float *dst = base + (int64_t)(p0 + min(row, n_rows - 1)) * ld + (int64_t)m * H + ucol(u, q);
                        if (row < n_rows) *reinterpret_cast<v4f *>(dst) = acc[rb * NB + nb][q];
                    }
                }
        };

        // =============== layer 0: act0 = celu(aev x W0^T + b0) over the flagged slabs ===============
        const v4f bnd = *(const gf4 *)(fs.bounds + 8 * m);   // operand bounds of this member
        // (round 5: the layer-0 biases are fetched behind the layer-0 k loop, not ahead of it -- sixteen registers less through
        // the loop, one spilled register less, -0.8 % of the stage in a same-box A/B)
        float bias0[NB][16];
        const uint32_t tmask_cur = tmask;   // (prefetch_w0 moves tmask on to the next item's)
        const int nact = __popc(tmask);
        const int npair = (nact + 2 * FR_GROUP - 1) / (2 * FR_GROUP);
        // (AEV slabs 0..5 and the first D weight fragments were requested during the previous item)
        // Owner order (L0B) with at most KEEP_SLABS flagged slabs -- every tile of a water box: the split planes of the tile's
        // AEV slabs are the same for all eight members, so they are staged ONCE, into slot 0 in an unpadded swizzled layout
        // (four slabs in the place of three padded ones), and the items of the other members neither fetch nor convert
        // nor store them again (3.6 KB per atom of fetches and ~1 k clocks per item)
        const bool keep = L0B && nact <= C::KEEP_SLABS;
        // (the kept path's first slab WRITES the accumulators; every other path -- and a tile of atoms without neighbors, whose
        // layer 0 multiplies nothing -- starts from zeros)
        if (!(keep && nact > 0)) zero_acc();
        bool staged = true;
        if (keep) {
            staged = staged_tile != tile;
            if (staged) {
                store_kept(va, 0);
                store_kept(vb, FR_GROUP);
                staged_tile = tile;
            }
        } else {
            store_group(va, slot(0));
            store_group(vb, slot(1));
            staged_tile = -1;
        }
        // (s_max is reset behind the barrier that follows its last read, below; an item that staged nothing has nothing to
        // publish either: the barrier at the end of the previous item already separates the two)
        if (staged) __syncthreads();   // slots 0 / 1 published
        ANIHIP_STAMP(trace, 2);
        // six slabs (two staging slots, 12 k steps) per barrier; the next six are fetched into registers
        // before the MFMAs of the current ones and staged after them
        for (int pr = 0; pr < npair; ++pr) {
            // (only when another pair follows: the ring requests of the loop below complete in order BEHIND these)
            if (pr + 1 < npair) {
                fetch_group(va);
                fetch_group(vb);
            }
            if (u1.nrb > 0) {
                const int lo_ = (u1.rb0 * 32 + n16) * FR_SLAB_LD + c4 * 8;   // this lane's fragment inside a staged slab
                const _Float16 *s0 = slot(2 * (pr & 1)) + lo_, *s1 = slot(2 * (pr & 1) + 1) + lo_;
                const int n_live = nact - 2 * FR_GROUP * pr;
                if (keep) {   // (wave-uniform) the kept copy: this lane's chunk of a slab's rows
                    const int rowk = u1.rb0 * 32 + n16, sw = (0 - (rowk >> 2)) & 3;
                    const _Float16 *k0 = slot0 + rowk * 32 + ((c4 ^ sw) << 3);
                    FR_UNIT(u1, (fr_l0_pair<RBA, D, ROWS, 4, 4 - D, true, true>(acc, rg, k0, k0, n_live, next_s2)))
                } else if (nact <= 4) {   // (wave-uniform)
                    FR_UNIT(u1, (fr_l0_pair<RBA, D, ROWS, 4, 4 - D>(acc, rg, s0, s1, n_live, next_s2)))
                } else {
                    FR_UNIT(u1, (fr_l0_pair<RBA, D, ROWS, 2 * FR_GROUP, 2 * FR_GROUP>(acc, rg, s0, s1, n_live, next_s2)))
                }
            }
            if (pr + 1 < npair) {   // (the last pair's successors are zeros nobody reads)
                store_group(va, slot(2 * ((pr + 1) & 1)));
                store_group(vb, slot(2 * ((pr + 1) & 1) + 1));
                __syncthreads();
            }
        }
        ANIHIP_STAMP(trace, 3);
        // weights of phase 1 start streaming during the layer-0 epilogue
        Ring r1;
        fr_ring<D>(r1, fs.w1, (int64_t)(H2 >> 5) * (H1 >> 4) * 2 * FRAG, m, H1 >> 4, u2.cb, lane, u2.nba);
        // celu and its derivative from one exponential: x > 0: (x, 1), else (alpha (e - 1), e), e = exp(x / alpha)
        const float ia_log2e = inv_alpha * 1.44269504f;
        // two elements at a time: bias + scale, the exponent argument and alpha (e - 1) as packed fp32 operations
        typedef float v2f __attribute__((ext_vector_type(2)));
        auto celu_d2 = [&](float a0, float a1, float osc, float b0, float b1, float &y0, float &y1, float &dd0,
                           float &dd1) {
            const v2f x = v2f{a0, a1} * osc + v2f{b0, b1};
            if constexpr (ACT == 1) {
                // gelu(x) = x Phi(x), gelu'(x) = Phi(x) + x phi(x)   (torch.nn.GELU(), approximate = 'none')
                const v2f ph = v2f{erff(x.x * 0.70710678f), erff(x.y * 0.70710678f)} * 0.5f + 0.5f;
                const v2f t = x * x * (-0.5f * 1.44269504f);
                const v2f g2 = v2f{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} * 0.39894228f;
                const v2f y = x * ph, d = ph + x * g2;
                y0 = y.x; y1 = y.y; dd0 = d.x; dd1 = d.y;
            } else {
                const v2f t = x * ia_log2e;
                const v2f e = v2f{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
                const v2f y = e * alpha - alpha;
                dd0 = fminf(e.x, 1.0f);
                dd1 = fminf(e.y, 1.0f);
                y0 = __builtin_amdgcn_fmed3f(x.x, y.x, 0.f);
                y1 = __builtin_amdgcn_fmed3f(x.y, y.y, 0.f);
            }
        };
        float d0f[NE][16];   // celu'(act0) of this lane's elements
        float a0max;         // tile max of |act0|
        {
            load_cols(fs.b0 + (int64_t)m * H1, bias0, u1);
            ANIHIP_STAMP(trace, 22);
            const float oscale = fs.is0 * 0.25f;
            // tile maximum of |act0| for the split scale: act0 >= -alpha (CELU) / >= -0.17 (GELU), so max(floor, max act0)
            // bounds it -- one v_max3_f32 per element pair instead of two |.| and three max (a quarter of this epilogue's
            // VALU instructions went into the absolute values)
            float vmax = ACT == 1 ? 0.17f : alpha;
            // (ONE branch per row block, not per element pair: with the test inside the pair loop the compiler emitted a
            // scalar branch between any two pairs and their dependent chains -- fma, mul, exp, fma, med3 -- ran one after
            // the other; inside one block the scheduler interleaves them)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                if (rb < u1.nrb) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                        for (int r = 0; r < 16; r += 2) {
                            const int i = rb * NB + nb;
                            float v0, v1;
                            celu_d2(ACC(acc[i], r), ACC(acc[i], r + 1), oscale, bias0[nb][r], bias0[nb][r + 1], v0, v1, d0f[i][r],
                                    d0f[i][r + 1]);
                            ACC(acc[i], r) = v0;
                            ACC(acc[i], r + 1) = v1;
                            if (NB == 1 || nb < u1.nba) vmax = __builtin_fmaxf(vmax, __builtin_fmaxf(v0, v1));
                        }
                } else {   // (defined on every path, like the rings)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) d0f[rb * NB + nb][r] = 0.f;
                }
            }
            if constexpr (TRAIN) store_rows(g.tr_act[0], g.tr_ld[0], H1, u1);
            ANIHIP_STAMP(trace, 23);
            a0max = tile_max(vmax);   // (barriers: every wave is past the staging slots)
            ANIHIP_STAMP(trace, 24);
        }
        const float s0 = pow2_scale_for(a0max);
        put_acc(X0, x0_plane, ld0, s0, u1);
        ANIHIP_STAMP(trace, 25);
        // the scales of the inner GEMM operands follow from a0max and the weight-norm bounds: no more reductions
        const float s1 = pow2_scale_for(__builtin_fmaf(a0max, bnd[0], bnd[1]));   // |act1| <= a0max ||W1||_inf + |b1|
        const float s2 = pow2_scale_for(bnd[2]);                                  // |d act2| <= max |w3| / M
        const float s3 = pow2_scale_for(bnd[3]);                                  // |d act1| <= [2] ||W2||_1
        __syncthreads();
        if (tid == 0) s_max = 0u;   // (every thread has read the tile maximum; the next item's atomics are many barriers away)
        ANIHIP_STAMP(trace, 4);

        // =============== phase 1: act1 = celu(act0 x W1^T + b1) ===============
        float bias1[NB][16];   // (per-column parameters travel during the GEMM)
        load_cols(fs.b1 + (int64_t)m * H2, bias1, u2);
        FR_UNIT(u2, (fr_gemm<RBA, D>(acc, X0 + u2.rb0 * 32 * ld0, ld0, x0_plane, r1, H1 >> 5, lane)))
        ANIHIP_STAMP(trace, 5);
        Ring r2;
        fr_ring<D>(r2, fs.w2, (int64_t)(H3 >> 5) * (H2 >> 4) * 2 * FRAG, m, H2 >> 4, u3.cb, lane, u3.nba);
        float d1f[NE][16];   // celu'(act1) of this lane's elements
        {
            const float oscale = fs.is1 / s0;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                if (rb < u2.nrb) {   // (one branch per row block: see the layer-0 epilogue)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                        for (int r = 0; r < 16; r += 2) {
                            const int i = rb * NB + nb;
                            float y0, y1;
                            celu_d2(ACC(acc[i], r), ACC(acc[i], r + 1), oscale, bias1[nb][r], bias1[nb][r + 1], y0, y1, d1f[i][r],
                                    d1f[i][r + 1]);
                            ACC(acc[i], r) = y0;
                            ACC(acc[i], r + 1) = y1;
                        }
                } else {   // (defined on every path, like the rings)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) d1f[rb * NB + nb][r] = 0.f;
                }
            }
            if constexpr (TRAIN) store_rows(g.tr_act[1], g.tr_ld[1], H2, u2);
            ANIHIP_STAMP(trace, 26);
            put_acc(X1, x1_plane, ld1, s1, u2);
            ANIHIP_STAMP(trace, 27);
        }
        __syncthreads();  // X1 complete; every wave is done reading X0 -> XU reusable
        ANIHIP_STAMP(trace, 6);

        // =============== phase 2: act2 = celu(act1 x W2^T + b2); output layer; backward seed ===============
        float bias2[NB][16], w3[NB][16];
        // (fetching these behind the GEMM as well frees 32 registers and the last two spills, and is 0.4 % SLOWER: measured)
        load_cols(fs.b2 + (int64_t)m * H3, bias2, u3);
        load_cols(fs.w3 + (int64_t)m * H3, w3, u3);
        FR_UNIT(u3, (fr_gemm<RBA, D>(acc, X1 + u3.rb0 * 32 * ld1, ld1, x1_plane, r2, H2 >> 5, lane)))
        ANIHIP_STAMP(trace, 7);
        Ring r3;   // (also without want_grad: see fr_ring)
        fr_ring<D>(r3, fs.w2t, (int64_t)(H2 >> 5) * (H3 >> 4) * 2 * FRAG, m, H3 >> 4, u2.cb, lane, u2.nba);
        {
            // e = sum_col act2 * w3 (+ b3): per-lane partial over its columns of each of its two rows of a row block, the four
            // lanes of a row (c4 = 0..3) combined with two lane swaps, the waves through LDS in a fixed order (deterministic sum).
            // seed: d act2 = w3 * celu'(act2) / M, kept in the accumulators
            const float osc2 = fs.is2 / s1;
            const float invM = 1.0f / (float)Mi;
            float e_loc[RB];   // [row block]: lanes with c4 = rt hold the sum of row half rt
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                float e[2] = {0.f, 0.f};
                if (rb < u3.nrb) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                        for (int r = 0; r < 16; r += 2) {
                            const int i = rb * NB + nb, rt = (r >> 2) & 1;
                            float y0, y1, dy0, dy1;
                            celu_d2(ACC(acc[i], r), ACC(acc[i], r + 1), osc2, bias2[nb][r], bias2[nb][r + 1], y0, y1, dy0, dy1);
                            e[rt] = __builtin_fmaf(y0, w3[nb][r], e[rt]);
                            e[rt] = __builtin_fmaf(y1, w3[nb][r + 1], e[rt]);
                            ACC(acc[i], r) = invM * w3[nb][r] * dy0;
                            ACC(acc[i], r + 1) = invM * w3[nb][r + 1] * dy1;
                            if constexpr (TRAIN) {   // act2 leaves from here (the accumulators take the backward seed)
                                const int row = urow(u3, rb, r >> 2);
                                float *dst = g.tr_act[2] + (int64_t)(p0 + min(row, n_rows - 1)) * g.tr_ld[2] + (int64_t)m * H3 +
                                             ucol(u3, r >> 2) + (r & 3);
                                if (row < n_rows) *reinterpret_cast<float2 *>(dst) = make_float2(y0, y1);
                            }
                        }
                    // the four lanes of a row (c4 = 0..3, sixteen lanes apart) through two permlane swaps on the VALU (no
                    // LDS round trips): the 16-lane rows of s are [e0.c0 + e0.c1, e1.c0 + e1.c1, e0.c2 + e0.c3, e1.c2 + e1.c3],
                    // then rows [e0 total, e1 total, e0 total, e1 total]
                    auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(e[0]), __float_as_uint(e[1]), false, false);
                    const float s = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);
                    auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
                    e[0] = __uint_as_float(s32[0]) + __uint_as_float(s32[1]);
                }
                e_loc[rb] = e[0];
            }
            // every wave writes its partial of every row of the tile (zero for the row blocks it has no unit in)
#pragma unroll
            for (int t = 0; t < RB; ++t) {
                float v = 0.f;
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
                    if (t - u3.rb0 == rb && rb < u3.nrb) v = e_loc[rb];
                if (c4 < 2) s_e[wave * ROWS + t * 32 + 16 * c4 + n16] = v;   // (lanes 0..31: row 16 c4 + n16 = lane)
            }
            if constexpr (TRAIN) store_rows(g.tr_dlt[2], g.tr_ld[2], H3, u3);
            ANIHIP_STAMP(trace, 28);
            if (g.want_grad) put_acc(X2, x2_plane, ld2, s2, u3);   // (XU: X0 is dead since the last barrier)
            ANIHIP_STAMP(trace, 29);
        }
        __syncthreads();
        if (tid < n_rows) {
            float e = fs.b3[m];
#pragma unroll
            for (int w8 = 0; w8 < NW; ++w8) e += s_e[w8 * ROWS + tid];
            g.member_part[(int64_t)(p0 + tid) * Mi + m] = e;
        }
        ANIHIP_STAMP(trace, 9);

        Ring r4;
        if (g.want_grad) {
            // =============== phase 3: d act1 = (d act2 x W2) * celu'(act1) ===============
            // (no zero fill: the first k2 step of fr_gemm writes the accumulators)
            FR_UNIT(u2, (fr_gemm<RBA, D, B2>(acc, X2 + u2.rb0 * 32 * ld2, ld2, x2_plane, r3, H3 >> 5, lane)))
            ANIHIP_STAMP(trace, 10);
        }
        fr_ring<D>(r4, fs.w1t, (int64_t)(H1 >> 5) * (H2 >> 4) * 2 * FRAG, m, H2 >> 4, u1.cb, lane, u1.nba);
        if (g.want_grad) {
            if (u2.nrb > 0) {
                const float osc3 = fs.is2 / s2;
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    if (rb >= u2.nrb) continue;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) ACC(acc[rb * NB + nb], r) *= osc3 * d1f[rb * NB + nb][r];
                }
                if constexpr (TRAIN) store_rows(g.tr_dlt[1], g.tr_ld[1], H2, u2);
                ANIHIP_STAMP(trace, 30);
                put_acc(X1, x1_plane, ld1, s3, u2);   // (X1: its last readers finished before the previous barrier)
                ANIHIP_STAMP(trace, 31);
            }
        }
        // the first layer-0 weight fragments of the next item: L2 hits, requested behind the phase-4 ring (the first
        // MFMAs of phase 4 do not wait for them) and ahead of its MFMA loop, so that the store phase at the end is left
        // with the stores and the AEV slabs (it is bound by the CU's vector-memory throughput)
        // (round 5: with phase 5 they are requested in ITS last pass instead -- 48 registers less through phases 4 and 5, no
        // spilled register left, -1.3 % of the stage in a same-box A/B)
        if constexpr (!L0B) prefetch_w0(te_n, mem_n);
        __syncthreads();
        ANIHIP_STAMP(trace, 11);
        // =============== phase 4: d act0 = (d act1 x W1) * celu'(act0)  -> global, or -> LDS for phase 5 ===============
        if (g.want_grad && u1.nrb > 0) {
            FR_UNIT(u1, (fr_gemm<RBA, D, B2>(acc, X1 + u1.rb0 * 32 * ld1, ld1, x1_plane, r4, H2 >> 5, lane)))
        }
        ANIHIP_STAMP(trace, 12);
        if constexpr (L0B) {
            {
            // =============== phase 5 (l0b): d E / d AEV += d act0 x W0 over the tile's flagged slabs ===============
            // d act0 goes into LDS as split planes (X0's place: the last readers of XU finished before the barrier ahead of
            // phase 4) instead of to HBM, and the layer-0 backward GEMM runs here: a column block is one flagged AEV slab,
            // K = this member's H1 columns.  The workgroup OWNS the tile through all members (owner order), so the sum
            // over the members is a plain read-add-write of the same lane on the same address, member after member in
            // a fixed order: no atomics, no d act0 round trip through HBM, no separate GEMM launch.
            const float s4 = pow2_scale_for(fs.bounds[8 * m + 4] * (ACT == 1 ? 1.13f : 1.0f));   // |d act0| <= [4] max act'
            // Work of a pass: FOUR flagged slabs x both row blocks x K = eight 16-column halves, one per wave: wave w takes column
            // half w & 1 of the pass's slab w >> 1 for all 64 rows and the whole of K -- four 16 x 16 tiles per wave, every weight
            // fragment {hi, lo} of its column half crosses the CU's 64 B/clk L2 port once and feeds twelve MFMAs.  (Rounds 4-5 split
            // K between the two waves of a SIMD because a 32 x 32 x 16 unit cannot be narrower than 32 columns; the partial tiles met
            // in LDS behind a barrier and left through a wave-private LDS tile: a hand-over, a barrier and two LDS round trips per
            // item that the 16-column units do not need.)  A lane ends up with row 16 t + n16 of the tile and the 16 bytes at
            // columns 4 c4 .. of its column half, t = 0..3: a store instruction covers 16 rows x 64 contiguous bytes, half a cache
            // line per row, and the wave of the other column half writes the other half of the same lines.
            const int KS5 = H1 >> 4;
            const int64_t mh5 = (int64_t)g.n_slabs * KS5 * (2 * FRAG);
            auto nth_slab = [&](int c) {   // c-th flagged slab of the tile (scalar), -1 past the end
                uint32_t mk = tmask_cur;
                for (int t = 0; t < c; ++t) mk &= mk - 1u;
                return mk ? (int)__builtin_ctz(mk) : -1;
            };
            const int ct5 = wave & 1;
            WRingHalf<D> r5;
            auto ring5 = [&](int sl_) {   // the first D k2 steps of slab sl_, this wave's column half
                r5.base = fs.w0t + (int64_t)m * mh5 + (int64_t)sl_ * KS5 * (2 * FRAG) + wring_lane_off(lane) + ct5 * 128;
#pragma unroll
                for (int sl = 0; sl < D; ++sl) r5.load(sl, min(sl, (KS5 >> 1) - 1));
            };
            int slab = nth_slab(wave >> 1);
            ring5(max(slab, 0));   // (travels during the epilogue below; requested behind it instead: no faster, measured)
            ANIHIP_STAMP(trace, 16);
            if (g.want_grad && u1.nrb > 0) {
                const float osc4 = fs.is1 / s3;
#pragma unroll
                for (int i = 0; i < NE; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ACC(acc[i], r) *= osc4 * d0f[i][r];
                put_acc(X0, x0_plane, ld0, s4, u1);
            }
            ANIHIP_STAMP(trace, 17);
            __syncthreads();   // d act0 complete
            ANIHIP_STAMP(trace, 13);
            if (g.want_grad) {
                const float osc5 = fs.is0 / s4;
                // (wave-uniform) a single-pass tile in owner order over single tiles: the members' sum stays in registers (gsum) and
                // only the last member needs the row pointers (LDS reads + 64-bit address arithmetic: 1.3 k clocks per item)
                const bool regsum_tile = nact <= 4 && g.owner == 1;
                float *orow[4];
                bool rok[4];
                if (!regsum_tile || m == Mi - 1) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int row = 16 * t + n16;
                        orow[t] = g.grad_aev + (int64_t)s_orow[par * ROWS + row] * g.L + 16 * ct5 + 4 * c4;
                        rok[t] = row < n_rows;    // (short tiles repeat their last atom: one writer per row only)
                    }
                } else {   // (defined on every path)
#pragma unroll
                    for (int t = 0; t < 4; ++t) { orow[t] = g.grad_aev; rok[t] = false; }
                }
                // (the last pass -- the only one of a water tile -- is peeled: what it prefetches for the next item must not be
                // defined under a condition inside a loop, or it is carried around the loop in registers)
                auto pass = [&](int c0, auto last_) {
                    constexpr bool LAST = decltype(last_)::value;
                    const bool live = slab >= 0;
                    const int sl = max(slab, 0);
                    const int col5 = g.kp_rad ? kp_col(g.kp_rad, sl) : 32 * sl;
                    const int nv5 = g.kp_rad ? kp_valid(g.kp_rad, sl) : min(32, (int)g.L - 32 * sl);
                    const bool cok = live && 16 * ct5 + 4 * c4 < nv5;
                    // (wave-uniform) the members' sum of a single-pass tile stays in registers (gsum)
                    const bool regsum = LAST && c0 == 0 && regsum_tile;
                    // what the members before this one left in the rows (this wave wrote it: L2 hits), requested ahead of the
                    // MFMA loop
                    v4f prev[4];
                    if (!regsum) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            // (lanes with nothing to read -- the first member, waves without a slab -- read a line that is hot in
                            // L2: loads return in order, and a miss to HBM here would hold up the weight ring's requests behind it)
                            const bool ok = cok && rok[t] && m > 0;
                            prev[t] = *(const gf4 *)(ok ? orow[t] + col5 : fs.bounds);
                            if (!ok) prev[t] = v4f{0.f, 0.f, 0.f, 0.f};
                        }
                    } else {
#pragma unroll
                        for (int t = 0; t < 4; ++t) prev[t] = v4f{0.f, 0.f, 0.f, 0.f};
                    }
                    v4f acc5[4];
                    ANIHIP_STAMP(trace, 19);
                    if (live) fr_gemm_half<D, B2>(acc5, X0, ld0, x0_plane, r5, KS5 >> 1, lane);   // (writes acc5: no zero fill)
                    else {
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc5[t] = v4f{0.f, 0.f, 0.f, 0.f};
                    }
                    ANIHIP_STAMP(trace, 20);
                    const int slab_n = nth_slab(c0 + 4 + (wave >> 1));
                    if constexpr (LAST) {
                        // the next item's AEV slabs: behind the last ring request of this item (loads return in order: a
                        // miss to HBM ahead of a ring request would stall the MFMA loop), ahead of the stores
                        // (not when the next item is another member of this tile and the tile's operand is kept in LDS)
                        prefetch_w0(te_n, mem_n);   // (the next item's first layer-0 weight fragments: L2 hits)
                        if (!(tile_n == tile && keep)) prefetch_aev(te_n, atom_n);
                        else no_aev();
                    } else {
                        ring5(max(slab_n, 0));
                    }
                    ANIHIP_STAMP(trace, 21);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        v4f v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(acc5[t][e], osc5, regsum ? (m > 0 ? gsum[t][e] : 0.f) : prev[t][e]);
                        gsum[t] = v;
                        if (cok && rok[t] && (!regsum || m == Mi - 1)) *reinterpret_cast<v4f *>(orow[t] + col5) = v;
                    }
                    slab = slab_n;
                };
                int c0 = 0;
                for (; c0 + 4 < nact; c0 += 4) pass(c0, std::false_type{});
                pass(c0, std::true_type{});
            } else {
                prefetch_w0(te_n, mem_n);
                if (!(tile_n == tile && keep)) prefetch_aev(te_n, atom_n);
                else no_aev();
            }
            ANIHIP_STAMP(trace, 8);
            if (spc == 0) s_orow[(par ^ 1) * ROWS + srow] = atom_n;
            // every wave is done with the LDS of this item: the next one may stage its slabs -- unless it stages nothing (another
            // member of this tile, operand kept): its first LDS writes are the act0 planes behind its own layer-0 loop and
            // tile-maximum barrier, which no wave passes before every wave has left this item's phase-5 k loop (the last reader
            // of the d act0 planes)
            if (!(tile_n == tile && keep)) __syncthreads();
            }
        } else {
        // the AEV slabs of the next item travel during the stores below (requested AFTER the last ring load of this
        // item: loads complete in order, and an HBM miss ahead of a ring request stalls the MFMA loop that waits
        // for it)
        prefetch_aev(te_n, atom_n);
        // every wave is done with the LDS of this item: the next one may stage its slabs
        __syncthreads();
        if (g.want_grad && u1.nrb > 0) {
            const float osc4 = fs.is1 / s3;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                if (rb >= u1.nrb) continue;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    if (nb >= u1.nba) continue;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {   // run q = 2 ct + rt: tile row 16 rt + n16 of the row block, columns 16 ct + 4 c4 ..
                        const int row = urow(u1, rb, q);
                        float *dst = g.d0 + (int64_t)(p0 + min(row, n_rows - 1)) * g.ld0 + (int64_t)m * H1 + ucol(u1, q);
                        if (g.d0_tm) {
                            // fragment order of the 32 x 16 A operand the layer-0 backward GEMMs read (tm_unit): column half ct = its
                            // k step, lane slot (columns 8 ..: 32 +) row of the block, floats (c4 & 1) * 4 ..
                            dst = g.d0 + tm_base + (int64_t)(((rel_tile >> 6) * Mi + m) * 64) * H1 +
                                  tm_unit(u1.cb, ((rel_tile >> 5) & 1) + u1.rb0 + rb, q >> 1) +
                                  ((c4 >> 1) * 32 + 16 * (q & 1) + n16) * 8 + 4 * (c4 & 1);
                        }
                        v4f v;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            v[e] = ACC(acc[rb * NB + nb], 4 * q + e) * osc4 * d0f[rb * NB + nb][4 * q + e];
                        if (row < n_rows) *reinterpret_cast<v4f *>(dst) = v;
                    }
                }
            }
        }
        }
        ANIHIP_STAMP(trace, 15);
        if (!has_next) break;
        par ^= 1;
        te = te_n;
        mem = mem_n;
        tile = tile_n;
        gj = gj_n; gsz = gsz_n;
        item = g.owner ? tile * Mi + mem : mem * n_tiles + tile;
    }
}
#undef FR_UNIT

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

__global__ void k_fused_finish(FinishArgs f)
{
    fused_finish(f, blockIdx.x * (int64_t)blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
}

// ---- output layer: energies + seed of the backward pass --------------------------------------------

struct HeadArgs {
    const float *w[MAX_S];     // [M][Hp]
    const float *bias[MAX_S];  // [M]
    int Hp[MAX_S];
    const int *ctl;
    const int *perm;
    float *act;   // last hidden activations [n][ld]; overwritten with d(mean energy)/d(activation)
    int64_t ld;
    float *seed;             // training pass: the backward seed goes here ([n][ld]) and act is kept (NULL: in place)
    const float *g_atom;     // training pass: upstream d Loss / d atomic_e per atom (NULL: 1)
    float *atomic_e;   // [n_atoms]
    float *member_e;   // [M][n_atoms] or NULL
    int64_t n_atoms;
    int S, M;
    float inv_alpha;
    int want_grad;
    unsigned *amax;   // f16x3: running max table (or NULL)
    int amax_out;
    int act_kind;          // ANIHIP_ACT_*: GELU reads the derivative from the pre-activations
    const float *zpre;     // [n][ld] pre-activations of the last hidden layer (GELU training passes)
};

__global__ __launch_bounds__(256) void k_head(HeadArgs h)
{
    const int lane = lane_id();
    const int64_t n = h.ctl[CTL_OFF + h.S];
    const int64_t nw = (int64_t)gridDim.x * 4;
    float gmax = 0.f;   // running max of the backward seed, flushed when the species changes
    int gs = -1;
    for (int64_t p = blockIdx.x * 4 + (threadIdx.x >> 6); p < n; p += nw) {
        int s = 0;
        while (s + 1 < h.S && p >= h.ctl[CTL_OFF + s + 1]) ++s;
        if (s != gs) {
            if (gs >= 0 && h.amax && h.want_grad) amax_update(h.amax, h.amax_out, gs, gmax);
            gs = s;
            gmax = 0.f;
        }
        const int Hp = h.Hp[s];
        float *row = h.act + p * h.ld;
        float *srow = h.seed ? h.seed + p * h.ld : row;
        const int atom = h.perm[p];
        float esum = 0.f;
        const float invM = 1.0f / (float)h.M;
        const float up = h.g_atom ? h.g_atom[atom] * invM : invM;
        for (int m = 0; m < h.M; ++m) {
            const float *w = h.w[s] + (int64_t)m * Hp;
            float part = 0.f;
            for (int o = lane; o < Hp; o += WAVE) {
                const float y = row[m * Hp + o];
                part += y * w[o];
                if (h.want_grad) {
                    float c1, c2;
                    act_derivs(h.act_kind, y, h.zpre ? h.zpre[p * h.ld + m * Hp + o] : 0.f, h.inv_alpha, c1, c2);
                    const float gq = up * w[o] * c1;
                    srow[m * Hp + o] = gq;
                    gmax = fmaxf(gmax, fabsf(gq));
                }
            }
            part = wave_sum(part) + h.bias[s][m];
            if (h.member_e && lane == 0) h.member_e[(int64_t)m * h.n_atoms + atom] = part;
            esum += part;
        }
        if (lane == 0) h.atomic_e[atom] = esum * invM;
    }
    if (gs >= 0 && h.amax && h.want_grad) amax_update(h.amax, h.amax_out, gs, gmax);
}

// ---- weight gradients (training pass) ----------------------------------------------------------------
// dW = X^T D over the rows (atoms) of one species: X = input of the layer (AEV rows gathered through the bucket
// list, or the activations of the previous layer), D = d Loss / d (pre-activation) of the layer.  The reduction
// index is the ROW index, so both operands are consumed exactly as they lie in memory: for two consecutive rows
// (k = lane >> 5) a lane reads the float2 at columns 2 (lane & 31) of X and of D, and
// v_mfma_f32_32x32x2_f32 accumulates the four (even/odd) x (even/odd) column combinations -- a 64 x 64 block of dW
// per wave with no LDS staging and no transposes.  A workgroup = 4 waves on the same 64 input columns and four
// neighbouring 64-column blocks of D (X is then shared through L1); rows are cut into chunks of WG_ROWS and the
// partial blocks are added to dW with float atomics (dW is zeroed by the host function).
constexpr int WG_ROWS = 2048;   // rows per workgroup
constexpr int WG_U = 8;         // row pairs in flight

struct WgradProblem {
    const float *X;   // input of the layer
    int64_t ldx;
    int x_boff;       // column offset of batch b in X rows = b * x_boff
    int K, k_valid;   // input width (padded), columns >= k_valid are neither read nor written
    const float *D;
    int64_t ldd;
    int d_boff;
    int N;            // output width per batch
    float *dW;        // [batch][N][ldw]: rows = output units (torch.nn.Linear layout), ldw >= k_valid
    int64_t ldw, w_bstride;
};
struct WgradArgs {
    WgradProblem prob[MAX_S];
    const int *ctl;
    const int *x_gather;   // sorted position -> source row of X (layer 0) or NULL
    int S, batch, ki_max, nj_max;
};

__device__ __forceinline__ bool chunk_lookup(const int *ctl, int S, int chunk, int rows_per_chunk, int &s, int &m0)
{
    int first = 0;
    for (s = 0; s < S; ++s) {
        const int nc = (ctl[CTL_CNT + s] + rows_per_chunk - 1) / rows_per_chunk;
        if (chunk < first + nc) {
            m0 = (chunk - first) * rows_per_chunk;
            return true;
        }
        first += nc;
    }
    return false;
}

__global__ __launch_bounds__(256, 2) void k_wgrad(WgradArgs g)
{
    int id = blockIdx.x;
    const int nj = id % g.nj_max; id /= g.nj_max;
    const int ki = id % g.ki_max; id /= g.ki_max;
    const int bb = id % g.batch;
    const int chunk = id / g.batch;
    int s, m0;
    if (!chunk_lookup(g.ctl, g.S, chunk, WG_ROWS, s, m0)) return;
    const WgradProblem &pr = g.prob[s];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int i0 = ki * 64, j0 = (nj * 4 + wave) * 64;
    if (i0 >= pr.k_valid || j0 >= pr.N) return;
    const int n_rows = min(WG_ROWS, g.ctl[CTL_CNT + s] - m0);
    const int p0 = g.ctl[CTL_OFF + s] + m0;
    const int c2 = 2 * (lane & 31), kk = lane >> 5;
    const bool x_ok = i0 + c2 < pr.k_valid;      // k_valid is even: the float2 is inside or outside as a whole
    const bool d_ok = j0 + c2 < pr.N;
    const int xc = x_ok ? i0 + c2 : 0, dc = d_ok ? j0 + c2 : 0;
    const float *Xb = pr.X + (int64_t)bb * pr.x_boff + xc;
    const float *Db = pr.D + (int64_t)bb * pr.d_boff + dc;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float2 xs[WG_U], ds[WG_U];
    auto load = [&](int r0) {
#pragma unroll
        for (int u = 0; u < WG_U; ++u) {
            const int r = r0 + 2 * u + kk;
            const bool v = r < n_rows;
            const int rr = v ? r : 0;                                  // clamped: always valid memory
            const int64_t xrow = g.x_gather ? (int64_t)g.x_gather[p0 + rr] : (int64_t)(p0 + rr);
            float2 x = *reinterpret_cast<const float2 *>(Xb + xrow * pr.ldx);
            float2 d = *reinterpret_cast<const float2 *>(Db + (int64_t)(p0 + rr) * pr.ldd);
            if (!(v && x_ok)) x = make_float2(0.f, 0.f);
            if (!d_ok) d = make_float2(0.f, 0.f);
            xs[u] = x;
            ds[u] = d;
        }
    };
    load(0);
    for (int r0 = 0; r0 < n_rows; r0 += 2 * WG_U) {
        float2 xc_[WG_U], dc_[WG_U];
#pragma unroll
        for (int u = 0; u < WG_U; ++u) { xc_[u] = xs[u]; dc_[u] = ds[u]; }
        if (r0 + 2 * WG_U < n_rows) load(r0 + 2 * WG_U);
#pragma unroll
        for (int u = 0; u < WG_U; ++u) {
            // D as the first operand: the accumulator holds the TRANSPOSED block (rows = output unit, columns =
            // input unit), i.e. torch.nn.Linear's [out][in] layout with the lanes along the contiguous index
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(dc_[u].x, xc_[u].x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(dc_[u].y, xc_[u].x, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(dc_[u].x, xc_[u].y, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(dc_[u].y, xc_[u].y, acc[1][1], 0, 0, 0);
        }
    }
    // C/D layout of 32x32: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    float *W = pr.dW + (int64_t)bb * pr.w_bstride;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int i = i0 + c2 + a;          // input unit: along the lanes
        if (i >= pr.k_valid) continue;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = j0 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * kk) + b;   // output unit
                if (j < pr.N) atomicAdd(W + (int64_t)j * pr.ldw + i, acc[a][b][r]);
            }
    }
}

// out[col] += sum over the rows of species s of  scale(row) * X[row][col]   (bias gradients: X = D, scale = 1;
// output layer: X = last activations, scale = upstream / M, and extra[m] += sum of the scales = d Loss / d b3)
constexpr int CR_ROWS = 64;
struct ColReduceArgs {
    const float *X[MAX_S];
    float *out[MAX_S];
    float *extra[MAX_S];
    int ncols[MAX_S];
    int64_t ldx;
    const int *ctl;
    const int *perm;
    const float *g_atom;   // scale source (NULL: 1)
    float inv_m;
    int S, M, ct_max;
    // out_mstride > 0: the destination arrays are per member, out_mstride floats apart: column c belongs to member
    // c / n_per[s] (anihip_species_grads.member_stride; 0: one packed [M][n_per] array)
    int64_t out_mstride;
    int n_per[MAX_S];
};

__global__ __launch_bounds__(256) void k_col_reduce(ColReduceArgs g)
{
    const int ct = blockIdx.x % g.ct_max, chunk = blockIdx.x / g.ct_max;
    int s, m0;
    if (!chunk_lookup(g.ctl, g.S, chunk, CR_ROWS, s, m0)) return;
    const int col = ct * 256 + threadIdx.x;
    if (ct * 256 >= g.ncols[s]) return;
    const int n_rows = min(CR_ROWS, g.ctl[CTL_CNT + s] - m0);
    const int p0 = g.ctl[CTL_OFF + s] + m0;
    const bool cv = col < g.ncols[s];
    const float *x = g.X[s] + (cv ? col : 0);
    // the rows' scales first (two dependent loads each: perm, then the atom's upstream gradient), once per block, so that the
    // row loop below is a chain of independent loads
    __shared__ float s_scale[CR_ROWS];
    if (threadIdx.x < CR_ROWS)
        s_scale[threadIdx.x] = threadIdx.x < n_rows ? (g.g_atom ? g.g_atom[g.perm[p0 + threadIdx.x]] : 1.0f) * g.inv_m : 0.f;
    __syncthreads();
    float acc = 0.f, sacc = 0.f;
#pragma unroll 8
    for (int r = 0; r < n_rows; ++r) {
        const float sc = s_scale[r];
        acc += sc * x[(int64_t)(p0 + r) * g.ldx];
        sacc += sc;
    }
    if (cv) {
        int64_t at = col;
        if (g.out_mstride > 0) {
            const int mem = col / g.n_per[s];
            at = (int64_t)mem * g.out_mstride + (col - mem * g.n_per[s]);
        }
        atomicAdd(g.out[s] + at, acc);
    }
    if (g.extra[s] && ct == 0 && threadIdx.x < g.M)
        atomicAdd(g.extra[s] + (g.out_mstride > 0 ? (int64_t)threadIdx.x * g.out_mstride : (int64_t)threadIdx.x), sacc);
}

// output layer of the tangent pass: p = w3 c'(a3) / M,  q = w3 c''(a3) zdot3 / M,  d atomic_e = sum_m w3 . adot3 / M
struct HeadTangentArgs {
    const float *w[MAX_S];   // [M][Hp]
    int Hp[MAX_S];
    const int *ctl;
    const int *perm;
    const float *act, *zd, *ad;   // last hidden layer: activations, zdot, adot  [n][ld]
    float *P, *Q;
    int64_t ld;
    float *datomic_e;             // [n_atoms] or NULL
    int S, M;
    float inv_alpha;
    int act_kind;
    const float *zpre;            // pre-activations of the last hidden layer (GELU)
};

__global__ __launch_bounds__(256) void k_head_tangent(HeadTangentArgs h)
{
    const int lane = lane_id();
    const int64_t n = h.ctl[CTL_OFF + h.S];
    const int64_t nw = (int64_t)gridDim.x * 4;
    const float invM = 1.0f / (float)h.M;
    for (int64_t p = blockIdx.x * 4 + (threadIdx.x >> 6); p < n; p += nw) {
        int s = 0;
        while (s + 1 < h.S && p >= h.ctl[CTL_OFF + s + 1]) ++s;
        const int Hp = h.Hp[s];
        float part = 0.f;
        for (int m = 0; m < h.M; ++m) {
            const float *w = h.w[s] + (int64_t)m * Hp;
            for (int o = lane; o < Hp; o += WAVE) {
                const int64_t idx = p * h.ld + m * Hp + o;
                const float y = h.act[idx], zd = h.zd[idx];
                float c1, c2;
                act_derivs(h.act_kind, y, h.zpre ? h.zpre[idx] : 0.f, h.inv_alpha, c1, c2);
                const float mu = invM * w[o];
                h.P[idx] = mu * c1;
                h.Q[idx] = mu * c2 * zd;
                part += mu * h.ad[idx];
            }
        }
        part = wave_sum(part);
        if (h.datomic_e && lane == 0) h.datomic_e[h.perm[p]] = part;
    }
}

// ---- parameter refresh (training) ----------------------------------------------------------------------
// Rewrites the packed fp32 arrays (w, wt, bias; layouts in include/anihip.h) from the torch.nn.Linear tensors they
// were packed from, after an optimizer step: one thread per source weight, two scattered stores.
struct RepackArgs {
    const float *const *src;   // device: [M][S][nl][2] pointers {weight [out][in], bias [out]}
    float *w[MAX_S][ANIHIP_MAX_LAYERS], *wt[MAX_S][ANIHIP_MAX_LAYERS], *bias[MAX_S][ANIHIP_MAX_LAYERS];
    int dims[MAX_S][ANIHIP_MAX_LAYERS + 1];   // padded widths
    int out[MAX_S][ANIHIP_MAX_LAYERS], in[MAX_S][ANIHIP_MAX_LAYERS];   // widths of the source tensors
    int S, M, nl, k0p;
};

__global__ __launch_bounds__(256) void k_repack(RepackArgs g)
{
    int id = blockIdx.y;
    const int l = id % g.nl; id /= g.nl;
    const int s = id % g.S;
    const int m = id / g.S;
    const int out = g.out[s][l], in = g.in[s][l];
    const float *W = g.src[((m * g.S + s) * g.nl + l) * 2 + 0];
    const float *b = g.src[((m * g.S + s) * g.nl + l) * 2 + 1];
    const int inp = g.dims[s][l], outp = g.dims[s][l + 1];
    for (int64_t e = blockIdx.x * 256 + threadIdx.x; e < (int64_t)out * in; e += (int64_t)gridDim.x * 256) {
        const int o = (int)(e / in), k = (int)(e - (int64_t)o * in);
        const float v = W[e];
        if (l == g.nl - 1) {
            g.w[s][l][(int64_t)m * inp + k] = v;
        } else if (l == 0) {
            g.w[s][l][(int64_t)k * ((int64_t)g.M * outp) + (int64_t)m * outp + o] = v;
            g.wt[s][l][((int64_t)m * outp + o) * g.k0p + k] = v;
        } else {
            g.w[s][l][((int64_t)m * inp + k) * outp + o] = v;
            g.wt[s][l][((int64_t)m * outp + o) * inp + k] = v;
        }
    }
    if (blockIdx.x == 0)
        for (int o = threadIdx.x; o < out; o += 256)
            g.bias[s][l][(l == g.nl - 1) ? m : (int64_t)m * outp + o] = b[o];
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

struct MlpWorkspace {
    float *zp[ANIHIP_MAX_LAYERS];   // training passes of GELU networks: pre-activations of the hidden layers (else NULL)
    int *ctl;
    unsigned *amax;
    float *member_part;
    int *perm;
    int4 *tile_tab;
    int *tile_rows;
    float *act[ANIHIP_MAX_LAYERS];
    int64_t ld[ANIHIP_MAX_LAYERS];
};

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// n_act: hidden-layer buffers carved (-1: all of them, what the layer-by-layer kernels and the training passes use; the fused
// kernel needs act[0] for its d E / d act0 hand-over, and none at all with the layer-0 backward inside)
static size_t mlp_carve(const anihip_mlp_desc *d, int64_t n, char *base, MlpWorkspace *w, int n_act = -1)
{
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char *p = base ? base + off : nullptr;
        off += align256(bytes);
        return p;
    };
    int *ctl = (int *)take(sizeof(int) * (CTL_WORDS + AMAX_WORDS));
    int *perm = (int *)take(sizeof(int) * (size_t)(n + 1));
    float *mpart = (float *)take(sizeof(float) * (size_t)(n + 1) * (size_t)d->n_members);
    const size_t tiles = (size_t)((n + 31) / 32) + ANIHIP_MAX_SPECIES;   // (finest tiling of the fused kernel)
    int4 *ttab = (int4 *)take(sizeof(int4) * tiles);
    // (row lists: 32 per tile at the finest tiling, 64 per tile at the coarsest, which has up to one partly filled tile
    // per species more rows than atoms)
    int *trows = (int *)take(sizeof(int) * (32 * tiles + 64 * (size_t)ANIHIP_MAX_SPECIES));
    if (w) {
        w->ctl = ctl; w->amax = (unsigned *)(ctl + CTL_WORDS); w->perm = perm; w->member_part = mpart;
        w->tile_tab = ttab; w->tile_rows = trows;
    }
    const int nh = d->net[0].n_layers - 1;  // hidden layers
    for (int l = 0; l < nh; ++l) {
        int mx = 0;
        for (int s = 0; s < d->num_species; ++s) mx = mx > d->net[s].dims[l + 1] ? mx : d->net[s].dims[l + 1];
        int64_t ld = (int64_t)mx * d->n_members;
        if (n_act >= 0 && l >= n_act) {
            if (w) { w->act[l] = nullptr; w->ld[l] = ld; }
            continue;
        }
        // (layer 0 doubles as the tile-major d E / d act0 buffer: one partly filled 64-row block per species)
        float *a = (float *)take(sizeof(float) * (size_t)ld * (size_t)(n + 1 + (l == 0 ? 64 * ANIHIP_MAX_SPECIES : 0)));
        if (w) { w->act[l] = a; w->ld[l] = ld; }
    }
    return off;
}

// pre-activation buffers of the hidden layers behind offset `off` (GELU training passes only; CELU: none)
static size_t carve_zp(const anihip_mlp_desc *d, int64_t n, char *base, size_t off, MlpWorkspace *w)
{
    const int nh = d->net[0].n_layers - 1;
    for (int l = 0; l < ANIHIP_MAX_LAYERS; ++l)
        if (w) w->zp[l] = nullptr;
    if (d->activation != ANIHIP_ACT_GELU) return off;
    for (int l = 0; l < nh; ++l) {
        int mx = 0;
        for (int s = 0; s < d->num_species; ++s) mx = mx > d->net[s].dims[l + 1] ? mx : d->net[s].dims[l + 1];
        if (w) w->zp[l] = base ? (float *)(base + off) : nullptr;
        off += align256(sizeof(float) * (size_t)mx * d->n_members * (size_t)(n + 1));
    }
    return off;
}

// training pass: the inference workspace + one gradient buffer per hidden layer (the activations are kept)
static size_t mlp_train_carve(const anihip_mlp_desc *d, int64_t n, char *base, MlpWorkspace *w,
                              float **dlt /* [ANIHIP_MAX_LAYERS] */)
{
    size_t off = align256(mlp_carve(d, n, base, w));
    off = carve_zp(d, n, base, off, w);
    const int nh = d->net[0].n_layers - 1;
    for (int l = 0; l < nh; ++l) {
        int mx = 0;
        for (int s = 0; s < d->num_species; ++s) mx = mx > d->net[s].dims[l + 1] ? mx : d->net[s].dims[l + 1];
        const size_t bytes = sizeof(float) * (size_t)mx * d->n_members * (size_t)(n + 1);
        if (dlt) dlt[l] = base ? (float *)(base + off) : nullptr;
        off += align256(bytes);
    }
    return off;
}

// tangent pass: the inference workspace + four more buffers per hidden layer (zdot, adot, p, q)
static size_t mlp_tangent_carve(const anihip_mlp_desc *d, int64_t n, char *base, MlpWorkspace *w,
                                float *(*buf)[ANIHIP_MAX_LAYERS] /* [4] */)
{
    size_t off = align256(mlp_carve(d, n, base, w));
    off = carve_zp(d, n, base, off, w);
    const int nh = d->net[0].n_layers - 1;
    for (int k = 0; k < 4; ++k)
        for (int l = 0; l < nh; ++l) {
            int mx = 0;
            for (int s = 0; s < d->num_species; ++s) mx = mx > d->net[s].dims[l + 1] ? mx : d->net[s].dims[l + 1];
            const size_t bytes = sizeof(float) * (size_t)mx * d->n_members * (size_t)(n + 1);
            if (buf) buf[k][l] = base ? (float *)(base + off) : nullptr;
            off += align256(bytes);
        }
    return off;
}

}  // namespace anihip

using namespace anihip;

extern "C" size_t anihip_mlp_workspace_bytes(const anihip_mlp_desc *d, int64_t n_central)
{
    if (!d || n_central < 0) return 0;
    return mlp_carve(d, n_central, nullptr, nullptr);
}

static int check_desc(const anihip_mlp_desc *d)
{
    ANIHIP_REQUIRE(d, "null descriptor");
    ANIHIP_REQUIRE(d->num_species >= 1 && d->num_species <= MAX_S - 1, "num_species must be 1..7");
    ANIHIP_REQUIRE(d->n_members >= 1 && d->n_members <= 64, "n_members must be 1..64");
    ANIHIP_REQUIRE(d->aev_len % BK == 0, "aev_len must be a multiple of %d", BK);
    ANIHIP_REQUIRE(d->precision == ANIHIP_MLP_FP32 || d->precision == ANIHIP_MLP_F16X3, "unknown precision");
    ANIHIP_REQUIRE(d->activation == ANIHIP_ACT_CELU || d->activation == ANIHIP_ACT_GELU, "unknown activation");
    ANIHIP_REQUIRE(d->aev_radial_len >= 0 && d->aev_radial_len <= d->aev_len &&
                       (d->aev_radial_len == 0 || (d->aev_len - d->aev_radial_len) % 32 == 0),
                   "aev_radial_len: the angular part must be a multiple of 32 long");
    const int nl = d->net[0].n_layers;
    ANIHIP_REQUIRE(nl >= 2 && nl <= ANIHIP_MAX_LAYERS, "n_layers must be 2..%d", ANIHIP_MAX_LAYERS);
    for (int s = 0; s < d->num_species; ++s) {
        const anihip_species_net &n = d->net[s];
        ANIHIP_REQUIRE(n.n_layers == nl, "all species must have the same depth");
        ANIHIP_REQUIRE(n.dims[0] == d->aev_len && n.dims[nl] == 1, "species %d: bad first/last width", s);
        for (int l = 1; l < nl; ++l)
            ANIHIP_REQUIRE(n.dims[l] > 0 && n.dims[l] % 32 == 0, "species %d: hidden width %d not padded to 32",
                           s, n.dims[l]);
        for (int l = 0; l < nl; ++l) {
            ANIHIP_REQUIRE(n.w[l] && n.bias[l], "species %d layer %d: null parameter pointer", s, l);
            if (l < nl - 1) ANIHIP_REQUIRE(n.wt[l], "species %d layer %d: null transposed weights", s, l);
            if (l < nl - 1 && d->precision == ANIHIP_MLP_F16X3)
                ANIHIP_REQUIRE(n.wh[l] && n.wth[l] && n.wh_scale[l] > 0.f,
                               "species %d layer %d: missing fp16 weight planes", s, l);
        }
    }
    return 0;
}

// widths the fused network kernel covers: 8 waves x one 32-column block
static bool fused_dims_supported(int H1, int H2, int H3)
{
    return H1 <= FR_MAXH && H2 <= FR_MAXH && H3 <= FR_MAXH;
}

// Which kernels one anihip_mlp_forward_backward call over n central atoms runs -- decided from the descriptor, n and
// whether d E / d AEV is wanted alone, so that the workspace query and the call agree.
struct FbPlan {
    bool fused;       // k_mlp_fused (f16x3, three hidden layers of width <= 256, at most 32 AEV slabs)
    bool fused_l0b;   // ... with the layer-0 backward as its phase 5 (no d E / d act0 buffer)
    bool big_tiles;   // 256 x 256 tiles for the layer-0 GEMMs outside the fused kernel
    int fused_rows;   // atoms per tile of the fused kernel
    int n_act;        // hidden-layer buffers of the workspace this call touches (mlp_carve)
};
static FbPlan fb_plan(const anihip_mlp_desc *d, int64_t n, bool want_grad)
{
    FbPlan p{};
    const int S = d->num_species, nh = d->net[0].n_layers - 1, L = d->aev_len;
    const bool h3 = d->precision == ANIHIP_MLP_F16X3;
    const int kp_rad = h3 ? d->aev_radial_len : 0;
    const int K0p = kp_rad > 0 ? 32 * ((kp_rad + 31) / 32 + (L - kp_rad) / 32) : ((L + 31) / 32) * 32;
    p.fused = h3 && nh == 3 && K0p <= 32 * 32 && L % 4 == 0;
    for (int s = 0; s < S && p.fused; ++s) {
        const anihip_species_net &nn = d->net[s];
        p.fused = p.fused && nn.whf[0] && nn.whf[1] && nn.whf[2] && nn.wthf[1] && nn.wthf[2] && nn.fused_bounds &&
                  fused_dims_supported(nn.dims[1], nn.dims[2], nn.dims[3]);
    }
    if (d->flags & ANIHIP_MLP_FLAG_NO_FUSED) p.fused = false;
    // 256 x 256 tiles for the layer-0 GEMMs once there are enough rows to fill the chip with them
    p.big_tiles = h3 && n >= 16384;
    if (d->flags & ANIHIP_MLP_FLAG_BIG_TILES) p.big_tiles = h3;
    if (d->flags & ANIHIP_MLP_FLAG_SMALL_TILES) p.big_tiles = false;
    p.fused_rows = 64;   // (the 32-atom / two-workgroups-per-CU tiling of rounds 1-4 was 3 % slower and spilled 300 registers: removed in round 5)
    // Layer-0 backward INSIDE the fused kernel (its phase 5): a workgroup owns a tile through all members and adds the
    // members' d E / d AEV in place -- no d act0 round trip through HBM (8 KB per atom written and read back), no layer-0
    // backward launch.  Tiles are then the unit of work (not tile x member items), so it needs enough of them to balance
    // over the CUs: from FUSED_L0B_MIN_ATOMS atoms on.  Smaller inputs keep the member-major sweep + a backward GEMM.
    // (phase 5 hands partial sums between waves through 32 KB of LDS in X1's place: 2 planes x 64 rows x (H2 + 8) halves)
    // (... and its k range is cut in two halves of >= 2 steps each for the two waves of a SIMD: first hidden layers of >= 64
    // columns -- with 32 the first half would be empty and its ring would read in front of the member's planes)
    auto l0b_ok = [&](int s) { return d->net[s].wthf[0] != nullptr && d->net[s].dims[2] >= 128 && d->net[s].dims[1] >= 64; };
    // (CELU networks: the GELU instantiation with phase 5 spilled registers and served ANI-2xr on > 65 536 atoms only --
    // removed in round 5, those systems take the d act0 hand-over + layer-0 backward GEMM)
    const bool celu = d->activation == ANIHIP_ACT_CELU;
    p.fused_l0b = p.fused && want_grad && celu && n >= FUSED_L0B_MIN_ATOMS &&
                  !(d->flags & (ANIHIP_MLP_FLAG_NO_FUSED_L0B | ANIHIP_MLP_FLAG_SMALL_TILES));
    if (d->flags & ANIHIP_MLP_FLAG_FUSED_L0B)   // (forced, e.g. by the tests on small inputs; the call checks that it can)
        p.fused_l0b = p.fused && want_grad && celu;
    for (int s = 0; s < S && p.fused_l0b; ++s) p.fused_l0b = l0b_ok(s);
    p.n_act = !p.fused ? nh : ((want_grad && !p.fused_l0b) ? 1 : 0);
    return p;
}

extern "C" size_t anihip_mlp_forward_backward_workspace_bytes(const anihip_mlp_desc *d, int64_t n_central, int want_grad)
{
    if (!d || n_central < 0 || d->net[0].n_layers < 2 || d->net[0].n_layers > ANIHIP_MAX_LAYERS) return 0;
    return mlp_carve(d, n_central, nullptr, nullptr, fb_plan(d, n_central, want_grad != 0).n_act);
}

template <int EPI>
static int launch_gemm_big(hipStream_t stream, GemmArgs &g, int64_t n_rows_total)
{
    // 256 x 256 tiles: recompute the tile upper bounds for this tiling
    const size_t lds = sizeof(_Float16) * 2 * H2_STAGE + 64;
    int nmax = 0;
    for (int s = 0; s < g.S; ++s) nmax = nmax > g.prob[s].N ? nmax : g.prob[s].N;
    GemmArgs h = g;
    h.ncol_max = (nmax + BN2 - 1) / BN2;
    // compacted output columns (layer-0 backward with slab masks): one workgroup per row tile walks the
    // groups of 8 active column blocks itself.  (Launching a workgroup per potential column tile and
    // letting the empty ones exit cost 3x the kernel time: every 128-KB-LDS workgroup occupies a CU slot.)
    if (EPI == EPI_SCATTER && g.stage_mask) h.ncol_max = 1;
    h.nrow_tiles_ub = (int)((n_rows_total + BM2 - 1) / BM2) + g.S;
    ANIHIP_CHECK_HIP(hipFuncSetAttribute((const void *)k_gemm_h2<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds));
    const int64_t total = (int64_t)h.nrow_tiles_ub * h.ncol_max * h.batch;
    if (EPI == EPI_SCATTER && g.stage_mask && g.a_tm_members > 0 && g.c_scatter && !g.a_gather && h.batch == 1) {
        // row tiles with few flagged slabs go to the skinny kernel, the others stay here (each kernel works out a
        // tile's mask and leaves the other kernel's tiles alone)
        h.skinny = 1;
        const size_t lds3 = sizeof(_Float16) * 2 * L0B_STAGE + 64;
        ANIHIP_CHECK_HIP(hipFuncSetAttribute((const void *)k_gemm_l0b, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)lds3));
        hipLaunchKernelGGL(k_gemm_l0b, dim3((unsigned)h.nrow_tiles_ub), dim3(L0B_THREADS), lds3, stream, h);
    }
    hipLaunchKernelGGL((k_gemm_h2<EPI>), dim3((unsigned)total), dim3(GEMM2_THREADS), lds, stream, h);
    return 0;
}

template <int EPI>
static void launch_gemm(hipStream_t stream, GemmArgs &g, bool f16x3)
{
    const int64_t total = (int64_t)g.nrow_tiles_ub * g.ncol_max * g.batch;
    if (f16x3) {
#ifdef ANIHIP_DEV_TRACE
        const bool tr = EPI == EPI_SCATTER && getenv("ANIHIP_GEMM_TRACE");
        if (tr) (void)hipMemsetAsync(g_gemm_trace, 0, sizeof(g_gemm_trace), stream);   // (symbol address: dev builds only)
#endif
        hipLaunchKernelGGL((k_gemm_h<EPI>), dim3((unsigned)total), dim3(GEMM_THREADS), 0, stream, g);
#ifdef ANIHIP_DEV_TRACE
        if (tr) {
            static unsigned long long host[8 * 4096];
            (void)hipStreamSynchronize(stream);
            (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gemm_trace), sizeof(host));
            double sum[4] = {0, 0, 0, 0}, cyc[4] = {0, 0, 0, 0};
            unsigned long long w0 = ~0ull, w1 = 0;
            int live = 0;
            for (int b = 0; b < 4096 && b < total; ++b) {
                const unsigned long long *h = host + 8 * b;
                if (!h[7]) continue;   // (exited before the end stamp)
                ++live;
                for (int k = 0; k < 3; ++k) {
                    sum[k] += (double)(h[2 * k + 3] - h[2 * k + 1]) * 10.0;   // ns (100 MHz)
                    cyc[k] += (double)(h[2 * k + 2] - h[2 * k]);
                }
                w0 = h[1] < w0 ? h[1] : w0;
                w1 = h[7] > w1 ? h[7] : w1;
            }
            if (live)
                fprintf(stderr, "k_gemm_h trace: %d live of %lld workgroups; mean ns  prologue %.0f  loop %.0f  epilogue %.0f;"
                        " mean shader clocks %.0f %.0f %.0f; first start -> last end %.0f ns\n", live, (long long)total,
                        sum[0] / live, sum[1] / live, sum[2] / live, cyc[0] / live, cyc[1] / live, cyc[2] / live,
                        (double)(w1 - w0) * 10.0);
        }
#endif
    } else
        hipLaunchKernelGGL((k_gemm<EPI>), dim3((unsigned)total), dim3(GEMM_THREADS), 0, stream, g);
}

template <int EPI>
static void launch_gemm_fp32(hipStream_t stream, GemmArgs &g)
{
    const int64_t total = (int64_t)g.nrow_tiles_ub * g.ncol_max * g.batch;
    hipLaunchKernelGGL((k_gemm<EPI>), dim3((unsigned)total), dim3(GEMM_THREADS), 0, stream, g);
}

extern "C" int anihip_mlp_forward_backward(void *stream_, const anihip_mlp_desc *d, int64_t n_atoms,
                                           int64_t lo, int64_t hi, const int32_t *species, const float *aev,
                                           const uint32_t *slab_mask, void *workspace, size_t workspace_bytes,
                                           float *atomic_e, float *grad_aev, float *member_e)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = check_desc(d)) return rc;
    ANIHIP_REQUIRE(species && aev && workspace && atomic_e, "null pointer argument");
    ANIHIP_REQUIRE(0 <= lo && lo <= hi && hi <= n_atoms, "central range outside 0..n_atoms");
    const int64_t n = hi - lo;
    if (n == 0) return 0;
    const FbPlan plan = fb_plan(d, n, grad_aev != nullptr);
    ANIHIP_REQUIRE(workspace_bytes >= mlp_carve(d, n, nullptr, nullptr, plan.n_act),
                   "workspace too small (anihip_mlp_forward_backward_workspace_bytes)");
    MlpWorkspace w;
    mlp_carve(d, n, (char *)workspace, &w, plan.n_act);
    const int S = d->num_species, M = d->n_members, nl = d->net[0].n_layers, nh = nl - 1;
    const int L = d->aev_len;
    const bool h3 = d->precision == ANIHIP_MLP_F16X3;
    // layer-0 reduction length: plain AEV order padded to 32, or the slab order of the fp16 planes
    const int kp_rad = h3 ? d->aev_radial_len : 0;
    const int K0p = kp_rad > 0 ? 32 * ((kp_rad + 31) / 32 + (L - kp_rad) / 32) : ((L + 31) / 32) * 32;
    const float alpha = d->celu_alpha, inv_alpha = 1.0f / d->celu_alpha;

    const int nrow_ub = (int)((n + BM - 1) / BM) + S;
    auto ncol_of = [&](int l_out, bool cat) {
        int mx = 0;
        for (int s = 0; s < S; ++s) {
            int N = d->net[s].dims[l_out] * (cat ? M : 1);
            mx = mx > N ? mx : N;
        }
        return (mx + BN - 1) / BN;
    };

    // fused network kernel (f16x3, three hidden layers of width <= 256, at most 32 AEV slabs): one kernel
    // from the AEV rows to d E / d act0, then the layer-0 backward GEMM
    const bool fused = plan.fused;
    // (the layer-by-layer kernels, the 32-atom tiling and the training passes implement CELU only)
    ANIHIP_REQUIRE(d->activation == ANIHIP_ACT_CELU || fused,
                   "GELU networks run through the fused network kernel only: f16x3 precision, 3 hidden layers <= 256 wide");
    // 256 x 256 tiles for the layer-0 GEMMs once there are enough rows to fill the chip with them
    int d0_tm = 0;
    const bool big_tiles = plan.big_tiles;
    // per-atom slab flags: honoured by the 256 x 256 kernels on slab-ordered planes
    const uint32_t *smask = (big_tiles && kp_rad > 0 && K0p <= 32 * 32) ? slab_mask : nullptr;
    if (d->flags & ANIHIP_MLP_FLAG_NO_SLAB_MASK) smask = nullptr;

    // 1. bucket by species (+ the tile table of the fused kernel and the padding rows, one launch for small inputs)
    const int fused_rows = plan.fused_rows;
    const int64_t fused_tiles = (n + fused_rows - 1) / fused_rows + S;
    const int n_slabs = K0p / 32;
    // per-atom slab flags for the fused kernel's tile masks: the ANI slab order (kp_rad > 0: anihip_aev_forward's flags for
    // the 16 / 32-column grids) or, for any other row layout, the plain 32-column slabs (kp_rad = 0: the flags of the
    // general AEV kernel; rows of at most 1024 columns)
    const uint32_t *tab_mask = ((kp_rad > 0 || (h3 && K0p <= 32 * 32)) && !(d->flags & ANIHIP_MLP_FLAG_NO_SLAB_MASK)) ? slab_mask : nullptr;
    const uint32_t all_slabs = n_slabs >= 32 ? 0xFFFFFFFFu : ((1u << n_slabs) - 1u);
    const bool small_prep = n <= SMALL_PREP_MAX;
    if (small_prep) {
        const size_t lds = sizeof(int) * (size_t)fused_tiles;
        ANIHIP_CHECK_HIP(hipFuncSetAttribute((const void *)k_small_prep, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)lds));
        int64_t pad_blocks = (n + SMALL_PREP_WAVES - 1) / SMALL_PREP_WAVES;   // one atom per wave
        if (pad_blocks < 1) pad_blocks = 1;
        hipLaunchKernelGGL(k_small_prep, dim3((unsigned)(1 + pad_blocks)), dim3(SMALL_PREP_WAVES * WAVE), lds, stream,
                           lo, hi, species, S, w.ctl, w.perm, tab_mask, all_slabs, (int)fused_tiles, fused_rows,
                           fused ? w.tile_tab : (int4 *)nullptr, w.tile_rows, atomic_e, grad_aev, L, member_e, M,
                           n_atoms);
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
    } else {
        zero_words_async(stream, w.ctl, sizeof(int) * (CTL_WORDS + AMAX_WORDS));
        const unsigned cblk = (unsigned)((n + 4 * SP_CHUNK - 1) / (4 * SP_CHUNK));
        // (scratch: the per-member energies buffer is written only later)
        int *chunk_cnt = reinterpret_cast<int *>(w.member_part);
        const int n_chunks = (int)((n + SP_CHUNK - 1) / SP_CHUNK);
        hipLaunchKernelGGL(k_sp_count, dim3(cblk), dim3(256), 0, stream, lo, hi, species, S, chunk_cnt);
        hipLaunchKernelGGL(k_sp_offsets, dim3(1), dim3(256), 0, stream, S, n_chunks, chunk_cnt, w.ctl);
        hipLaunchKernelGGL(k_sp_scatter, dim3(cblk), dim3(256), 0, stream, lo, hi, species, S, w.ctl, chunk_cnt, w.perm,
                           atomic_e, grad_aev, L, member_e, M, n_atoms);
    }

    // 2. forward through the hidden layers
    for (int l = 0; l < (fused ? 0 : nh); ++l) {
        GemmArgs g{};
        g.ctl = w.ctl; g.S = S; g.alpha = alpha; g.inv_alpha = inv_alpha;
        g.nrow_tiles_ub = nrow_ub;
        g.C = w.act[l]; g.ldc = w.ld[l]; g.c_scatter = nullptr; g.n_store = 0;
        if (l == 0) {
            g.A = aev; g.lda = L; g.a_gather = w.perm; g.batch = 1;
            g.ncol_max = ncol_of(1, true);
        } else {
            g.A = w.act[l - 1]; g.lda = w.ld[l - 1]; g.a_gather = nullptr; g.batch = M;
            g.ncol_max = ncol_of(l + 1, false);
        }
        for (int s = 0; s < S; ++s) {
            const anihip_species_net &nn = d->net[s];
            GemmProblem &p = g.prob[s];
            p.B = nn.w[l]; p.bias = nn.bias[l];
            if (l == 0) {
                p.K = nn.dims[0]; p.N = nn.dims[1] * M; p.ldb = p.N;
                p.a_boff = 0; p.c_boff = 0; p.b_stride = 0; p.bias_stride = 0;
            } else {
                p.K = nn.dims[l]; p.N = nn.dims[l + 1]; p.ldb = p.N;
                p.a_boff = nn.dims[l]; p.c_boff = nn.dims[l + 1];
                p.b_stride = (int64_t)p.K * p.N; p.bias_stride = p.N;
            }
            if (h3) {  // B planes shaped like wt[l]: [N][K], K contiguous
                p.Bh = (const _Float16 *)nn.wh[l];
                p.k_valid = p.K;
                p.w_inv_scale = 1.0f / nn.wh_scale[l];
                if (l == 0) {
                    p.K = K0p; p.ldbh = K0p; p.bh_stride = 0; p.bh_plane = (int64_t)p.N * K0p;
                } else {
                    p.ldbh = p.K; p.bh_stride = (int64_t)p.N * p.K; p.bh_plane = (int64_t)M * p.N * p.K;
                }
            }
        }
        g.amax = w.amax; g.amax_out = h3 ? l : -1; g.amax_in = (h3 && l > 0) ? l - 1 : -1;
        g.a_static_scale = 4.0f;  // layer-0 input: |aev| < 16376 by construction (see include/anihip.h)
        if (l == 0) { g.kp_rad = kp_rad; g.stage_mask = smask; }
        if (h3 && l == 0 && big_tiles) {
            if (int rc = launch_gemm_big<EPI_BIAS_CELU>(stream, g, n)) return rc;
        } else {
            launch_gemm<EPI_BIAS_CELU>(stream, g, h3);
        }
    }

    // layer-0 backward inside the fused kernel (fb_plan)
    const bool fused_l0b = plan.fused_l0b;
    if (d->flags & ANIHIP_MLP_FLAG_FUSED_L0B)
        ANIHIP_REQUIRE(fused_l0b, "ANIHIP_MLP_FLAG_FUSED_L0B needs the fused kernel with CELU networks, wthf[0], first hidden layers of >= 64 and second hidden layers of >= 128 columns");

    FinishArgs fin{};
    // few atoms: the layer-0 backward runs in the 8-wave 128 x 128 kernel (needs the slab flags for its column compaction)
    const bool use_l0s = grad_aev && !fused_l0b && h3 && !big_tiles && kp_rad > 0 && K0p <= 32 * 32 && slab_mask &&
                         !(d->flags & ANIHIP_MLP_FLAG_NO_SLAB_MASK);
    if (fused) {
        FusedArgs f{};
        size_t lds = 0;
        // tiling: 64 atoms x 8 waves, one workgroup per CU
        const int rows = fused_rows;
        for (int s = 0; s < S; ++s) {
            const anihip_species_net &nn = d->net[s];
            FusedSpecies &fs = f.sp[s];
            fs.H1 = nn.dims[1]; fs.H2 = nn.dims[2]; fs.H3 = nn.dims[3];
            fs.w0 = (const _Float16 *)nn.whf[0];
            fs.w1 = (const _Float16 *)nn.whf[1]; fs.w2 = (const _Float16 *)nn.whf[2];
            fs.w2t = (const _Float16 *)nn.wthf[2]; fs.w1t = (const _Float16 *)nn.wthf[1];
            fs.w0t = (const _Float16 *)nn.wthf[0];
            fs.is0 = 1.0f / nn.wh_scale[0]; fs.is1 = 1.0f / nn.wh_scale[1]; fs.is2 = 1.0f / nn.wh_scale[2];
            fs.b0 = nn.bias[0]; fs.b1 = nn.bias[1]; fs.b2 = nn.bias[2]; fs.w3 = nn.w[3]; fs.b3 = nn.bias[3];
            fs.bounds = nn.fused_bounds;
            const size_t xu = fs.H1 > fs.H3 ? fs.H1 : fs.H3;
            size_t halves = 2 * (size_t)rows * (fs.H2 + FR_XPAD) + 2 * (size_t)rows * (xu + FR_XPAD);
            const size_t slab = 2 * (size_t)rows * FR_SLAB_LD;
            if (halves < 3 * FR_GROUP * slab) halves = 3 * FR_GROUP * slab;   // staging slots 1..3
            halves += FusedCfg<2, 1>::FIXED_HALVES;
            lds = lds > halves * 2 ? lds : halves * 2;
        }
        f.ctl = w.ctl; f.amax = w.amax; f.aev = aev; f.L = L; f.kp_rad = kp_rad; f.n_slabs = n_slabs;
        f.slab_mask = tab_mask;
        f.d0 = w.act[0]; f.ld0 = w.ld[0]; f.perm = w.perm;
        // tile-major hand-over to the 256 x 256 layer-0 backward GEMM (the 128 x 128 kernel of small inputs reads rows)
        f.d0_tm = (big_tiles && grad_aev) ? 1 : 0;
        if (d->flags & ANIHIP_MLP_FLAG_D0_ROWS) f.d0_tm = 0;
        d0_tm = f.d0_tm;
        f.tile_tab = w.tile_tab; f.tile_rows = w.tile_rows;
        f.member_part = w.member_part; f.S = S; f.M = M; f.alpha = alpha; f.inv_alpha = inv_alpha;
        f.want_grad = grad_aev ? 1 : 0;
        f.owner = 0;   // (member-major sweep; the layer-0 backward inside the kernel switches to owner order below)
        f.l0b = fused_l0b ? 1 : 0;
        f.grad_aev = grad_aev;
        if (fused_l0b) { f.owner = FUSED_OWNER_GROUP; f.d0 = nullptr; }
        const bool gelu = d->activation == ANIHIP_ACT_GELU;
        // (off by default: the backward GEMMs of the large-system path with two products -- forces then differ from the
        // three-product result by ~1e-6 Ha/A, inside north_star's 1e-4 gate and outside this package's 5e-6 regression gate)
        const bool bwd2 = fused_l0b && (d->flags & ANIHIP_MLP_FLAG_BWD_TWO_PRODUCTS);
        const void *kfn = fused_l0b ? (bwd2 ? (const void *)k_mlp_fused<2, 1, 0, true, false, true> : (const void *)k_mlp_fused<2, 1, 0, true>)
                                    : (gelu ? (const void *)k_mlp_fused<2, 1, 1, false> : (const void *)k_mlp_fused<2, 1, 0, false>);
        ANIHIP_CHECK_HIP(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int64_t tiles = fused_tiles;
        f.tiles_total = (int)tiles;
        // persistent workgroups over the (member, tile) items, as many as are resident at once
        static int n_cus = 0;
        if (n_cus == 0) {
            int dev = 0, v = 0;
            ANIHIP_CHECK_HIP(hipGetDevice(&dev));
            ANIHIP_CHECK_HIP(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
            n_cus = v > 0 ? v : 256;
        }
        const int64_t items = tiles * M;
        const int64_t resident = (int64_t)n_cus * (2 * lds <= 160 * 1024 ? 2 : 1);
        const int64_t units = f.owner ? tiles : items;
        const int64_t grid = units < resident ? units : resident;
        if (!small_prep)
            hipLaunchKernelGGL(k_tile_table, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, stream, w.ctl, S, w.perm,
                               f.slab_mask, all_slabs, (int)tiles, rows, w.tile_tab, w.tile_rows);
#ifdef ANIHIP_DEV_TRACE   // development builds only (tools/fused_trace.py): per-item phase stamps, allocates and synchronises
        const char *trace_path = getenv("ANIHIP_FUSED_TRACE");
        const size_t trace_words = (size_t)32 * 8 * items;   // [item][wave][32]
        if (trace_path) {
            ANIHIP_CHECK_HIP(hipMalloc((void **)&f.trace, sizeof(unsigned long long) * trace_words));
            ANIHIP_CHECK_HIP(hipMemset(f.trace, 0, sizeof(unsigned long long) * trace_words));
        }
#endif
        if (bwd2) hipLaunchKernelGGL((k_mlp_fused<2, 1, 0, true, false, true>), dim3((unsigned)grid), dim3(512), lds, stream, f);
        else if (fused_l0b) hipLaunchKernelGGL((k_mlp_fused<2, 1, 0, true>), dim3((unsigned)grid), dim3(512), lds, stream, f);
        else if (gelu) hipLaunchKernelGGL((k_mlp_fused<2, 1, 1, false>), dim3((unsigned)grid), dim3(512), lds, stream, f);
        else hipLaunchKernelGGL((k_mlp_fused<2, 1, 0, false>), dim3((unsigned)grid), dim3(512), lds, stream, f);
#ifdef ANIHIP_DEV_TRACE
        if (trace_path) {
            ANIHIP_CHECK_HIP(hipStreamSynchronize(stream));
            std::vector<unsigned long long> host(trace_words);
            ANIHIP_CHECK_HIP(hipMemcpy(host.data(), f.trace, host.size() * 8, hipMemcpyDeviceToHost));
            ANIHIP_CHECK_HIP(hipFree(f.trace));
            if (FILE *fp = fopen(trace_path, "wb")) {
                fwrite(host.data(), 8, host.size(), fp);
                fclose(fp);
            }
        }
#endif
        fin.ctl = w.ctl; fin.perm = w.perm; fin.member_part = w.member_part; fin.atomic_e = atomic_e;
        fin.member_e = member_e; fin.n_atoms = n_atoms; fin.S = S; fin.M = M; fin.first_block = 0;
        if (!use_l0s) {   // (the 8-wave layer-0 backward of small inputs does this in extra workgroups)
            int64_t fb = (n + 255) / 256;
            if (fb > 2048) fb = 2048;
            hipLaunchKernelGGL(k_fused_finish, dim3((unsigned)fb), dim3(256), 0, stream, fin);
        }
    }

    // 3. output layer (+ seed of the backward pass, written in place over the last activations)
    if (!fused) {
        HeadArgs h{};
        for (int s = 0; s < S; ++s) {
            h.w[s] = d->net[s].w[nl - 1];
            h.bias[s] = d->net[s].bias[nl - 1];
            h.Hp[s] = d->net[s].dims[nl - 1];
        }
        h.ctl = w.ctl; h.perm = w.perm; h.act = w.act[nh - 1]; h.ld = w.ld[nh - 1];
        h.atomic_e = atomic_e; h.member_e = member_e; h.n_atoms = n_atoms; h.S = S; h.M = M;
        h.inv_alpha = inv_alpha; h.want_grad = grad_aev ? 1 : 0;
        h.amax = h3 ? w.amax : nullptr; h.amax_out = 3;
        int64_t blocks = (n + 3) / 4;
        if (blocks > 256 * 8) blocks = 256 * 8;
        hipLaunchKernelGGL(k_head, dim3((unsigned)blocks), dim3(256), 0, stream, h);
    }

    // 4. backward to the AEV rows
    if (grad_aev && !fused_l0b) {
        for (int l = fused ? 0 : nh - 1; l >= 0; --l) {
            GemmArgs g{};
            g.ctl = w.ctl; g.S = S; g.alpha = alpha; g.inv_alpha = inv_alpha;
            g.nrow_tiles_ub = nrow_ub;
            g.A = w.act[l]; g.lda = w.ld[l]; g.a_gather = nullptr;
            if (l == 0) {
                g.batch = 1; g.C = grad_aev; g.ldc = L; g.c_scatter = w.perm; g.n_store = L;
                g.ncol_max = (K0p + BN - 1) / BN;
            } else {
                g.batch = M; g.C = w.act[l - 1]; g.ldc = w.ld[l - 1]; g.c_scatter = nullptr;
                g.ncol_max = ncol_of(l, false);
            }
            for (int s = 0; s < S; ++s) {
                const anihip_species_net &nn = d->net[s];
                GemmProblem &p = g.prob[s];
                p.B = nn.wt[l]; p.bias = nullptr; p.bias_stride = 0;
                if (l == 0) {
                    p.K = nn.dims[1] * M; p.N = K0p; p.ldb = p.N;
                    p.a_boff = 0; p.c_boff = 0; p.b_stride = 0;
                } else {
                    p.K = nn.dims[l + 1]; p.N = nn.dims[l]; p.ldb = p.N;
                    p.a_boff = nn.dims[l + 1]; p.c_boff = nn.dims[l];
                    p.b_stride = (int64_t)p.K * p.N;
                }
                if (h3) {  // B planes shaped like w[l]: [N = layer input index][K = layer output index]
                    p.Bh = (const _Float16 *)nn.wth[l];
                    p.k_valid = p.K;
                    p.w_inv_scale = 1.0f / nn.wh_scale[l];
                    p.ldbh = p.K;
                    p.bh_stride = l == 0 ? 0 : (int64_t)p.N * p.K;
                    p.bh_plane = (l == 0 ? 1 : (int64_t)M) * p.N * p.K;
                }
            }
            g.amax = w.amax;
            g.amax_in = h3 ? 3 + (nh - 1 - l) : -1;
            g.amax_out = (h3 && l > 0) ? 3 + (nh - l) : -1;
            g.a_static_scale = 1.0f;
            if (l == 0) {
                // (the layer-0 backward compacts its output columns in both tilings; the forward honours the flags only
                // in the 256 x 256 kernel)
                g.kp_rad = kp_rad;
                g.stage_mask = (h3 && kp_rad > 0 && K0p <= 32 * 32 && !(d->flags & ANIHIP_MLP_FLAG_NO_SLAB_MASK)) ? slab_mask : nullptr;
            }
            if (l == 0 && d0_tm) {
                g.a_tm_members = M;
                for (int s = 0; s < S; ++s) g.a_tm_h[s] = d->net[s].dims[1];
            }
            if (l == 0 && h3 && big_tiles) {
                if (int rc = launch_gemm_big<EPI_SCATTER>(stream, g, n)) return rc;
            } else if (l == 0 && use_l0s) {
                const int64_t total = (int64_t)g.nrow_tiles_ub * g.ncol_max;
                const size_t lds = sizeof(_Float16) * L0S_BUFS * 4 * H_PLANE;
                ANIHIP_CHECK_HIP(hipFuncSetAttribute((const void *)k_gemm_l0s, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     (int)lds));
                int64_t fb = 0;
                if (fused) {
                    fb = (n + L0S_THREADS - 1) / L0S_THREADS;
                    fin.first_block = (int)total;
                }
                hipLaunchKernelGGL(k_gemm_l0s, dim3((unsigned)(total + fb)), dim3(L0S_THREADS), lds, stream, g, fin);
            } else if (l == 0) {
                launch_gemm<EPI_SCATTER>(stream, g, h3);
            } else {
                launch_gemm<EPI_DCELU>(stream, g, h3);
            }
        }
    }
    ANIHIP_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" size_t anihip_mlp_train_workspace_bytes(const anihip_mlp_desc *d, int64_t n_central)
{
    if (!d || n_central < 0) return 0;
    return mlp_train_carve(d, n_central, nullptr, nullptr, nullptr);
}

// bucketing + exact-fp32 forward with the activations kept in the workspace (first half of the training pass)
static int train_forward(hipStream_t stream, const anihip_mlp_desc *d, int64_t n_atoms, int64_t lo, int64_t hi,
                         const int32_t *species, const float *aev, MlpWorkspace &w, float *atomic_e,
                         float *grad_aev)
{
    const int S = d->num_species, M = d->n_members, nl = d->net[0].n_layers, nh = nl - 1, L = d->aev_len;
    const int64_t n = hi - lo;
    zero_words_async(stream, w.ctl, sizeof(int) * (CTL_WORDS + AMAX_WORDS));
    const unsigned nblk = (unsigned)((n + 255) / 256);
    const unsigned cblk = (unsigned)((n + 4 * SP_CHUNK - 1) / (4 * SP_CHUNK));
    {   // (scratch: the per-member energies buffer is written only later)
        int *chunk_cnt = reinterpret_cast<int *>(w.member_part);
        const int n_chunks = (int)((n + SP_CHUNK - 1) / SP_CHUNK);
        hipLaunchKernelGGL(k_sp_count, dim3(cblk), dim3(256), 0, stream, lo, hi, species, S, chunk_cnt);
        hipLaunchKernelGGL(k_sp_offsets, dim3(1), dim3(256), 0, stream, S, n_chunks, chunk_cnt, w.ctl);
        hipLaunchKernelGGL(k_sp_scatter, dim3(cblk), dim3(256), 0, stream, lo, hi, species, S, w.ctl, chunk_cnt, w.perm,
                           atomic_e, grad_aev, L, (float *)nullptr, M, n_atoms);
    }
    for (int l = 0; l < nh; ++l) {
        GemmArgs g{};
        g.ctl = w.ctl; g.S = S; g.alpha = d->celu_alpha; g.inv_alpha = 1.0f / d->celu_alpha;
        g.nrow_tiles_ub = (int)((n + BM - 1) / BM) + S;
        g.amax_in = g.amax_out = -1;
        g.C = w.act[l]; g.ldc = w.ld[l];
        g.act = d->activation; g.Xout = w.zp[l];
        int wmax = 0;
        for (int s = 0; s < S; ++s) wmax = wmax > d->net[s].dims[l + 1] ? wmax : d->net[s].dims[l + 1];
        if (l == 0) {
            g.A = aev; g.lda = L; g.a_gather = w.perm; g.batch = 1;
            g.ncol_max = (wmax * M + BN - 1) / BN;
        } else {
            g.A = w.act[l - 1]; g.lda = w.ld[l - 1]; g.batch = M;
            g.ncol_max = (wmax + BN - 1) / BN;
        }
        for (int s = 0; s < S; ++s) {
            const anihip_species_net &nn = d->net[s];
            GemmProblem &p = g.prob[s];
            p.B = nn.w[l]; p.bias = nn.bias[l];
            if (l == 0) {
                p.K = nn.dims[0]; p.N = nn.dims[1] * M; p.ldb = p.N;
            } else {
                p.K = nn.dims[l]; p.N = nn.dims[l + 1]; p.ldb = p.N;
                p.a_boff = nn.dims[l]; p.c_boff = nn.dims[l + 1];
                p.b_stride = (int64_t)p.K * p.N; p.bias_stride = p.N;
            }
        }
        launch_gemm<EPI_BIAS_CELU>(stream, g, false);
    }
    return 0;
}

// Does the training pass of this descriptor run through the fused network kernel?  A split-fp16 CELU pack of the shape the
// fused kernel covers (three hidden layers <= 256 wide, <= 32 AEV slabs, every fragment-ordered plane present).
static bool train_fused(const anihip_mlp_desc *d)
{
    if (d->precision != ANIHIP_MLP_F16X3 || d->activation != ANIHIP_ACT_CELU || d->net[0].n_layers != 4) return false;
    anihip_mlp_desc c = *d;
    c.flags = 0;
    return fb_plan(&c, 1 << 16, true).fused;
}

// First half of a training step on the fast path: species buckets, tile table, ONE k_mlp_fused<.., TRAIN> launch (split-fp16
// MFMA like inference: forward AND the backward down to d e / d z0 for a unit upstream gradient -- the backward does not
// depend on the loss, only its per-atom scale does, and that enters the weight-gradient GEMMs as a row scale), per-atom
// energies.  Left in the workspace: act[0..2] (hidden activations) and dlt[0..2] (d e / d pre-activation), fp32 rows in
// sorted order.
static int train_forward_fused(hipStream_t stream, const anihip_mlp_desc *d, int64_t n_atoms, int64_t lo, int64_t hi,
                               const int32_t *species, const float *aev, MlpWorkspace &w, float **dlt, float *atomic_e)
{
    const int S = d->num_species, M = d->n_members, L = d->aev_len;
    const int64_t n = hi - lo;
    const int kp_rad = d->aev_radial_len;
    const int K0p = kp_rad > 0 ? 32 * ((kp_rad + 31) / 32 + (L - kp_rad) / 32) : ((L + 31) / 32) * 32;
    const int n_slabs = K0p / 32;
    const uint32_t all_slabs = n_slabs >= 32 ? 0xFFFFFFFFu : ((1u << n_slabs) - 1u);
    constexpr int rows = 64;
    zero_words_async(stream, w.ctl, sizeof(int) * (CTL_WORDS + AMAX_WORDS));
    const unsigned cblk = (unsigned)((n + 4 * SP_CHUNK - 1) / (4 * SP_CHUNK));
    {
        int *chunk_cnt = reinterpret_cast<int *>(w.member_part);
        const int n_chunks = (int)((n + SP_CHUNK - 1) / SP_CHUNK);
        hipLaunchKernelGGL(k_sp_count, dim3(cblk), dim3(256), 0, stream, lo, hi, species, S, chunk_cnt);
        hipLaunchKernelGGL(k_sp_offsets, dim3(1), dim3(256), 0, stream, S, n_chunks, chunk_cnt, w.ctl);
        hipLaunchKernelGGL(k_sp_scatter, dim3(cblk), dim3(256), 0, stream, lo, hi, species, S, w.ctl, chunk_cnt, w.perm,
                           atomic_e, (float *)nullptr, L, (float *)nullptr, M, n_atoms);
    }
    const int64_t tiles = (n + rows - 1) / rows + S;
    // (species that do not occur in the system flag nothing: valid when the call covers the whole system)
    const int ani_species = (lo == 0 && hi == n_atoms && kp_rad == 16 * S && kp_rad > 0) ? S : 0;
    hipLaunchKernelGGL(k_tile_table, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, stream, w.ctl, S, w.perm,
                       (const uint32_t *)nullptr, all_slabs, (int)tiles, rows, w.tile_tab, w.tile_rows, ani_species);
    FusedArgs f{};
    size_t lds = 0;
    for (int s = 0; s < S; ++s) {
        const anihip_species_net &nn = d->net[s];
        FusedSpecies &fs = f.sp[s];
        fs.H1 = nn.dims[1]; fs.H2 = nn.dims[2]; fs.H3 = nn.dims[3];
        fs.w0 = (const _Float16 *)nn.whf[0];
        fs.w1 = (const _Float16 *)nn.whf[1]; fs.w2 = (const _Float16 *)nn.whf[2];
        fs.w2t = (const _Float16 *)nn.wthf[2]; fs.w1t = (const _Float16 *)nn.wthf[1];
        fs.w0t = (const _Float16 *)nn.wthf[0];
        fs.is0 = 1.0f / nn.wh_scale[0]; fs.is1 = 1.0f / nn.wh_scale[1]; fs.is2 = 1.0f / nn.wh_scale[2];
        fs.b0 = nn.bias[0]; fs.b1 = nn.bias[1]; fs.b2 = nn.bias[2]; fs.w3 = nn.w[3]; fs.b3 = nn.bias[3];
        fs.bounds = nn.fused_bounds;
        const size_t xu = fs.H1 > fs.H3 ? fs.H1 : fs.H3;
        size_t halves = 2 * (size_t)rows * (fs.H2 + FR_XPAD) + 2 * (size_t)rows * (xu + FR_XPAD);
        const size_t slab = 2 * (size_t)rows * FR_SLAB_LD;
        if (halves < 3 * FR_GROUP * slab) halves = 3 * FR_GROUP * slab;
        halves += FusedCfg<2, 1>::FIXED_HALVES;
        lds = lds > halves * 2 ? lds : halves * 2;
    }
    f.ctl = w.ctl; f.amax = w.amax; f.aev = aev; f.L = L; f.kp_rad = kp_rad; f.n_slabs = n_slabs;
    f.slab_mask = nullptr;
    f.d0 = dlt[0]; f.ld0 = w.ld[0]; f.d0_tm = 0; f.perm = w.perm;
    f.tile_tab = w.tile_tab; f.tile_rows = w.tile_rows;
    f.member_part = w.member_part; f.S = S; f.M = M; f.alpha = d->celu_alpha; f.inv_alpha = 1.0f / d->celu_alpha;
    f.want_grad = 1; f.owner = 0; f.l0b = 0; f.grad_aev = nullptr;
    f.tiles_total = (int)tiles;
    for (int l = 0; l < 3; ++l) { f.tr_act[l] = w.act[l]; f.tr_ld[l] = w.ld[l]; f.tr_dlt[l] = dlt[l]; }
    const void *kfn = (const void *)k_mlp_fused<2, 1, 0, false, true>;
    ANIHIP_CHECK_HIP(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int dev = 0, n_cus = 0;
    ANIHIP_CHECK_HIP(hipGetDevice(&dev));
    ANIHIP_CHECK_HIP(hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (n_cus <= 0) n_cus = 256;
    const int64_t items = tiles * M;
    const int64_t grid = items < n_cus ? items : n_cus;
    hipLaunchKernelGGL((k_mlp_fused<2, 1, 0, false, true>), dim3((unsigned)grid), dim3(512), lds, stream, f);
    FinishArgs fin{};
    fin.ctl = w.ctl; fin.perm = w.perm; fin.member_part = w.member_part; fin.atomic_e = atomic_e;
    fin.member_e = nullptr; fin.n_atoms = n_atoms; fin.S = S; fin.M = M; fin.first_block = 0;
    int64_t fb = (n + 255) / 256;
    if (fb > 2048) fb = 2048;
    hipLaunchKernelGGL(k_fused_finish, dim3((unsigned)fb), dim3(256), 0, stream, fin);
    return 0;
}

static void train_head(hipStream_t stream, const anihip_mlp_desc *d, int64_t n_atoms, int64_t n, MlpWorkspace &w,
                       float *seed, const float *g_atom, float *atomic_e)
{
    const int S = d->num_species, nl = d->net[0].n_layers, nh = nl - 1;
    HeadArgs h{};
    for (int s = 0; s < S; ++s) {
        h.w[s] = d->net[s].w[nl - 1];
        h.bias[s] = d->net[s].bias[nl - 1];
        h.Hp[s] = d->net[s].dims[nl - 1];
    }
    h.ctl = w.ctl; h.perm = w.perm; h.act = w.act[nh - 1]; h.ld = w.ld[nh - 1];
    h.seed = seed; h.g_atom = g_atom;
    h.atomic_e = atomic_e; h.member_e = nullptr; h.n_atoms = n_atoms; h.S = S; h.M = d->n_members;
    h.inv_alpha = 1.0f / d->celu_alpha; h.want_grad = seed ? 1 : 0; h.amax = nullptr; h.amax_out = 0;
    h.act_kind = d->activation; h.zpre = w.zp[nh - 1];
    int64_t blocks = (n + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(k_head, dim3((unsigned)blocks), dim3(256), 0, stream, h);
}

extern "C" int anihip_mlp_train_forward(void *stream_, const anihip_mlp_desc *d, int64_t n_atoms, int64_t lo,
                                        int64_t hi, const int32_t *species, const float *aev, void *workspace,
                                        size_t workspace_bytes, float *atomic_e)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = check_desc(d)) return rc;
    ANIHIP_REQUIRE(species && aev && workspace && atomic_e, "null pointer argument");
    ANIHIP_REQUIRE(0 <= lo && lo <= hi && hi <= n_atoms, "central range outside 0..n_atoms");
    const int64_t n = hi - lo;
    if (n == 0) return 0;
    ANIHIP_REQUIRE(workspace_bytes >= mlp_train_carve(d, n, nullptr, nullptr, nullptr), "workspace too small");
    MlpWorkspace w;
    float *dlt[ANIHIP_MAX_LAYERS];
    mlp_train_carve(d, n, (char *)workspace, &w, dlt);
    if (train_fused(d)) {
        if (int rc = train_forward_fused(stream, d, n_atoms, lo, hi, species, aev, w, dlt, atomic_e)) return rc;
        ANIHIP_CHECK_HIP(hipGetLastError());
        return 0;
    }
    ANIHIP_REQUIRE(d->precision == ANIHIP_MLP_FP32 || d->activation == ANIHIP_ACT_CELU,
                   "training passes of GELU networks need an ANIHIP_MLP_FP32 descriptor");
    if (int rc = train_forward(stream, d, n_atoms, lo, hi, species, aev, w, atomic_e, nullptr)) return rc;
    train_head(stream, d, n_atoms, n, w, nullptr, nullptr, atomic_e);
    ANIHIP_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int anihip_mlp_repack(void *stream_, const anihip_mlp_desc *d, const void *const *src,
                                 const int32_t *out_in, int32_t *status, int32_t flags)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = check_desc(d)) return rc;
    ANIHIP_REQUIRE(src && out_in, "null pointer argument");
    if (d->precision == ANIHIP_MLP_F16X3) return repack_f16(stream, d, src, out_in, status, flags);
    RepackArgs a{};
    a.src = (const float *const *)src;
    a.S = d->num_species; a.M = d->n_members; a.nl = d->net[0].n_layers;
    a.k0p = ((d->aev_len + 31) / 32) * 32;
    int64_t biggest = 0;
    for (int s = 0; s < a.S; ++s) {
        const anihip_species_net &nn = d->net[s];
        for (int l = 0; l <= a.nl; ++l) a.dims[s][l] = nn.dims[l];
        for (int l = 0; l < a.nl; ++l) {
            a.out[s][l] = out_in[(s * a.nl + l) * 2 + 0];
            a.in[s][l] = out_in[(s * a.nl + l) * 2 + 1];
            ANIHIP_REQUIRE(a.out[s][l] >= 1 && a.out[s][l] <= nn.dims[l + 1] && a.in[s][l] >= 1 &&
                               a.in[s][l] <= nn.dims[l],
                           "species %d layer %d: source shape outside the packed shape", s, l);
            a.w[s][l] = const_cast<float *>(nn.w[l]);
            a.wt[s][l] = const_cast<float *>(nn.wt[l]);
            a.bias[s][l] = const_cast<float *>(nn.bias[l]);
            const int64_t e = (int64_t)a.out[s][l] * a.in[s][l];
            biggest = biggest > e ? biggest : e;
        }
    }
    unsigned bx = (unsigned)((biggest + 256 * 8 - 1) / (256 * 8));
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(k_repack, dim3(bx, (unsigned)(a.M * a.S * a.nl)), dim3(256), 0, stream, a);
    ANIHIP_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int anihip_mlp_weight_grads(void *stream_, const anihip_mlp_desc *d, int64_t n_atoms, int64_t lo,
                                       int64_t hi, const int32_t *species, const float *aev,
                                       const float *grad_atomic_e, void *workspace, size_t workspace_bytes,
                                       const anihip_species_grads *grads, float *atomic_e, float *grad_aev,
                                       int32_t forward_done)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = check_desc(d)) return rc;
    ANIHIP_REQUIRE(species && aev && grad_atomic_e && workspace && grads && atomic_e, "null pointer argument");
    ANIHIP_REQUIRE(0 <= lo && lo <= hi && hi <= n_atoms, "central range outside 0..n_atoms");
    const int S = d->num_species, M = d->n_members, nl = d->net[0].n_layers, nh = nl - 1, L = d->aev_len;
    for (int s = 0; s < S; ++s)
        for (int l = 0; l < nl; ++l)
            ANIHIP_REQUIRE(grads[s].gw[l] && grads[s].gbias[l], "species %d layer %d: null gradient pointer", s, l);
    // (a caller that wants d Loss / d aev as well gets the exact-fp32 backward; it runs on the activations of either forward)
    const bool fast = train_fused(d) && !grad_aev;
    const int64_t mstride = grads[0].member_stride;
    const bool accumulate = grads[0].accumulate != 0;
    for (int s = 0; s < S; ++s)
        ANIHIP_REQUIRE(grads[s].member_stride == mstride && (grads[s].accumulate != 0) == accumulate,
                       "member_stride / accumulate must be the same for every species");
    ANIHIP_REQUIRE(mstride >= 0 && (mstride == 0 || fast),
                   "per-member gradient arrays (member_stride) are written by the split-fp16 training pass only");
    ANIHIP_REQUIRE(mstride == 0 || accumulate, "per-member gradient arrays are accumulated into (the caller zeroes them)");
    // gradients are overwritten (zero, then accumulate with atomics) unless the caller accumulates
    for (int s = 0; s < S && !accumulate; ++s) {
        const anihip_species_net &nn = d->net[s];
        for (int l = 0; l < nl; ++l) {
            const size_t nw = (size_t)M * nn.dims[l] * nn.dims[l + 1], nb = (size_t)M * nn.dims[l + 1];
            zero_words_async(stream, grads[s].gw[l], sizeof(float) * nw);
            zero_words_async(stream, grads[s].gbias[l], sizeof(float) * nb);
        }
    }
    const int64_t n = hi - lo;
    if (n == 0) return 0;
    ANIHIP_REQUIRE(workspace_bytes >= mlp_train_carve(d, n, nullptr, nullptr, nullptr), "workspace too small");
    MlpWorkspace w;
    float *dlt[ANIHIP_MAX_LAYERS];
    mlp_train_carve(d, n, (char *)workspace, &w, dlt);
    const float alpha = d->celu_alpha, inv_alpha = 1.0f / d->celu_alpha;

    if (fast) {
        // Fast path: forward + unit-gradient backward by the fused kernel (anihip_mlp_train_forward, or here), then per layer
        // ONE column reduction (bias gradients) and ONE bf16 x 3 weight-gradient launch, both scaling the rows of
        // d e / d z by the upstream d Loss / d atomic_e of their atoms.  No GEMM runs in this call.
        if (!forward_done)
            if (int rc = train_forward_fused(stream, d, n_atoms, lo, hi, species, aev, w, dlt, atomic_e)) return rc;
        const int cr_chunks = (int)((n + CR_ROWS - 1) / CR_ROWS) + S;
        for (int l = nl - 1; l >= 0; --l) {
            const bool output_layer = l == nl - 1;
            ColReduceArgs c{};
            int mx = 0;
            for (int s = 0; s < S; ++s) {
                const int width = d->net[s].dims[output_layer ? nl - 1 : l + 1];
                c.X[s] = output_layer ? w.act[nh - 1] : dlt[l];
                c.ncols[s] = M * width;
                c.n_per[s] = width;
                c.out[s] = output_layer ? grads[s].gw[nl - 1] : grads[s].gbias[l];
                c.extra[s] = output_layer ? grads[s].gbias[nl - 1] : nullptr;
                mx = mx > c.ncols[s] ? mx : c.ncols[s];
            }
            c.ldx = output_layer ? w.ld[nh - 1] : w.ld[l];
            c.ctl = w.ctl; c.perm = w.perm; c.S = S; c.M = M;
            c.g_atom = grad_atomic_e;
            c.inv_m = output_layer ? 1.0f / (float)M : 1.0f;   // (d e / d z of the hidden layers carries the 1 / M already)
            c.ct_max = (mx + 255) / 256;
            c.out_mstride = mstride;
            // (the bias gradients of the hidden layers are column sums of the rows the weight-gradient kernel stages anyway)
            if (output_layer) {
                hipLaunchKernelGGL(k_col_reduce, dim3((unsigned)(cr_chunks * c.ct_max)), dim3(256), 0, stream, c);
                continue;
            }
            WgradB3Args a{};
            a.ctl = w.ctl; a.perm = w.perm; a.S = S; a.g_atom = grad_atomic_e;
            a.x_gather = l == 0 ? w.perm : nullptr;
            a.batch = l == 0 ? 1 : M;
            int kmax = 0, nmax = 0;
            for (int s = 0; s < S; ++s) {
                const anihip_species_net &nn = d->net[s];
                WgradB3Problem &p = a.prob[s];
                p.D = dlt[l]; p.ldd = w.ld[l]; p.dW = grads[s].gw[l];
                p.n_per = nn.dims[l + 1];
                p.ldw = nn.dims[l];
                p.w_mstride = mstride > 0 ? mstride : (int64_t)nn.dims[l] * nn.dims[l + 1];
                if (l == 0) {
                    p.X = aev; p.ldx = L; p.x_boff = 0; p.k_valid = L; p.d_boff = 0; p.N = nn.dims[1] * M;
                } else {
                    p.X = w.act[l - 1]; p.ldx = w.ld[l - 1]; p.x_boff = nn.dims[l]; p.k_valid = nn.dims[l];
                    p.d_boff = nn.dims[l + 1]; p.N = nn.dims[l + 1];
                }
                kmax = kmax > p.k_valid ? kmax : p.k_valid;
                nmax = nmax > p.N ? nmax : p.N;
            }
            a.ki_max = (kmax + 127) / 128;
            a.nj_max = (nmax + 127) / 128;
            // layer 0 of a whole system: only the AEV slabs of species (pairs) that occur in it (train.h)
            a.ani_species = S;
            a.x_slab_rad = (l == 0 && lo == 0 && hi == n_atoms && d->aev_radial_len == 16 * S && S * (S + 1) / 2 + (16 * S + 31) / 32 <= 32)
                               ? d->aev_radial_len : 0;
            for (int s = 0; s < S; ++s) {
                a.gbias[s] = grads[s].gbias[l];
                a.b_mstride[s] = mstride > 0 ? mstride : (int64_t)d->net[s].dims[l + 1];   // (packed: [M][n_per])
            }
            // atoms per workgroup: enough workgroups for several rounds over the chip's 512 slots (the tail of the last
            // round is what an uneven split costs), few enough that the float atomics of the partial tiles stay cheap
            {
                const int64_t tiles = (int64_t)a.batch * a.ki_max * a.nj_max;
                // (measured on the config-5 batch, whole graphed step: 3072 workgroups for every layer 3.24 ms, 1536 3.20,
                // 6144 3.30, 12288 3.40 -- more chunks cost more atomics than their finer tail saves; 1536 for the wide layer-0
                // launch with 640 for the small hidden layers 3.37: those are chains of dependent stages and want MANY short
                // workgroups)
                const int64_t target_wgs = tiles >= 64 ? 1536 : 3072;
                int64_t rows = (n * tiles / target_wgs + 255) / 256 * 256;
                a.rows_per_chunk = (int)(rows < 512 ? 512 : (rows > 4096 ? 4096 : rows));
            }
            launch_wgrad_b3(stream, a, n);
        }
        ANIHIP_CHECK_HIP(hipGetLastError());
        return 0;
    }

    // 1.-2. bucket by species, forward in exact fp32 with the activations kept (or reuse anihip_mlp_train_forward's)
    if (!forward_done) {
        if (int rc = train_forward(stream, d, n_atoms, lo, hi, species, aev, w, atomic_e, grad_aev)) return rc;
    } else if (grad_aev) {
        hipLaunchKernelGGL(k_zero_padding, dim3(zero_pad_blocks(n)), dim3(256), 0, stream, lo, hi, species,
                           atomic_e, grad_aev, L, (float *)nullptr, M, n_atoms);
    }
    const int nrow_ub = (int)((n + BM - 1) / BM) + S;
    auto width_max = [&](int l) {
        int mx = 0;
        for (int s = 0; s < S; ++s) mx = mx > d->net[s].dims[l] ? mx : d->net[s].dims[l];
        return mx;
    };
    auto gemm_base = [&]() {
        GemmArgs g{};
        g.ctl = w.ctl; g.S = S; g.alpha = alpha; g.inv_alpha = inv_alpha; g.nrow_tiles_ub = nrow_ub;
        g.amax_in = g.amax_out = -1;
        g.act = d->activation;
        return g;
    };

    // 3. output layer: energies, seed of the backward pass (scaled by the upstream gradient), d w_out, d b_out
    train_head(stream, d, n_atoms, n, w, dlt[nh - 1], grad_atomic_e, atomic_e);
    const int cr_chunks = (int)((n + CR_ROWS - 1) / CR_ROWS) + S;
    auto col_reduce = [&](int l, const float *X, int64_t ldx, bool output_layer) {
        ColReduceArgs c{};
        int mx = 0;
        for (int s = 0; s < S; ++s) {
            c.X[s] = X;
            c.ncols[s] = M * d->net[s].dims[output_layer ? nl - 1 : l + 1];
            c.out[s] = output_layer ? grads[s].gw[nl - 1] : grads[s].gbias[l];
            c.extra[s] = output_layer ? grads[s].gbias[nl - 1] : nullptr;
            mx = mx > c.ncols[s] ? mx : c.ncols[s];
        }
        c.ldx = ldx; c.ctl = w.ctl; c.perm = w.perm; c.S = S; c.M = M;
        c.g_atom = output_layer ? grad_atomic_e : nullptr;
        c.inv_m = output_layer ? 1.0f / (float)M : 1.0f;
        c.ct_max = (mx + 255) / 256;
        hipLaunchKernelGGL(k_col_reduce, dim3((unsigned)(cr_chunks * c.ct_max)), dim3(256), 0, stream, c);
    };
    col_reduce(nl - 1, w.act[nh - 1], w.ld[nh - 1], true);

    // 4. backward through the hidden layers (gradients in their own buffers), weight and bias gradients
    const int wg_chunks = (int)((n + WG_ROWS - 1) / WG_ROWS) + S;
    for (int l = nh - 1; l >= 0; --l) {
        col_reduce(l, dlt[l], w.ld[l], false);
        WgradArgs a{};
        a.ctl = w.ctl; a.S = S;
        a.x_gather = l == 0 ? w.perm : nullptr;
        a.batch = l == 0 ? 1 : M;
        int kmax = 0, nmax = 0;
        for (int s = 0; s < S; ++s) {
            const anihip_species_net &nn = d->net[s];
            WgradProblem &p = a.prob[s];
            p.D = dlt[l]; p.ldd = w.ld[l]; p.dW = grads[s].gw[l];
            if (l == 0) {
                p.X = aev; p.ldx = L; p.x_boff = 0; p.K = L; p.k_valid = L;
                p.d_boff = 0; p.N = nn.dims[1] * M; p.ldw = L; p.w_bstride = 0;
            } else {
                p.X = w.act[l - 1]; p.ldx = w.ld[l - 1]; p.x_boff = nn.dims[l]; p.K = nn.dims[l]; p.k_valid = p.K;
                p.d_boff = nn.dims[l + 1]; p.N = nn.dims[l + 1]; p.ldw = p.K; p.w_bstride = (int64_t)p.K * p.N;
            }
            kmax = kmax > p.K ? kmax : p.K;
            nmax = nmax > p.N ? nmax : p.N;
        }
        a.ki_max = (kmax + 63) / 64;
        a.nj_max = (nmax + 255) / 256;
        const int64_t total = (int64_t)wg_chunks * a.batch * a.ki_max * a.nj_max;
        hipLaunchKernelGGL(k_wgrad, dim3((unsigned)total), dim3(256), 0, stream, a);

        if (l == 0 && !grad_aev) break;
        GemmArgs g = gemm_base();
        g.A = dlt[l]; g.lda = w.ld[l];
        if (l == 0) {
            g.batch = 1; g.C = grad_aev; g.ldc = L; g.c_scatter = w.perm; g.n_store = L;
            g.ncol_max = (((L + 31) / 32) * 32 + BN - 1) / BN;
        } else {
            g.batch = M; g.C = dlt[l - 1]; g.ldc = w.ld[l - 1]; g.Y = w.act[l - 1]; g.ldy = w.ld[l - 1];
            g.X = w.zp[l - 1];
            g.ncol_max = (width_max(l) + BN - 1) / BN;
        }
        for (int s = 0; s < S; ++s) {
            const anihip_species_net &nn = d->net[s];
            GemmProblem &p = g.prob[s];
            p.B = nn.wt[l];
            if (l == 0) {
                p.K = nn.dims[1] * M; p.N = ((L + 31) / 32) * 32; p.ldb = p.N;
            } else {
                p.K = nn.dims[l + 1]; p.N = nn.dims[l]; p.ldb = p.N;
                p.a_boff = nn.dims[l + 1]; p.c_boff = nn.dims[l];
                p.b_stride = (int64_t)p.K * p.N;
            }
        }
        if (l == 0)
            launch_gemm<EPI_SCATTER>(stream, g, false);
        else
            launch_gemm<EPI_DCELU>(stream, g, false);
    }
    ANIHIP_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" size_t anihip_mlp_tangent_workspace_bytes(const anihip_mlp_desc *d, int64_t n_central)
{
    if (!d || n_central < 0) return 0;
    return mlp_tangent_carve(d, n_central, nullptr, nullptr, nullptr);
}

extern "C" int anihip_mlp_tangent_weight_grads(void *stream_, const anihip_mlp_desc *d, int64_t n_atoms, int64_t lo,
                                               int64_t hi, const int32_t *species, const float *aev,
                                               const float *tangent, void *workspace, size_t workspace_bytes,
                                               const anihip_species_grads *grads, float *datomic_e)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = check_desc(d)) return rc;
    ANIHIP_REQUIRE(species && aev && tangent && workspace && grads && datomic_e, "null pointer argument");
    for (int s = 0; s < d->num_species; ++s)
        ANIHIP_REQUIRE(grads[s].member_stride == 0 && grads[s].accumulate == 0,
                       "the second-order pass overwrites packed [M][...] gradient arrays (member_stride = accumulate = 0)");
    ANIHIP_REQUIRE(0 <= lo && lo <= hi && hi <= n_atoms, "central range outside 0..n_atoms");
    const int S = d->num_species, M = d->n_members, nl = d->net[0].n_layers, nh = nl - 1, L = d->aev_len;
    for (int s = 0; s < S; ++s)
        for (int l = 0; l < nl; ++l)
            ANIHIP_REQUIRE(grads[s].gw[l] && grads[s].gbias[l], "species %d layer %d: null gradient pointer", s, l);
    for (int s = 0; s < S; ++s) {
        const anihip_species_net &nn = d->net[s];
        for (int l = 0; l < nl; ++l) {
            zero_words_async(stream, grads[s].gw[l], sizeof(float) * (size_t)M * nn.dims[l] * nn.dims[l + 1]);
            zero_words_async(stream, grads[s].gbias[l], sizeof(float) * (size_t)M * nn.dims[l + 1]);
        }
    }
    const int64_t n = hi - lo;
    if (n == 0) return 0;
    ANIHIP_REQUIRE(workspace_bytes >= mlp_tangent_carve(d, n, nullptr, nullptr, nullptr), "workspace too small");
    MlpWorkspace w;
    float *buf[4][ANIHIP_MAX_LAYERS];
    mlp_tangent_carve(d, n, (char *)workspace, &w, buf);
    float **zd = buf[0], **ad = buf[1], **P = buf[2], **Q = buf[3];
    const float alpha = d->celu_alpha, inv_alpha = 1.0f / d->celu_alpha;

    // 1. bucket by species, activations a_l (datomic_e doubles as the array whose padding entries get zeroed)
    if (int rc = train_forward(stream, d, n_atoms, lo, hi, species, aev, w, datomic_e, nullptr)) return rc;
    const int nrow_ub = (int)((n + BM - 1) / BM) + S;
    auto width_max = [&](int l) {
        int mx = 0;
        for (int s = 0; s < S; ++s) mx = mx > d->net[s].dims[l] ? mx : d->net[s].dims[l];
        return mx;
    };
    auto gemm_base = [&]() {
        GemmArgs g{};
        g.ctl = w.ctl; g.S = S; g.alpha = alpha; g.inv_alpha = inv_alpha; g.nrow_tiles_ub = nrow_ub;
        g.amax_in = g.amax_out = -1;
        g.act = d->activation;
        return g;
    };
    // 2. tangents: zdot_l = W_l adot_{l-1}, adot_l = c'(z_l) zdot_l   (adot_0 = tangent rows)
    for (int l = 0; l < nh; ++l) {
        GemmArgs g = gemm_base();
        g.C = zd[l]; g.C2 = ad[l]; g.ldc = w.ld[l]; g.Y = w.act[l]; g.ldy = w.ld[l]; g.X = w.zp[l];
        if (l == 0) {
            g.A = tangent; g.lda = L; g.a_gather = w.perm; g.batch = 1;
            g.ncol_max = (width_max(1) * M + BN - 1) / BN;
        } else {
            g.A = ad[l - 1]; g.lda = w.ld[l - 1]; g.batch = M;
            g.ncol_max = (width_max(l + 1) + BN - 1) / BN;
        }
        for (int s = 0; s < S; ++s) {
            const anihip_species_net &nn = d->net[s];
            GemmProblem &p = g.prob[s];
            p.B = nn.w[l];
            if (l == 0) {
                p.K = nn.dims[0]; p.N = nn.dims[1] * M; p.ldb = p.N;
            } else {
                p.K = nn.dims[l]; p.N = nn.dims[l + 1]; p.ldb = p.N;
                p.a_boff = nn.dims[l]; p.c_boff = nn.dims[l + 1];
                p.b_stride = (int64_t)p.K * p.N;
            }
        }
        launch_gemm_fp32<EPI_TANGENT>(stream, g);
    }
    // 3. output layer: adjoint seeds p, q of the last hidden layer, d atomic_e, d w_out (d b_out = 0)
    {
        HeadTangentArgs h{};
        for (int s = 0; s < S; ++s) { h.w[s] = d->net[s].w[nl - 1]; h.Hp[s] = d->net[s].dims[nl - 1]; }
        h.ctl = w.ctl; h.perm = w.perm; h.act = w.act[nh - 1]; h.zd = zd[nh - 1]; h.ad = ad[nh - 1];
        h.P = P[nh - 1]; h.Q = Q[nh - 1]; h.ld = w.ld[nh - 1]; h.datomic_e = datomic_e; h.S = S; h.M = M;
        h.inv_alpha = inv_alpha; h.act_kind = d->activation; h.zpre = w.zp[nh - 1];
        int64_t blocks = (n + 3) / 4;
        if (blocks > 256 * 8) blocks = 256 * 8;
        hipLaunchKernelGGL(k_head_tangent, dim3((unsigned)blocks), dim3(256), 0, stream, h);
    }
    const int cr_chunks = (int)((n + CR_ROWS - 1) / CR_ROWS) + S;
    auto col_reduce = [&](const float *X, int64_t ldx, int l_out, bool weights, float scale) {
        ColReduceArgs c{};
        int mx = 0;
        for (int s = 0; s < S; ++s) {
            c.X[s] = X;
            c.ncols[s] = M * d->net[s].dims[l_out];
            c.out[s] = weights ? grads[s].gw[nl - 1] : grads[s].gbias[l_out - 1];
            c.extra[s] = nullptr;
            mx = mx > c.ncols[s] ? mx : c.ncols[s];
        }
        c.ldx = ldx; c.ctl = w.ctl; c.perm = w.perm; c.S = S; c.M = M; c.g_atom = nullptr; c.inv_m = scale;
        c.ct_max = (mx + 255) / 256;
        hipLaunchKernelGGL(k_col_reduce, dim3((unsigned)(cr_chunks * c.ct_max)), dim3(256), 0, stream, c);
    };
    col_reduce(ad[nh - 1], w.ld[nh - 1], nl - 1, true, 1.0f / (float)M);   // d S / d w_out = sum adot_last / M

    // 4. adjoints down the layers; d S / d W_l = p_l adot_{l-1}^T + q_l a_{l-1}^T, d S / d b_l = sum q_l
    const int wg_chunks = (int)((n + WG_ROWS - 1) / WG_ROWS) + S;
    for (int l = nh - 1; l >= 0; --l) {
        col_reduce(Q[l], w.ld[l], l + 1, false, 1.0f);
        for (int term = 0; term < 2; ++term) {
            WgradArgs a{};
            a.ctl = w.ctl; a.S = S;
            a.x_gather = l == 0 ? w.perm : nullptr;
            a.batch = l == 0 ? 1 : M;
            int kmax = 0, nmax = 0;
            for (int s = 0; s < S; ++s) {
                const anihip_species_net &nn = d->net[s];
                WgradProblem &p = a.prob[s];
                p.D = term == 0 ? P[l] : Q[l]; p.ldd = w.ld[l]; p.dW = grads[s].gw[l];
                if (l == 0) {
                    p.X = term == 0 ? tangent : aev; p.ldx = L; p.x_boff = 0; p.K = L; p.k_valid = L;
                    p.d_boff = 0; p.N = nn.dims[1] * M; p.ldw = L; p.w_bstride = 0;
                } else {
                    p.X = term == 0 ? ad[l - 1] : w.act[l - 1]; p.ldx = w.ld[l - 1]; p.x_boff = nn.dims[l];
                    p.K = nn.dims[l]; p.k_valid = p.K;
                    p.d_boff = nn.dims[l + 1]; p.N = nn.dims[l + 1]; p.ldw = p.K; p.w_bstride = (int64_t)p.K * p.N;
                }
                kmax = kmax > p.K ? kmax : p.K;
                nmax = nmax > p.N ? nmax : p.N;
            }
            a.ki_max = (kmax + 63) / 64;
            a.nj_max = (nmax + 255) / 256;
            const int64_t total = (int64_t)wg_chunks * a.batch * a.ki_max * a.nj_max;
            hipLaunchKernelGGL(k_wgrad, dim3((unsigned)total), dim3(256), 0, stream, a);
        }
        if (l == 0) break;
        for (int pass = 0; pass < 2; ++pass) {   // mu = W^T p -> (p, mu c'' zdot);  nu = W^T q -> q += nu c'
            GemmArgs g = gemm_base();
            g.A = pass == 0 ? P[l] : Q[l]; g.lda = w.ld[l]; g.batch = M;
            g.C = pass == 0 ? P[l - 1] : Q[l - 1]; g.C2 = Q[l - 1]; g.ldc = w.ld[l - 1];
            g.Y = w.act[l - 1]; g.Z = zd[l - 1]; g.ldy = w.ld[l - 1]; g.X = w.zp[l - 1];
            g.ncol_max = (width_max(l) + BN - 1) / BN;
            for (int s = 0; s < S; ++s) {
                const anihip_species_net &nn = d->net[s];
                GemmProblem &p = g.prob[s];
                p.B = nn.wt[l];
                p.K = nn.dims[l + 1]; p.N = nn.dims[l]; p.ldb = p.N;
                p.a_boff = nn.dims[l + 1]; p.c_boff = nn.dims[l];
                p.b_stride = (int64_t)p.K * p.N;
            }
            if (pass == 0)
                launch_gemm_fp32<EPI_ADJ_P>(stream, g);
            else
                launch_gemm_fp32<EPI_ADJ_Q>(stream, g);
        }
    }
    ANIHIP_CHECK_HIP(hipGetLastError());
    return 0;
}

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

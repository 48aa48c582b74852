// Per-species MLP ensemble forward + input-gradient backward on the gfx950 matrix cores.
//
// Replaces mnp::run (csrc/mnp.cpp:32-232) and BmmEnsemble (nn/_infer.py:61-216).  Common to all paths:
// the atoms of the shard are bucketed by species on the device (ballot ranks, no host sync, no
// nonzero()/index_select like nn/_containers.py:406-416).  Then, by network shape and precision:
//
//   A. split-fp16 ("f16x3", default), three hidden layers of width <= 256 (ANI-1x / ANI-2x):
//        >= 16384 atoms: k_tile_table -> k_mlp_fused<RB,NB> -> k_fused_finish -> k_gemm_l0b + k_gemm_h2<EPI_SCATTER>
//        >= 24000 atoms: k_tile_table (-> k_tile_order: tiles by falling cost, drawn from a queue) -> k_mlp_fused<2,1,ACT,L0B = true>
//                        (layer-0 backward inside: phase 5; ANIHIP_MLP_FLAG_SHAPED: one launch per species with compile-time widths,
//                        queued on two streams from four rounds of tiles on) -> k_fused_finish
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
// The bucketing, tile-table, finish and energy-sum kernels live in csrc/mlp_prep.hip (mlp_prep.h), the fused network kernel in
// csrc/mlp_fused.hip (mlp_fused.h); this file keeps the host entry points and the layer-by-layer GEMM / training kernels.
#include <stdlib.h>

#include "anihip_common.h"
#include "train.h"
#include "mlp_fused.h"
#include "mlp_prep.h"

#include <type_traits>
#include <vector>
#include <cstdio>
#include <mutex>

namespace anihip {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const v4f gf4;  // global-memory float4 (forces global_load_dwordx4)

constexpr int BN = 128, BK = 16;   // (BM = 128 rows: mlp_prep.h)
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

// (FinishArgs, the bucketing / tile-table / finish kernels and their launchers: mlp_prep.h, mlp_prep.hip)

// (control block layout of the workspace: CTL_* in train.h)
// running |max| of every intermediate tensor (per stage, per species), spread over slots to keep the
// atomics off a single address; lives right behind the control block

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

__device__ __forceinline__ int64_t tm_species_base(const int *ctl, const int *H, int M, int s)
{
    int64_t base = 0;
    for (int t = 0; t < s; ++t) base += (int64_t)((ctl[8 * 0 + t] + 63) >> 6) * 64 * M * H[t];   // ctl[CTL_CNT + t]
    return base;
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

struct MlpWorkspace {
    float *zp[ANIHIP_MAX_LAYERS];   // training passes of GELU networks: pre-activations of the hidden layers (else NULL)
    int *ctl;
    unsigned *amax;
    float *member_part;
    int *perm;
    int4 *tile_tab;
    int *tile_rows;
    int4 *tile_tab2;    // the table sorted by falling tile cost (k_tile_order; up to FUSED_TILE_QUEUE_MAX 64-row tiles)
    int *tile_rows2;
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
    const size_t tiles64 = (size_t)((n + 63) / 64) + ANIHIP_MAX_SPECIES;
    const size_t qtiles = tiles64 <= (size_t)FUSED_TILE_QUEUE_MAX ? tiles64 : 0;
    int4 *ttab2 = (int4 *)take(sizeof(int4) * qtiles);
    int *trows2 = (int *)take(sizeof(int) * 64 * qtiles);
    if (w) {
        w->ctl = ctl; w->amax = (unsigned *)(ctl + CTL_WORDS); w->perm = perm; w->member_part = mpart;
        w->tile_tab = ttab; w->tile_rows = trows; w->tile_tab2 = qtiles ? ttab2 : nullptr; w->tile_rows2 = qtiles ? trows2 : nullptr;
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

#ifndef ANIHIP_WGRAD_F16
#define ANIHIP_WGRAD_F16 1   // 0: the weight gradients of the fast training path on six bf16 products (development A/B)
#endif

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

#ifndef ANIHIP_SHAPED_OVERLAP
#define ANIHIP_SHAPED_OVERLAP 1   // 0: the per-species launches of the fused kernel one after the other, static tile order (development A/B)
#endif
// second stream + fork / join events of the per-species launches, one set per device, created on first use (never destroyed:
// they live as long as the process; NULL on failure -- the launches then stay on the caller's stream).  A caller holds the
// set's mutex from its fork to its join: two host threads of one process share the stream and the events, and an event
// re-recorded by the other thread between a record and its wait would order the second stream behind the wrong work.
struct OverlapSet {
    hipStream_t st;
    hipEvent_t fork, join;
    int state;
    std::mutex mu;
};
static OverlapSet *overlap_resources()
{
    static OverlapSet res[64];
    static std::mutex create;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(create);
    OverlapSet &r = res[dev];
    if (r.state == 0) {
        r.state = -1;
        if (hipStreamCreateWithFlags(&r.st, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&r.fork, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&r.join, hipEventDisableTiming) == hipSuccess)
            r.state = 1;
        else
            (void)hipGetLastError();
    }
    return r.state == 1 ? &r : nullptr;
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
        if (int rc = launch_small_prep(stream, lo, hi, species, S, w.ctl, w.perm, tab_mask, all_slabs, (int)fused_tiles, fused_rows,
                                       fused ? w.tile_tab : (int4 *)nullptr, w.tile_rows, atomic_e, grad_aev, L, member_e, M,
                                       n_atoms))
            return rc;
    } else {
        // (scratch of the counting sort: the per-member energies buffer is written only later)
        launch_bucketing(stream, lo, hi, species, S, w.ctl, CTL_WORDS + AMAX_WORDS, reinterpret_cast<int *>(w.member_part), w.perm,
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
        const int variant = fused_l0b ? (bwd2 ? FUSED_CELU_L0B_B2 : FUSED_CELU_L0B) : (gelu ? FUSED_GELU : FUSED_CELU);
        const void *kfn = fused_kernel(variant);
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
            launch_tile_table(stream, w.ctl, S, w.perm, f.slab_mask, all_slabs, (int)tiles, rows, w.tile_tab, w.tile_rows);
#ifdef ANIHIP_DEV_TRACE   // development builds only (tools/fused_trace.py): per-item phase stamps, allocates and synchronises
        const char *trace_path = getenv("ANIHIP_FUSED_TRACE");
        const size_t trace_words = (size_t)32 * 8 * items;   // [item][wave][32]
        if (trace_path) {
            ANIHIP_CHECK_HIP(hipMalloc((void **)&f.trace, sizeof(unsigned long long) * trace_words));
            ANIHIP_CHECK_HIP(hipMemset(f.trace, 0, sizeof(unsigned long long) * trace_words));
        }
#endif
        f.only_species = -1;
        f.queue = nullptr;
        if (ANIHIP_TILE_QUEUE && f.owner == 1 && variant == FUSED_CELU_L0B && !(d->flags & ANIHIP_MLP_FLAG_SHAPED) && rows == 64 && w.tile_tab2 &&
            tiles <= FUSED_TILE_QUEUE_MAX && tiles > grid && !small_prep) {
            TileOrderArgs to{};
            to.tile_tab = w.tile_tab; to.tile_rows = w.tile_rows; to.tile_tab2 = w.tile_tab2; to.tile_rows2 = w.tile_rows2;
            to.tiles_total = (int)tiles;
            for (int s = 0; s < S; ++s) to.H1[s] = f.sp[s].H1;
            launch_tile_order(stream, to);
            f.tile_tab = w.tile_tab2; f.tile_rows = w.tile_rows2; f.queue = w.ctl + CTL_QUEUE;
        }
        if (variant == FUSED_CELU_L0B && (d->flags & ANIHIP_MLP_FLAG_SHAPED)) {
            // one launch per species, restricted to its tiles, with the network widths as compile-time constants where an
            // instantiation exists (every ANI-2x network and ANI-1x hydrogen); a species without atoms exits at once.
            // Every launch ends with a partly filled last round of the CUs (the hydrogen tiles of the 2.34 M-atom water box: 16
            // of 256 workgroups, eight items long).  The launches alternate between the caller's stream and a second one
            // (forked and joined with events; they touch disjoint atoms) and draw their tiles from a queue per species: the next
            // species' workgroups start on the CUs as they come free and take fewer tiles the later they start.  (Without the
            // queue the overlap buys nothing: a late workgroup then carries its static share to the end.)  Not inside a stream
            // capture (the step then stays a chain of kernel nodes).
            OverlapSet *ov = nullptr;
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            // (from four rounds of tiles on: below, the fork / join and the draws cost more than they balance -- water boxes of
            // 24 000 / 41 472 atoms 0.399 / 0.606 ms against 0.385 / 0.581 with plain launches; 81 000: 1.01 against 1.10)
            if (ANIHIP_SHAPED_OVERLAP && tiles >= 4 * grid && S <= CTL_WORDS - CTL_QUEUE) {
                if (hipStreamIsCapturing(stream, &cap) != hipSuccess)
                    (void)hipGetLastError();   // (not a reason to fail the call: plain launches)
                else if (cap == hipStreamCaptureStatusNone)
                    ov = overlap_resources();
            }
            std::unique_lock<std::mutex> ov_lock;
            hipStream_t aux = nullptr;
            if (ov) {
                ov_lock = std::unique_lock<std::mutex>(ov->mu);
                aux = ov->st;
                ANIHIP_CHECK_HIP(hipEventRecord(ov->fork, stream));
                ANIHIP_CHECK_HIP(hipStreamWaitEvent(aux, ov->fork, 0));
            }
            for (int s = 0; s < S; ++s) {
                const FusedSpecies &fs = f.sp[s];
                int v = FUSED_CELU_L0B;
                if (fs.H1 == 256 && fs.H2 == 192 && fs.H3 == 160) v = FUSED_CELU_L0B_256;
                else if (fs.H1 == 192 && fs.H2 == 160 && fs.H3 == 128) v = FUSED_CELU_L0B_192;
                else if (fs.H1 == 224 && fs.H2 == 192 && fs.H3 == 160) v = FUSED_CELU_L0B_224;
                else if (fs.H1 == 160 && fs.H2 == 128 && fs.H3 == 96) v = FUSED_CELU_L0B_160;
                if (v != FUSED_CELU_L0B) ANIHIP_CHECK_HIP(hipFuncSetAttribute(fused_kernel(v), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                f.only_species = s;
                f.queue = aux ? w.ctl + CTL_QUEUE + s : nullptr;   // (zeroed with the control block by the bucketing)
                launch_fused(v, (unsigned)grid, lds, (aux && (s & 1)) ? aux : stream, f);
            }
            if (ov) {
                ANIHIP_CHECK_HIP(hipEventRecord(ov->join, aux));
                ANIHIP_CHECK_HIP(hipStreamWaitEvent(stream, ov->join, 0));
            }
        } else {
            launch_fused(variant, (unsigned)grid, lds, stream, f);
        }
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
        if (!use_l0s) launch_fused_finish(stream, fin, n);   // (the 8-wave layer-0 backward of small inputs does this in extra workgroups)
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
    // (scratch of the counting sort: the per-member energies buffer is written only later)
    launch_bucketing(stream, lo, hi, species, S, w.ctl, CTL_WORDS + AMAX_WORDS, reinterpret_cast<int *>(w.member_part), w.perm,
                     atomic_e, grad_aev, L, (float *)nullptr, M, n_atoms);
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
    launch_bucketing(stream, lo, hi, species, S, w.ctl, CTL_WORDS + AMAX_WORDS, reinterpret_cast<int *>(w.member_part), w.perm,
                     atomic_e, (float *)nullptr, L, (float *)nullptr, M, n_atoms);
    const int64_t tiles = (n + rows - 1) / rows + S;
    // (species that do not occur in the system flag nothing: valid when the call covers the whole system)
    const int ani_species = (lo == 0 && hi == n_atoms && kp_rad == 16 * S && kp_rad > 0) ? S : 0;
    launch_tile_table(stream, w.ctl, S, w.perm, nullptr, all_slabs, (int)tiles, rows, w.tile_tab, w.tile_rows, ani_species);
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
    const void *kfn = fused_kernel(FUSED_TRAIN);
    ANIHIP_CHECK_HIP(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int dev = 0, n_cus = 0;
    ANIHIP_CHECK_HIP(hipGetDevice(&dev));
    ANIHIP_CHECK_HIP(hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (n_cus <= 0) n_cus = 256;
    const int64_t items = tiles * M;
    const int64_t grid = items < n_cus ? items : n_cus;
    launch_fused(FUSED_TRAIN, (unsigned)grid, lds, stream, f);
    FinishArgs fin{};
    fin.ctl = w.ctl; fin.perm = w.perm; fin.member_part = w.member_part; fin.atomic_e = atomic_e;
    fin.member_e = nullptr; fin.n_atoms = n_atoms; fin.S = S; fin.M = M; fin.first_block = 0;
    launch_fused_finish(stream, fin, n);
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
        // (max |d Loss / d atomic_e| over the atoms of this call: the fp16 scale of the weight-gradient kernels' D operand)
        if (ANIHIP_WGRAD_F16) launch_absmax(stream, grad_atomic_e + lo, n, w.amax, AMAX_STAGE_GATOM);
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
            // three fp16 products with scales from bounds (train.hip) where every species has its operand bounds; the six
            // bf16 products (no scales) otherwise or on request
            a.layer = l; a.M = M; a.amax = nullptr;
            if (ANIHIP_WGRAD_F16) {
                bool have = true;
                for (int s = 0; s < S; ++s) { a.bounds[s] = d->net[s].fused_bounds; have = have && a.bounds[s]; }
                if (have) a.amax = w.amax;
            }
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
#ifndef ANIHIP_WG_T1
#define ANIHIP_WG_T1 1536
#define ANIHIP_WG_T2 3072
#endif
                const int64_t target_wgs = tiles >= 64 ? ANIHIP_WG_T1 : ANIHIP_WG_T2;
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
        launch_zero_padding(stream, lo, hi, species, atomic_e, grad_aev, L, nullptr, M, n_atoms);
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

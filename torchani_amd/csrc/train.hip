// Training passes of the ensemble that have no counterpart in the reference's native code (csrc/mnp.cpp has no weight
// gradients: the reference trains through eager autograd, nn/_core.py:146-149, with torch.optim.Adam,
// tools/training-aev-benchmark.py:88,120-150):
//   k_wgrad_x3        dW = D^T X on v_mfma_f32_32x32x16_{f16, bf16}: both operands split on the fly -- two fp16 planes with power-of-two
//                     scales from bounds (three products), or three bf16 planes without scales (six products)
//   k_adam            one launch over the flat parameter / gradient / moment buffers (torch.optim.Adam's update)
//   k_repack_f16      refresh EVERY layout of an ANIHIP_MLP_F16X3 pack from the nn.Linear tensors after an optimizer step
//   k_fused_bounds    ... and the operand bounds of the fused network kernel
#include "train.h"

#include <math.h>

namespace anihip {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) const v4f gf4;

// ---- weight gradients, bf16 x 3 ---------------------------------------------------------------------------------------
// dW[j][i] = sum over the atoms a of one species of  g_a D[a][j] X[a][i]   (D = d e / d pre-activation for a unit upstream
// gradient, as the fused network kernel leaves it; g_a = d Loss / d atomic_e; X = the layer's input).  The reduction index
// is the ATOM, and an MFMA wants eight consecutive reduction indices per lane: both operands are transposed on their way
// into LDS -- a thread loads the same four columns of two consecutive rows (two 16-byte loads), and v_cvt_pk_bf16_f32 of
// the two rows' values IS the packed pair the transposed image wants: [column][32 atoms] bf16, one dword per atom pair.
// Precision: every fp32 value is split into THREE bf16 numbers x = hi + mid + lo (8 + 8 + 8 mantissa bits: exact), and a
// product is the six terms of magnitude >= 2^-16: hi hi + hi mid + mid hi + mid mid + hi lo + lo hi, fp32 accumulation
// (what is dropped is 2^-24 of the product).  bf16 has fp32's exponent range, so -- unlike the split-fp16 planes of the
// inference kernels -- no operand needs a scale: gradients of 1e-9 and activations of 1e+3 go through the same code.
// Six MFMAs per product is twice the fp16 split and still 2.6 x the rate of v_mfma_f32_32x32x2_f32.
// Workgroup: 4 waves, 128 D columns x 128 X columns (a wave: 64 x 64 = 4 accumulators), rows_per_chunk atoms, stages of 32
// atoms; partial tiles are added to dW with float atomics (dW is zeroed, or accumulated into, by the caller).
constexpr int WB_KS = 32;                  // atoms per LDS stage (two MFMA k steps)
constexpr int WB_STR = 40;                 // bf16 per staged column: 32 atoms + 8 of padding (80 B: conflict-free 16-B reads)
constexpr int WB_T = 128;                  // columns of D / of X per workgroup
constexpr int WB_PLANE = WB_T * WB_STR;    // bf16 per plane
constexpr int WB_THREADS = 256;

__device__ __forceinline__ unsigned bf2_bits(v2f v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2)); }

// ---- the same on three fp16 products (round 6) --------------------------------------------------------------------------
// x s = hi + lo in fp16 with ONE power-of-two scale s per launch-side operand and workgroup: the split keeps 2^-25 of the scaled
// maximum (2^13 .. 2^14) in absolute terms, i.e. 2^-39 of the bound the scale was taken from -- a bound that is loose by a factor
// of a thousand still leaves 2^-29 of the true maximum, far below the fp32 accumulation of the products.  The bounds need no
// pass over the data: |D| <= fused_bounds[2 | 3 | 4] (weight norms: include/anihip.h) x max |g_atom| (k_absmax, one tiny
// launch), |X| <= 16376 for the AEV rows (static scale 4, as in the inference kernels), max |act0| as measured by the fused
// training kernel (one atomic per tile), |act1| <= max |act0| [0] + [1].  Three products hi hi + hi lo + lo hi (2^-22 relative),
// two planes per operand instead of three.
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split2h(float x0, float x1, float scale, unsigned &hi, unsigned &lo)
{
    const v2f x = v2f{x0, x1} * scale;
    const h2 h = __builtin_convertvector(x, h2);
    const v2f r = x - __builtin_convertvector(h, v2f);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, h2));
}
__device__ __forceinline__ float pow2_scale_of(float mx)   // mx * scale in [2^13, 2^14); 1 for mx = 0 (or not finite: nothing to save)
{
    if (!(mx > 0.f) || !(mx < 3.0e38f)) return 1.0f;
    const int e = (int)(__float_as_uint(mx) >> 23) - 127;
    return __uint_as_float((unsigned)(127 + 13 - e) << 23);
}
// max |x| over n floats into the running-max table (slot of the stage, species 0)
__global__ __launch_bounds__(256) void k_absmax(const float *x, int64_t n, unsigned *amax, int stage)
{
    float m = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0.f)
        atomicMax(amax + (stage * MAX_S) * AMAX_SLOTS + (blockIdx.x & (AMAX_SLOTS - 1)), __float_as_uint(m));
}
// the running maximum of (stage, s): every lane reads a slot, wave max
__device__ __forceinline__ float amax_value(const unsigned *amax, int stage, int s)
{
    unsigned v = amax[(stage * MAX_S + s) * AMAX_SLOTS + (threadIdx.x & (AMAX_SLOTS - 1))];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o));
    return __uint_as_float(v);
}

// (x0, x1) -> packed {hi, mid, lo} pairs: low half = x0, high half = x1
__device__ __forceinline__ void split3(float x0, float x1, unsigned &hi, unsigned &mid, unsigned &lo)
{
    hi = bf2_bits(v2f{x0, x1});
    const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xFFFF0000u);
    mid = bf2_bits(v2f{r0, r1});
    const float t0 = r0 - __uint_as_float(mid << 16), t1 = r1 - __uint_as_float(mid & 0xFFFF0000u);
    lo = bf2_bits(v2f{t0, t1});
}

#ifndef ANIHIP_WGRAD_WGS
#define ANIHIP_WGRAD_WGS 2   // workgroups per CU the fp16 instantiation is compiled for
#endif
template <bool F16>
__global__ __launch_bounds__(WB_THREADS, F16 ? ANIHIP_WGRAD_WGS : 2) void k_wgrad_x3(WgradB3Args g)
{
    constexpr int NPL = F16 ? 2 : 3;   // planes per operand
    __shared__ __attribute__((aligned(16))) unsigned s_all[2 * NPL * WB_PLANE / 2];   // D planes {hi, (mid,) lo} | X planes
    unsigned *sD = s_all, *sX = s_all + NPL * (WB_PLANE / 2);
    int id = blockIdx.x;
    const int nj = id % g.nj_max; id /= g.nj_max;
    const int ki = id % g.ki_max; id /= g.ki_max;
    const int bb = id % g.batch;
    const int chunk = id / g.batch;
    int s = 0, m0 = 0, first = 0;
    bool found = false;
    for (; s < g.S; ++s) {
        const int nc = (g.ctl[CTL_CNT + s] + g.rows_per_chunk - 1) / g.rows_per_chunk;
        if (chunk < first + nc) {
            m0 = (chunk - first) * g.rows_per_chunk;
            found = true;
            break;
        }
        first += nc;
    }
    if (!found) return;
    const WgradB3Problem &pr = g.prob[s];
    const int i0 = ki * WB_T, j0 = nj * WB_T;
    // X columns: plain, or the compacted list of the AEV slabs that can be non-zero (layer 0, ANI layout; scalar work)
    uint32_t act = 0u;
    const int rad = g.x_slab_rad, nrs = (rad + 31) >> 5;
    if (rad > 0) {
        for (int a = 0; a < g.ani_species; ++a) {
            if (g.ctl[CTL_CNT + a] <= 0) continue;
            act |= 1u << (a >> 1);
            for (int b = a; b < g.ani_species; ++b)
                if (g.ctl[CTL_CNT + b] > 0) act |= 1u << (nrs + a * g.ani_species - a * (a - 1) / 2 + (b - a));
        }
    }
    const int k_cols = rad > 0 ? 32 * __popc(act) : pr.k_valid;   // columns of the (compacted) X index
    if (i0 >= k_cols || j0 >= pr.N) return;
    // compacted column c -> (AEV column, is it a column of the row); plain: the identity
    auto x_column = [&](int c, int &col) {
        if (rad <= 0) { col = c; return c < pr.k_valid; }
        uint32_t mk = act;
        for (int t = 0; t < (c >> 5); ++t) mk &= mk - 1u;
        if (!mk) { col = 0; return false; }
        const int slab = (int)__builtin_ctz(mk);
        const int start = slab < nrs ? 32 * slab : rad + 32 * (slab - nrs);
        const int valid = slab == nrs - 1 ? rad - 32 * (nrs - 1) : 32;
        col = start + (c & 31);
        return (c & 31) < valid && col < pr.k_valid;
    };
    const int n_rows = min(g.rows_per_chunk, g.ctl[CTL_CNT + s] - m0);
    const int p0 = g.ctl[CTL_OFF + s] + m0;
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
    const int wn = wave & 1, wk = wave >> 1;

    // F16: the workgroup's two scales (wave-uniform; layer 0 holds all members side by side: the largest bound)
    float scale_d = 1.0f, scale_x = 1.0f;
    if constexpr (F16) {
        const float gam = amax_value(g.amax, AMAX_STAGE_GATOM, 0), a0m = amax_value(g.amax, AMAX_STAGE_ACT0, s);
        const float *bnd = g.bounds[s];
        float bd = 0.f, bx = 0.f;
        const int m_lo = g.layer == 0 ? 0 : bb, m_hi = g.layer == 0 ? g.M : bb + 1;
        for (int m = m_lo; m < m_hi; ++m) {
            bd = fmaxf(bd, bnd[8 * m + (g.layer == 0 ? 4 : (g.layer == 1 ? 3 : 2))]);
            bx = fmaxf(bx, g.layer == 1 ? a0m : __builtin_fmaf(a0m, bnd[8 * m + 0], bnd[8 * m + 1]));
        }
        scale_d = pow2_scale_of(bd * gam);
        scale_x = g.layer == 0 ? 4.0f : pow2_scale_of(bx);
    }
    // staging role: atom pair `pair` of the stage, column groups cg0 and cg0 + 16 (four columns each) of D and of X
    const int pair = tid & 15, cg0 = tid >> 4;
    const float *Db = pr.D + (int64_t)bb * pr.d_boff, *Xb = pr.X + (int64_t)bb * pr.x_boff;
    int dcol[2], xcol[2];
    bool dok[2], xok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int c = 4 * (cg0 + 16 * u);
        dok[u] = j0 + c < pr.N;
        dcol[u] = dok[u] ? j0 + c : 0;
        int col;
        xok[u] = x_column(i0 + c, col);   // (four consecutive columns lie inside one slab: the first decides)
        xcol[u] = xok[u] ? col : 0;
    }
    struct Rows { v4f rd[2][2], rx[2][2]; float ga[2]; };   // a stage's rows of this thread: [column group][row of the pair]
    Rows ra;
    // The row indices of a stage (atom of the row for g_atom, source row of X for layer 0) are loads of their own that the data
    // loads depend on: fetched ONE STAGE AHEAD of the data (load_idx), or the data "prefetch" waits for them right where it is
    // issued -- a microsecond per 32-atom stage in front of its MFMAs (round 6: the stage took 9 k cycles for 3.6 k of work)
    int i_atom[2];
    int64_t i_xrow[2];
    auto load_idx = [&](int r0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = r0 + 2 * pair + h;
            const int p = p0 + (r < n_rows ? r : 0);   // (clamped: always valid memory)
            i_xrow[h] = g.x_gather ? (int64_t)g.x_gather[p] : (int64_t)p;
            i_atom[h] = g.g_atom ? g.perm[p] : 0;
        }
    };
    auto load = [&](Rows &R, int r0) {   // (with the indices load_idx(r0) fetched)
        v4f (&rd)[2][2] = R.rd, (&rx)[2][2] = R.rx;
        float (&ga)[2] = R.ga;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = r0 + 2 * pair + h;
            const bool v = r < n_rows;
            const int p = p0 + (v ? r : 0);
            const int64_t xrow = i_xrow[h];
            ga[h] = v ? (g.g_atom ? g.g_atom[i_atom[h]] : 1.0f) : 0.f;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                // (unconditional, clamped columns: a load under a condition -- even a wave-uniform one, tried for the empty
                // 64-column halves of the hidden layers' edge tiles -- is waited for right behind its issue; measured: the whole
                // gain of the early indices gone)
                v4f d = *(const gf4 *)(Db + (int64_t)p * pr.ldd + dcol[u]);
                v4f x = *(const gf4 *)(Xb + xrow * pr.ldx + xcol[u]);
                if (!dok[u]) d = v4f{0.f, 0.f, 0.f, 0.f};
                if (!(xok[u] && v)) x = v4f{0.f, 0.f, 0.f, 0.f};
                rd[u][h] = d;
                rx[u][h] = x;
            }
        }
    };
    float bsum[2][4];   // column sums of the scaled D rows this thread stages (bias gradients; the first X tile's workgroups)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) bsum[u][e] = 0.f;
    auto store = [&](const Rows &R) {
        const v4f (&rd)[2][2] = R.rd, (&rx)[2][2] = R.rx;
        const float (&ga)[2] = R.ga;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int at = (4 * (cg0 + 16 * u) + e) * (WB_STR / 2) + pair;   // dword of this column's atom pair
                unsigned hi, mid, lo;
                const float d0 = rd[u][0][e] * ga[0], d1 = rd[u][1][e] * ga[1];
                bsum[u][e] += d0 + d1;
                if constexpr (F16) {
                    split2h(d0, d1, scale_d, hi, lo);
                    sD[at] = hi; sD[at + WB_PLANE / 2] = lo;
                    split2h(rx[u][0][e], rx[u][1][e], scale_x, hi, lo);
                    sX[at] = hi; sX[at + WB_PLANE / 2] = lo;
                } else {
                    split3(d0, d1, hi, mid, lo);
                    sD[at] = hi; sD[at + WB_PLANE / 2] = mid; sD[at + 2 * (WB_PLANE / 2)] = lo;
                    split3(rx[u][0][e], rx[u][1][e], hi, mid, lo);
                    sX[at] = hi; sX[at + WB_PLANE / 2] = mid; sX[at + 2 * (WB_PLANE / 2)] = lo;
                }
            }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // 32-column blocks of this wave that exist (wave-uniform)
    const int nbv = max(0, min(2, (pr.N - (j0 + wn * 64) + 31) >> 5));
    const int kbv = max(0, min(2, (k_cols - (i0 + wk * 64) + 31) >> 5));
    const bool active = nbv > 0 && kbv > 0;
    // fragment of a block: column (lane & 31), eight atoms 8 (lane >> 5) .. of the k step
    const int fcol = lane & 31, fk = lane >> 5;
    const unsigned short *hD = reinterpret_cast<const unsigned short *>(sD), *hX = reinterpret_cast<const unsigned short *>(sX);

    auto mfma_stage = [&]() {
        if (active) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if constexpr (F16) {
                    h8 a[2][2], b[2][2];
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl) {
                            a[q][pl] = *reinterpret_cast<const h8 *>(hD + pl * WB_PLANE + (wn * 64 + q * 32 + fcol) * WB_STR + ks * 16 + fk * 8);
                            b[q][pl] = *reinterpret_cast<const h8 *>(hX + pl * WB_PLANE + (wk * 64 + q * 32 + fcol) * WB_STR + ks * 16 + fk * 8);
                        }
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        if (nb >= nbv) continue;
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb) {
                            if (kb >= kbv) continue;
                            f32x16 c = acc[nb][kb];
                            // (the small terms first)
                            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[nb][1], b[kb][0], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[nb][0], b[kb][1], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[nb][0], b[kb][0], c, 0, 0, 0);
                            acc[nb][kb] = c;
                        }
                    }
                } else {
                bf8 a[2][3], b[2][3];
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        a[q][pl] = *reinterpret_cast<const bf8 *>(hD + pl * WB_PLANE + (wn * 64 + q * 32 + fcol) * WB_STR + ks * 16 + fk * 8);
                        b[q][pl] = *reinterpret_cast<const bf8 *>(hX + pl * WB_PLANE + (wk * 64 + q * 32 + fcol) * WB_STR + ks * 16 + fk * 8);
                    }
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    if (nb >= nbv) continue;
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        if (kb >= kbv) continue;
                        f32x16 c = acc[nb][kb];
                        // (the small terms first)
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[nb][2], b[kb][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[nb][0], b[kb][2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[nb][1], b[kb][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[nb][1], b[kb][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[nb][0], b[kb][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[nb][0], b[kb][0], c, 0, 0, 0);
                        acc[nb][kb] = c;
                    }
                }
                }
            }
        }
    };
    // stage pipeline: the indices two stages, the data one stage ahead of the MFMAs.  (Two register sets -- data two stages
    // ahead -- do not fit: 256 registers with 70 spilled.)
    load_idx(0);
    load(ra, 0);
    load_idx(WB_KS);
    for (int r0 = 0; r0 < n_rows; r0 += WB_KS) {
        __syncthreads();   // (every wave is done reading the previous stage)
        store(ra);
        __syncthreads();
        load(ra, r0 + WB_KS);   // (unconditional: rows behind the chunk read clamped addresses and count as zeros)
        load_idx(r0 + 2 * WB_KS);
        mfma_stage();
    }
    if (ki == 0 && g.gbias[s]) {
        // bias gradients: the sixteen atom pairs of a column lie in the sixteen lanes of a DPP row
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = bsum[u][e];
                v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, true));   // row_shr:1
                v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xF, 0xF, true));   // row_shr:2
                v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xF, 0xF, true));   // row_shr:4
                v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xF, 0xF, true));   // row_shr:8
                const int j = j0 + 4 * (cg0 + 16 * u) + e;
                if (pair == 15 && j < pr.N) {
                    const int mem = j / pr.n_per, row = j - mem * pr.n_per;
                    atomicAdd(g.gbias[s] + (int64_t)(bb + mem) * g.b_mstride[s] + row, v);
                }
            }
    }
    if (!active) return;
    const float unscale = F16 ? 1.0f / (scale_d * scale_x) : 1.0f;   // (powers of two: exact)
    // accumulator element r of lane l: row (output unit) (r & 3) + 8 (r >> 2) + 4 (l >> 5), column (input unit) l & 31
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        if (nb >= nbv) continue;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (kb >= kbv) continue;
            int i;
            if (!x_column(i0 + wk * 64 + kb * 32 + fcol, i)) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = j0 + wn * 64 + nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                if (j >= pr.N) continue;
                const int mem = j / pr.n_per, row = j - mem * pr.n_per;
                atomicAdd(pr.dW + (int64_t)(bb + mem) * pr.w_mstride + (int64_t)row * pr.ldw + i, acc[nb][kb][r] * unscale);
            }
        }
    }
}

void launch_wgrad_b3(hipStream_t stream, const WgradB3Args &a, int64_t rows_total)
{
    const int64_t chunks = (rows_total + a.rows_per_chunk - 1) / a.rows_per_chunk + a.S;
    const int64_t total = chunks * a.batch * a.ki_max * a.nj_max;
    if (total <= 0) return;
    if (a.amax && a.bounds[0]) {
        // (max |g_atom| of the rows of this call: the scale of the D operand needs it -- one small launch per weight-gradient launch)
        hipLaunchKernelGGL(k_wgrad_x3<true>, dim3((unsigned)total), dim3(WB_THREADS), 0, stream, a);
    } else {
        hipLaunchKernelGGL(k_wgrad_x3<false>, dim3((unsigned)total), dim3(WB_THREADS), 0, stream, a);
    }
}

void launch_absmax(hipStream_t stream, const float *x, int64_t n, unsigned *amax, int stage)
{
    if (n <= 0) return;
    int64_t blocks = (n + 1023) / 1024;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(k_absmax, dim3((unsigned)blocks), dim3(256), 0, stream, x, n, amax, stage);
}

// ---- Adam ---------------------------------------------------------------------------------------------------------------
// torch.optim.Adam's update (amsgrad = False, maximize = False; weight_decay adds wd * p to the gradient) over flat
// buffers: 28 bytes of traffic per parameter in ONE launch instead of a dozen foreach launches over 448 tensors; optionally
// the gradients are zeroed behind the update (the next backward accumulates into them: no memset launch per step).
// step: device counter of the updates done so far (the bias corrections need it; on the device so that a captured HIP graph
// of the training step advances it on replay).
__global__ __launch_bounds__(256) void k_adam(float *p, float *g, float *m, float *v, int64_t n, double lr, double b1d,
                                              double b2d, float eps, float wd, const int32_t *step, int zero_grads)
{
    __shared__ float s_c[2];
    if (threadIdx.x == 0) {
        const double t = (double)(*step + 1);
        const double bc1 = 1.0 - pow(b1d, t), bc2 = 1.0 - pow(b2d, t);
        s_c[0] = (float)(lr / bc1);   // step size
        s_c[1] = (float)sqrt(bc2);
    }
    __syncthreads();
    const float step_size = s_c[0], bc2s = s_c[1];
    // (1 - beta in double, like torch's lerp_ / addcmul_ weights: 1.0f - 0.999f is 1.3e-5 off)
    const float b2 = (float)b2d, ob1 = (float)(1.0 - b1d), ob2 = (float)(1.0 - b2d);
    const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
        v4f pp = reinterpret_cast<v4f *>(p)[i], mm = reinterpret_cast<v4f *>(m)[i], vv = reinterpret_cast<v4f *>(v)[i];
        const v4f gg = reinterpret_cast<const v4f *>(g)[i];
        if (zero_grads) reinterpret_cast<v4f *>(g)[i] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gr = gg[e] + wd * pp[e];
            mm[e] = mm[e] + ob1 * (gr - mm[e]);              // lerp_(grad, 1 - beta1)
            vv[e] = b2 * vv[e] + ob2 * gr * gr;             // mul_(beta2).addcmul_(grad, grad, 1 - beta2)
            const float denom = sqrtf(vv[e]) / bc2s + eps;
            pp[e] = pp[e] - step_size * (mm[e] / denom);
        }
        reinterpret_cast<v4f *>(p)[i] = pp;
        reinterpret_cast<v4f *>(m)[i] = mm;
        reinterpret_cast<v4f *>(v)[i] = vv;
    }
    // tail (n not a multiple of 4)
    for (int64_t i = (n4 << 2) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gr = g[i] + wd * p[i];
        const float mm = m[i] + ob1 * (gr - m[i]);
        const float vv = b2 * v[i] + ob2 * gr * gr;
        m[i] = mm;
        v[i] = vv;
        p[i] = p[i] - step_size * (mm / (sqrtf(vv) / bc2s + eps));
        if (zero_grads) g[i] = 0.f;
    }
}

__global__ void k_step_inc(int32_t *step)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) *step += 1;
}

// ---- parameter refresh of a split-fp16 pack ---------------------------------------------------------------------------
// Every layout anihip_mlp_pack (csrc/pack.hip) derives from the nn.Linear tensors -- fp32 w / wt / bias, the {hi, lo} fp16
// planes wh / wth, their MFMA fragment orders whf / wthf -- rewritten in place from the tensors (one thread per source
// weight, ten scattered stores), with the power-of-two weight scales the pack was built with: a weight that has outgrown
// the fp16 range of its scale sets bit 0 of *status (the caller re-packs on the host: new scales).
struct RepackF16Args {
    const float *const *src;   // device: [M][S][nl][2] pointers {weight [out][in], bias [out]}
    float *w[MAX_S][ANIHIP_MAX_LAYERS], *wt[MAX_S][ANIHIP_MAX_LAYERS], *bias[MAX_S][ANIHIP_MAX_LAYERS];
    _Float16 *wh[MAX_S][ANIHIP_MAX_LAYERS], *wth[MAX_S][ANIHIP_MAX_LAYERS], *whf[MAX_S][ANIHIP_MAX_LAYERS],
        *wthf[MAX_S][ANIHIP_MAX_LAYERS];
    float scale[MAX_S][ANIHIP_MAX_LAYERS];
    int dims[MAX_S][ANIHIP_MAX_LAYERS + 1];                            // padded widths
    int out[MAX_S][ANIHIP_MAX_LAYERS], in[MAX_S][ANIHIP_MAX_LAYERS];   // widths of the source tensors
    int S, M, nl, k0p, K0h, R, Rpad;
    int fused_only;   // ANIHIP_REPACK_FUSED_ONLY: bias, output layer, whf, wthf of the hidden layers (+ the bounds)
    int32_t *status;
};

// index of element (n, k) of plane pl of member m in the fragment order of pack.hip's to_fragments:
// [M][N / 32][K / 16][plane][k half][32 rows][8]
__device__ __forceinline__ int64_t frag_index(int N, int K, int m, int n, int k, int pl)
{
    return ((((((int64_t)m * (N >> 5) + (n >> 5)) * (K >> 4) + (k >> 4)) * 2 + pl) * 2 + ((k >> 3) & 1)) * 32 + (n & 31)) * 8 + (k & 7);
}

__global__ __launch_bounds__(256) void k_repack_f16(RepackF16Args g)
{
    int id = blockIdx.y;
    const int l = id % g.nl; id /= g.nl;
    const int s = id % g.S;
    const int m = id / g.S;
    const int out = g.out[s][l], in = g.in[s][l];
    const float *W = g.src[((m * g.S + s) * g.nl + l) * 2 + 0];
    const float *b = g.src[((m * g.S + s) * g.nl + l) * 2 + 1];
    const int inp = g.dims[s][l], outp = g.dims[s][l + 1];
    const bool f16 = l < g.nl - 1 && g.wh[s][l] != nullptr;
    const float scale = g.scale[s][l];
    bool over = false;
    for (int64_t e = blockIdx.x * 256 + threadIdx.x; e < (int64_t)out * in; e += (int64_t)gridDim.x * 256) {
        const int o = (int)(e / in), k = (int)(e - (int64_t)o * in);
        const float v = W[e];
        if (l == g.nl - 1) {
            g.w[s][l][(int64_t)m * inp + k] = v;
            continue;
        }
        // a weight that has outgrown the fp16 range of its layer's scale is CLAMPED to it (and reported): unclamped, hi = inf
        // and lo = -inf would put NaN into this step's forward, its gradients and -- through the optimizer -- the parameters
        // before the host reads the status word one step later and packs again with new scales (round-5 advice)
        const float xs = v * scale;
        over = over || !(fabsf(xs) < 65504.f);
        const float x = fminf(fmaxf(xs, -65504.f), 65504.f);   // (NaN parameters stay NaN: fminf / fmaxf return the other operand, -65504)
        const _Float16 hi = (_Float16)x, lo = (_Float16)(x - (float)hi);
        if (l == 0) {
            const int64_t ld = (int64_t)g.M * outp, col = (int64_t)m * outp + o;
            const int kc = g.R ? (k < g.R ? k : g.Rpad + (k - g.R)) : k;   // slab order of the AEV columns
            if (!g.fused_only) {
                g.w[s][l][(int64_t)k * ld + col] = v;
                g.wt[s][l][col * g.k0p + k] = v;
                if (f16) {
                    const int64_t n_el = ld * g.K0h;
                    g.wh[s][l][col * g.K0h + kc] = hi;
                    g.wh[s][l][n_el + col * g.K0h + kc] = lo;
                    g.wth[s][l][(int64_t)kc * ld + col] = hi;
                    g.wth[s][l][n_el + (int64_t)kc * ld + col] = lo;
                }
            }
            if (!f16) continue;
            g.whf[s][l][frag_index(outp, g.K0h, m, o, kc, 0)] = hi;
            g.whf[s][l][frag_index(outp, g.K0h, m, o, kc, 1)] = lo;
            if (g.wthf[s][l] && !g.fused_only) {   // W0 transposed per member: N = the AEV columns in slab order, K = the member's H1 columns
                g.wthf[s][l][frag_index(g.K0h, outp, m, kc, o, 0)] = hi;
                g.wthf[s][l][frag_index(g.K0h, outp, m, kc, o, 1)] = lo;
            }
        } else {
            if (!g.fused_only) {
                g.w[s][l][((int64_t)m * inp + k) * outp + o] = v;
                g.wt[s][l][((int64_t)m * outp + o) * inp + k] = v;
                if (f16) {
                    const int64_t n_el = (int64_t)g.M * outp * inp;
                    g.wh[s][l][((int64_t)m * outp + o) * inp + k] = hi;
                    g.wh[s][l][n_el + ((int64_t)m * outp + o) * inp + k] = lo;
                    g.wth[s][l][((int64_t)m * inp + k) * outp + o] = hi;
                    g.wth[s][l][n_el + ((int64_t)m * inp + k) * outp + o] = lo;
                }
            }
            if (!f16) continue;
            g.whf[s][l][frag_index(outp, inp, m, o, k, 0)] = hi;
            g.whf[s][l][frag_index(outp, inp, m, o, k, 1)] = lo;
            if (g.wthf[s][l]) {
                g.wthf[s][l][frag_index(inp, outp, m, k, o, 0)] = hi;
                g.wthf[s][l][frag_index(inp, outp, m, k, o, 1)] = lo;
            }
        }
    }
    if (over && g.status) atomicOr(reinterpret_cast<int *>(g.status), 1);
    if (blockIdx.x == 0)
        for (int o = threadIdx.x; o < out; o += 256)
            g.bias[s][l][(l == g.nl - 1) ? m : (int64_t)m * outp + o] = b[o];
}

// operand bounds of the fused kernel's inner GEMMs (include/anihip.h, fused_bounds; pack.hip computes the same on the
// host): one workgroup per (species, member)
struct BoundsArgs {
    const float *const *src;
    float *bounds[MAX_S];
    int H1[MAX_S], H2[MAX_S], H3[MAX_S];
    int S, M, nl;
    float dmax;
};

__device__ __forceinline__ void block_max(unsigned *slot, float v)
{
    atomicMax(slot, __float_as_uint(v));   // (non-negative values: integer order)
}

// (1024 threads: the launch is 56 workgroups of dependent chains, 77 us with 256 threads each -- 3 % of a training step)
constexpr int FB_THREADS = 1024;
__global__ __launch_bounds__(FB_THREADS) void k_fused_bounds(BoundsArgs g)
{
    __shared__ unsigned s_mx[5];   // row1, bmax, col1, col2, w3max
    const int s = blockIdx.x % g.S, m = blockIdx.x / g.S;
    if (!g.bounds[s]) return;
    if (threadIdx.x < 5) s_mx[threadIdx.x] = 0u;
    __syncthreads();
    const float *W1 = g.src[((m * g.S + s) * g.nl + 1) * 2 + 0], *b1 = g.src[((m * g.S + s) * g.nl + 1) * 2 + 1];
    const float *W2 = g.src[((m * g.S + s) * g.nl + 2) * 2 + 0], *w3 = g.src[((m * g.S + s) * g.nl + 3) * 2 + 0];
    const int H1 = g.H1[s], H2 = g.H2[s], H3 = g.H3[s];
    float row1 = 0.f, bmax = 0.f, col1 = 0.f, col2 = 0.f, w3max = 0.f;
    // row sums of |W1| (a wave per row, lanes along the contiguous index), column sums of |W1| and |W2| (a thread per column:
    // consecutive threads read consecutive addresses)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int j = wv; j < H2; j += FB_THREADS / 64) {
        float rs = 0.f;
        for (int k = lane; k < H1; k += 64) rs += fabsf(W1[(int64_t)j * H1 + k]);
        row1 = fmaxf(row1, wave_sum(rs));
    }
    for (int j = threadIdx.x; j < H2; j += FB_THREADS) bmax = fmaxf(bmax, fabsf(b1[j]));
    // (column sums: four threads per column, each a quarter of the rows, eight independent loads in flight)
    {
        const int q = threadIdx.x & 3;
        for (int k = threadIdx.x >> 2; k < H1; k += FB_THREADS / 4) {
            float cs = 0.f;
#pragma unroll 8
            for (int j = q; j < H2; j += 4) cs += fabsf(W1[(int64_t)j * H1 + k]);
            cs += __shfl_xor(cs, 1);
            cs += __shfl_xor(cs, 2);
            col1 = fmaxf(col1, cs);
        }
        for (int k = threadIdx.x >> 2; k < H2; k += FB_THREADS / 4) {
            float cs = 0.f;
#pragma unroll 8
            for (int j = q; j < H3; j += 4) cs += fabsf(W2[(int64_t)j * H2 + k]);
            cs += __shfl_xor(cs, 1);
            cs += __shfl_xor(cs, 2);
            col2 = fmaxf(col2, cs);
        }
    }
    for (int j = threadIdx.x; j < H3; j += FB_THREADS) w3max = fmaxf(w3max, fabsf(w3[j]));
    block_max(&s_mx[0], row1); block_max(&s_mx[1], bmax); block_max(&s_mx[2], col1); block_max(&s_mx[3], col2);
    block_max(&s_mx[4], w3max);
    __syncthreads();
    if (threadIdx.x == 0) {
        float *q = g.bounds[s] + 8 * (int64_t)m;
        // (a hair above the host packer's sums: the summation orders differ, and these are upper bounds)
        const float r1 = __uint_as_float(s_mx[0]) * 1.0001f, c1 = __uint_as_float(s_mx[2]) * 1.0001f,
                    c2 = __uint_as_float(s_mx[3]) * 1.0001f;
        const float g2 = __uint_as_float(s_mx[4]) / (float)g.M * g.dmax, g3 = g2 * c2 * g.dmax, g4 = g3 * c1;
        q[0] = r1; q[1] = __uint_as_float(s_mx[1]); q[2] = g2; q[3] = g3; q[4] = g4; q[5] = q[6] = q[7] = 0.f;
    }
}

int repack_f16(hipStream_t stream, const anihip_mlp_desc *d, const void *const *src, const int32_t *out_in, int32_t *status,
               int32_t flags)
{
    RepackF16Args a{};
    a.src = (const float *const *)src;
    a.S = d->num_species; a.M = d->n_members; a.nl = d->net[0].n_layers;
    const int K0 = d->aev_len;
    a.k0p = ((K0 + 31) / 32) * 32;
    a.R = d->aev_radial_len;
    a.Rpad = ((a.R + 31) / 32) * 32;
    a.K0h = a.R ? a.Rpad + (K0 - a.R) : a.k0p;
    a.status = status;
    a.fused_only = (flags & ANIHIP_REPACK_FUSED_ONLY) ? 1 : 0;
    int64_t biggest = 0;
    bool fused = a.nl == 4;
    for (int s = 0; s < a.S; ++s) {
        const anihip_species_net &nn = d->net[s];
        for (int l = 0; l <= a.nl; ++l) a.dims[s][l] = nn.dims[l];
        for (int l = 0; l < a.nl; ++l) {
            a.out[s][l] = out_in[(s * a.nl + l) * 2 + 0];
            a.in[s][l] = out_in[(s * a.nl + l) * 2 + 1];
            ANIHIP_REQUIRE(a.out[s][l] >= 1 && a.out[s][l] <= nn.dims[l + 1] && a.in[s][l] >= 1 && a.in[s][l] <= nn.dims[l],
                           "species %d layer %d: source shape outside the packed shape", s, l);
            a.w[s][l] = const_cast<float *>(nn.w[l]);
            a.wt[s][l] = const_cast<float *>(nn.wt[l]);
            a.bias[s][l] = const_cast<float *>(nn.bias[l]);
            a.wh[s][l] = (_Float16 *)const_cast<void *>(nn.wh[l]);
            a.wth[s][l] = (_Float16 *)const_cast<void *>(nn.wth[l]);
            a.whf[s][l] = (_Float16 *)const_cast<void *>(nn.whf[l]);
            a.wthf[s][l] = (_Float16 *)const_cast<void *>(nn.wthf[l]);
            a.scale[s][l] = nn.wh_scale[l];
            if (l < a.nl - 1)
                ANIHIP_REQUIRE(nn.wh[l] && nn.wth[l] && nn.whf[l] && nn.wh_scale[l] > 0.f,
                               "species %d layer %d: not a split-fp16 pack", s, l);
            const int64_t e = (int64_t)a.out[s][l] * a.in[s][l];
            biggest = biggest > e ? biggest : e;
        }
        fused = fused && nn.fused_bounds != nullptr;
    }
    unsigned bx = (unsigned)((biggest + 256 * 8 - 1) / (256 * 8));
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(k_repack_f16, dim3(bx, (unsigned)(a.M * a.S * a.nl)), dim3(256), 0, stream, a);
    if (fused) {
        BoundsArgs b{};
        b.src = a.src; b.S = a.S; b.M = a.M; b.nl = a.nl;
        b.dmax = d->activation == ANIHIP_ACT_GELU ? 1.13f : 1.0f;
        for (int s = 0; s < a.S; ++s) {
            b.bounds[s] = const_cast<float *>(d->net[s].fused_bounds);
            b.H1[s] = a.out[s][0]; b.H2[s] = a.out[s][1]; b.H3[s] = a.out[s][2];
        }
        hipLaunchKernelGGL(k_fused_bounds, dim3((unsigned)(a.S * a.M)), dim3(FB_THREADS), 0, stream, b);
    }
    ANIHIP_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace anihip

using namespace anihip;

extern "C" int anihip_adam_step(void *stream_, float *params, float *grads, float *exp_avg, float *exp_avg_sq,
                                int64_t n, double lr, double beta1, double beta2, double eps, double weight_decay,
                                int32_t *step, int32_t zero_grads)
{
    hipStream_t stream = (hipStream_t)stream_;
    ANIHIP_REQUIRE(params && grads && exp_avg && exp_avg_sq && step, "null pointer argument");
    ANIHIP_REQUIRE(n >= 0, "negative parameter count");
    ANIHIP_REQUIRE(lr >= 0. && beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1. && eps >= 0. && weight_decay >= 0.,
                   "invalid Adam hyper-parameters");
    ANIHIP_REQUIRE(((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0,
                   "parameter, gradient and moment buffers must be 16-byte aligned");
    if (n > 0) {
        int64_t blocks = (n / 4 + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(256), 0, stream, params, grads, exp_avg, exp_avg_sq, n, lr,
                           beta1, beta2, (float)eps, (float)weight_decay, step, (int)zero_grads);
    }
    hipLaunchKernelGGL(k_step_inc, dim3(1), dim3(64), 0, stream, step);
    ANIHIP_CHECK_HIP(hipGetLastError());
    return 0;
}

// Radial + angular AEV forward and analytic backward for gfx950 (wave64).
//
// One wave owns one central atom at a time (persistent waves stride over the shard).  The atom's species-sorted neighbor
// row (written by nbr.hip) is loaded with one coalesced 16-B load per lane -- header and entries of the NEXT atom are
// in flight while the current one is computed -- turned into unit vectors / cutoff factors once per neighbor and staged
// in the wave's private LDS.  Three kernels:
//
//   k_aev_fwd3  forward of the energy / force path: one lane per (j, k) pair over a flat slot assignment of all species-pair
//               blocks, log-domain cutoff product, packed outer product, segmented reduction through LDS.  No atomics.
//   k_aev_bwd   analytic backward: radial part by symmetric gather (each atom finishes its own radial force from the
//               64-B block of every neighbor's dE/dAEV row), angular part over all neighbor pairs in one tournament
//               with lane-owned j and plain LDS read-add-write for k, dE/dAEV staged block-wise; the few remaining
//               global accumulations are float atomics or (ANIHIP_BWD_FIXED_POINT) 64-bit integer atomics.
//   k_aev_fwd<.., JVP = true>  forward-mode derivative J t (the reference's double backward), four lanes per pair.
//
// Maths restated from the reference (paths relative to /root/reference/torchani/):
//   aev/_terms.py:99-104,171-186 (radial), :34-55,324-325,339-343 (angular), cutoffs.py:80-81,
//   aev/_computer.py:183-191,302-350 (species / species-pair binning and layout).  cos(theta - ShfZ)
//   is expanded as cos(theta)cos(ShfZ)+sin(theta)sin(ShfZ) with cos(theta)=0.95 cos(angle), which
//   is algebraically identical to acos + cos (aev/_terms.py:339-343) and needs no acos.
#include "anihip_common.h"

namespace anihip {

// Workgroups are dealt to the 8 XCDs round-robin (workgroup b runs on XCD b mod 8) and every XCD has its own L2: with
// "workgroup b takes atoms 4 b .. 4 b + 3" the neighbors that consecutive atoms share are fetched into all eight L2s.
// Remapped, the workgroups of one XCD take a contiguous run of atoms in every sweep of the grid.
__device__ __forceinline__ int xcd_block()
{
#ifdef ANIHIP_NO_XCD_MAP
    return (int)blockIdx.x;
#else
    const int nb = (int)gridDim.x;
    if (nb & 7) return (int)blockIdx.x;
    return ((int)blockIdx.x & 7) * (nb >> 3) + ((int)blockIdx.x >> 3);
#endif
}

// The persistent waves of the tuned AEV kernels take their atoms from a QUEUE of their workgroup.  A workgroup is a whole
// CU's worth of waves (QW = 16: four per SIMD, one workgroup per CU) and owns the atoms  first + 16 b + r + 16 blocks k
// (r = 0..15, k = 0, 1, ..: the same 16-atom groups at a stride of all workgroups as with a fixed share per wave); position
// p = 16 k + r of that list goes to whichever wave asks next -- an LDS counter, one returning ds_add of lane 0 per atom,
// issued an atom ahead.  With a FIXED share per wave the waves of a SIMD finish far apart (the issue arbiter favours the
// oldest wave: wave totals from 0.74 to 1.30 of their mean, tools/fwd3_trace.py) and the SIMD spends the last fifth of
// the kernel with three, two, one wave(s) to hide latencies with; from the queue all of a CU's waves work until its atoms
// run out (totals within 1 %).  (A device-wide counter does the same for the whole chip but its atomics execute in memory
// -- the L2s of the eight XCDs are not coherent -- at ~10 ns each, serialised: 0.6 M requests per launch were slower than the
// imbalance they removed.)
constexpr int QW = 16;
struct AtomQueue {
    uint32_t *ctr;        // LDS: the next position nobody has taken
    int64_t base, stride;
    int lane;
    uint32_t v;           // lane 0: the position a request in flight returns
    __device__ __forceinline__ AtomQueue(uint32_t *c, int64_t first, int block, int blocks, int lane_)
        : ctr(c), base(first + (int64_t)block * QW), stride((int64_t)blocks * QW), lane(lane_), v(0u) {}
    __device__ __forceinline__ int64_t atom(uint32_t p) const { return base + (int64_t)(p >> 4) * stride + (p & 15u); }
    __device__ __forceinline__ void request() { if (lane == 0) v = atomicAdd(ctr, 1u); }
    __device__ __forceinline__ int64_t granted() const { return atom((uint32_t)__builtin_amdgcn_readfirstlane((int)v)); }
};

constexpr int FWD_WPB = 4;      // k_aev_fwd (tangent pass)
constexpr int FWD3_WPB = QW;    // k_aev_fwd3
constexpr int BWD_WPB = QW;     // k_aev_bwd
constexpr int STAGE_FLOATS = 1024;  // >= L (S<=7: 1008)
constexpr float LOG2E = 1.4426950408889634f;
constexpr float PI_F = 3.14159265358979323846f;

struct AevArgs {
    int S, NR, L, radlen;
    float Rcr, Rca, kR, kA, EtaR, EtaA, Zeta;
    float qR, qA;   // sqrt(eta log2 e): exp(-eta x^2) = exp2(-(q x)^2)
    int smooth;  // cutoff_kind: 0 = CutoffCosine, 1 = CutoffSmooth (order 2, eps 1e-10)
    int update;  // k_aev_fwd3: rows are UPDATED in place -- slab_mask[i] on entry = the slabs of row i that may hold non-zero
                 // data from the previous call on these buffers; only those and the slabs flagged now are written
};

// CutoffSmooth (cutoffs.py:84-101, csrc/aev.cu:150-178): fc = exp(1 - 1/max(eps, 1 - (r/Rc)^2)); 1 - q^2 is formed
// as (1-q)(1+q) to keep its relative error at one ulp close to the cutoff.  Returns {fc, dfc/dr}.
#define SMOOTH_EPS 1e-10f
__device__ __forceinline__ float2 smooth_cutoff(float r, float inv_rc)
{
    const float q = r * inv_rc;
    const float m1 = (1.0f - q) * (1.0f + q);
    const float im = 1.0f / fmaxf(SMOOTH_EPS, m1);
    const float f = __builtin_amdgcn_exp2f((1.0f - im) * LOG2E);
    const float df = m1 - SMOOTH_EPS >= 0.0f ? -2.0f * r * inv_rc * inv_rc * f * im * im : 0.0f;
    return make_float2(f, df);
}

template <int Q>
__device__ __forceinline__ float quad_bcast(float v)
{
    // DPP quad_perm: every lane of a quad reads quad-lane Q
    return __int_as_float(
        __builtin_amdgcn_update_dpp(0, __float_as_int(v), Q * 0x55, 0xF, 0xF, true));
}

__device__ __forceinline__ float quad_bcast_rt(float v, int q)
{
    switch (q) {
        case 0: return quad_bcast<0>(v);
        case 1: return quad_bcast<1>(v);
        case 2: return quad_bcast<2>(v);
        default: return quad_bcast<3>(v);
    }
}

__device__ __forceinline__ int cnt_of(uint64_t pk, int t) { return (int)((pk >> (8 * t)) & 255u); }
// byte (sh / 8) of a wave-uniform 64-bit word for a per-lane bit shift sh = 0, 8, .. 56, in 32-bit operations (a per-lane
// 64-bit shift of a scalar pair keeps two more registers alive per use)
__device__ __forceinline__ int byte_at(uint64_t pk, int sh)
{
    const uint32_t w = sh < 32 ? (uint32_t)pk : (uint32_t)(pk >> 32);
    return (int)((w >> (sh & 31)) & 255u);
}

// (j,k) of the t-th pair of a block: rectangle for two different species, circular tournament for
// pairs inside one species (every unordered pair exactly once, no sqrt / triangular-index decode)
__device__ __forceinline__ void decode_pair(bool same, int t, int nj, int nk, float inv_div, int div,
                                            int &jr, int &kr)
{
    if (!same) {
        jr = (int)(((float)t + 0.5f) * inv_div);  // t / nk
        kr = t - jr * nk;
    } else {
        const int rect = nj * div;  // div = (n-1)/2 partners per row
        if (t < rect) {
            jr = (int)(((float)t + 0.5f) * inv_div);
            kr = jr + 1 + (t - jr * div);
            kr = kr >= nj ? kr - nj : kr;
        } else {  // n even: the n/2 diameters
            jr = t - rect;
            kr = jr + (nj >> 1);
        }
    }
}

// ---- cross-lane helpers (gfx950) -------------------------------------------------------------------
// v + (v shifted right by N lanes inside each 16-lane DPP row, zero fill)
template <int N>
__device__ __forceinline__ float row_shr_add(float v)
{
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + N, 0xF, 0xF, true));
}
// rows (16 lanes) of the result: [x.r0+x.r1, y.r0+y.r1, x.r2+x.r3, y.r2+y.r3]
__device__ __forceinline__ float sum16(float x, float y)
{
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// rows of the result: [x.r0+x.r2, x.r1+x.r3, y.r0+y.r2, y.r1+y.r3]
__device__ __forceinline__ float sum32(float x, float y)
{
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// sums inside a quad / an aligned group of 8 lanes on the VALU (DPP), no LDS round trip
template <int CTRL>
__device__ __forceinline__ float dpp_perm(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_sum(float v)
{
    v += dpp_perm<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_perm<0x4E>(v);   // quad_perm [2,3,0,1]
    return v;
}
__device__ __forceinline__ float oct_sum(float v)
{
    v = quad_sum(v);
    v += dpp_perm<0x141>(v);  // row_half_mirror: lane l <-> 7 - l of each 8-lane half row
    return v;
}

// branch-free (j,k) decode of pair t of a block (see decode_pair); `rect` = INT_MAX for rectangles
__device__ __forceinline__ void decode_pair2(bool same, int t, int nj, int div, float inv_div, int rect,
                                             int half, int &jr, int &kr)
{
    const int q = (int)(((float)t + 0.5f) * inv_div);
    const int rem = t - q * div;
    int k2 = same ? q + 1 + rem : rem;
    k2 = (same && k2 >= nj) ? k2 - nj : k2;
    const bool diam = t >= rect;
    jr = diam ? t - rect : q;
    kr = diam ? t - rect + half : k2;
}

// The same decode advanced incrementally: pair t = qd * div + rem of slot p moves on by 16 per step, so the quotient /
// remainder follow from two adds and a carry instead of a float division and an integer multiply per step.
// 16 / div for the wave-uniform group size div (0 for div == 0), on the scalar unit
__device__ __forceinline__ int quot16(int div)
{
    return div > 16 ? 0 : div > 8 ? 1 : div > 5 ? 2 : div == 5 ? 3 : div == 4 ? 4 : div == 3 ? 5 : div == 2 ? 8
         : div == 1 ? 16 : 0;
}
struct PairIter {
    int t, qd, rem;
};
__device__ __forceinline__ PairIter pair_begin(int t, int div, float inv_div)
{
    PairIter it;
    it.t = t;
    it.qd = (int)(((float)t + 0.5f) * inv_div);
    it.rem = t - it.qd * div;
    return it;
}
__device__ __forceinline__ void pair_next(PairIter &it, int div, int q16, int r16)
{
    it.t += 16;
    it.rem += r16;
    it.qd += q16;
    const bool carry = it.rem >= div;   // (div == 0: only diameter pairs, qd / rem are not used)
    it.rem -= carry ? div : 0;
    it.qd += carry ? 1 : 0;
}
// (j, k) of the current pair, clamped into the groups (slots past the last pair read valid entries and are masked)
__device__ __forceinline__ void pair_get(const PairIter &it, bool same, int nj, int rect, int half, int jmax,
                                         int kmax, int &jr, int &kr)
{
    int j = it.qd, k = it.rem;
    if (same) {   // wave-uniform
        int k2 = it.qd + 1 + it.rem;
        k2 -= k2 >= nj ? nj : 0;
        const bool diam = it.t >= rect;
        j = diam ? it.t - rect : it.qd;
        k = diam ? it.t - rect + half : k2;
    }
    jr = min(j, jmax);
    kr = min(k, kmax);
}

// per-atom header prefetched one iteration ahead: lanes 0..5 hold the meta words, lane 6 the species
struct AtomHdr {
    uint32_t start;
    int nA, nF, sp;
    uint64_t pkA, pkF;
};
__device__ __forceinline__ uint32_t hdr_load(const uint32_t *meta, const int32_t *species, int64_t i, bool ok)
{
    int lane = lane_id();
    // (opaque: the per-lane base address below is loop invariant; hoisted out of the atom loop of the backward kernel it is a
    // 64-bit register pair held -- or spilled: the reload then waits with vmcnt(0) for the prefetch just issued -- through
    // every phase, for the sake of four vector instructions per atom)
    asm volatile("" : "+v"(lane));
    // ONE load instruction: lanes 0..5 the meta words, lane 6 the species.  (Two predicated loads into the same register
    // make the second wait for the first -- s_waitcnt vmcnt(0) right behind the issue, i.e. no prefetch at all.)
    const uint32_t *src = lane == META_W ? reinterpret_cast<const uint32_t *>(species + i) : meta + (size_t)i * META_W + lane;
    uint32_t w = 0;
    if (ok && lane <= META_W) w = *src;
    return w;
}
__device__ __forceinline__ AtomHdr hdr_decode(uint32_t w)
{
    AtomHdr h;
    h.start = (uint32_t)__builtin_amdgcn_readlane((int)w, 0);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)w, 1);
    h.nA = (int)(c & 0xFFFFu);
    h.nF = (int)(c >> 16);
    h.pkA = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)w, 3) << 32) |
            (uint32_t)__builtin_amdgcn_readlane((int)w, 2);
    h.pkF = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)w, 5) << 32) |
            (uint32_t)__builtin_amdgcn_readlane((int)w, 4);
    h.sp = __builtin_amdgcn_readlane((int)w, 6);
    return h;
}

// ---------------------------------------------------------------------------------------------------
// JVP = true turns the kernel into the forward-mode derivative  out = (d aev / d r) . tang  (J t) for a coordinate-space
// direction tang [n_atoms][3]: the reference's cuaev double backward (csrc/aev.cu:1986-2015 and the is_double_backward
// kernel variants) -- the derivative of grad_coords = J^T grad_aev with respect to grad_aev, contracted with the
// gradient arriving at the forces.  Same lane layout and reductions; every neighbor additionally carries
// r' = u . d', u' = (d' - u r') / r, fc' r' (d' = t_j - t_i), every term is replaced by its directional derivative.
template <int NA, int NZ, bool JVP>
__global__ __launch_bounds__(FWD_WPB * WAVE, JVP ? 4 : 7) void k_aev_fwd(
    AevArgs a, const float *__restrict__ tab, int64_t lo, int64_t hi,
    const int32_t *__restrict__ species, const uint32_t *__restrict__ meta,
    const float4 *__restrict__ ent, float *__restrict__ aev, uint32_t *__restrict__ slab_mask,
    const float *__restrict__ tang)
{
    static_assert(NA % 4 == 0 && NZ % 4 == 0 && NA * NZ == 32, "angular tiling");
    constexpr int AQ = NA / 4, ZQ = NZ / 4;
    __shared__ float4 s_ang[FWD_WPB][MAXA];   // ux uy uz r
    __shared__ float s_afc[FWD_WPB][MAXA];    // fc(r, Rca)
    __shared__ float2 s_rad[FWD_WPB][MAXR];   // r, 0.25 fc(r, Rcr)            (JVP: r, 0.25 fc r')
    __shared__ float4 s_angd[JVP ? FWD_WPB : 1][JVP ? MAXA : 1];   // JVP: u'x u'y u'z r'
    __shared__ float s_afcd[JVP ? FWD_WPB : 1][JVP ? MAXA : 1];    // JVP: fc'(r, Rca) r'
    __shared__ float s_radb[JVP ? FWD_WPB : 1][JVP ? MAXR : 1];    // JVP: 0.25 fc'(r, Rcr) r'

    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();   // scalar LDS bases
    float4 *ang = s_ang[wib];
    float *afc = s_afc[wib];
    float2 *rad = s_rad[wib];
    float4 *angd = s_angd[JVP ? wib : 0];
    float *afcd = s_afcd[JVP ? wib : 0];
    float *radb = s_radb[JVP ? wib : 0];

    // per-lane constants
    const int rp = lane >> 3, rsq = lane & 7;  // radial: neighbor slot, shift pair
    const float shfR0 = tab[TAB_SHFR + rsq], shfR1 = tab[TAB_SHFR + rsq + 8];
    const int p = lane >> 2, q = lane & 3;  // angular: pair slot, quarter
    float shfA[AQ], cosZ[ZQ], sinZ[ZQ];
#pragma unroll
    for (int u = 0; u < AQ; ++u) shfA[u] = tab[TAB_SHFA + q + 4 * u];
#pragma unroll
    for (int v = 0; v < ZQ; ++v) {
        cosZ[v] = tab[TAB_COSZ + q + 4 * v];
        sinZ[v] = tab[TAB_SINZ + q + 4 * v];
    }
    // cutoffs through v_cos_f32 (argument in revolutions): cos(pi r / Rc) = cos(2 pi * r / (2 Rc))
    const float rev_rcr = 0.5f / a.Rcr, rev_rca = 0.5f / a.Rca;
    // output positions of the reduced angular block / radial species rows (see the reductions below)
    const int row = lane >> 4;
    const bool ang_writer = (lane & 15) >= 12;   // lanes 12..15 of every DPP row hold the totals
    const int ang_o0 = (NZ == 4) ? (q * 4 + row) : (q * 8 + row);            // value index `row`
    const int ang_o1 = (NZ == 4) ? ((q + 4) * 4 + row) : (q * 8 + 4 + row);  // value index 4 + `row`
    const bool rad_writer = (lane & 8) && row < 2;
    const int rad_o = row * 8 + rsq;

    const int64_t nw = (int64_t)gridDim.x * FWD_WPB;
    int64_t i = lo + xcd_block() * (int64_t)FWD_WPB + wib;
    // software pipeline over atoms: header of atom i+nw and the first 128 entries of atom i+nw are in
    // flight while atom i is being computed
    uint32_t hw = hdr_load(meta, species, i, i < hi);
    AtomHdr h = hdr_decode(hw);
    float4 e0 = make_float4(1.f, 0.f, 0.f, 0.f), e1 = e0;
    if (i < hi && h.sp >= 0) {
        if (lane < h.nA + h.nF) e0 = ent[h.start + lane];
        if (lane + WAVE < h.nA + h.nF) e1 = ent[h.start + lane + WAVE];
    }
    uint32_t hw_next = hdr_load(meta, species, i + nw, i + nw < hi);

    for (; i < hi; i += nw) {
        float *out = aev + (size_t)i * a.L;
        const int nA = h.nA, nR = h.nA + h.nF;
        const uint64_t pkA = h.pkA, pkF = h.pkF;
        const bool padding = h.sp < 0;
        const uint32_t start = h.start;

        // ---- per-neighbor precompute -> LDS ----
        if (!padding) {
            for (int c0 = 0; c0 < nR; c0 += WAVE) {
                const int e = c0 + lane;
                float4 d = c0 == 0 ? e0 : (c0 == WAVE ? e1 : make_float4(1.f, 0.f, 0.f, 0.f));
                if (c0 >= 2 * WAVE && e < nR) d = ent[start + e];
                if (!JVP && e < nR) {
                    const float r = sqrtf(d.x * d.x + d.y * d.y + d.z * d.z);
                    rad[e] = make_float2(r, a.smooth ? 0.25f * smooth_cutoff(r, 1.0f / a.Rcr).x
                                                     : 0.125f * __builtin_amdgcn_cosf(r * rev_rcr) + 0.125f);
                    if (e < nA) {
                        const float inv = 1.0f / r;
                        ang[e] = make_float4(d.x * inv, d.y * inv, d.z * inv, r);
                        afc[e] = a.smooth ? smooth_cutoff(r, 1.0f / a.Rca).x
                                          : 0.5f * __builtin_amdgcn_cosf(r * rev_rca) + 0.5f;
                    }
                }
                if (JVP && e < nR) {
                    const float r = sqrtf(d.x * d.x + d.y * d.y + d.z * d.z), inv = 1.0f / r;
                    const float ux = d.x * inv, uy = d.y * inv, uz = d.z * inv;
                    const float *tj = tang + 3 * (size_t)(__float_as_uint(d.w) & IDX_MASK), *ti = tang + 3 * (size_t)i;
                    const float tx = tj[0] - ti[0], ty = tj[1] - ti[1], tz = tj[2] - ti[2];
                    const float rd = ux * tx + uy * ty + uz * tz;
                    const float pi_rcr = PI_F / a.Rcr, pi_rca = PI_F / a.Rca;
                    float2 cr = a.smooth ? smooth_cutoff(r, 1.0f / a.Rcr)
                                         : make_float2(0.5f * __builtin_amdgcn_cosf(r * rev_rcr) + 0.5f,
                                                       -0.5f * pi_rcr * __builtin_amdgcn_sinf(r * rev_rcr));
                    rad[e] = make_float2(r, 0.25f * cr.x * rd);
                    radb[e] = 0.25f * cr.y * rd;
                    if (e < nA) {
                        const float2 ca = a.smooth ? smooth_cutoff(r, 1.0f / a.Rca)
                                                   : make_float2(0.5f * __builtin_amdgcn_cosf(r * rev_rca) + 0.5f,
                                                                 -0.5f * pi_rca * __builtin_amdgcn_sinf(r * rev_rca));
                        ang[e] = make_float4(ux, uy, uz, r);
                        angd[e] = make_float4((tx - ux * rd) * inv, (ty - uy * rd) * inv, (tz - uz * rd) * inv, rd);
                        afc[e] = ca.x;
                        afcd[e] = ca.y * rd;
                    }
                }
            }
        }
        // ---- prefetch the next atom ----
        h = hdr_decode(hw_next);
        {
            const int64_t in = i + nw;
            e0 = make_float4(1.f, 0.f, 0.f, 0.f);
            e1 = e0;
            if (in < hi && h.sp >= 0) {
                if (lane < h.nA + h.nF) e0 = ent[h.start + lane];
                if (lane + WAVE < h.nA + h.nF) e1 = ent[h.start + lane + WAVE];
            }
            hw_next = hdr_load(meta, species, in + nw, in + nw < hi);
        }
        if (padding) {
            float4 *out4 = reinterpret_cast<float4 *>(out);
            for (int f = lane; f < (a.L >> 2); f += WAVE) out4[f] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (slab_mask && lane == 0) slab_mask[i] = 0u;
            continue;
        }
        wave_sync();
        uint32_t smask = 0u;   // 32-wide slabs of this row that are not identically zero (include/anihip.h)
        const int rslabs = (a.S + 1) >> 1;

        // ---- radial ----
        {
            int oA = 0, oF = 0;
            for (int t = 0; t < a.S; ++t) {
                const int cA = cnt_of(pkA, t), cF = cnt_of(pkF, t), n = cA + cF;
                float acc0 = 0.f, acc1 = 0.f;
                if (n == 0) {   // no neighbor of this species: zero block, no reduction
                    if (rad_writer) out[t * 16 + rad_o] = 0.f;
                    continue;
                }
                smask |= 1u << (t >> 1);
                for (int b = 0; b < n; b += 8) {
                    const int idx = b + rp;
                    const bool v = idx < n;
                    int e = idx < cA ? oA + idx : nA + oF + (idx - cA);
                    e = v ? e : 0;
                    const float2 rf = rad[e];
                    const float f = v ? rf.y : 0.f;
                    const float d0 = rf.x - shfR0, d1 = rf.x - shfR1;
                    if (!JVP) {
                        acc0 += __builtin_amdgcn_exp2f(a.kR * d0 * d0) * f;
                        acc1 += __builtin_amdgcn_exp2f(a.kR * d1 * d1) * f;
                    } else {   // d/dt [0.25 exp(-eta d^2) fc] = exp(..) (0.25 fc' r' - 2 eta d 0.25 fc r')
                        const float fb = v ? radb[e] : 0.f;
                        acc0 += __builtin_amdgcn_exp2f(a.kR * d0 * d0) * (fb - 2.0f * a.EtaR * d0 * f);
                        acc1 += __builtin_amdgcn_exp2f(a.kR * d1 * d1) * (fb - 2.0f * a.EtaR * d1 * f);
                    }
                }
                // 8 slots -> 1: inside the DPP row, then across rows.  Row 0 ends with the totals of
                // acc0, row 1 with those of acc1 (lanes 8..15 = shift pair rsq).
                acc0 = row_shr_add<8>(acc0);
                acc1 = row_shr_add<8>(acc1);
                float x = sum16(acc0, acc1);
                x = sum32(x, x);
                if (rad_writer) out[t * 16 + rad_o] = x;
                oA += cA;
                oF += cF;
            }
        }

        // ---- angular ----
        {
            int P = 0, oj = 0;
            for (int tj = 0; tj < a.S; ++tj) {
                const int nj = cnt_of(pkA, tj);
                int ok = oj;
                for (int tk = tj; tk < a.S; ++tk, ++P) {
                    const int nk = cnt_of(pkA, tk);
                    const bool same = (tk == tj);
                    const int np = same ? (nj * (nj - 1)) >> 1 : nj * nk;
                    float *blk = out + a.radlen + P * 32;
                    if (np == 0) {
                        if (lane < 32) blk[lane] = 0.f;
                        ok += nk;
                        continue;
                    }
                    smask |= 1u << ((rslabs + P) & 31);
                    const int div = same ? ((nj - 1) >> 1) : nk;
                    const float inv_div = div > 0 ? 1.0f / (float)div : 0.f;
                    const int rect = same ? nj * div : 0x7FFFFFFF;
                    const int half = nj >> 1;
                    float acc[AQ][NZ];
#pragma unroll
                    for (int u = 0; u < AQ; ++u)
#pragma unroll
                        for (int z = 0; z < NZ; ++z) acc[u][z] = 0.f;
                    // software-pipelined over the steps: LDS reads of step s+1 are issued before the
                    // arithmetic of step s
                    int jr, kr;
                    const int q16 = quot16(div), r16 = 16 - q16 * div;
                    const int jmax = nj - 1, kmax = (same ? nj : nk) - 1;
                    PairIter it = pair_begin(p, div, inv_div);
                    pair_get(it, same, nj, rect, half, jmax, kmax, jr, kr);
                    if (!JVP) {
                        float4 J = ang[oj + jr], K = ang[ok + kr];
                        float fj = afc[oj + jr], fk = afc[ok + kr];
                        for (int t0 = 0; t0 < np; t0 += 16) {
                            const bool v = it.t < np;
                            const float4 Jc = J, Kc = K;
                            const float fcc = v ? 2.0f * fj * fk : 0.f;
                            if (t0 + 16 < np) {
                                pair_next(it, div, q16, r16);
                                pair_get(it, same, nj, rect, half, jmax, kmax, jr, kr);
                                J = ang[oj + jr];
                                K = ang[ok + kr];
                                fj = afc[oj + jr];
                                fk = afc[ok + kr];
                            }
                            const float c = Jc.x * Kc.x + Jc.y * Kc.y + Jc.z * Kc.z;
                            const float ct = 0.95f * c;
                            const float st = __builtin_amdgcn_sqrtf(fmaxf(1.0f - ct * ct, 0.f));
                            const float rm = 0.5f * (Jc.w + Kc.w);
                            float f1t[ZQ], f2[AQ];
    #pragma unroll
                            for (int vz = 0; vz < ZQ; ++vz) {
                                const float cz = ct * cosZ[vz] + st * sinZ[vz];
                                const float hh = fmaxf(0.5f + 0.5f * cz, 0.f);
                                f1t[vz] = __builtin_amdgcn_exp2f(a.Zeta * __builtin_amdgcn_logf(hh)) * fcc;
                            }
    #pragma unroll
                            for (int u = 0; u < AQ; ++u) {
                                const float d = rm - shfA[u];
                                f2[u] = __builtin_amdgcn_exp2f(a.kA * d * d);
                            }
    #pragma unroll
                            for (int z = 0; z < NZ; ++z) {
                                const float f1 = quad_bcast_rt(f1t[z >> 2], z & 3);
    #pragma unroll
                                for (int u = 0; u < AQ; ++u) acc[u][z] += f2[u] * f1;
                            }
                        }
                    } else {
                        for (int t0 = 0; t0 < np; t0 += 16) {
                            const bool v = it.t < np;
                            const float4 Jc = ang[oj + jr], Kc = ang[ok + kr], Jd = angd[oj + jr], Kd = angd[ok + kr];
                            const float fj = afc[oj + jr], fk = afc[ok + kr], fjd = afcd[oj + jr], fkd = afcd[ok + kr];
                            pair_next(it, div, q16, r16);
                            pair_get(it, same, nj, rect, half, jmax, kmax, jr, kr);
                            const float fcc = v ? 2.0f * fj * fk : 0.f;                    // (the 2 of f1 = 2 h^zeta)
                            const float fccd = v ? 2.0f * (fjd * fk + fj * fkd) : 0.f;
                            const float c = Jc.x * Kc.x + Jc.y * Kc.y + Jc.z * Kc.z;
                            const float cd = Jd.x * Kc.x + Jd.y * Kc.y + Jd.z * Kc.z + Jc.x * Kd.x + Jc.y * Kd.y + Jc.z * Kd.z;
                            const float ct = 0.95f * c;
                            const float st2 = fmaxf(1.0f - ct * ct, 1e-12f);
                            const float rst = __builtin_amdgcn_rsqf(st2);
                            const float st = st2 * rst;
                            const float thd = -0.95f * cd * rst;                           // d theta / dt
                            const float rm = 0.5f * (Jc.w + Kc.w), rmd = 0.5f * (Jd.w + Kd.w);
                            float At[ZQ], Bt[ZQ], f2[AQ], df2[AQ];
#pragma unroll
                            for (int vz = 0; vz < ZQ; ++vz) {
                                const float cz = ct * cosZ[vz] + st * sinZ[vz];   // cos(theta - ShfZ)
                                const float sz = st * cosZ[vz] - ct * sinZ[vz];   // sin(theta - ShfZ)
                                const float hh = fmaxf(0.5f + 0.5f * cz, 0.f);
                                const float p1 = __builtin_amdgcn_exp2f((a.Zeta - 1.0f) * __builtin_amdgcn_logf(hh));
                                const float f1 = hh * p1, df1 = -0.5f * a.Zeta * p1 * sz;  // h^zeta and its theta derivative
                                At[vz] = df1 * thd * fcc + f1 * fccd;
                                Bt[vz] = f1 * rmd * fcc;
                            }
#pragma unroll
                            for (int u = 0; u < AQ; ++u) {
                                const float d = rm - shfA[u];
                                f2[u] = __builtin_amdgcn_exp2f(a.kA * d * d);
                                df2[u] = -2.0f * a.EtaA * d * f2[u];
                            }
#pragma unroll
                            for (int z = 0; z < NZ; ++z) {
                                const float Az = quad_bcast_rt(At[z >> 2], z & 3), Bz = quad_bcast_rt(Bt[z >> 2], z & 3);
#pragma unroll
                                for (int u = 0; u < AQ; ++u) acc[u][z] += f2[u] * Az + df2[u] * Bz;
                            }
                        }
                    }
                    // 16 slots -> 1.  Inside each DPP row (4 slots): two shifted adds leave the row
                    // totals in lanes 12..15 (= quarter q).  Across the 4 rows: permlane swaps reduce two
                    // values per instruction; row r ends with the total of value r (x0) / 4+r (x1).
                    float v8[8];
#pragma unroll
                    for (int u = 0; u < AQ; ++u)
#pragma unroll
                        for (int z = 0; z < NZ; ++z) v8[u * NZ + z] = row_shr_add<8>(row_shr_add<4>(acc[u][z]));
                    const float x0 = sum32(sum16(v8[0], v8[1]), sum16(v8[2], v8[3]));
                    const float x1 = sum32(sum16(v8[4], v8[5]), sum16(v8[6], v8[7]));
                    if (ang_writer) {
                        blk[ang_o0] = x0;
                        blk[ang_o1] = x1;
                    }
                    ok += nk;
                }
                oj += nj;
            }
        }
        if (slab_mask && lane == 0) slab_mask[i] = smask;
        wave_sync();
    }
}

typedef float v2f __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------------
// k_aev_fwd3 (round 3): ONE lane per (j, k) pair and a FLAT slot assignment over all species-pair blocks of the atom.
//   * k_aev_fwd2 walks the species-pair blocks one after the other, 32 two-lane pair slots at a time: on the water box
//     an atom has 53 + 58 + 12 pairs (HH, HO, OO) in 2 + 2 + 1 = 5 iterations of 91 instructions -- 77 % of the slots
//     hold a pair -- and every block pays its own 32-slot transpose-reduce.  Both lanes of a slot repeat the pair's
//     geometry (index decode, two LDS reads, dot product, sqrt).
//   * here a lane owns a pair slot for the whole atom: block b of the atom gets ns_b = pad4(ceil(np_b / I)) of the 64
//     slots (I = iterations, the smallest for which the blocks fit; a slot walks I consecutive pairs of ITS block), the
//     lane evaluates all NA Gaussians and all NZ angle factors of its pair -- every constant is wave-uniform now and
//     lives in scalar registers -- and keeps the whole NA x NZ block of sums (32 registers).  The same atom takes
//     ceil(123 / 64) = 2 iterations, 96 % of the slots busy, the geometry once per pair.
//   * the 64 x 32 sums leave through LDS as a SEGMENTED reduction, 16 values per round: every lane writes its row
//     (stride 20 floats: conflict-free 16-B accesses), lane group g = lane / 4 adds the four rows 4 g .. 4 g + 3 (blocks
//     are padded to multiples of four slots, so a group never straddles two blocks), a two-step segmented scan over the
//     four groups of a DPP row (v += [same block] row_shr:4 / :8) and a three-entry LDS table of the rows' trailing sums
//     carry the partial sums to the LAST group of every block, whose four lanes store 64 B of the AEV row.  No atomics,
//     fixed order, any number of blocks for the price of one reduction.
//   * atoms whose blocks do not fit 64 slots (many species with a few pairs each) are done in batches of blocks.
// (k_aev_fwd2, the round-2 kernel described in the first bullet, is gone: 4.28 ms against 4.06 ms on the 2.34 M-atom box.)
#ifndef ANIHIP_FWD3_WAVES
#define ANIHIP_FWD3_WAVES 4
#endif
#ifndef ANIHIP_FWD3_REC
#define ANIHIP_FWD3_REC 1   // 0: never use the Gaussian recurrence (development A/B)
#endif
// sum over the wave of small non-negative integers held by the lanes: scan inside the DPP rows, four readlanes
__device__ __forceinline__ int wave_isum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
    return __builtin_amdgcn_readlane(v, 15) + __builtin_amdgcn_readlane(v, 31) + __builtin_amdgcn_readlane(v, 47) +
           __builtin_amdgcn_readlane(v, 63);
}
// slots of a block of np pairs walked in I iterations: ceil(np / I) rounded up to a multiple of four (np <= 8128, I <= 127)
__device__ __forceinline__ int block_slots(int np, int I, float inv_I)
{
    int c = (int)(((float)np + 0.5f) * inv_I);   // floor(np / I) (the half keeps the product off the integers)
    c += (__mul24(c, I) < np) ? 1 : 0;   // (24-bit multiply: full rate; v_mul_lo_u32 issues at a quarter of it)
    return (c + 3) & ~3;
}

#ifdef ANIHIP_TRACE
// development: per-phase shader-clock sums of wave 0 of every block (s_memtime stamps), read back by anihip_dev_trace_read
__device__ unsigned long long g_fwd3_trace[2048][10];
#define TR_STAMP(k_)                                                       \
    {                                                                      \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();      \
        if (wib == 0) tr_sum[k_] += now_ - tr_last;                        \
        tr_last = now_;                                                    \
    }
#else
#define TR_STAMP(k_)
#endif
template <int NA, int NZ, bool REC>
__global__ __launch_bounds__(FWD3_WPB * WAVE, ANIHIP_FWD3_WAVES) void k_aev_fwd3(
    AevArgs a, const float *__restrict__ tab, int64_t lo, int64_t hi,
    const int32_t *__restrict__ species, const uint32_t *__restrict__ meta,
    const float4 *__restrict__ ent, float *__restrict__ aev, uint32_t *__restrict__ slab_mask,
    const uint32_t *__restrict__ prev_mask)
{
    static_assert(NA % 4 == 0 && NZ % 4 == 0 && NA * NZ == 32, "angular tiling");
    constexpr int ZP = NZ / 2;          // packed pairs of angle shifts
    constexpr int RS = 20;              // floats per row of the reduction buffer: 16 values + 4 (bank spread)
    constexpr int XOFF = 64 * RS;       // trailing sums of DPP rows 0..2: 3 x 16 floats
    __shared__ float4 s_ang[FWD3_WPB][MAXA + 1];   // ux uy uz, 0.5 qA r          (entry nA = dummy)
    __shared__ float s_lfc[FWD3_WPB][MAXA + 2];    // log2 fc(r, Rca) + 0.5       (dummy: -inf)
    __shared__ __attribute__((aligned(16))) float s_red[FWD3_WPB][XOFF + 48];   // reduction rows (radial terms, then pair sums)
    __shared__ __attribute__((aligned(16))) float s_rst[FWD3_WPB][MAX_S * 16];  // radial part of the row until it is stored

    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
    float4 *ang = s_ang[wib];
    float *lfc = s_lfc[wib];
    float *red = s_red[wib];
    float *rst = s_rst[wib];

    const float qR = a.qR, qA = a.qA;
    float shfAq[NA], cZ[NZ], sZ[NZ];           // wave-uniform: scalar registers
#pragma unroll
    for (int u = 0; u < NA; ++u) shfAq[u] = tab[TAB_SHFAQ + u];
#pragma unroll
    for (int z = 0; z < NZ; ++z) {   // h(theta) = 0.5 + 0.5 cos(theta - ShfZ)
        cZ[z] = tab[TAB_COSZH + z];
        sZ[z] = tab[TAB_SINZH + z];
    }
    // Gaussian recurrence of the pair loop (REC: equally spaced shifts, checked by anihip_aev_table_pack)
    const float gD = shfAq[1] - shfAq[0], gq = __builtin_amdgcn_exp2f(-2.0f * gD * gD);
    const float rev_rcr = 0.5f / a.Rcr, rev_rca = 0.5f / a.Rca;   // v_cos_f32 takes revolutions
    const int row = lane >> 4;
    float2 *rad = reinterpret_cast<float2 *>(red);   // qR r, 0.25 fc(r, Rcr): dead before the first reduction
    const int rp = lane >> 3, rsq = lane & 7;        // radial: neighbor slot, shift pair
    const float shfR0 = tab[TAB_SHFRQ + rsq], shfR1 = tab[TAB_SHFRQ + rsq + 8];
    const bool rad_writer = (lane & 8) && row < 2;
    const int rad_o = row * 8 + rsq;
    const int w4 = lane & 3;                      // reduction: value quad of a round
    const bool row_last = (lane & 15) >= 12;      // last lane group of its DPP row

    // "needed block" bookkeeping: bit t < 7 radial block of species t, bit 7 + P angular block of species pair P;
    // lane b < 35 decides bit b and (b >= 7) is the BLOCK LANE of species pair P = b - 7
    int nd_tj = 7, nd_tk = 7;
    if (lane < 7) {
        nd_tj = nd_tk = lane;
    } else {
        int P = lane - 7, tj = 0;
        while (tj < a.S && P >= a.S - tj) { P -= a.S - tj; ++tj; }
        if (tj < a.S) { nd_tj = tj; nd_tk = tj + P; }
    }
    const bool blk_same = nd_tj == nd_tk;
    const int L4 = a.L >> 2, R4 = a.radlen >> 2;
    const int rslabs = (a.S + 1) >> 1;

    // atoms from the workgroup's queue (AtomQueue); the first three positions of a wave are its own
    __shared__ uint32_t s_queue;
    if (threadIdx.x == 0) s_queue = 3u * FWD3_WPB;
    __syncthreads();
    AtomQueue q(&s_queue, lo, xcd_block(), (int)gridDim.x, lane);
    int64_t i = q.atom(wib), i1 = q.atom(FWD3_WPB + wib), i2 = q.atom(2 * FWD3_WPB + wib);
    uint32_t hw = hdr_load(meta, species, i, i < hi);
    AtomHdr hd = hdr_decode(hw);
    const float4 dummy4 = make_float4(1.f, 0.f, 0.f, 0.f);
    // the first 128 entries of the row travel one atom ahead.  Every lane loads (index clamped into the row): a load under
    // a per-lane condition merges with the old value afterwards, and that move waits for the load on the spot
    float4 e0 = dummy4, e1 = dummy4;
    if (i < hi && hd.sp >= 0 && hd.nA + hd.nF > 0) {
        const int nl = hd.nA + hd.nF - 1;
        e0 = ent[hd.start + min(lane, nl)];
        e1 = ent[hd.start + min(lane + WAVE, nl)];
    }
    uint32_t hw_next = hdr_load(meta, species, i1, i1 < hi);
    // rows updated in place (a.update): what the previous call left in this row, one atom ahead like the header (wave-uniform
    // address: a scalar load); otherwise every slab counts as dirty and the whole row is written
    // (prev_mask is a buffer of its own, never written here: wave-uniform reads of it are scalar loads)
    uint32_t pm = (a.update && i < hi) ? prev_mask[i] : 0xFFFFFFFFu;
    uint32_t pm_next = (a.update && i1 < hi) ? prev_mask[i1] : 0xFFFFFFFFu;
    // Memory order of an atom: [loads for the NEXT atom] ... arithmetic ... [wait for those loads] [ALL stores of this atom].
    // Vector-memory operations retire in order and the compiler waits with vmcnt(0) for whatever it cannot count, so a
    // load consumed after stores were issued drains those stores first (HBM write latency, once per atom and wave).  With
    // the stores last and the prefetched registers "used" right before them, nothing in the loop ever waits for a store:
    // an atom's 4 KB row drains while the next atom is computed.
#define ANIHIP_FWD3_ARRIVED()                                                                                         \
    asm volatile("" ::"v"(e0.x), "v"(e0.y), "v"(e0.z), "v"(e0.w), "v"(e1.x), "v"(e1.y), "v"(e1.z), "v"(e1.w), "v"(hw_next) \
                 : "memory")
    ANIHIP_FWD3_ARRIVED();

#ifdef ANIHIP_TRACE
    unsigned long long tr_sum[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tr_last = __builtin_amdgcn_s_memtime();
#endif
    for (; i < hi; i = i1, i1 = i2, i2 = q.granted()) {
        TR_STAMP(9)   // loop overhead / tail of the previous atom
        float *out = aev + (size_t)i * a.L;
        const int nA = hd.nA, nR = hd.nA + hd.nF;
        const uint64_t pkA = hd.pkA, pkF = hd.pkF;
        const bool padding = hd.sp < 0;
        const uint32_t start = hd.start;

        // ---- per-neighbor precompute -> LDS ----
        uint64_t need = 0ull;
        const int cj = (int)((pkA >> (8 * nd_tj)) & 255u), ck = (int)((pkA >> (8 * nd_tk)) & 255u);
        if (!padding) {
            const int cf = (int)((pkF >> (8 * nd_tj)) & 255u);
            const bool nd = lane < 7 ? (cj + cf > 0) : (blk_same ? cj >= 2 : (cj >= 1 && ck >= 1));
            need = __ballot(nd && lane < 35);
            // The radial list is kept GROUPED BY SPECIES, every group padded to a multiple of 8 entries with fc = 0 (the row
            // itself is sorted {angular, far} x species): the radial loop below then walks 8 consecutive entries per step
            // with no index arithmetic and no validity select (17 -> 9 vector instructions per step, 8 steps per water atom).
            // Entry e of the row goes to e + shift, the shift of its {class, species} segment: two compare-selects per
            // present species (the last segment whose start is <= e wins; wave-uniform bounds, scalar loop).
            const uint64_t prA2 = pkA * 0x0101010101010100ull, prF2 = pkF * 0x0101010101010100ull;
            int rad_total = 0;
            auto rad_pos = [&](int e) {
                int shA = 0, shF = 0, base = 0;
                for (uint32_t m_ = (uint32_t)need & 0x7Fu; m_; m_ &= m_ - 1) {
                    const int t = __builtin_ctz(m_);
                    const int cA = cnt_of(pkA, t), cF = cnt_of(pkF, t);
                    const int oA = cnt_of(prA2, t), oF = nA + cnt_of(prF2, t);
                    shA = e >= oA ? base - oA : shA;
                    shF = e >= oF ? base + cA - oF : shF;
                    base += (cA + cF + 7) & ~7;
                }
                rad_total = base;
                return e + (e < nA ? shA : shF);
            };
            const int pos0 = rad_pos(lane);
            for (int k = 0; k < rad_total; k += WAVE) rad[k + lane] = make_float2(0.f, 0.f);   // (<= 256 + 49 of 664 entries)
            for (int c0 = 0; c0 < nR; c0 += WAVE) {
                const int e = c0 + lane;
                float4 d = c0 == 0 ? e0 : (c0 == WAVE ? e1 : dummy4);
                if (c0 >= 2 * WAVE && e < nR) d = ent[start + e];   // (> 128 neighbors: rare, waits on the spot)
                const int pos = c0 == 0 ? pos0 : rad_pos(e);
                if (e < nR) {
                    const float r2 = d.x * d.x + d.y * d.y + d.z * d.z;
                    const float inv = __builtin_amdgcn_rsqf(r2);
                    const float r = r2 * inv;
                    const float fcr = a.smooth ? 0.25f * smooth_cutoff(r, 1.0f / a.Rcr).x
                                               : 0.125f * __builtin_amdgcn_cosf(r * rev_rcr) + 0.125f;
                    rad[pos] = make_float2(qR * r, fcr);
                    if (e < nA) {
                        ang[e] = make_float4(d.x * inv, d.y * inv, d.z * inv, 0.5f * qA * r);
                        float lf;
                        if (a.smooth) {   // log2 exp(1 - 1 / m)
                            const float q_ = r / a.Rca;
                            lf = (1.0f - 1.0f / fmaxf(SMOOTH_EPS, (1.0f - q_) * (1.0f + q_))) * LOG2E;
                        } else {
                            lf = __builtin_amdgcn_logf(0.5f * __builtin_amdgcn_cosf(r * rev_rca) + 0.5f);
                        }
                        lfc[e] = lf + 0.5f;
                    }
                }
                if (c0 == 0 && lane == 0) {
                    ang[nA] = make_float4(0.f, 0.f, 0.f, 0.f);
                    lfc[nA] = -__builtin_inff();
                }
            }
            wave_sync();
            TR_STAMP(0)   // neighbor terms
            // radial, lane = (8 neighbor slots) x (8 shift pairs), one species group of the list after the other
            const float2 *rgrp = rad + rp;
            for (uint32_t rm_ = (uint32_t)need & 0x7Fu; rm_; rm_ &= rm_ - 1) {
                const int t = __builtin_ctz(rm_);
                const int n8 = (cnt_of(pkA, t) + cnt_of(pkF, t) + 7) & ~7;
                float acc0 = 0.f, acc1 = 0.f;
                for (int b_ = 0; b_ < n8; b_ += 8) {
                    const float2 rf = rgrp[b_];
                    const float d0 = rf.x - shfR0, d1 = rf.x - shfR1;
                    acc0 += __builtin_amdgcn_exp2f(-d0 * d0) * rf.y;
                    acc1 += __builtin_amdgcn_exp2f(-d1 * d1) * rf.y;
                }
                rgrp += n8;
                acc0 = row_shr_add<8>(acc0);
                acc1 = row_shr_add<8>(acc1);
                float x = sum16(acc0, acc1);
                x = sum32(x, x);
                if (rad_writer) rst[t * 16 + rad_o] = x;
            }
            wave_sync();   // (the radial list is dead from here on: the reduction rows overwrite it)
        }
        TR_STAMP(1)   // radial sums
        // ---- prefetch the next atom ----
        hd = hdr_decode(hw_next);
        q.request();   // (the position of the atom after i2: granted by the end of this atom)
        {
            if (i1 < hi && hd.sp >= 0 && hd.nA + hd.nF > 0) {   // (wave-uniform; else the registers keep stale, unused values)
                const int nl = hd.nA + hd.nF - 1;
                e0 = ent[hd.start + min(lane, nl)];
                e1 = ent[hd.start + min(lane + WAVE, nl)];
            }
            hw_next = hdr_load(meta, species, i2, i2 < hi);
        }
        const uint32_t prev_m = pm;
        pm = pm_next;
        pm_next = (a.update && i2 < hi) ? prev_mask[i2] : 0xFFFFFFFFu;
        // results that wait for the end of the atom: radial part in LDS (rst), the angular blocks of the last batch in the
        // neighbor table's LDS (dead by then: ang[lane] / ang[64 + lane] of the lanes flagged hold_last, destination hold_dst)
        bool hold_last = false;
        float *hold_dst = out;
        if (!padding) {
        const uint64_t prA = pkA * 0x0101010101010100ull;

        TR_STAMP(2)   // prefetch issue
        // ---- angular ----
        uint32_t remaining = (uint32_t)(need >> 7);
        if (remaining) {
            // block lanes: pairs of the block, where its two groups start in the row
            const bool is_blk = (need >> lane) & 1ull && lane >= 7;
            const int np_b = is_blk ? (blk_same ? (cj * (cj - 1)) >> 1 : cj * ck) : 0;
            const int oj_b = (int)((prA >> (8 * nd_tj)) & 255u), ok_b = (int)((prA >> (8 * nd_tk)) & 255u);
            const int word_b = oj_b | (ok_b << 8) | (cj << 16) | (ck << 24);
            const int T = (nA * (nA - 1)) >> 1;
            int I = (T + 63) >> 6;   // (>= 1: a block is flagged)
            float inv_I = __builtin_amdgcn_rcpf((float)I);
            int slots_b = block_slots(np_b, I, inv_I);
            if (wave_isum(slots_b) > 64) {   // padding pushed the blocks over the wave: one more iteration if that fits
                const float inv_I1 = __builtin_amdgcn_rcpf((float)(I + 1));
                const int s1 = block_slots(np_b, I + 1, inv_I1);
                if (wave_isum(s1) <= 64) {
                    I += 1;
                    inv_I = inv_I1;
                    slots_b = s1;
                }   // (else: batches of blocks, I iterations each)
            }
            while (remaining) {
                // -- deal the slots of this batch: blocks in ascending order while they fit --
                int s_run = 0, myblk = 0, mys0 = 0;   // myblk: block LANE (7 + P), 0 = none
                while (remaining) {
                    const int P = __builtin_ctz(remaining);
                    const int ns = __builtin_amdgcn_readlane(slots_b, 7 + P);
                    if (s_run + ns > 64) break;
                    const bool mine = lane >= s_run;
                    myblk = mine ? 7 + P : myblk;
                    mys0 = mine ? s_run : mys0;
                    s_run += ns;
                    remaining &= remaining - 1;
                }
                myblk = lane < s_run ? myblk : 0;
                TR_STAMP(3)   // slot dealing
                // -- the lane's block: group offsets and sizes, pair iterator --
                const int word = __builtin_amdgcn_ds_bpermute(myblk << 2, word_b);
                const bool same = __builtin_amdgcn_ds_bpermute(myblk << 2, (int)blk_same) != 0;
                const int oj = word & 255, ok = (word >> 8) & 255, nj = (word >> 16) & 255, nk = (word >> 24) & 255;
                const int np = myblk ? (same ? (nj * (nj - 1)) >> 1 : nj * nk) : 0;
                const int div = same ? ((nj - 1) >> 1) : nk;
                const float inv_div = div > 0 ? __builtin_amdgcn_rcpf((float)div) : 0.f;
                const int rect = same ? nj * div : 0x7FFFFFFF;
                const int half = nj >> 1;
                int t = __mul24(lane - mys0, I);
                int qd = (int)(((float)t + 0.5f) * inv_div);
                int rem = t - __mul24(qd, div);
                // (no zero fill of the 32 sums: the first pair of a slot writes them, the others add -- the zero fill was 64
                // moves per batch, the compiler cleared the registers once for the loop and once for "no iteration")
                v2f acc[NA][ZP];
                TR_STAMP(4)   // pair iterator setup
                auto pair_step = [&](auto first_) {
                    // (j, k) of pair t inside the two groups; slots past the last pair read the dummy neighbor
                    int k2 = qd + 1 + rem;
                    k2 = (int)min((uint32_t)k2, (uint32_t)(k2 - nj));   // k2 >= nj ? k2 - nj : k2
                    const bool diam = t >= rect;
                    const int jr = (same && diam) ? t - rect : qd;
                    const int kr = same ? (diam ? t - rect + half : k2) : rem;
                    const bool v = t < np;
                    const int ej = v ? oj + jr : nA, ek = v ? ok + kr : nA;
                    const float4 J = ang[ej], K = ang[ek];
                    const float lf = lfc[ej] + lfc[ek];
                    // advance to the next pair of the block
                    t += 1;
                    rem += 1;
                    {
                        const bool carry = rem >= div;
                        rem = carry ? 0 : rem;
                        qd += carry ? 1 : 0;
                    }
                    const float c = J.x * K.x + J.y * K.y + J.z * K.z;
                    const float ct = 0.95f * c;
                    const float st = __builtin_amdgcn_sqrtf(fmaxf(1.0f - ct * ct, 0.f));
                    const float sr = J.w + K.w;
                    v2f f1[ZP];
#pragma unroll
                    for (int vp = 0; vp < ZP; ++vp) {
                        const float h0 = 0.5f + ct * cZ[2 * vp] + st * sZ[2 * vp];
                        const float h1 = 0.5f + ct * cZ[2 * vp + 1] + st * sZ[2 * vp + 1];
                        f1[vp] = (v2f){__builtin_amdgcn_exp2f(a.Zeta * __builtin_amdgcn_logf(__builtin_fabsf(h0)) + lf),
                                       __builtin_amdgcn_exp2f(a.Zeta * __builtin_amdgcn_logf(__builtin_fabsf(h1)) + lf)};
                    }
                    float f2[NA];
                    if (REC) {
                        // equally spaced shifts s_u = s_c + (u - c) D:  g_u = exp2(-(x - (u - c) D)^2),  x = sr - s_c, and
                        // g_(u+1) / g_u = exp2(2 D x_u - D^2) =: r_u  with  r_(u+1) = r_u exp2(-2 D^2): three exponentials
                        // and 2 (NA - 2) + 1 multiplications instead of NA exponentials (a product of at most NA / 2
                        // rounded factors: 3e-7 relative)
                        constexpr int C = NA / 2 - 1;
                        const float x = sr - shfAq[C];
                        // (clamped like the backward's: inside the cutoff |e| < 100 by anihip_aev_table_pack's check)
                        const float e = __builtin_amdgcn_fmed3f((2.0f * gD) * x - gD * gD, -100.0f, 100.0f);
                        f2[C] = __builtin_amdgcn_exp2f(-x * x);
                        float ru = __builtin_amdgcn_exp2f(e), rd = __builtin_amdgcn_exp2f(-e);
#pragma unroll
                        for (int u = C + 1; u < NA; ++u) {
                            f2[u] = f2[u - 1] * ru;
                            ru *= gq;
                        }
#pragma unroll
                        for (int u = C - 1; u >= 0; --u) {
                            rd *= gq;
                            f2[u] = f2[u + 1] * rd;
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < NA; ++u) {
                            const float dd = sr - shfAq[u];
                            f2[u] = __builtin_amdgcn_exp2f(-dd * dd);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < NA; ++u)
#pragma unroll
                        for (int vp = 0; vp < ZP; ++vp) {
                            if constexpr (decltype(first_)::value) acc[u][vp] = (v2f){f2[u], f2[u]} * f1[vp];
                            else acc[u][vp] += (v2f){f2[u], f2[u]} * f1[vp];
                        }
                };
                pair_step(std::true_type{});   // (I >= 1: a block is flagged)
                for (int it = 1; it < I; ++it) pair_step(std::false_type{});
                TR_STAMP(5)   // pair loop
                // -- segmented reduction of the 64 x 32 sums, 16 values per round --
                // same-block predicates between lane groups (all four slots of a group belong to one block)
                const int b_m4 = __builtin_amdgcn_update_dpp(0, myblk, 0x114, 0xF, 0xF, true);   // group - 1 (0 at the row start)
                const int b_m8 = __builtin_amdgcn_update_dpp(0, myblk, 0x118, 0xF, 0xF, true);   // group - 2
                const int b_p4 = __builtin_amdgcn_update_dpp(0, myblk, 0x104, 0xF, 0xF, true);   // group + 1 (0 at the row end)
                const float m1 = (myblk && b_m4 == myblk) ? 1.f : 0.f, m2 = (myblk && b_m8 == myblk) ? 1.f : 0.f;
                const int e0b = __builtin_amdgcn_readlane(myblk, 15), e1b = __builtin_amdgcn_readlane(myblk, 31),
                          e2b = __builtin_amdgcn_readlane(myblk, 47);
                const int n1b = __builtin_amdgcn_readlane(myblk, 16), n2b = __builtin_amdgcn_readlane(myblk, 32),
                          n3b = __builtin_amdgcn_readlane(myblk, 48);
                const float c0 = (myblk && row >= 1 && e0b == myblk) ? 1.f : 0.f;
                const float c1 = (myblk && row >= 2 && e1b == myblk) ? 1.f : 0.f;
                const float c2 = (myblk && row >= 3 && e2b == myblk) ? 1.f : 0.f;
                const int nxt = row_last ? (row == 0 ? n1b : row == 1 ? n2b : row == 2 ? n3b : 0) : b_p4;
                const bool blk_last = myblk && nxt != myblk;
                float *dst = out + a.radlen + (myblk - 7) * 32 + 4 * w4;
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    float4 *mine4 = reinterpret_cast<float4 *>(red + lane * RS);
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) {   // values 16 rr + 4 c4 .. + 3 of the block, value = u * NZ + z
                        const int v0 = 16 * rr + 4 * c4, u = v0 / NZ, z0 = v0 % NZ;
                        mine4[c4] = make_float4(acc[u][z0 / 2].x, acc[u][z0 / 2].y, acc[u][z0 / 2 + 1].x, acc[u][z0 / 2 + 1].y);
                    }
                    wave_sync();
                    const float4 *g4 = reinterpret_cast<const float4 *>(red + (lane & ~3) * RS + 4 * w4);   // row 4 g, quad w4
                    const float4 r0 = g4[0], r1 = g4[RS / 4], r2 = g4[2 * (RS / 4)], r3 = g4[3 * (RS / 4)];
                    float4 p = make_float4((r0.x + r1.x) + (r2.x + r3.x), (r0.y + r1.y) + (r2.y + r3.y),
                                           (r0.z + r1.z) + (r2.z + r3.z), (r0.w + r1.w) + (r2.w + r3.w));
                    // segmented inclusive scan over the four groups of the DPP row
                    p.x += m1 * dpp_perm<0x114>(p.x); p.y += m1 * dpp_perm<0x114>(p.y);
                    p.z += m1 * dpp_perm<0x114>(p.z); p.w += m1 * dpp_perm<0x114>(p.w);
                    p.x += m2 * dpp_perm<0x118>(p.x); p.y += m2 * dpp_perm<0x118>(p.y);
                    p.z += m2 * dpp_perm<0x118>(p.z); p.w += m2 * dpp_perm<0x118>(p.w);
                    // trailing sums of rows 0..2 carry into the later rows of the same block
                    if (row_last && row < 3) *reinterpret_cast<float4 *>(red + XOFF + row * 16 + 4 * w4) = p;
                    wave_sync();
                    const float4 x0 = *reinterpret_cast<const float4 *>(red + XOFF + 4 * w4);
                    const float4 x1 = *reinterpret_cast<const float4 *>(red + XOFF + 16 + 4 * w4);
                    const float4 x2 = *reinterpret_cast<const float4 *>(red + XOFF + 32 + 4 * w4);
                    p.x += c0 * x0.x + c1 * x1.x + c2 * x2.x;
                    p.y += c0 * x0.y + c1 * x1.y + c2 * x2.y;
                    p.z += c0 * x0.z + c1 * x1.z + c2 * x2.z;
                    p.w += c0 * x0.w + c1 * x1.w + c2 * x2.w;
                    if (remaining) {   // (blocks that did not fit the wave: an earlier batch stores on the spot)
                        if (blk_last) *reinterpret_cast<float4 *>(dst + 16 * rr) = p;
                    } else {
                        ang[64 * rr + lane] = p;
                    }
                    wave_sync();
                }
                hold_last = blk_last;
                hold_dst = dst;
            }
        }
        }   // !padding
        wave_sync();
        TR_STAMP(6)   // reduction
        // ---- the next atom's data has had this atom's arithmetic to arrive; then all stores of the row ----
        ANIHIP_FWD3_ARRIVED();
        TR_STAMP(7)   // wait for the prefetch
        // 32-wide slabs of this row that are not identically zero (include/anihip.h)
        uint32_t now_m;
        {
            const uint32_t r7 = (uint32_t)need & 0x7Fu;
            const uint32_t pr = (r7 | (r7 >> 1)) & 0x55u;   // bit 2 s: species 2 s or 2 s + 1 present
            const uint32_t rs = (pr & 1u) | ((pr >> 1) & 2u) | ((pr >> 2) & 4u) | ((pr >> 3) & 8u);
            now_m = rs | ((uint32_t)(need >> 7) << rslabs);
        }
        {
            // Zero fill: a slab that is not flagged now needs zeros only if it may still hold data (prev_m: every slab when
            // the row is written for the first time, else the flags the previous call left) -- 27 of a water atom's 32 slabs
            // are never flagged, 3.4 KB of its 4 KB row that an update in place does not write again.  Inside a flagged
            // radial slab the 16 columns of an absent species are zeroed as before.
            float4 *out4 = reinterpret_cast<float4 *>(out);
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            const uint32_t touch = prev_m | now_m;
            // (an update whose flags did not lose a slab -- the steady state of a simulation -- has nothing to clear outside
            // the radial part: one pass instead of four, wave-uniform)
            const int m_end = (prev_m & ~now_m) == 0u ? 1 : 4;
            for (int m = 0; m < m_end; ++m) {   // float4 slot f of the row belongs to block bit sb, slab sl
                const int f = lane + WAVE * m;
                const int sb = f < R4 ? (f >> 2) : 7 + ((f - R4) >> 3);
                const int sl = f < R4 ? (f >> 3) : rslabs + ((f - R4) >> 3);
                const bool nd = (need >> sb) & 1ull;
                if (m == 0 && f < R4) {   // radial part: 16-B stores from the staged row
                    const float4 rv = *reinterpret_cast<const float4 *>(rst + 4 * (lane & 31));
                    if ((touch >> sl) & 1u)
                        out4[f] = make_float4(nd ? rv.x : 0.f, nd ? rv.y : 0.f, nd ? rv.z : 0.f, nd ? rv.w : 0.f);
                } else if (f < L4 && !nd && ((prev_m >> sl) & 1u)) {
                    out4[f] = z4;
                }
            }
        }
        if (hold_last) {
            *reinterpret_cast<float4 *>(hold_dst) = ang[lane];
            *reinterpret_cast<float4 *>(hold_dst + 16) = ang[64 + lane];
        }
        if (slab_mask && lane == 0) slab_mask[i] = now_m;
        wave_sync();
        TR_STAMP(8)   // stores issued
    }
#ifdef ANIHIP_TRACE
    if (wib == 0 && lane == 0 && blockIdx.x < 2048)
        for (int q_ = 0; q_ < 10; ++q_) g_fwd3_trace[blockIdx.x][q_] = tr_sum[q_];
#endif
#undef ANIHIP_FWD3_ARRIVED
}

// ---------------------------------------------------------------------------------------------------
// Backward.  One wave per central atom i, three phases:
//   1. lane = neighbor: geometry + both cutoffs once per neighbor, and the WHOLE radial backward.  The list is full, so
//      the pair (i, j) contributes  sum_s (g_i[sp_j, s] + g_j[sp_i, s]) d/dr [0.25 exp(-eta (r - s)^2) fc(r)]  to the
//      force on i: atom i GATHERS the 64-B block g_j[sp_i, :] of every neighbor row and finishes its own radial force;
//      nothing is pushed to j (no atomics, deterministic).  Only when row j is not available (outside this launch's
//      rows lo..hi, i.e. owned by another rank, or not flagged in slab_mask) the own term is pushed to j instead.
//   2. lane = (angular neighbor j, part): the unordered pairs {j, k} of the n angular-range neighbors in the circular
//      tournament order k = j + 1 + rem (mod n), rem = part + parts * step.  A lane keeps its j for the whole atom
//      (J-side data and the gradient on j stay in registers); inside one part all k of a step are distinct, so the
//      gradient on k is a plain LDS read-add-write into the part's plane -- no LDS atomics (a ds_add_f32 costs ~40
//      LDS cycles per wave on gfx950), no cross-lane reduction.  The species-pair block of dE/dAEV a pair needs is
//      looked up per lane (8x8 table) and read from the row staged in LDS, so a step mixes all species pairs.
//   3. lane = angular neighbor: planes summed, one global float atomic per component per neighbor, minus the total
//      (and the radial force) on the central atom.
// Only the blocks of the dE/dAEV row that the atom's neighbor species can reach are loaded (5 of 35 for water).
// VIRIAL: also accumulate  W[a][b] = sum_ij (d E_i / d d_ij)[a] d_ij[b]  (the "fdotr" virial, ase.py:164-168) over the
// central atoms of this launch: six per-lane running sums (W is symmetric), one double atomic per wave at the end.
// accumulate one gradient component of atom `at`: float atomic, or (FIXED) a 64-bit integer atomic on a fixed-point
// accumulator in units of 2^-32 -- integer addition is associative, so the result does not depend on the order in which
// the waves arrive (run-to-run reproducible forces); every value pushed is itself computed in a fixed order
template <bool FIXED>
__device__ __forceinline__ void push_grad(float *grad_coords, size_t at, int comp, float v)
{
    if (FIXED) {
        unsigned long long *acc = reinterpret_cast<unsigned long long *>(grad_coords) + 3 * at + comp;
        atomicAdd(acc, (unsigned long long)__float2ll_rn(v * 4294967296.0f));
    } else {
        atomicAdd(grad_coords + 3 * at + comp, v);
    }
}

#ifndef ANIHIP_BWD_WAVES
#define ANIHIP_BWD_WAVES 4
#endif
#ifndef ANIHIP_REC_SCOPE
#define ANIHIP_REC_SCOPE 0
#endif
#ifndef ANIHIP_BWD_REC
#define ANIHIP_BWD_REC 1   // 0: never use the Gaussian recurrences (development A/B)
#endif
// REC (ANIHIP_AEV_REC_BWD: equally spaced ShfR and ShfA, constants in the table): Gaussians by recurrence.
//   radial, 16 shifts s_k = s_0 + k D: four anchors a = 1, 5, 9, 13 evaluated directly, f_a = exp2(-x_a^2), x_a = q r - s_a,
//   their neighbors as f_(a+m) = f_a gu_a^m K_m, m = -1, 1, 2, with gu_a = exp2(2 D x_a) = gu_9 exp2(-2 D (s_a - s_9)) and
//   K_m = exp2(-(m D)^2): 6 exponentials and 20 products instead of 16 exponentials; chained in the order (f_a gu) gu, whose
//   partial products are Gaussians over K_m (never overflow; an anchor that underflows has no significant neighbor).  The
//   K_m and the shifts ride on the packed FMA: sum_k w_k f_k (1, q r - s_k) = (A, q r A - B), (A, B) = sum (w_k h_k) (K_m, K_m s_k).
//   angular, NA shifts: ONE anchor c = NA / 2 - 1 and powers of gu = exp2(2 D x), gd = 1 / gu; f_c is common to all NA terms and
//   multiplies the contracted sums once:  sum_u w_u f_u (1, x - m D) = f_c (X, x X - Y), (X, Y) = sum (w_u gu^m) (K_m, m D K_m):
//   3 exponentials and 2 NA - 3 products instead of NA exponentials and 3 NA products.
template <int NA, int NZ, bool VIRIAL, bool FIXED, bool REC>
__global__ __launch_bounds__(BWD_WPB * WAVE, ANIHIP_BWD_WAVES) void k_aev_bwd(
    AevArgs a, const float *__restrict__ tab, int64_t lo64, int64_t hi64,
    const int32_t *__restrict__ species, const uint32_t *__restrict__ meta,
    const float4 *__restrict__ ent, const float *__restrict__ grad_aev, float *__restrict__ grad_coords,
    double *__restrict__ virial, const uint32_t *__restrict__ slab_mask, int64_t glo64, int64_t ghi64)
{
    // (32-bit atom indices inside the kernel -- rows hold 28-bit neighbor indices anyway: a 64-bit "less than" is a vector
    // compare whose operands end up spilled)
    const int lo = (int)lo64, hi = (int)hi64, glo = (int)glo64, ghi = (int)ghi64;
    float vxx = 0.f, vyy = 0.f, vzz = 0.f, vxy = 0.f, vxz = 0.f, vyz = 0.f;
    constexpr int ZQ = NZ / 4;
    __shared__ float4 s_nb[BWD_WPB][MAXA];    // ux uy uz r
    __shared__ float4 s_af[BWD_WPB][MAXA];    // fc, fc', 1/r, bits(j | species << 28)    (Rca)
    __shared__ float s_g[BWD_WPB][3][MAXA];   // gradient on neighbor k accumulated by part: [component][part * n + k]
    __shared__ __attribute__((aligned(16))) float s_stage[BWD_WPB][STAGE_FLOATS];
    __shared__ uint16_t s_ptab[64];           // byte offset of the angular block of species pair (sj, sk) in a row
    __shared__ uint32_t s_queue;              // AtomQueue: the next position of this workgroup's atom list

    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
    float4 *nb = s_nb[wib];
    float4 *af = s_af[wib];
    float *gx = s_g[wib][0], *gy = s_g[wib][1], *gz = s_g[wib][2];
    float *stage = s_stage[wib];
    if (threadIdx.x < 64) {
        const int sj = threadIdx.x >> 3, sk = threadIdx.x & 7;
        const int l_ = min(sj, sk), h_ = max(sj, sk);
        const int P = l_ * a.S - ((l_ * (l_ - 1)) >> 1) + (h_ - l_);
        s_ptab[threadIdx.x] = h_ < a.S ? (uint16_t)((a.radlen + 32 * P) * 4) : (uint16_t)0;
        if (threadIdx.x == 0) s_queue = 3u * BWD_WPB;
    }
    __syncthreads();

    // exp(-eta x^2) = exp2(-(q x)^2) with q = sqrt(eta log2 e): distances and shifts are kept pre-scaled
    const float qR = a.qR, qA = a.qA;
    float shfAq[NA], cZh[NZ], sZh[NZ], shfRq[16];   // wave-uniform (scalar registers)
#pragma unroll
    for (int u = 0; u < NA; ++u) shfAq[u] = REC ? 0.f : tab[TAB_SHFAQ + u];
#pragma unroll
    for (int v = 0; v < NZ; ++v) {   // halves: h = 0.5 + 0.5 cos(theta - ShfZ)
        cZh[v] = tab[TAB_COSZH + v];
        sZh[v] = tab[TAB_SINZH + v];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) shfRq[k] = REC ? 0.f : tab[TAB_SHFRQ + k];
    // REC: wave-uniform constants of the recurrences (scalar registers)
    constexpr int CA = NA / 2 - 1;
    // (REC: the constants of the recurrences are fetched by scalar loads at the head of the phase that uses them, from an opaque
    // copy of the table pointer -- hoisted out of the atom loop they overflow the scalar register file together with the other
    // phase's, and every use of a spilled one is a v_readlane on the vector ALU this kernel is bound by)
    const float pi_rcr = PI_F / a.Rcr, pi_rca = PI_F / a.Rca;
    // factors pulled out of the inner sums (see phase 2): d f2 / d rm = kap (q d) f2, d f1 / d theta = -2 zeta p1 (sz / 2)
    const float kap = -2.0f * a.EtaA / qA, kth0 = 2.0f * 0.95f * a.Zeta, kR2 = -2.0f * a.EtaR / qR;
    const float rev_rcr = 0.5f / a.Rcr, rev_rca = 0.5f / a.Rca;  // v_sin/v_cos take revolutions
    const int L4 = a.L >> 2, R4 = a.radlen >> 2;

    // "needed block" bookkeeping.  Bit t < 7: radial block of species t; bit 7 + P: angular block of species pair P.
    // Lane b < 35 decides bit b of an atom's mask from the per-species neighbor counts (ballot).
    int nd_tj = 7, nd_tk = 7;   // species 7 never occurs: count 0
    if (lane < 7) {
        nd_tj = nd_tk = lane;
    } else {
        int P = lane - 7, tj = 0;
        while (tj < a.S && P >= a.S - tj) { P -= a.S - tj; ++tj; }
        if (tj < a.S) { nd_tj = tj; nd_tk = tj + P; }
    }
    int nd_pack = (8 * nd_tj) | ((8 * nd_tk) << 8);   // bit shifts of the lane's two count bytes, one register
    // float4 slot f = lane + 64 m of a row belongs to block bit slot_of(f) (63 = beyond the row); worked out where it is
    // used -- four registers held across the atom otherwise
    auto slot_of = [&](int f) -> int { return f >= L4 ? 63 : (f < R4 ? (f >> 2) : 7 + ((f - R4) >> 3)); };
    auto need_of = [&](uint64_t pkA_, uint64_t pkF_) -> uint64_t {
        // (opaque: what the compiler derives from the shifts -- byte masks, comparisons -- is loop invariant, gets hoisted
        // out of the atom loop and spilled; the reload then sits behind the prefetch loads and waits for them)
        asm volatile("" : "+v"(nd_pack));
        const int sj = nd_pack & 255, sk = nd_pack >> 8;
        const int cj = byte_at(pkA_, sj), ck = byte_at(pkA_, sk), cf = byte_at(pkF_, sj);
        const bool nd = lane < 7 ? (cj + cf > 0) : (sj == sk ? cj >= 2 : (cj >= 1 && ck >= 1));
        return __ballot(nd && lane < 35);
    };

    // atoms from the workgroup's queue (AtomQueue); the first three positions of a wave are its own
    // (s_queue is set before the barrier above)
    AtomQueue q(&s_queue, lo, xcd_block(), (int)gridDim.x, lane);
    int i = (int)q.atom(wib), i1 = (int)q.atom(BWD_WPB + wib), i2 = (int)q.atom(2 * BWD_WPB + wib);
    // software pipeline over atoms: the header, the first 128 neighbor entries and the needed blocks of the dE/dAEV
    // row of the wave's next atom are in flight while atom i is processed
    uint32_t hw = hdr_load(meta, species, i, i < hi);
    AtomHdr h = hdr_decode(hw);
    // (the prefetched values are kept as 128-bit vector values, not float4 structs: a struct is split into four scalars
    // that the register allocator places apart, and the 16-byte load then goes to a temporary tuple whose copy-out waits
    // for the load right behind its issue)
    typedef float v4f_ __attribute__((ext_vector_type(4)));
    const v4f_ zero4 = {1.f, 0.f, 0.f, 0.f};
    v4f_ e0 = zero4, gr0 = zero4, gr1 = zero4, gr2 = zero4, gr3 = zero4;   // (e0: the first 64 entries of the row)
    // (a macro, not a lambda: captured register arrays would be spilled to scratch)
#define ANIHIP_BWD_ISSUE(ia, hh)                                                                         \
    {                                                                                                    \
        /* every lane loads, index clamped into the row: a load under a per-lane condition is merged with the old value  \
           afterwards, and that move waits for the load right behind its issue (no prefetch at all) */          \
        const bool ok_ = (ia) < hi && (hh).sp >= 0 && (hh).nA + (hh).nF > 0;                             \
        const int n_ = ok_ ? (hh).nA + (hh).nF : 0;                                                      \
        if (n_ > 0) {                                                                                    \
            e0 = *reinterpret_cast<const v4f_ *>(ent + (hh).start + min(lane, n_ - 1));                  \
        }                                                                                                \
        const uint64_t need_ = ok_ ? need_of((hh).pkA, (hh).pkF) : 0ull;                                 \
        const v4f_ *g4_ = reinterpret_cast<const v4f_ *>(grad_aev + (size_t)((ia) < hi ? (ia) : lo) * a.L); \
        if ((need_ >> slot_of(lane)) & 1ull) gr0 = g4_[lane];                                              \
        if ((need_ >> slot_of(lane + WAVE)) & 1ull) gr1 = g4_[lane + WAVE];                                       \
        if ((need_ >> slot_of(lane + 2 * WAVE)) & 1ull) gr2 = g4_[lane + 2 * WAVE];                                   \
        if ((need_ >> slot_of(lane + 3 * WAVE)) & 1ull) gr3 = g4_[lane + 3 * WAVE];                                   \
    }
    ANIHIP_BWD_ISSUE(i, h)
    uint32_t hw_next = hdr_load(meta, species, i1, i1 < hi);
#ifdef ANIHIP_TRACE
    unsigned long long tr_sum[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tr_last = __builtin_amdgcn_s_memtime();
#endif

    for (; i < hi; i = i1, i1 = i2, i2 = (int)q.granted()) {
        TR_STAMP(0)   // loop head
        const int nA = h.nA, nR = h.nA + h.nF;
        const bool skip = h.sp < 0 || nR == 0;
        const uint32_t start = h.start;
        const int spi = h.sp;
        float sx = 0.f, sy = 0.f, sz_ = 0.f;   // minus the gradient on the central atom, per lane
        if (!skip) {
            v4f_ *st4 = reinterpret_cast<v4f_ *>(stage);
            st4[lane] = gr0;
            st4[lane + WAVE] = gr1;
            st4[lane + 2 * WAVE] = gr2;
            if (lane + 3 * WAVE < L4) st4[lane + 3 * WAVE] = gr3;
            gx[lane] = 0.f; gx[lane + WAVE] = 0.f;
            gy[lane] = 0.f; gy[lane + WAVE] = 0.f;
            gz[lane] = 0.f; gz[lane + WAVE] = 0.f;
            wave_sync();
            TR_STAMP(1)   // stage own dE/dAEV row
            // ---- phase 1 (lane = neighbor) ----
            const float *tab1 = tab;
#if ANIHIP_REC_SCOPE & 1
            asm volatile("" : "+s"(tab1));
#endif
            const float twoDR = tab1[TAB_RECR], ebmax = tab1[TAB_RECR + 9];
            const float KR1 = tab1[TAB_RECR + 1], KR1D = tab1[TAB_RECR + 2], KR2 = tab1[TAB_RECR + 3], KR2D = tab1[TAB_RECR + 4];
            // gu_a / gu_9, gd_a / gd_9 of anchor a = 4 g + 1
            const float GU[4] = {tab1[TAB_RECR + 6], tab1[TAB_RECR + 5], 1.0f, tab1[TAB_RECR + 7]};
            const float GD[4] = {tab1[TAB_RECR + 8], tab1[TAB_RECR + 7], 1.0f, tab1[TAB_RECR + 5]};
            const float sA[4] = {tab1[TAB_SHFRQ + 1], tab1[TAB_SHFRQ + 5], tab1[TAB_SHFRQ + 9], tab1[TAB_SHFRQ + 13]};
            const float *grow0 = grad_aev + (size_t)(spi < 0 ? 0 : spi) * 16;
            const int sbit = spi >> 1;
            for (int c0 = 0; c0 < nR; c0 += WAVE) {
                const int e = c0 + lane;
                v4f_ d = c0 == 0 ? e0 : zero4;
                if (c0 >= WAVE && e < nR) d = *reinterpret_cast<const v4f_ *>(ent + start + e);   // (> 64 neighbors: waits here)
                const bool ve = e < nR;
                const uint32_t wbits = __float_as_uint(d.w);
                const int jn = ve ? (int)(wbits & IDX_MASK) : lo;
                // the neighbor's block for MY species: issued first, consumed at the end of the pass
                const bool inrow = ve && jn >= glo && jn < ghi;   // (empty range: asymmetric list, push everything)
                float4 G0 = make_float4(0.f, 0.f, 0.f, 0.f), G1 = G0, G2 = G0, G3 = G0;
                uint32_t jmask = 0xFFFFFFFFu;
                if (inrow) {
                    const float4 *gj4 = reinterpret_cast<const float4 *>(grow0 + (size_t)jn * a.L);
                    G0 = gj4[0]; G1 = gj4[1]; G2 = gj4[2]; G3 = gj4[3];
                    if (slab_mask) jmask = slab_mask[jn];
                }
                const float r2 = d.x * d.x + d.y * d.y + d.z * d.z;
                const float inv = __builtin_amdgcn_rsqf(r2);
                const float r = r2 * inv;
                const float ux = d.x * inv, uy = d.y * inv, uz = d.z * inv;
                const int t = ve ? (int)(wbits >> 28) : 0;
                float fcr, dfcr;  // 0.25 fc, 0.25 fc' of the radial cutoff
                if (a.smooth) {
                    const float2 cr = smooth_cutoff(r, 1.0f / a.Rcr);
                    fcr = 0.25f * cr.x; dfcr = 0.25f * cr.y;
                } else {
                    fcr = 0.125f * __builtin_amdgcn_cosf(r * rev_rcr) + 0.125f;
                    dfcr = -0.125f * pi_rcr * __builtin_amdgcn_sinf(r * rev_rcr);
                }
                const float4 *wo = reinterpret_cast<const float4 *>(stage + t * 16);
                const float4 W0 = wo[0], W1 = wo[1], W2 = wo[2], W3 = wo[3];
                const bool gat = inrow && ((jmask >> sbit) & 1u);
                if (!gat) G0 = G1 = G2 = G3 = make_float4(0.f, 0.f, 0.f, 0.f);   // (never 0 * unwritten memory)
                float wk[16] = {W0.x + G0.x, W0.y + G0.y, W0.z + G0.z, W0.w + G0.w, W1.x + G1.x, W1.y + G1.y,
                                W1.z + G1.z, W1.w + G1.w, W2.x + G2.x, W2.y + G2.y, W2.z + G2.z, W2.w + G2.w,
                                W3.x + G3.x, W3.y + G3.y, W3.z + G3.z, W3.w + G3.w};
                // d/dr [exp(-eta d^2) fc] = exp(..) (fc' - 2 eta d fc):  dR = fc' sum w e - 2 eta fc sum w e d
                float rq = qR * r;
                // (opaque: left alone the compiler folds q r - s_k into one FMA with TWO scalar operands, which the vector ALU
                // cannot take -- one more move per shift, 16 per pass)
                asm volatile("" : "+v"(rq));
                v2f AB = (v2f){0.f, 0.f};   // sum w e, sum w e (q d)
                if constexpr (REC) {
                    // (clamped: distances inside the cutoff stay far from the bound -- anihip_aev_table_pack checked that --, an entry
                    // beyond it, which no row builder of this library produces, gets finite nonsense instead of inf x 0 = NaN)
                    const float eb = __builtin_amdgcn_fmed3f(twoDR * (rq - sA[2]), -ebmax, ebmax);
                    const float gub = __builtin_amdgcn_exp2f(eb), gdb = __builtin_amdgcn_exp2f(-eb);
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        // the group's four shifts s_a + m D, m = -1 .. 2:  P = sum_m (w h)_m (K_m, m D K_m), and
                        // sum_m w f (1, x_a - m D) = (P.x, x_a P.x - P.y)
                        // (one group after the other: interleaved, the four groups' temporaries push the next atom's prefetched
                        // rows out of the register file)
                        __builtin_amdgcn_sched_barrier(0);
                        const float4 Wg = wo[g4], Gg = g4 == 0 ? G0 : g4 == 1 ? G1 : g4 == 2 ? G2 : G3;
                        const float xg = rq - sA[g4];
                        const float gu = g4 == 2 ? gub : gub * GU[g4], gd = g4 == 2 ? gdb : gdb * GD[g4];
                        const float h1 = __builtin_amdgcn_exp2f(-xg * xg);
                        const float h0 = h1 * gd, h2 = h1 * gu, h3 = h2 * gu;
                        const float w0 = (Wg.x + Gg.x) * h0, w1 = (Wg.y + Gg.y) * h1, w2 = (Wg.z + Gg.z) * h2, w3 = (Wg.w + Gg.w) * h3;
                        v2f P = (v2f){w1, 0.f};
                        P += (v2f){w0, w0} * (v2f){KR1, -KR1D};
                        P += (v2f){w2, w2} * (v2f){KR1, KR1D};
                        P += (v2f){w3, w3} * (v2f){KR2, KR2D};
                        AB += (v2f){P.x, P.x} * (v2f){1.0f, xg};
                        AB.y -= P.y;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const float dq = rq - shfRq[k];
                        const float we = wk[k] * __builtin_amdgcn_exp2f(-dq * dq);
                        AB += (v2f){we, we} * (v2f){1.0f, dq};
                    }
                }
                float dR = dfcr * AB.x + kR2 * fcr * AB.y;
                dR = ve ? dR : 0.f;
                const float Gx = dR * ux, Gy = dR * uy, Gz = dR * uz;
                // an own term that must be PUSHED to an angular-range neighbor travels with the angular part
                // (plane 0 of the per-neighbor sums) and is counted in phase 3
                const bool via3 = ve && e < nA && !gat;
                if (!via3) {
                    sx += Gx; sy += Gy; sz_ += Gz;
                    if (VIRIAL) {   // a gathered pair is seen from both of its atoms: half the virial each time
                        const float vf = gat ? 0.5f : 1.0f;
                        vxx += vf * Gx * d.x; vyy += vf * Gy * d.y; vzz += vf * Gz * d.z;
                        vxy += vf * Gx * d.y; vxz += vf * Gx * d.z; vyz += vf * Gy * d.z;
                    }
                }
                if (ve) {
                    if (e < nA) {
                        nb[e] = make_float4(ux, uy, uz, 0.5f * qA * r);
                        const float2 ca = a.smooth ? smooth_cutoff(r, 1.0f / a.Rca)
                                                   : make_float2(0.5f * __builtin_amdgcn_cosf(r * rev_rca) + 0.5f,
                                                                 -0.5f * pi_rca * __builtin_amdgcn_sinf(r * rev_rca));
                        af[e] = make_float4(ca.x, ca.y, inv, d.w);
                        if (!gat) { gx[e] = Gx; gy[e] = Gy; gz[e] = Gz; }   // pushed with the angular part (plane 0)
                    } else if (!gat) {
                        push_grad<FIXED>(grad_coords, (size_t)jn, 0, Gx);
                        push_grad<FIXED>(grad_coords, (size_t)jn, 1, Gy);
                        push_grad<FIXED>(grad_coords, (size_t)jn, 2, Gz);
                    }
                }
            }
        }
        // prefetch the next atom
        h = hdr_decode(hw_next);
        TR_STAMP(2)   // phase 1: neighbor terms, radial gather
        q.request();   // (the position of the atom after i2: granted by the end of this atom)
        ANIHIP_BWD_ISSUE(i1, h)
        hw_next = hdr_load(meta, species, i2, i2 < hi);
        if (skip) continue;
        wave_sync();
        TR_STAMP(3)   // prefetch issue

        // ---- phase 2 (lane = (angular neighbor j, part)) ----
        const float *tab2 = tab;
#if ANIHIP_REC_SCOPE & 2
        asm volatile("" : "+s"(tab2));
#endif
        const float twoDA = tab2[TAB_RECA], sC = tab2[TAB_SHFAQ + CA];
        float KA[5], KAD[5];         // K_m, m D K_m, m = 0 .. 4
        KA[0] = 1.0f; KAD[0] = 0.f;
#pragma unroll
        for (int m = 1; m <= 4; ++m) {
            KA[m] = tab2[TAB_RECAK + 2 * (m - 1)];
            KAD[m] = tab2[TAB_RECAK + 2 * (m - 1) + 1];
        }
        const int n = nA;
        const int dv = (n - 1) >> 1;                 // partners per neighbor in the tournament
        const bool even = (n & 1) == 0;
        const int nrem = dv + (even ? 1 : 0);        // rem == dv: the n / 2 diameters of an even n
        const int halfn = n >> 1;
        const int npl = n < WAVE ? n : WAVE;         // lanes per part
        int parts = n > 0 ? WAVE / npl : 1;
        parts = parts > nrem ? (nrem > 0 ? nrem : 1) : parts;
        const int part = n > 0 ? (int)(((float)lane + 0.5f) / (float)npl) : 0;
        const int jl = lane - part * npl;
        const int pbase = part * n;                  // (n > 64: one part, base 0)
        const int nsteps = nrem > 0 ? (nrem + parts - 1) / parts : 0;
        for (int jb = 0; jb < n && nsteps > 0; jb += WAVE) {
            const int j = jb + jl;
            const bool vj = part < parts && j < n;
            const int jc = vj ? j : 0;
            const float4 Jv = nb[jc], FJ = af[jc];
            const int sj8 = (int)((__float_as_uint(FJ.w) >> 25) & 0x38u);
            float gjx = 0.f, gjy = 0.f, gjz = 0.f;
            for (int s = 0; s < nsteps; ++s) {
                const int rem = part + parts * s;
                const bool v = vj && (rem < dv || (even && rem == dv && j < halfn));
                int k = j + 1 + rem;
                k = k >= n ? k - n : k;
                k = v ? k : 0;
                const float4 Kv = nb[k], FK = af[k];
                const int sk = (int)(__float_as_uint(FK.w) >> 28);
                const float4 *wb = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(stage) +
                                                                     s_ptab[sj8 + sk]);
                const float c = Jv.x * Kv.x + Jv.y * Kv.y + Jv.z * Kv.z;
                const float ct = 0.95f * c;
                const float st2 = fmaxf(1.0f - ct * ct, 1e-12f);
                const float rst = __builtin_amdgcn_rsqf(st2);   // 1 / sin(theta)
                const float st = st2 * rst;
                const float srq = Jv.w + Kv.w;                  // q rm
                // f1h = h^zeta (= f1 / 2), f1d = p1 sin(theta - ShfZ) / 2 (= -(d f1 / d theta) / (2 zeta))
                v2f fd[NZ];
#pragma unroll
                for (int z = 0; z < NZ; ++z) {
                    // (h - 0.5, sin(theta - ShfZ) / 2) = ct (cos, -sin) / 2 + st (sin, cos) / 2
                    const v2f hs = (v2f){ct, ct} * (v2f){cZh[z], -sZh[z]} + (v2f){st, st} * (v2f){sZh[z], cZh[z]};
                    const float hh = 0.5f + hs.x;
                    const float p1 = __builtin_amdgcn_exp2f((a.Zeta - 1.0f) * __builtin_amdgcn_logf(__builtin_fabsf(hh)));
                    fd[z] = (v2f){p1, p1} * (v2f){hh, hs.y};
                }
                // contract the radial-shift index first: (X_z, Y_z) = sum_a w[a][z] (f2[a], q d f2[a])
                v2f XY[NZ];
#pragma unroll
                for (int z = 0; z < NZ; ++z) XY[z] = (v2f){0.f, 0.f};
                float xC = 0.f, fC = 1.0f;   // REC: scaled distance from the anchor shift, the anchor's Gaussian
                if constexpr (REC) {
                    xC = srq - sC;
                    // (see the radial clamp: gu^(NA / 2) stays finite; the table packer admits |e| NA / 2 < 110)
                    const float e = __builtin_amdgcn_fmed3f(twoDA * xC, -240.0f / NA, 240.0f / NA);
                    fC = __builtin_amdgcn_exp2f(-xC * xC);
                    float pw[NA];   // gu^m of shift u = CA + m
                    pw[CA] = 1.0f;
                    pw[CA + 1] = __builtin_amdgcn_exp2f(e);
#pragma unroll
                    for (int u = CA + 2; u < NA; ++u) pw[u] = pw[u - 1] * pw[CA + 1];
                    if constexpr (CA > 0) {
                        pw[CA - 1] = __builtin_amdgcn_exp2f(-e);
#pragma unroll
                        for (int u = CA - 2; u >= 0; --u) pw[u] = pw[u + 1] * pw[CA - 1];
                    }
#pragma unroll
                    for (int u = 0; u < NA; ++u) {
                        const int m = u - CA;
                        const v2f F = (v2f){pw[u], pw[u]} * (v2f){KA[m < 0 ? -m : m], m < 0 ? -KAD[-m] : KAD[m]};
#pragma unroll
                        for (int zq = 0; zq < ZQ; ++zq) {
                            const float4 w4 = wb[u * ZQ + zq];
                            XY[4 * zq + 0] += (v2f){w4.x, w4.x} * F;
                            XY[4 * zq + 1] += (v2f){w4.y, w4.y} * F;
                            XY[4 * zq + 2] += (v2f){w4.z, w4.z} * F;
                            XY[4 * zq + 3] += (v2f){w4.w, w4.w} * F;
                        }
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < NA; ++u) {
                        const float dq = srq - shfAq[u];
                        const float f2 = __builtin_amdgcn_exp2f(-dq * dq);
                        const v2f F = (v2f){f2, dq * f2};
#pragma unroll
                        for (int zq = 0; zq < ZQ; ++zq) {
                            const float4 w4 = wb[u * ZQ + zq];
                            XY[4 * zq + 0] += (v2f){w4.x, w4.x} * F;
                            XY[4 * zq + 1] += (v2f){w4.y, w4.y} * F;
                            XY[4 * zq + 2] += (v2f){w4.z, w4.z} * F;
                            XY[4 * zq + 3] += (v2f){w4.w, w4.w} * F;
                        }
                    }
                }
                // C0 = sum w f1 f2 = 2 c0h, Cth = sum w f1' f2 = -2 zeta cth, CR = sum w f1 f2' = 2 kap crh
                v2f cc = (v2f){0.f, 0.f};
                float crh = 0.f;
#pragma unroll
                for (int z = 0; z < NZ; ++z) {
                    cc += (v2f){XY[z].x, XY[z].x} * fd[z];
                    crh += XY[z].y * fd[z].x;
                }
                if constexpr (REC) {   // (X, Y) -> f_c (X, x X - Y)
                    crh = fC * (xC * cc.x - crh);
                    cc = cc * fC;
                }
                const float m = v ? 1.0f : 0.0f;
                const float fcc = m * FJ.x * FK.x;
                const float kth = cc.y * fcc * (kth0 * rst);
                const float krr = kap * crh * fcc, c02 = 2.0f * m * cc.x;
                const float k1 = krr + c02 * (FJ.y * FK.x);
                const float k2 = krr + c02 * (FJ.x * FK.y);
                const float aj = kth * FJ.z, ak = kth * FK.z;
                gjx += aj * (Kv.x - c * Jv.x) + k1 * Jv.x;
                gjy += aj * (Kv.y - c * Jv.y) + k1 * Jv.y;
                gjz += aj * (Kv.z - c * Jv.z) + k1 * Jv.z;
                const float gkx = ak * (Jv.x - c * Kv.x) + k2 * Kv.x;
                const float gky = ak * (Jv.y - c * Kv.y) + k2 * Kv.y;
                const float gkz = ak * (Jv.z - c * Kv.z) + k2 * Kv.z;
                // inside a part the k of one step are all different: plain read-add-write
                const int gi = (v ? pbase : 0) + k;
                const float ox = gx[gi], oy = gy[gi], oz = gz[gi];
                if (v) {   // (LDS operations of one wave execute in order: the next step sees these)
                    gx[gi] = ox + gkx;
                    gy[gi] = oy + gky;
                    gz[gi] = oz + gkz;
                }
            }
            if (vj) {
                const int gi = pbase + j;
                gx[gi] += gjx;
                gy[gi] += gjy;
                gz[gi] += gjz;
            }
            wave_sync();
        }
        TR_STAMP(4)   // phase 2: angular pairs
        // ---- phase 3 (lane = angular neighbor): +G to the neighbor, -sum(all G) to the central atom ----
        // Vector-memory operations retire in order and the compiler waits with vmcnt(0) for what it cannot count: the next
        // atom's prefetched registers are "used" HERE, before this atom's atomics are issued, so that the top of the loop
        // has nothing left to wait for (it would wait for the atomics' round trip to L2 otherwise, once per atom and wave).
        asm volatile("" ::"v"(e0), "v"(gr0), "v"(gr1), "v"(gr2), "v"(gr3), "v"(hw_next) : "memory");
        for (int e = lane; e < nA; e += WAVE) {
            float x = 0.f, y = 0.f, z = 0.f;
            const int np_ = n <= WAVE ? parts : 1;
            for (int pp = 0; pp < np_; ++pp) {
                x += gx[pp * n + e];
                y += gy[pp * n + e];
                z += gz[pp * n + e];
            }
            const float4 fa = af[e];
            const size_t jat = (size_t)(__float_as_uint(fa.w) & IDX_MASK);
            push_grad<FIXED>(grad_coords, jat, 0, x);
            push_grad<FIXED>(grad_coords, jat, 1, y);
            push_grad<FIXED>(grad_coords, jat, 2, z);
            sx += x; sy += y; sz_ += z;
            if (VIRIAL) {
                const float4 u = nb[e];   // unit vector, 0.5 qA r
                const float rr = u.w * (2.0f / qA);
                const float dx = u.x * rr, dy = u.y * rr, dz = u.z * rr;
                vxx += x * dx; vyy += y * dy; vzz += z * dz;
                vxy += x * dy; vxz += x * dz; vyz += y * dz;
            }
        }
        sx = wave_sum(sx); sy = wave_sum(sy); sz_ = wave_sum(sz_);
        if (lane == 0) {
            push_grad<FIXED>(grad_coords, (size_t)i, 0, -sx);
            push_grad<FIXED>(grad_coords, (size_t)i, 1, -sy);
            push_grad<FIXED>(grad_coords, (size_t)i, 2, -sz_);
        }
        wave_sync();
        TR_STAMP(5)   // phase 3: atomics issued
    }
#ifdef ANIHIP_TRACE
    if (wib == 0 && lane == 0 && blockIdx.x < 2048)
        for (int q_ = 0; q_ < 10; ++q_) g_fwd3_trace[blockIdx.x][q_] = tr_sum[q_];
#endif
    if (VIRIAL) {
        vxx = wave_sum(vxx); vyy = wave_sum(vyy); vzz = wave_sum(vzz);
        vxy = wave_sum(vxy); vxz = wave_sum(vxz); vyz = wave_sum(vyz);
        if (lane == 0) {
            atomicAdd(virial + 0, (double)vxx); atomicAdd(virial + 4, (double)vyy); atomicAdd(virial + 8, (double)vzz);
            atomicAdd(virial + 1, (double)vxy); atomicAdd(virial + 3, (double)vxy);
            atomicAdd(virial + 2, (double)vxz); atomicAdd(virial + 6, (double)vxz);
            atomicAdd(virial + 5, (double)vyz); atomicAdd(virial + 7, (double)vyz);
        }
    }
}

// any other grid: aev_generic.hip
int aev_forward_generic(hipStream_t stream, const anihip_aev_params *p, const float *table, int64_t lo, int64_t hi,
                        const int32_t *species, const uint32_t *meta, const float *ent, float *aev, const float *tangent,
                        uint32_t *slab_mask);
int aev_backward_generic(hipStream_t stream, const anihip_aev_params *p, const float *table, int64_t lo, int64_t hi,
                         const int32_t *species, const uint32_t *meta, const float *ent, const float *grad_aev,
                         float *grad_coords, double *virial, bool fixed);

// the grids the tuned kernels are built for: 16 radial shifts, 8 x 4 (ANI-2x) or 4 x 8 (ANI-1x) angular terms
static bool tuned_grid(const anihip_aev_params *p)
{
    return p->n_shf_r == 16 && ((p->n_shf_a == 8 && p->n_shf_z == 4) || (p->n_shf_a == 4 && p->n_shf_z == 8));
}

}  // namespace anihip

using namespace anihip;

extern "C" int anihip_aev_table_pack(anihip_aev_params *p, const float *ShfR, const float *ShfA,
                                     const float *ShfZ, float *t)
{
    ANIHIP_REQUIRE(p && ShfR && ShfA && ShfZ && t, "null pointer argument");
    ANIHIP_REQUIRE(p->n_shf_r >= 1 && p->n_shf_r <= 32 && p->n_shf_a >= 1 && p->n_shf_a <= 16 && p->n_shf_z >= 1 &&
                       p->n_shf_z <= 16,
                   "symmetry-function grid outside n_shf_r <= 32, n_shf_a <= 16, n_shf_z <= 16 (got %d, %d x %d)",
                   p->n_shf_r, p->n_shf_a, p->n_shf_z);
    for (int k = 0; k < ANIHIP_AEV_TABLE_FLOATS; ++k) t[k] = 0.f;
    for (int k = 0; k < p->n_shf_r; ++k) t[TAB_SHFR + k] = ShfR[k];
    for (int k = 0; k < p->n_shf_a; ++k) t[TAB_SHFA + k] = ShfA[k];
    for (int k = 0; k < p->n_shf_z; ++k) {
        t[TAB_COSZ + k] = (float)cos((double)ShfZ[k]);
        t[TAB_SINZ + k] = (float)sin((double)ShfZ[k]);
    }
    p->flags = 0;
    if (!tuned_grid(p)) return 0;   // (the general kernels read the plain shifts)
    // pre-scaled copies (wave-uniform operands of the kernels stay in scalar registers)
    const float qR = sqrtf(p->EtaR * LOG2E), qA = sqrtf(p->EtaA * LOG2E);
    for (int k = 0; k < p->n_shf_r; ++k) t[TAB_SHFRQ + k] = qR * ShfR[k];
    for (int k = 0; k < p->n_shf_a; ++k) t[TAB_SHFAQ + k] = qA * ShfA[k];
    for (int k = 0; k < p->n_shf_z; ++k) {
        t[TAB_COSZH + k] = 0.5f * t[TAB_COSZ + k];
        t[TAB_SINZH + k] = 0.5f * t[TAB_SINZ + k];
    }
    // Gaussian recurrence of the forward pair loop (k_aev_fwd3): equally spaced ShfA, and exponents inside fp32's range
    // for every scaled mean distance 0 .. qA Rca the kernel can meet
    {
        const int n = p->n_shf_a, c = n / 2 - 1;
        const float D = t[TAB_SHFAQ + 1] - t[TAB_SHFAQ];
        bool ok = D > 0.f;
        for (int k = 1; k + 1 < n; ++k) ok = ok && fabsf((t[TAB_SHFAQ + k + 1] - t[TAB_SHFAQ + k]) - D) < 1e-4f * D;
        const float xm = fmaxf(fabsf(qA * p->Rca - t[TAB_SHFAQ + c]), fabsf(t[TAB_SHFAQ + c]));
        if (ok && xm * xm < 100.f && 2.0f * D * xm < 100.f) p->flags |= ANIHIP_AEV_UNIFORM_SHFA;
    }
    // Gaussian recurrences of the backward kernel (k_aev_bwd<.., REC>): both shift arrays equally spaced.  Everything in double
    // from the fp32 arrays: a spacing taken as the difference of two fp32 products is 3e-7 off, and it is multiplied by
    // scaled distances of up to 13.
    {
        const int na = p->n_shf_a, ca = na / 2 - 1, nr = p->n_shf_r;
        const double qa = sqrt((double)p->EtaA * (double)LOG2E), qr = sqrt((double)p->EtaR * (double)LOG2E);
        const double DA = qa * ((double)ShfA[na - 1] - (double)ShfA[0]) / (na - 1);
        const double DR = qr * ((double)ShfR[nr - 1] - (double)ShfR[0]) / (nr - 1);
        bool ok = DA > 0. && DR > 0.;
        for (int k = 0; k < na; ++k) ok = ok && fabs(qa * ((double)ShfA[k] - (double)ShfA[0]) - k * DA) < 1e-5 * DA;
        for (int k = 0; k < nr; ++k) ok = ok && fabs(qr * ((double)ShfR[k] - (double)ShfR[0]) - k * DR) < 1e-5 * DR;
        // angular: powers gu^m, |m| <= NA / 2, of gu = exp2(2 D_A x), x = q_A (r_mean - ShfA[ca]), r_mean in 0 .. Rca
        const double xa = fmax(fabs(qa * ((double)p->Rca - (double)ShfA[ca])), fabs(qa * (double)ShfA[ca]));
        ok = ok && (na / 2) * 2. * DA * xa < 110. && (na / 2) * (na / 2) * DA * DA < 110.;
        // radial: gu = exp2(2 D_R x) exp2(+-16 D_R^2), x = q_R (r - ShfR[9]), r in 0 .. Rcr; chained values <= exp2(4 D_R^2)
        const double xr = fmax(fabs(qr * ((double)p->Rcr - (double)ShfR[9])), fabs(qr * (double)ShfR[9]));
        ok = ok && 2. * DR * xr + 16. * DR * DR < 110.;
        if (ok) {
            p->flags |= ANIHIP_AEV_REC_BWD;
            t[TAB_RECA] = (float)(2. * DA);
            for (int m = 1; m <= 4; ++m) {
                const double K = exp2(-(m * DA) * (m * DA));
                t[TAB_RECAK + 2 * (m - 1)] = (float)K;
                t[TAB_RECAK + 2 * (m - 1) + 1] = (float)(m * DA * K);
            }
            const double K1 = exp2(-DR * DR), K2 = exp2(-4. * DR * DR);
            t[TAB_RECR + 0] = (float)(2. * DR);
            t[TAB_RECR + 1] = (float)K1;
            t[TAB_RECR + 2] = (float)(DR * K1);
            t[TAB_RECR + 3] = (float)K2;
            t[TAB_RECR + 4] = (float)(2. * DR * K2);
            t[TAB_RECR + 5] = (float)exp2(8. * DR * DR);
            t[TAB_RECR + 6] = (float)exp2(16. * DR * DR);
            t[TAB_RECR + 7] = (float)exp2(-8. * DR * DR);
            t[TAB_RECR + 8] = (float)exp2(-16. * DR * DR);
            t[TAB_RECR + 9] = (float)(120. - 16. * DR * DR);   // clamp of |2 D_R x|: gu_9 exp2(16 D_R^2) stays finite
        }
    }
    return 0;
}

static int make_args(const anihip_aev_params *p, AevArgs *a)
{
    ANIHIP_REQUIRE(p->num_species >= 1 && p->num_species <= MAX_S - 1, "num_species must be 1..7");
    ANIHIP_REQUIRE(p->n_shf_r == 16, "n_shf_r must be 16");
    ANIHIP_REQUIRE((p->n_shf_a == 8 && p->n_shf_z == 4) || (p->n_shf_a == 4 && p->n_shf_z == 8),
                   "angular grid must be 8x4 (ANI-2x) or 4x8 (ANI-1x)");
    a->S = p->num_species;
    a->NR = p->n_shf_r;
    a->radlen = a->S * a->NR;
    a->L = a->radlen + (a->S * (a->S + 1) / 2) * 32;
    a->Rcr = p->Rcr; a->Rca = p->Rca;
    a->EtaR = p->EtaR; a->EtaA = p->EtaA; a->Zeta = p->Zeta;
    a->kR = -p->EtaR * LOG2E;
    a->kA = -p->EtaA * LOG2E;
    a->qR = sqrtf(p->EtaR * LOG2E);
    a->qA = sqrtf(p->EtaA * LOG2E);
    ANIHIP_REQUIRE(p->cutoff_kind == ANIHIP_CUTOFF_COSINE || p->cutoff_kind == ANIHIP_CUTOFF_SMOOTH,
                   "cutoff_kind must be ANIHIP_CUTOFF_COSINE or ANIHIP_CUTOFF_SMOOTH");
    a->smooth = p->cutoff_kind == ANIHIP_CUTOFF_SMOOTH;
    return 0;
}

static int persistent_blocks(int64_t n_central, int wpb, int blocks_per_cu)
{
    int64_t b = (n_central + wpb - 1) / wpb;
    if (b < 1) b = 1;
    if (b > 256 * blocks_per_cu) b = 256 * blocks_per_cu;
    return (int)b;
}

static int aev_forward(void *stream, const anihip_aev_params *p, const float *table,
                       int64_t n_atoms, int64_t lo, int64_t hi, const int32_t *species,
                       const uint32_t *meta, const float *ent, float *aev, uint32_t *slab_mask,
                       uint32_t *status, bool update, const uint32_t *prev_mask = nullptr)
{
    ANIHIP_REQUIRE(p && table && species && meta && ent && aev, "null pointer argument");
    ANIHIP_REQUIRE(0 <= lo && lo <= hi && hi <= n_atoms, "central range outside 0..n_atoms");
    if (!tuned_grid(p)) {   // (general grids: flags of the plain 32-column slabs, rows of at most 1024 columns)
        ANIHIP_REQUIRE(!update, "rows are updated in place on the 16 / 8x4 / 4x8 grids only");
        return aev_forward_generic((hipStream_t)stream, p, table, lo, hi, species, meta, ent, aev, nullptr, slab_mask);
    }
    ANIHIP_REQUIRE(!slab_mask || (p->num_species + 1) / 2 + p->num_species * (p->num_species + 1) / 2 <= 32,
                   "slab_mask needs at most 32 slabs (num_species <= 7)");
    AevArgs a;
    if (int rc = make_args(p, &a)) return rc;
    a.update = update ? 1 : 0;
    if (hi == lo) return 0;
    // (one workgroup of 16 waves per CU: ANIHIP_FWD3_WAVES per SIMD)
    dim3 grid(persistent_blocks(hi - lo, FWD3_WPB, (ANIHIP_FWD3_WAVES * 4 + FWD3_WPB - 1) / FWD3_WPB)), block(FWD3_WPB * WAVE);
    const bool rec = (p->flags & ANIHIP_AEV_UNIFORM_SHFA) != 0 && ANIHIP_FWD3_REC;
#define ANIHIP_LAUNCH_FWD3(NA_, NZ_, REC_)                                                                              \
    hipLaunchKernelGGL((k_aev_fwd3<NA_, NZ_, REC_>), grid, block, 0, (hipStream_t)stream, a, table, lo, hi, species, meta, \
                       (const float4 *)ent, aev, slab_mask, prev_mask)
    if (p->n_shf_a == 8) {
        if (rec) ANIHIP_LAUNCH_FWD3(8, 4, true); else ANIHIP_LAUNCH_FWD3(8, 4, false);
    } else {
        if (rec) ANIHIP_LAUNCH_FWD3(4, 8, true); else ANIHIP_LAUNCH_FWD3(4, 8, false);
    }
#undef ANIHIP_LAUNCH_FWD3
    ANIHIP_CHECK_HIP(hipGetLastError());
    (void)status;
    return 0;
}

extern "C" int anihip_aev_forward(void *stream, const anihip_aev_params *p, const float *table,
                                  int64_t n_atoms, int64_t lo, int64_t hi, const int32_t *species,
                                  const uint32_t *meta, const float *ent, float *aev, uint32_t *slab_mask,
                                  uint32_t *status)
{
    return aev_forward(stream, p, table, n_atoms, lo, hi, species, meta, ent, aev, slab_mask, status, false);
}

extern "C" int anihip_aev_forward_update(void *stream, const anihip_aev_params *p, const float *table,
                                         int64_t n_atoms, int64_t lo, int64_t hi, const int32_t *species,
                                         const uint32_t *meta, const float *ent, float *aev, const uint32_t *prev_mask,
                                         uint32_t *slab_mask, uint32_t *status)
{
    ANIHIP_REQUIRE(slab_mask && prev_mask && slab_mask != prev_mask,
                   "anihip_aev_forward_update needs the slab flags of the previous call (prev_mask) and a second buffer for this call's");
    ANIHIP_REQUIRE(p && tuned_grid(p), "rows are updated in place on the 16 / 8x4 / 4x8 grids only");
    return aev_forward(stream, p, table, n_atoms, lo, hi, species, meta, ent, aev, slab_mask, status, true, prev_mask);
}

extern "C" int anihip_aev_jvp(void *stream, const anihip_aev_params *p, const float *table, int64_t n_atoms,
                              int64_t lo, int64_t hi, const int32_t *species, const uint32_t *meta,
                              const float *ent, const float *tangent, float *daev, uint32_t *status)
{
    ANIHIP_REQUIRE(p && table && species && meta && ent && tangent && daev, "null pointer argument");
    ANIHIP_REQUIRE(0 <= lo && lo <= hi && hi <= n_atoms, "central range outside 0..n_atoms");
    if (!tuned_grid(p))   // any other grid: the JVP instantiation of the general kernel
        return aev_forward_generic((hipStream_t)stream, p, table, lo, hi, species, meta, ent, daev, tangent, nullptr);
    AevArgs a;
    if (int rc = make_args(p, &a)) return rc;
    if (hi == lo) return 0;
    dim3 grid(persistent_blocks(hi - lo, FWD_WPB, 4)), block(FWD_WPB * WAVE);
    if (p->n_shf_a == 8)
        hipLaunchKernelGGL((k_aev_fwd<8, 4, true>), grid, block, 0, (hipStream_t)stream, a, table, lo, hi, species,
                           meta, (const float4 *)ent, daev, (uint32_t *)nullptr, tangent);
    else
        hipLaunchKernelGGL((k_aev_fwd<4, 8, true>), grid, block, 0, (hipStream_t)stream, a, table, lo, hi, species,
                           meta, (const float4 *)ent, daev, (uint32_t *)nullptr, tangent);
    ANIHIP_CHECK_HIP(hipGetLastError());
    (void)status;
    return 0;
}

static int aev_backward(void *stream, const anihip_aev_params *p, const float *table, int64_t n_atoms, int64_t lo,
                        int64_t hi, const int32_t *species, const uint32_t *meta, const float *ent,
                        const float *grad_aev, float *grad_coords, double *virial, const uint32_t *slab_mask,
                        int32_t flags)
{
    ANIHIP_REQUIRE(p && table && species && meta && ent && grad_aev && grad_coords, "null pointer argument");
    ANIHIP_REQUIRE(0 <= lo && lo <= hi && hi <= n_atoms, "central range outside 0..n_atoms");
    if (!tuned_grid(p)) {
        ANIHIP_REQUIRE(!slab_mask, "slab masks exist for the 16 / 8x4 / 4x8 grids only");
        if (virial) zero_words_async((hipStream_t)stream, virial, 9 * sizeof(double));
        return aev_backward_generic((hipStream_t)stream, p, table, lo, hi, species, meta, ent, grad_aev, grad_coords, virial,
                                    (flags & ANIHIP_BWD_FIXED_POINT) != 0);
    }
    AevArgs a;
    if (int rc = make_args(p, &a)) return rc;
    if (virial) zero_words_async((hipStream_t)stream, virial, 9 * sizeof(double));
    if (hi == lo) return 0;
    dim3 grid(persistent_blocks(hi - lo, BWD_WPB, (ANIHIP_BWD_WAVES * 4 + BWD_WPB - 1) / BWD_WPB)), block(BWD_WPB * WAVE);
    const float4 *e4 = (const float4 *)ent;
    hipStream_t st = (hipStream_t)stream;
    const bool symmetric = (flags & ANIHIP_BWD_SYMMETRIC) != 0, fixed = (flags & ANIHIP_BWD_FIXED_POINT) != 0;
    const int64_t glo = symmetric ? lo : 0, ghi = symmetric ? hi : 0;   // rows the radial gather may read
    const bool rec = (p->flags & ANIHIP_AEV_REC_BWD) != 0 && ANIHIP_BWD_REC;
#define ANIHIP_LAUNCH_BWD(NA_, NZ_, VIR_, FIX_)                                                                              \
    if (rec)                                                                                                                 \
        hipLaunchKernelGGL((k_aev_bwd<NA_, NZ_, VIR_, FIX_, true>), grid, block, 0, st, a, table, lo, hi, species, meta, e4, \
                           grad_aev, grad_coords, virial, slab_mask, glo, ghi);                                              \
    else                                                                                                                     \
        hipLaunchKernelGGL((k_aev_bwd<NA_, NZ_, VIR_, FIX_, false>), grid, block, 0, st, a, table, lo, hi, species, meta, e4, \
                           grad_aev, grad_coords, virial, slab_mask, glo, ghi)
    const int variant = (p->n_shf_a == 8 ? 0 : 4) + (virial ? 2 : 0) + (fixed ? 1 : 0);
    switch (variant) {
        case 0: ANIHIP_LAUNCH_BWD(8, 4, false, false); break;
        case 1: ANIHIP_LAUNCH_BWD(8, 4, false, true); break;
        case 2: ANIHIP_LAUNCH_BWD(8, 4, true, false); break;
        case 3: ANIHIP_LAUNCH_BWD(8, 4, true, true); break;
        case 4: ANIHIP_LAUNCH_BWD(4, 8, false, false); break;
        case 5: ANIHIP_LAUNCH_BWD(4, 8, false, true); break;
        case 6: ANIHIP_LAUNCH_BWD(4, 8, true, false); break;
        default: ANIHIP_LAUNCH_BWD(4, 8, true, true); break;
    }
#undef ANIHIP_LAUNCH_BWD
    ANIHIP_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int anihip_aev_backward(void *stream, const anihip_aev_params *p, const float *table,
                                   int64_t n_atoms, int64_t lo, int64_t hi, const int32_t *species,
                                   const uint32_t *meta, const float *ent, const float *grad_aev,
                                   const uint32_t *slab_mask, int32_t flags, float *grad_coords,
                                   uint32_t *status)
{
    (void)status;
    return aev_backward(stream, p, table, n_atoms, lo, hi, species, meta, ent, grad_aev, grad_coords, nullptr,
                        slab_mask, flags);
}

extern "C" int anihip_aev_backward_virial(void *stream, const anihip_aev_params *p, const float *table,
                                          int64_t n_atoms, int64_t lo, int64_t hi, const int32_t *species,
                                          const uint32_t *meta, const float *ent, const float *grad_aev,
                                          const uint32_t *slab_mask, int32_t flags, float *grad_coords,
                                          double *virial, uint32_t *status)
{
    (void)status;
    ANIHIP_REQUIRE(virial, "null pointer argument");
    return aev_backward(stream, p, table, n_atoms, lo, hi, species, meta, ent, grad_aev, grad_coords, virial,
                        slab_mask, flags);
}

#ifdef ANIHIP_TRACE
extern "C" int anihip_dev_trace_read(unsigned long long *dst /* host, 2048 x 10 */)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(anihip::g_fwd3_trace), sizeof(unsigned long long) * 2048 * 10);
}
#endif

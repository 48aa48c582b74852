// The fused network kernel of libanihip: one workgroup takes a 64-atom tile of one species through one ensemble member from the
// AEV rows to d E / d AEV (csrc/mlp.hip holds the host entry points that launch it, the layer-by-layer kernels and the
// preparation kernels).  Replaces mnp::run (csrc/mnp.cpp:32-232) and BmmEnsemble (nn/_infer.py:61-216).
#include "mlp_fused.h"

#include <type_traits>

namespace anihip {

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

// wave-uniform dispatch on the active part of a wave's accumulators in a phase (compile-time inside)
#define FR_UNIT(u, CALL)                                                                    \
    if ((u).nrb == RB && (u).nba >= NB) { constexpr int RBA = RB, NBA = NB; CALL; }         \
    else if ((u).nrb == RB && (u).nba == 1) { constexpr int RBA = RB, NBA = 1; CALL; }      \
    else if ((u).nrb == 1) { constexpr int RBA = 1, NBA = 1; CALL; }

// L0B: the layer-0 backward as phase 5 of the kernel (owner order, d E / d AEV accumulated in place; RB = 2, NB = 1 only)
// TRAIN: the forward half of a training step -- the hidden activations and the backward's per-layer gradients are also
// written to global memory (FusedArgs::tr_*), from the registers of the epilogues that produce them
// B2: the backward GEMMs (phases 3, 4, 5) with two products instead of three (ANIHIP_MLP_FLAG_BWD_TWO_PRODUCTS)
// H1C / H2C / H3C > 0: the widths of the networks as compile-time constants (launches restricted to ONE species,
// FusedArgs::only_species: the k loops unroll, the unit dealing folds) -- instantiated for the ANI-2x hydrogen (256 / 192 / 160) and
// oxygen / nitrogen (192 / 160 / 128), carbon (224 / 192 / 160) and sulfur / halogen (160 / 128 / 96, also ANI-1x hydrogen) networks;
// 0: read from the species table
template <int RB, int NB, int ACT, bool L0B, bool TRAIN = false, bool B2 = false, int H1C = 0, int H2C = 0, int H3C = 0>   // ACT: 0 = CELU(alpha), 1 = GELU (exact, erf)
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
    int t_lo = 0;      // owner order restricted to ONE species (only_species >= 0): its tiles t_lo .. n_tiles - 1
    for (int t = 0; t < g.S; ++t) {
        const int nt = (g.ctl[CTL_CNT + t] + ROWS - 1) / ROWS;
        if (g.owner && g.only_species >= 0) {
            if (t < g.only_species) t_lo += nt;
            if (t <= g.only_species) n_tiles += nt;
        } else {
            n_tiles += nt;
        }
    }
    const int n_items = n_tiles * g.M;
    int item = t_lo + blockIdx.x;
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
    // tile queue (owner order over single tiles): s_tab[1] = position in tile_order of the tile this workgroup takes next
    // (only the run-time-width instantiation with phase 5 serves mid-size systems.  The draw is inline assembly: a global atomic
    // the compiler can see counts as a possible write to every table this kernel reads through scalar loads -- tile entries,
    // bounds, biases -- and turns them all into per-lane vector loads: 116 spilled registers)
    constexpr bool DYN = L0B && !B2;
#define FR_DYN() (DYN && g.queue != nullptr)
    auto draw_tile = [&]() {   // one lane: the next position of the queue
        unsigned q;
        const int zero = 0, one = 1;
        // (s_nop: the pointer may come out of a v_readlane right ahead of this block -- the scalar register file is spilled to
        // vector lanes in this kernel -- and a vector-memory instruction must not read a scalar register a vector instruction
        // wrote less than five wait states earlier; the compiler pads its own instructions, not the inside of an asm block)
        asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(q) : "v"(zero), "v"(one), "s"(g.queue));
        return q;
    };
    if (FR_DYN() && threadIdx.x == 0) s_tab[1] = draw_tile();
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
                    // (the tile drawn while the tile before this one was in its last member)
                    // (the counter starts at zero: position q of the queue is tile t_lo + grid + q of the launch's range)
                    if (FR_DYN()) tile_n = min(t_lo + (int)gridDim.x + __builtin_amdgcn_readfirstlane((int)s_tab[1]), n_tiles);
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
        const int H1 = H1C ? H1C : fs.H1, H2 = H2C ? H2C : fs.H2, H3 = H3C ? H3C : fs.H3;
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
            if constexpr (TRAIN) {
                // the weight-gradient kernels take the fp16 scale of their act0 / act1 operands from the largest |act0| of the
                // launch (train.h AMAX_STAGE_ACT0): one atomic per item.  Inline assembly for the reason given at draw_tile
                // (a visible global atomic turns the kernel's scalar loads into vector loads), s_nop for the same hazard.
                if (tid == 0) {
                    const int off = ((AMAX_STAGE_ACT0 * MAX_S + s) * AMAX_SLOTS + (int)(blockIdx.x & (AMAX_SLOTS - 1))) * 4;
                    const unsigned bits = __float_as_uint(a0max);
                    asm volatile("s_nop 4\n\tglobal_atomic_umax %0, %1, %2" : : "v"(off), "v"(bits), "s"(g.amax));
                }
            }
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
        // (every thread has also read the queue position at the head of this item: the last member draws the one after it)
        if (FR_DYN() && tid == 0 && mem == Mi - 1) s_tab[1] = draw_tile();
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
#undef FR_DYN

const void *fused_kernel(int variant)
{
    switch (variant) {
        case FUSED_CELU_L0B: return (const void *)k_mlp_fused<2, 1, 0, true>;
        case FUSED_CELU_L0B_B2: return (const void *)k_mlp_fused<2, 1, 0, true, false, true>;
        case FUSED_GELU: return (const void *)k_mlp_fused<2, 1, 1, false>;
        case FUSED_TRAIN: return (const void *)k_mlp_fused<2, 1, 0, false, true>;
        case FUSED_CELU_L0B_256: return (const void *)k_mlp_fused<2, 1, 0, true, false, false, 256, 192, 160>;
        case FUSED_CELU_L0B_192: return (const void *)k_mlp_fused<2, 1, 0, true, false, false, 192, 160, 128>;
        case FUSED_CELU_L0B_224: return (const void *)k_mlp_fused<2, 1, 0, true, false, false, 224, 192, 160>;
        case FUSED_CELU_L0B_160: return (const void *)k_mlp_fused<2, 1, 0, true, false, false, 160, 128, 96>;
        default: return (const void *)k_mlp_fused<2, 1, 0, false>;
    }
}

void launch_fused(int variant, unsigned grid, size_t lds_bytes, hipStream_t stream, const FusedArgs &f)
{
    switch (variant) {
        case FUSED_CELU_L0B: hipLaunchKernelGGL((k_mlp_fused<2, 1, 0, true>), dim3(grid), dim3(512), lds_bytes, stream, f); break;
        case FUSED_CELU_L0B_B2: hipLaunchKernelGGL((k_mlp_fused<2, 1, 0, true, false, true>), dim3(grid), dim3(512), lds_bytes, stream, f); break;
        case FUSED_GELU: hipLaunchKernelGGL((k_mlp_fused<2, 1, 1, false>), dim3(grid), dim3(512), lds_bytes, stream, f); break;
        case FUSED_TRAIN: hipLaunchKernelGGL((k_mlp_fused<2, 1, 0, false, true>), dim3(grid), dim3(512), lds_bytes, stream, f); break;
        case FUSED_CELU_L0B_256: hipLaunchKernelGGL((k_mlp_fused<2, 1, 0, true, false, false, 256, 192, 160>), dim3(grid), dim3(512), lds_bytes, stream, f); break;
        case FUSED_CELU_L0B_192: hipLaunchKernelGGL((k_mlp_fused<2, 1, 0, true, false, false, 192, 160, 128>), dim3(grid), dim3(512), lds_bytes, stream, f); break;
        case FUSED_CELU_L0B_224: hipLaunchKernelGGL((k_mlp_fused<2, 1, 0, true, false, false, 224, 192, 160>), dim3(grid), dim3(512), lds_bytes, stream, f); break;
        case FUSED_CELU_L0B_160: hipLaunchKernelGGL((k_mlp_fused<2, 1, 0, true, false, false, 160, 128, 96>), dim3(grid), dim3(512), lds_bytes, stream, f); break;
        default: hipLaunchKernelGGL((k_mlp_fused<2, 1, 0, false>), dim3(grid), dim3(512), lds_bytes, stream, f); break;
    }
}

}  // namespace anihip

// AEV forward and analytic backward for ANY symmetry-function grid (n_shf_r <= 32, n_shf_a <= 16, n_shf_z <= 16): the
// twin of the reference's templated cuAEV kernels for arbitrary ShfR / ShfA / ShfZ lengths (csrc/aev.cu:1687-1777,
// aev/_computer.py:602-666 `from_constants`).  The grids of the published models (16 radial shifts, 8 x 4 or 4 x 8 angular
// terms) run through the tuned kernels of aev.hip; these kernels are the general path: same neighbor rows, same layout
// of the AEV row (radial blocks by species, angular blocks by the triu index of the species pair, sub-index a * nZ + z),
// same contract of the backward (central atoms lo..hi, gradients PUSHED to the neighbors with atomics so that a shard
// needs nothing from other ranks; optional virial; optional 64-bit fixed-point accumulation).
//
// One wave per central atom.  Forward: neighbor terms once per neighbor into the wave's LDS; every output element is a
// sum over the neighbors (pairs) of one species (pair), taken with lanes = neighbors (pairs) and one DPP wave sum -- no
// atomics, deterministic.  Backward: lanes = neighbors (radial) / pairs (angular), per-neighbor gradient sums in LDS
// (ds_add_f32), one global atomic per component per neighbor.
//
// Maths restated from the reference (paths relative to /root/reference/torchani/): aev/_terms.py:99-104,171-186 (radial),
// :34-55,324-325,339-343 (angular; cos(theta - ShfZ) expanded with cos(theta) = 0.95 cos(angle)), cutoffs.py:71-101,
// aev/_computer.py:302-350 (layout); the derivatives are the chain rule on those expressions.
#include "anihip_common.h"

namespace anihip {

constexpr int GEN_WPB = 4;
constexpr float G_LOG2E = 1.4426950408889634f;
constexpr float G_PI = 3.14159265358979323846f;
constexpr int GEN_MAXA = 16, GEN_MAXZ = 16, GEN_MAXR = 32;

struct GenArgs {
    int S, nR, nA, nZ, L, radlen;
    float Rcr, Rca, EtaR, EtaA, Zeta;
    int smooth;
};

struct GenHdr {
    uint32_t start;
    int nA, nF;
    uint64_t pkA, pkF;
};

__device__ __forceinline__ GenHdr gen_hdr(const uint32_t *meta, int64_t i)
{
    const uint32_t *m = meta + (size_t)i * META_W;
    GenHdr h;
    h.start = m[0];
    h.nA = (int)(m[1] & 0xFFFFu);
    h.nF = (int)(m[1] >> 16);
    h.pkA = (uint64_t)m[2] | ((uint64_t)m[3] << 32);
    h.pkF = (uint64_t)m[4] | ((uint64_t)m[5] << 32);
    return h;
}

__device__ __forceinline__ int gen_cnt(uint64_t pk, int t) { return (int)((pk >> (8 * t)) & 255u); }

// {fc, d fc / d r} of either cutoff envelope
__device__ __forceinline__ float2 gen_cutoff(float r, float rc, bool smooth)
{
    if (smooth) {   // cutoffs.py:84-101
        const float q = r / rc;
        const float m1 = (1.0f - q) * (1.0f + q);
        const float im = 1.0f / fmaxf(1e-10f, m1);
        const float f = __builtin_amdgcn_exp2f((1.0f - im) * G_LOG2E);
        const float df = m1 - 1e-10f >= 0.0f ? -2.0f * r / (rc * rc) * f * im * im : 0.0f;
        return make_float2(f, df);
    }
    const float x = r / rc;   // cutoffs.py:71-81; cosf / sinf of the hardware take revolutions
    return make_float2(0.5f * __builtin_amdgcn_cosf(0.5f * x) + 0.5f, -0.5f * G_PI / rc * __builtin_amdgcn_sinf(0.5f * x));
}

// (j, k) of the t-th pair of a block: rectangle for two species, row-major upper triangle inside one
__device__ __forceinline__ void gen_pair(bool same, int t, int n1, int n2, int &j, int &k)
{
    if (!same) {
        j = t / n2;
        k = t - j * n2;
    } else {   // t = j (2 n - j - 1) / 2 + (k - j - 1), 0 <= j < k < n
        const float nn = (float)(2 * n1 - 1);
        j = (int)((nn - sqrtf(fmaxf(nn * nn - 8.0f * (float)t, 0.f))) * 0.5f);
        j = max(0, min(j, n1 - 2));
        while (j > 0 && (j * (2 * n1 - j - 1)) / 2 > t) --j;
        while (((j + 1) * (2 * n1 - j - 2)) / 2 <= t) ++j;
        k = j + 1 + (t - (j * (2 * n1 - j - 1)) / 2);
    }
}

__device__ __forceinline__ int gen_triu(int S, int a, int b) { return a * S - (a * (a - 1)) / 2 + (b - a); }

// ---- staging of a row: unit vector + distance, angular cutoff (value, derivative) ------------------------------------
struct GenStage {
    float4 *ur;     // [MAXR] unit vector, r
    float2 *fca;    // [MAXR] fc_A, fc_A'   (angular-range entries)
    float2 *fcr;    // [MAXR] fc_R, fc_R'
    int *jat;       // [MAXR] neighbor atom
};

__device__ __forceinline__ void gen_stage_row(const GenArgs &a, const GenHdr &h, const float4 *ent, const GenStage &s)
{
    const int lane = lane_id(), nR = h.nA + h.nF;
    for (int e = lane; e < nR; e += WAVE) {
        const float4 d = ent[h.start + e];
        const float r2 = d.x * d.x + d.y * d.y + d.z * d.z;
        const float r = sqrtf(r2), inv = 1.0f / r;
        s.ur[e] = make_float4(d.x * inv, d.y * inv, d.z * inv, r);
        s.fcr[e] = gen_cutoff(r, a.Rcr, a.smooth != 0);
        s.fca[e] = e < h.nA ? gen_cutoff(r, a.Rca, a.smooth != 0) : make_float2(0.f, 0.f);
        s.jat[e] = (int)(__float_as_uint(d.w) & IDX_MASK);
    }
    wave_sync();
}

// ======================================================================================================================
// forward
// ======================================================================================================================
// JVP: instead of the row, its derivative along the coordinate direction `tangent` [n_atoms][3] (J t: the reference's
// cuaev double backward, csrc/aev.cu:1986-2015, templated there on any grid as well).  With d = r_j - r_i of a neighbor:
// d' = t_j - t_i, r' = u . d', u' = (d' - u r') / r; every radial / angular term is replaced by its derivative along that
// motion (oracle/ani_oracle.c ani_oracle_aev_jvp states the same chain rule in fp64).
// slab_mask (optional, rows of at most 1024 columns): bit j of slab_mask[i] <=> columns 32 j .. 32 j + 31 of row i can be
// non-zero -- PLAIN column order (a general grid has no 16 / 32-column blocks to line up with slabs: a block of a present
// species (pair) flags every slab it overlaps); the networks' layer 0 skips the slabs no atom of a tile flags.
template <bool JVP>
__global__ __launch_bounds__(GEN_WPB * WAVE) void k_aev_fwd_gen(GenArgs a, const float *__restrict__ tab, int64_t lo,
                                                                int64_t hi, const int32_t *__restrict__ species,
                                                                const uint32_t *__restrict__ meta,
                                                                const float4 *__restrict__ ent, float *__restrict__ aev,
                                                                const float *__restrict__ tangent,
                                                                uint32_t *__restrict__ slab_mask)
{
    __shared__ float4 s_ur[GEN_WPB][MAXR];
    __shared__ float2 s_fca[GEN_WPB][MAXR];
    __shared__ float2 s_fcr[GEN_WPB][MAXR];
    __shared__ int s_j[GEN_WPB][MAXR];
    __shared__ float4 s_td[JVP ? GEN_WPB : 1][JVP ? MAXR : 1];   // u', r' of every neighbor (JVP)
    const int wib = threadIdx.x >> 6, lane = lane_id();
    const GenStage st{s_ur[wib], s_fca[wib], s_fcr[wib], s_j[wib]};
    float4 *td = s_td[JVP ? wib : 0];
    const int64_t nw = (int64_t)gridDim.x * GEN_WPB;
    const int nAZ = a.nA * a.nZ;
    auto flag_cols = [&](int c0, int c1) {   // slabs overlapped by the columns c0 .. c1 - 1
        const int j0 = c0 >> 5, j1 = (c1 - 1) >> 5;
        return (j1 >= 31 ? 0xFFFFFFFFu : ((1u << (j1 + 1)) - 1u)) & ~((1u << j0) - 1u);
    };
    for (int64_t i = lo + blockIdx.x * (int64_t)GEN_WPB + wib; i < hi; i += nw) {
        float *out = aev + (size_t)i * a.L;   // (the caller's pointer is that of row 0, also for a shard's [hi - lo, L] buffer)
        GenHdr h = gen_hdr(meta, i);
        if (species[i] < 0) { h.nA = 0; h.nF = 0; h.pkA = 0ull; h.pkF = 0ull; }
        uint32_t flags = 0u;
        // (every element of the row is written exactly once: blocks without a neighbor (pair) as zeros)
        if (h.nA + h.nF > 0) gen_stage_row(a, h, ent, st);
        if (JVP && h.nA + h.nF > 0) {
            const float tix = tangent[3 * i], tiy = tangent[3 * i + 1], tiz = tangent[3 * i + 2];
            for (int e = lane; e < h.nA + h.nF; e += WAVE) {
                const size_t jn = (size_t)st.jat[e];
                const float4 U = st.ur[e];
                const float dx = tangent[3 * jn] - tix, dy = tangent[3 * jn + 1] - tiy, dz = tangent[3 * jn + 2] - tiz;
                const float rd = U.x * dx + U.y * dy + U.z * dz, ir = 1.0f / U.w;
                td[e] = make_float4((dx - U.x * rd) * ir, (dy - U.y * rd) * ir, (dz - U.z * rd) * ir, rd);
            }
            wave_sync();
        }
        // ---- radial: element (s, k) = sum over the neighbors of species s (its angular-range run and its far run) ----
        int offA = 0, offF = h.nA;
        for (int s = 0; s < a.S; ++s) {
            const int cA = gen_cnt(h.pkA, s), cF = gen_cnt(h.pkF, s);
            if (cA + cF > 0) {
                float keep = 0.f;   // lane k keeps element (s, k)   (nR <= 32)
                for (int k = 0; k < a.nR; ++k) {
                    const float sh = tab[TAB_SHFR + k];
                    float acc = 0.f;
                    for (int q = lane; q < cA + cF; q += WAVE) {
                        const int e = q < cA ? offA + q : offF + (q - cA);
                        const float dr = st.ur[e].w - sh;
                        const float ex = 0.25f * __builtin_amdgcn_exp2f(-a.EtaR * G_LOG2E * dr * dr);
                        if (JVP) acc += (ex * st.fcr[e].y - 2.0f * a.EtaR * dr * ex * st.fcr[e].x) * td[e].w;
                        else acc += ex * st.fcr[e].x;
                    }
                    const float tot = wave_sum(acc);
                    if (lane == k) keep = tot;
                }
                if (lane < a.nR) out[s * a.nR + lane] = keep;
                flags |= flag_cols(s * a.nR, (s + 1) * a.nR);
            } else if (lane < a.nR) {
                out[s * a.nR + lane] = 0.f;
            }
            offA += cA;
            offF += cF;
        }
        // ---- angular: block (s1 <= s2), element (a, z) = sum over the pairs of the block ----
        int o1 = 0;
        for (int s1 = 0; s1 < a.S; ++s1) {
            const int n1 = gen_cnt(h.pkA, s1);
            int o2 = o1;
            for (int s2 = s1; s2 < a.S; ++s2) {
                const int n2 = gen_cnt(h.pkA, s2);
                const bool same = s1 == s2;
                const int np = same ? (n1 * (n1 - 1)) / 2 : n1 * n2;
                float *blk = out + a.radlen + gen_triu(a.S, s1, s2) * nAZ;
                if (np == 0) {
                    for (int q = lane; q < nAZ; q += WAVE) blk[q] = 0.f;
                } else {
                    float keep[(GEN_MAXA * GEN_MAXZ) / WAVE];
#pragma unroll
                    for (int c = 0; c < (GEN_MAXA * GEN_MAXZ) / WAVE; ++c) keep[c] = 0.f;
                    for (int t0 = 0; t0 < np; t0 += WAVE) {
                        const int t = t0 + lane;
                        const bool v = t < np;
                        int j = 0, k = same ? 1 : 0;
                        if (v) gen_pair(same, t, n1, n2, j, k);
                        const int e1 = o1 + j, e2 = (same ? o1 : o2) + k;
                        const float4 U1 = st.ur[v ? e1 : 0], U2 = st.ur[v ? e2 : 0];
                        const float2 F1 = st.fca[v ? e1 : 0], F2 = st.fca[v ? e2 : 0];
                        const float fcc = v ? F1.x * F2.x : 0.f;
                        const float ct = 0.95f * (U1.x * U2.x + U1.y * U2.y + U1.z * U2.z);
                        const float sn = sqrtf(fmaxf(1.0f - ct * ct, 0.f));
                        const float rm = 0.5f * (U1.w + U2.w);
                        // JVP: derivatives of cos, sin, the mean distance and the cutoff product along the motion
                        float ctd = 0.f, snd = 0.f, rmd = 0.f, fcd = 0.f;
                        if (JVP) {
                            const float4 T1 = td[v ? e1 : 0], T2 = td[v ? e2 : 0];
                            ctd = 0.95f * (T1.x * U2.x + T1.y * U2.y + T1.z * U2.z + U1.x * T2.x + U1.y * T2.y + U1.z * T2.z);
                            snd = -ct * ctd / sn;   // (sn >= 0.31: |cos| is scaled by 0.95)
                            rmd = 0.5f * (T1.w + T2.w);
                            fcd = v ? F1.y * T1.w * F2.x + F1.x * F2.y * T2.w : 0.f;
                        }
                        float f2[GEN_MAXA], g2[JVP ? GEN_MAXA : 1];   // f2 = Gaussian x cutoffs; g2 = its derivative (JVP)
#pragma unroll
                        for (int u = 0; u < GEN_MAXA; ++u) {
                            const float dr = rm - tab[TAB_SHFA + (u < a.nA ? u : 0)];
                            const float ga = __builtin_amdgcn_exp2f(-a.EtaA * G_LOG2E * dr * dr);
                            f2[u] = ga * fcc;
                            if (JVP) g2[u] = ga * (fcd - 2.0f * a.EtaA * dr * rmd * fcc);
                        }
                        for (int z = 0; z < a.nZ; ++z) {
                            const float cz = tab[TAB_COSZ + z], sz = tab[TAB_SINZ + z];
                            const float hh = fmaxf(0.5f + 0.5f * (ct * cz + sn * sz), 1e-30f);
                            const float lg = __builtin_amdgcn_logf(hh);
                            const float f1 = 2.0f * __builtin_amdgcn_exp2f(a.Zeta * lg);
                            // d f1 / dt = 2 Zeta h^(Zeta - 1) h',  h' = (cos' cos ShfZ + sin' sin ShfZ) / 2
                            const float f1d = JVP ? a.Zeta * __builtin_amdgcn_exp2f((a.Zeta - 1.0f) * lg) * (ctd * cz + snd * sz) : 0.f;
#pragma unroll
                            for (int u = 0; u < GEN_MAXA; ++u) {
                                if (u < a.nA) {   // (wave-uniform)
                                    const float term = JVP ? f1d * f2[u] + f1 * g2[u] : f1 * f2[u];
                                    const float tot = wave_sum(v ? term : 0.f);
                                    const int q = u * a.nZ + z;
#pragma unroll
                                    for (int c = 0; c < (GEN_MAXA * GEN_MAXZ) / WAVE; ++c)
                                        if ((q >> 6) == c && (q & 63) == lane) keep[c] += tot;
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int c = 0; c < (GEN_MAXA * GEN_MAXZ) / WAVE; ++c)
                        if (c * WAVE + lane < nAZ) blk[c * WAVE + lane] = keep[c];
                    const int c0 = a.radlen + gen_triu(a.S, s1, s2) * nAZ;
                    flags |= flag_cols(c0, c0 + nAZ);
                }
                o2 += n2;
            }
            o1 += n1;
        }
        if (slab_mask && lane == 0) slab_mask[i] = flags;
        wave_sync();
    }
}

// ======================================================================================================================
// backward
// ======================================================================================================================
template <bool FIXED>
__device__ __forceinline__ void gen_push(float *grad_coords, size_t at, int comp, float v)
{
    if (FIXED) {
        unsigned long long *acc = reinterpret_cast<unsigned long long *>(grad_coords) + 3 * at + comp;
        atomicAdd(acc, (unsigned long long)__float2ll_rn(v * 4294967296.0f));
    } else {
        atomicAdd(grad_coords + 3 * at + comp, v);
    }
}

template <bool VIRIAL, bool FIXED>
__global__ __launch_bounds__(GEN_WPB * WAVE) void k_aev_bwd_gen(GenArgs a, const float *__restrict__ tab, int64_t lo,
                                                                int64_t hi, const int32_t *__restrict__ species,
                                                                const uint32_t *__restrict__ meta,
                                                                const float4 *__restrict__ ent,
                                                                const float *__restrict__ grad_aev,
                                                                float *__restrict__ grad_coords,
                                                                double *__restrict__ virial)
{
    __shared__ float4 s_ur[GEN_WPB][MAXR];
    __shared__ float2 s_fca[GEN_WPB][MAXR];
    __shared__ float2 s_fcr[GEN_WPB][MAXR];
    __shared__ int s_j[GEN_WPB][MAXR];
    __shared__ float s_g[GEN_WPB][3][MAXR];   // gradient on the neighbors of the row
    const int wib = threadIdx.x >> 6, lane = lane_id();
    const GenStage st{s_ur[wib], s_fca[wib], s_fcr[wib], s_j[wib]};
    float *gx = s_g[wib][0], *gy = s_g[wib][1], *gz = s_g[wib][2];
    const int64_t nw = (int64_t)gridDim.x * GEN_WPB;
    const int nAZ = a.nA * a.nZ;
    float vir[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) vir[q] = 0.f;
    for (int64_t i = lo + blockIdx.x * (int64_t)GEN_WPB + wib; i < hi; i += nw) {
        if (species[i] < 0) continue;
        const GenHdr h = gen_hdr(meta, i);
        const int nR = h.nA + h.nF;
        if (nR == 0) continue;
        gen_stage_row(a, h, ent, st);
        const float *w = grad_aev + (size_t)i * a.L;   // (pointer of row 0, as in the forward)
        float ox = 0.f, oy = 0.f, oz = 0.f;   // minus the gradient on the central atom, per lane
        // ---- radial (lane = neighbor): dE/dr_ij = sum_k w[s_j, k] d/dr [0.25 exp(-eta (r - s_k)^2) fc(r)] ----
        for (int e = lane; e < nR; e += WAVE) {
            const float4 U = st.ur[e];
            const float2 fc = st.fcr[e];
            const int t = (int)(__float_as_uint(ent[h.start + e].w) >> 28);
            const float *wr = w + t * a.nR;
            float dR = 0.f;
            for (int k = 0; k < a.nR; ++k) {
                const float dr = U.w - tab[TAB_SHFR + k];
                const float ex = 0.25f * __builtin_amdgcn_exp2f(-a.EtaR * G_LOG2E * dr * dr);
                dR += wr[k] * (ex * fc.y - 2.0f * a.EtaR * dr * ex * fc.x);
            }
            const float vx = dR * U.x, vy = dR * U.y, vz = dR * U.z;
            gx[e] = vx; gy[e] = vy; gz[e] = vz;
            ox += vx; oy += vy; oz += vz;
            if (VIRIAL) {
                const float dx = U.x * U.w, dy = U.y * U.w, dz = U.z * U.w;
                vir[0] += vx * dx; vir[1] += vx * dy; vir[2] += vx * dz;
                vir[3] += vy * dx; vir[4] += vy * dy; vir[5] += vy * dz;
                vir[6] += vz * dx; vir[7] += vz * dy; vir[8] += vz * dz;
            }
        }
        wave_sync();
        // ---- angular (lane = pair) ----
        int o1 = 0;
        for (int s1 = 0; s1 < a.S; ++s1) {
            const int n1 = gen_cnt(h.pkA, s1);
            int o2 = o1;
            for (int s2 = s1; s2 < a.S; ++s2) {
                const int n2 = gen_cnt(h.pkA, s2);
                const bool same = s1 == s2;
                const int np = same ? (n1 * (n1 - 1)) / 2 : n1 * n2;
                const float *ww = w + a.radlen + gen_triu(a.S, s1, s2) * nAZ;
                for (int t0 = 0; t0 < np; t0 += WAVE) {
                    const int t = t0 + lane;
                    if (t < np) {
                        int j, k;
                        gen_pair(same, t, n1, n2, j, k);
                        const int e1 = o1 + j, e2 = (same ? o1 : o2) + k;
                        const float4 U1 = st.ur[e1], U2 = st.ur[e2];
                        const float2 F1 = st.fca[e1], F2 = st.fca[e2];
                        const float c = U1.x * U2.x + U1.y * U2.y + U1.z * U2.z;
                        const float ct = 0.95f * c;
                        const float sn = sqrtf(fmaxf(1.0f - ct * ct, 0.f));
                        const float rm = 0.5f * (U1.w + U2.w);
                        float C0 = 0.f, Cth = 0.f, CR = 0.f;   // sum w f1 f2, sum w f1' f2, sum w f1 f2'
                        for (int z = 0; z < a.nZ; ++z) {
                            const float cz = tab[TAB_COSZ + z], sz = tab[TAB_SINZ + z];
                            const float hh = fmaxf(0.5f + 0.5f * (ct * cz + sn * sz), 1e-30f);
                            const float lg = __builtin_amdgcn_logf(hh);
                            const float f1 = 2.0f * __builtin_amdgcn_exp2f(a.Zeta * lg);
                            // d f1 / d theta = -Zeta h^(Zeta - 1) sin(theta - ShfZ)
                            const float df1 = -a.Zeta * __builtin_amdgcn_exp2f((a.Zeta - 1.0f) * lg) * (sn * cz - ct * sz);
                            for (int u = 0; u < a.nA; ++u) {
                                const float dr = rm - tab[TAB_SHFA + u];
                                const float f2 = __builtin_amdgcn_exp2f(-a.EtaA * G_LOG2E * dr * dr);
                                const float wz = ww[u * a.nZ + z];
                                C0 += wz * f1 * f2;
                                Cth += wz * df1 * f2;
                                CR += wz * f1 * (-2.0f * a.EtaA * dr * f2);
                            }
                        }
                        const float fcc = F1.x * F2.x;
                        const float kth = Cth * fcc * (-0.95f / sn);   // dE / d cos(angle)
                        const float k1 = 0.5f * CR * fcc + C0 * F1.y * F2.x;
                        const float k2 = 0.5f * CR * fcc + C0 * F1.x * F2.y;
                        const float i1 = 1.0f / U1.w, i2 = 1.0f / U2.w;
                        const float g1x = kth * (U2.x - c * U1.x) * i1 + k1 * U1.x;
                        const float g1y = kth * (U2.y - c * U1.y) * i1 + k1 * U1.y;
                        const float g1z = kth * (U2.z - c * U1.z) * i1 + k1 * U1.z;
                        const float g2x = kth * (U1.x - c * U2.x) * i2 + k2 * U2.x;
                        const float g2y = kth * (U1.y - c * U2.y) * i2 + k2 * U2.y;
                        const float g2z = kth * (U1.z - c * U2.z) * i2 + k2 * U2.z;
                        atomicAdd(&gx[e1], g1x); atomicAdd(&gy[e1], g1y); atomicAdd(&gz[e1], g1z);
                        atomicAdd(&gx[e2], g2x); atomicAdd(&gy[e2], g2y); atomicAdd(&gz[e2], g2z);
                        ox += g1x + g2x; oy += g1y + g2y; oz += g1z + g2z;
                        if (VIRIAL) {
                            const float d1x = U1.x * U1.w, d1y = U1.y * U1.w, d1z = U1.z * U1.w;
                            const float d2x = U2.x * U2.w, d2y = U2.y * U2.w, d2z = U2.z * U2.w;
                            vir[0] += g1x * d1x + g2x * d2x; vir[1] += g1x * d1y + g2x * d2y; vir[2] += g1x * d1z + g2x * d2z;
                            vir[3] += g1y * d1x + g2y * d2x; vir[4] += g1y * d1y + g2y * d2y; vir[5] += g1y * d1z + g2y * d2z;
                            vir[6] += g1z * d1x + g2z * d2x; vir[7] += g1z * d1y + g2z * d2y; vir[8] += g1z * d1z + g2z * d2z;
                        }
                    }
                }
                o2 += n2;
            }
            o1 += n1;
        }
        wave_sync();
        // ---- push: every neighbor its sum, the central atom minus the total ----
        for (int e = lane; e < nR; e += WAVE) {
            const size_t jn = (size_t)st.jat[e];
            gen_push<FIXED>(grad_coords, jn, 0, gx[e]);
            gen_push<FIXED>(grad_coords, jn, 1, gy[e]);
            gen_push<FIXED>(grad_coords, jn, 2, gz[e]);
        }
        const float tx = wave_sum(ox), ty = wave_sum(oy), tz = wave_sum(oz);
        if (lane == 0) {
            gen_push<FIXED>(grad_coords, (size_t)i, 0, -tx);
            gen_push<FIXED>(grad_coords, (size_t)i, 1, -ty);
            gen_push<FIXED>(grad_coords, (size_t)i, 2, -tz);
        }
        wave_sync();
    }
    if (VIRIAL) {
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const float tot = wave_sum(vir[q]);
            if (lane == 0 && tot != 0.f) atomicAdd(virial + q, (double)tot);
        }
    }
}

static int gen_blocks(int64_t n_central)
{
    int64_t b = (n_central + GEN_WPB - 1) / GEN_WPB;
    if (b < 1) b = 1;
    if (b > 256 * 4) b = 256 * 4;
    return (int)b;
}

static int gen_args(const anihip_aev_params *p, GenArgs *a)
{
    ANIHIP_REQUIRE(p->num_species >= 1 && p->num_species <= MAX_S - 1, "num_species must be 1..7");
    ANIHIP_REQUIRE(p->n_shf_r >= 1 && p->n_shf_r <= GEN_MAXR && p->n_shf_a >= 1 && p->n_shf_a <= GEN_MAXA &&
                       p->n_shf_z >= 1 && p->n_shf_z <= GEN_MAXZ,
                   "symmetry-function grid outside n_shf_r <= 32, n_shf_a <= 16, n_shf_z <= 16 (got %d, %d x %d)",
                   p->n_shf_r, p->n_shf_a, p->n_shf_z);
    a->S = p->num_species;
    a->nR = p->n_shf_r; a->nA = p->n_shf_a; a->nZ = p->n_shf_z;
    a->radlen = a->S * a->nR;
    a->L = a->radlen + (a->S * (a->S + 1) / 2) * a->nA * a->nZ;
    a->Rcr = p->Rcr; a->Rca = p->Rca; a->EtaR = p->EtaR; a->EtaA = p->EtaA; a->Zeta = p->Zeta;
    ANIHIP_REQUIRE(p->cutoff_kind == ANIHIP_CUTOFF_COSINE || p->cutoff_kind == ANIHIP_CUTOFF_SMOOTH,
                   "cutoff_kind must be ANIHIP_CUTOFF_COSINE or ANIHIP_CUTOFF_SMOOTH");
    a->smooth = p->cutoff_kind == ANIHIP_CUTOFF_SMOOTH;
    return 0;
}

// (called by anihip_aev_forward / anihip_aev_backward* of aev.hip for grids the tuned kernels do not cover)
int aev_forward_generic(hipStream_t stream, const anihip_aev_params *p, const float *table, int64_t lo, int64_t hi,
                        const int32_t *species, const uint32_t *meta, const float *ent, float *aev, const float *tangent,
                        uint32_t *slab_mask)
{
    GenArgs a;
    if (int rc = gen_args(p, &a)) return rc;
    ANIHIP_REQUIRE(!slab_mask || a.L <= 32 * 32, "slab masks need rows of at most 1024 columns (got %d)", a.L);
    if (hi == lo) return 0;
    const dim3 grid(gen_blocks(hi - lo)), block(GEN_WPB * WAVE);
    if (tangent)
        hipLaunchKernelGGL(k_aev_fwd_gen<true>, grid, block, 0, stream, a, table, lo, hi, species, meta, (const float4 *)ent, aev,
                           tangent, slab_mask);
    else
        hipLaunchKernelGGL(k_aev_fwd_gen<false>, grid, block, 0, stream, a, table, lo, hi, species, meta, (const float4 *)ent,
                           aev, tangent, slab_mask);
    ANIHIP_CHECK_HIP(hipGetLastError());
    return 0;
}

int aev_backward_generic(hipStream_t stream, const anihip_aev_params *p, const float *table, int64_t lo, int64_t hi,
                         const int32_t *species, const uint32_t *meta, const float *ent, const float *grad_aev,
                         float *grad_coords, double *virial, bool fixed)
{
    GenArgs a;
    if (int rc = gen_args(p, &a)) return rc;
    if (hi == lo) return 0;
    const dim3 grid(gen_blocks(hi - lo)), block(GEN_WPB * WAVE);
    const float4 *e4 = (const float4 *)ent;
#define ANIHIP_LAUNCH_GEN(VIR_, FIX_)                                                                                \
    hipLaunchKernelGGL((k_aev_bwd_gen<VIR_, FIX_>), grid, block, 0, stream, a, table, lo, hi, species, meta, e4, grad_aev, \
                       grad_coords, virial)
    if (virial) {
        if (fixed) ANIHIP_LAUNCH_GEN(true, true); else ANIHIP_LAUNCH_GEN(true, false);
    } else {
        if (fixed) ANIHIP_LAUNCH_GEN(false, true); else ANIHIP_LAUNCH_GEN(false, false);
    }
#undef ANIHIP_LAUNCH_GEN
    ANIHIP_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace anihip

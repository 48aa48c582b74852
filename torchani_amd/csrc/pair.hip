// Pair potentials evaluated on the neighbor rows of nbr.hip (cutoff envelope and half-per-atom bookkeeping of
// potentials/core.py:155-207): the analytic family k_pair<KIND> -- xTB repulsion (potentials/xtb.py:17-77), ZBL screened
// nuclear repulsion (zbl.py:10-81), Lennard-Jones 12 / 6 terms (lj.py:42-108), fixed-charge Coulomb with optional MNOK
// damping (fixed_coulomb.py:8-75) -- and DFT-D3(BJ) dispersion below.
//
// One wave per central atom, lane = neighbor.  The rows are a FULL symmetric list, so atom i finishes everything that
// concerns itself from its own row: atomic energy sum_j e_ij / 2 (core.py:195-198) and gradient
// sum_j e'(d_ij) d r_ij / d r_i -- no atomics, deterministic.  (Rows from a LAMMPS full list are not symmetric: there the
// pair term is pushed to the neighbor with float atomics, flag ANIHIP_PAIR_PUSH.)
#include "anihip_common.h"

namespace anihip {

constexpr float A2B = 1.8897261258369282f;   // torchani/units.py:41

struct PairExtra {
    float v[8];   // ZBL: screening coefficients c_0..3 and exponents b_0..3
};

// bare pair energy (no envelope) and its derivative with respect to r [Angstrom]; p = table entry of the species pair
template <int KIND>
__device__ __forceinline__ void pair_eval(const float4 p, const PairExtra &x, float r, float &base, float &dbase)
{
    if constexpr (KIND == ANIHIP_PAIR_XTB) {            // {y_ab, sqrt(alpha_ab), k_ab}: y / d exp(-sqrt(alpha) d^k), d [Bohr]
        const float rb = r * A2B;
        const float pw = __builtin_amdgcn_exp2f(p.z * __builtin_amdgcn_logf(rb));   // rb^k
        const float ex = __expf(-p.y * pw);
        base = p.x / rb * ex;
        dbase = base * (-1.0f / rb - p.y * p.z * pw / rb) * A2B;
    } else if constexpr (KIND == ANIHIP_PAIR_ZBL) {     // {Za Zb, (Za^kz + Zb^kz) / k}: Za Zb / d sum_i c_i exp(-b_i d s), d [Bohr]
        const float rb = r * A2B, xr = rb * p.y;
        float phi = 0.f, dphi = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float t = x.v[i] * __expf(-x.v[4 + i] * xr);
            phi += t;
            dphi -= x.v[4 + i] * t;
        }
        const float cl = p.x / rb;
        base = cl * phi;
        dbase = (cl * dphi * p.y - cl / rb * phi) * A2B;
    } else if constexpr (KIND == ANIHIP_PAIR_LJ) {      // {4 eps_ab, sigma_ab, c12, c6}: 4 eps (c12 x^12 + c6 x^6), x = sigma / r
        const float ir = 1.0f / r, xs = p.y * ir, x2 = xs * xs, x6 = x2 * x2 * x2, x12 = x6 * x6;
        base = p.x * (p.z * x12 + p.w * x6);
        dbase = -p.x * (12.0f * p.z * x12 + 6.0f * p.w * x6) * ir;
    } else {                                            // Coulomb {q_a q_b / dielectric, 1 / eta_ab}: qq / sqrt(d^2 + 1 / eta^2), d [Bohr]
        const float rb = r * A2B;
        const float is = __builtin_amdgcn_rsqf(rb * rb + p.y * p.y);
        base = p.x * is;
        dbase = -p.x * rb * is * is * is * A2B;
    }
}

template <int KIND>
__global__ __launch_bounds__(256) void k_pair(int64_t lo, int64_t hi, const int32_t *__restrict__ species,
                                              const uint32_t *__restrict__ meta, const float4 *__restrict__ ent,
                                              const float *__restrict__ tab /* [8][8][4] per species pair */,
                                              PairExtra extra, float cutoff, int smooth, int push, int clamp_r,
                                              float *__restrict__ atomic_e, float *__restrict__ grad_coords,
                                              double *__restrict__ virial)
{
    const int lane = lane_id();
    const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    float vxx = 0.f, vyy = 0.f, vzz = 0.f, vxy = 0.f, vxz = 0.f, vyz = 0.f;
    const float inv_rc = 1.0f / cutoff, rev_rc = 0.5f / cutoff, pi_rc = 3.14159265358979f / cutoff;
    for (int64_t i = lo + blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6); i < hi; i += nw) {
        const int si = species[i];
        if (si < 0) continue;
        const uint32_t start = meta[(size_t)i * META_W], c = meta[(size_t)i * META_W + 1];
        const int nR = (int)(c & 0xFFFFu) + (int)(c >> 16);
        float e = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
        for (int e0 = 0; e0 < nR; e0 += WAVE) {
            const int k = e0 + lane;
            if (k >= nR) continue;
            const float4 d = ent[start + k];
            const uint32_t w = __float_as_uint(d.w);
            const int sj = (int)(w >> 28);
            const float r2 = d.x * d.x + d.y * d.y + d.z * d.z;
            const float inv = __builtin_amdgcn_rsqf(r2);
            const float r = clamp_r ? fmaxf(r2 * inv, 1e-7f) : r2 * inv;   // (core.py:138-139 clamp)
            if (r > cutoff) continue;
            float fc, dfc;   // envelope and its derivative (cutoffs.py:74-101)
            if (smooth) {
                const float q = r * inv_rc, m1 = (1.0f - q) * (1.0f + q);
                const float im = 1.0f / fmaxf(1e-10f, m1);
                fc = __expf(1.0f - im);
                dfc = m1 - 1e-10f >= 0.0f ? -2.0f * r * inv_rc * inv_rc * fc * im * im : 0.0f;
            } else {
                fc = 0.5f * __builtin_amdgcn_cosf(r * rev_rc) + 0.5f;
                dfc = -0.5f * pi_rc * __builtin_amdgcn_sinf(r * rev_rc);
            }
            const float4 p = reinterpret_cast<const float4 *>(tab)[si * 8 + sj];
            float base, dbase;
            pair_eval<KIND>(p, extra, r, base, dbase);
            const float eij = base * fc;
            const float de = dbase * fc + base * dfc;
            e += 0.5f * eij;
            // d r_ij / d r_i = -u_ij, u = d / r;  the pair contributes e_ij / 2 to BOTH atoms: gradient on i = -de u
            const float ux = d.x * inv, uy = d.y * inv, uz = d.z * inv;
            gx -= de * ux; gy -= de * uy; gz -= de * uz;
            if (push && grad_coords) {   // asymmetric rows: this row's half of the pair acts on the neighbor too
                float *gj = grad_coords + 3 * (size_t)(w & IDX_MASK);
                atomicAdd(gj + 0, 0.5f * de * ux); atomicAdd(gj + 1, 0.5f * de * uy); atomicAdd(gj + 2, 0.5f * de * uz);
            }
            if (virial) {   // sum over ordered pairs of (dE_i / d d_ij) (x) d_ij with E_i = sum_j e_ij / 2
                const float h = 0.5f * de;
                vxx += h * ux * d.x; vyy += h * uy * d.y; vzz += h * uz * d.z;
                vxy += h * ux * d.y; vxz += h * ux * d.z; vyz += h * uy * d.z;
            }
        }
        e = wave_sum(e);
        if (lane == 0 && atomic_e) atomic_e[i] += e;
        if (grad_coords) {
            const float sc = push ? 0.5f : 1.0f;   // (symmetric rows: the partner's row supplies the other half)
            gx = wave_sum(gx) * sc; gy = wave_sum(gy) * sc; gz = wave_sum(gz) * sc;
            if (lane == 0) {
                float *gi = grad_coords + 3 * (size_t)i;
                if (push) { atomicAdd(gi + 0, gx); atomicAdd(gi + 1, gy); atomicAdd(gi + 2, gz); }
                else { gi[0] += gx; gi[1] += gy; gi[2] += gz; }
            }
        }
    }
    if (virial) {
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

// ---- DFT-D3(BJ) two-body dispersion (potentials/dftd3.py:113-330) -------------------------------------------------
// Three passes over the rows, one wave per central atom, lane = neighbor, all gathers (the rows are full and symmetric):
//   k_d3_cn     CN_i = sum_j count(d_ij)                                                    (:256-279 _coordnums)
//   k_d3_pair   per pair the C6 interpolation over the 5 x 5 references (:281-330), e_ij, its derivative at fixed C6
//               (energy and direct gradient of atom i), and gcn_i = dE / dCN_i = sum_j (de_ij / dC6) (dC6_ij / dCN_i)
//               -- the pair (i, j) sits in both rows and C6_ji(CN_j, CN_i) = C6_ij(CN_i, CN_j), so the two halves of
//               dE / dCN_i are equal and row i alone gives the whole of it
//   k_d3_cngrad grad_i += sum_j (gcn_i + gcn_j) count'(d_ij) d d_ij / d r_i
struct D3P {
    float s6, s8, a1, a2;
    float cov[8], sq[8];
};
constexpr float D3_K1 = 16.0f, D3_K2 = 4.0f / 3.0f, D3_K3 = 4.0f, D3_EPS = 1e-35f;

__device__ __forceinline__ void d3_envelope(float r, float cutoff, int smooth, float &fc, float &dfc)
{
    if (smooth) {
        const float inv_rc = 1.0f / cutoff, q = r * inv_rc, m1 = (1.0f - q) * (1.0f + q);
        const float im = 1.0f / fmaxf(1e-10f, m1);
        fc = __expf(1.0f - im);
        dfc = m1 - 1e-10f >= 0.0f ? -2.0f * r * inv_rc * inv_rc * fc * im * im : 0.0f;
    } else {
        const float rev_rc = 0.5f / cutoff;
        fc = 0.5f * __builtin_amdgcn_cosf(r * rev_rc) + 0.5f;
        dfc = -0.5f * (3.14159265358979f / cutoff) * __builtin_amdgcn_sinf(r * rev_rc);
    }
}

// count(d) and d count / d d (d in Bohr)
__device__ __forceinline__ float d3_count(float rsum, float d, float &dcnt)
{
    const float t = __expf(-D3_K1 * (D3_K2 * rsum / d - 1.0f));
    const float c = 1.0f / (1.0f + t);
    dcnt = -c * (1.0f - c) * D3_K1 * D3_K2 * rsum / (d * d);
    return c;
}

__global__ __launch_bounds__(256) void k_d3_cn(int64_t n, const int32_t *__restrict__ species,
                                               const uint32_t *__restrict__ meta, const float4 *__restrict__ ent, D3P p,
                                               float cutoff, float *__restrict__ cn)
{
    const int lane = lane_id();
    const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t i = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6); i < n; i += nw) {
        const int si = species[i];
        float acc = 0.f;
        if (si >= 0) {
            const uint32_t start = meta[(size_t)i * META_W], c = meta[(size_t)i * META_W + 1];
            const int nR = (int)(c & 0xFFFFu) + (int)(c >> 16);
            for (int k = lane; k < nR; k += WAVE) {
                const float4 d = ent[start + k];
                const int sj = (int)(__float_as_uint(d.w) >> 28);
                const float r = fmaxf(sqrtf(d.x * d.x + d.y * d.y + d.z * d.z), 1e-7f);
                if (r > cutoff) continue;
                float dc;
                acc += d3_count(p.cov[si] + p.cov[sj], r * A2B, dc);
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) cn[i] = acc;
    }
}

__global__ __launch_bounds__(256) void k_d3_pair(int64_t n, int64_t lo, int64_t hi, const int32_t *__restrict__ species,
                                                 const uint32_t *__restrict__ meta, const float4 *__restrict__ ent,
                                                 const float4 *__restrict__ tab, D3P p, float cutoff, int smooth,
                                                 const float *__restrict__ cn, float *__restrict__ gcn,
                                                 float *__restrict__ atomic_e, float *__restrict__ grad_coords,
                                                 double *__restrict__ virial)
{
    const int lane = lane_id();
    const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    float vxx = 0.f, vyy = 0.f, vzz = 0.f, vxy = 0.f, vxz = 0.f, vyz = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6); i < n; i += nw) {
        const int si = species[i];
        if (si < 0) {
            if (lane == 0) gcn[i] = 0.f;
            continue;
        }
        const bool own = i >= lo && i < hi;
        const uint32_t start = meta[(size_t)i * META_W], c = meta[(size_t)i * META_W + 1];
        const int nR = (int)(c & 0xFFFFu) + (int)(c >> 16);
        const float cni = cn[i];
        float e = 0.f, gx = 0.f, gy = 0.f, gz = 0.f, gc = 0.f;
        for (int k = lane; k < nR; k += WAVE) {
            const float4 d = ent[start + k];
            const uint32_t w = __float_as_uint(d.w);
            const int sj = (int)(w >> 28);
            const float r2 = d.x * d.x + d.y * d.y + d.z * d.z;
            const float inv = __builtin_amdgcn_rsqf(r2);
            const float r = fmaxf(r2 * inv, 1e-7f);
            if (r > cutoff) continue;
            const float cnj = cn[w & IDX_MASK];
            // C6 and d C6 / d CN_i from the 25 reference pairs
            const float4 *t = tab + (si * 8 + sj) * 25;
            const int nv = (int)t[0].w;   // the pair's references with c6ref > 0 come first (H-H: 4 of the 25)
            // (the weights are formed relative to the largest one: coordination numbers far from every reference --
            // dense random geometries -- give weights below the fp32 range, which the reference's fp64 still resolves;
            // its 1e-35 guards, dftd3.py:322-330, are applied to the rescaled sums)
            float amin = 3.0e38f;
            for (int q = 0; q < nv; ++q) {
                const float4 ref = t[q];
                const float da = cni - ref.y, db = cnj - ref.z;
                amin = fminf(amin, D3_K3 * (da * da + db * db));
            }
            float W = 0.f, Z = 0.f, dW = 0.f, dZ = 0.f;
            for (int q = 0; q < nv; ++q) {
                const float4 ref = t[q];
                const float da = cni - ref.y, db = cnj - ref.z;
                const float L = __expf(amin - D3_K3 * (da * da + db * db));
                W += L; Z += ref.x * L;
                dW += L * da; dZ += ref.x * L * da;
            }
            const float sc = __expf(-amin);   // (0 when there is no reference at all: C6 = eps / eps = 1 like the reference)
            W = W * sc + D3_EPS; Z = Z * sc + D3_EPS;
            const float iW = 1.0f / W;
            const float c6 = Z * iW;
            const float dc6 = -2.0f * D3_K3 * sc * (dZ - c6 * dW) * iW;   // d C6 / d CN_i
            const float qab = p.sq[si] * p.sq[sj];
            const float R = p.a1 * sqrtf(3.0f * qab) + p.a2;
            const float R2 = R * R, R6 = R2 * R2 * R2, R8 = R6 * R2;
            const float rb = r * A2B, rb2 = rb * rb, rb6 = rb2 * rb2 * rb2, rb8 = rb6 * rb2;
            const float i6 = 1.0f / (rb6 + R6), i8 = 1.0f / (rb8 + R8);
            const float k6 = p.s6, k8 = 3.0f * p.s8 * qab;
            float fc, dfc;
            d3_envelope(r, cutoff, smooth, fc, dfc);
            const float per_c6 = -(k6 * i6 + k8 * i8);                 // e / C6 without the envelope
            const float bare = c6 * per_c6;
            // d bare / d r [Angstrom] at fixed C6
            const float dbare = c6 * (k6 * 6.0f * rb2 * rb2 * rb * i6 * i6 + k8 * 8.0f * rb6 * rb * i8 * i8) * A2B;
            const float de = dbare * fc + bare * dfc;
            gc += per_c6 * fc * dc6;
            if (own) {
                e += 0.5f * bare * fc;
                const float ux = d.x * inv, uy = d.y * inv, uz = d.z * inv;
                gx -= de * ux; gy -= de * uy; gz -= de * uz;
                if (virial) {
                    const float h = 0.5f * de;
                    vxx += h * ux * d.x; vyy += h * uy * d.y; vzz += h * uz * d.z;
                    vxy += h * ux * d.y; vxz += h * ux * d.z; vyz += h * uy * d.z;
                }
            }
        }
        gc = wave_sum(gc);
        if (lane == 0) gcn[i] = gc;
        if (own) {
            e = wave_sum(e);
            if (lane == 0 && atomic_e) atomic_e[i] += e;
            if (grad_coords) {
                gx = wave_sum(gx); gy = wave_sum(gy); gz = wave_sum(gz);
                if (lane == 0) {
                    float *gi = grad_coords + 3 * (size_t)i;
                    gi[0] += gx; gi[1] += gy; gi[2] += gz;
                }
            }
        }
    }
    if (virial) {
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

__global__ __launch_bounds__(256) void k_d3_cngrad(int64_t lo, int64_t hi, const int32_t *__restrict__ species,
                                                   const uint32_t *__restrict__ meta, const float4 *__restrict__ ent,
                                                   D3P p, float cutoff, const float *__restrict__ gcn,
                                                   float *__restrict__ grad_coords, double *__restrict__ virial)
{
    const int lane = lane_id();
    const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    float vxx = 0.f, vyy = 0.f, vzz = 0.f, vxy = 0.f, vxz = 0.f, vyz = 0.f;
    for (int64_t i = lo + blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6); i < hi; i += nw) {
        const int si = species[i];
        if (si < 0) continue;
        const uint32_t start = meta[(size_t)i * META_W], c = meta[(size_t)i * META_W + 1];
        const int nR = (int)(c & 0xFFFFu) + (int)(c >> 16);
        const float gi_ = gcn[i];
        float gx = 0.f, gy = 0.f, gz = 0.f;
        for (int k = lane; k < nR; k += WAVE) {
            const float4 d = ent[start + k];
            const uint32_t w = __float_as_uint(d.w);
            const int sj = (int)(w >> 28);
            const float r2 = d.x * d.x + d.y * d.y + d.z * d.z;
            const float inv = __builtin_amdgcn_rsqf(r2);
            const float r = fmaxf(r2 * inv, 1e-7f);
            if (r > cutoff) continue;
            float dc;
            d3_count(p.cov[si] + p.cov[sj], r * A2B, dc);
            const float de = (gi_ + gcn[w & IDX_MASK]) * dc * A2B;   // d E / d r_ij [Angstrom] through the CNs
            const float ux = d.x * inv, uy = d.y * inv, uz = d.z * inv;
            gx -= de * ux; gy -= de * uy; gz -= de * uz;
            if (virial) {
                const float h = 0.5f * de;
                vxx += h * ux * d.x; vyy += h * uy * d.y; vzz += h * uz * d.z;
                vxy += h * ux * d.y; vxz += h * ux * d.z; vyz += h * uy * d.z;
            }
        }
        gx = wave_sum(gx); gy = wave_sum(gy); gz = wave_sum(gz);
        if (lane == 0) {
            float *gi = grad_coords + 3 * (size_t)i;
            gi[0] += gx; gi[1] += gy; gi[2] += gz;
        }
    }
    if (virial) {
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

}  // namespace anihip

using namespace anihip;

extern "C" int anihip_pair_analytic(void *stream, int32_t kind, int64_t n_atoms, int64_t lo, int64_t hi,
                                    const int32_t *species, const uint32_t *meta, const float *ent,
                                    const float *pair_table, const float *extra, float cutoff, int32_t cutoff_kind,
                                    int32_t flags, float *atomic_e, float *grad_coords, double *virial)
{
    ANIHIP_REQUIRE(species && meta && ent && pair_table, "null pointer argument");
    ANIHIP_REQUIRE(0 <= lo && lo <= hi && hi <= n_atoms, "central range outside 0..n_atoms");
    ANIHIP_REQUIRE(cutoff > 0.f, "cutoff must be positive (the rows hold pairs up to their own radial cutoff)");
    ANIHIP_REQUIRE(cutoff_kind == ANIHIP_CUTOFF_COSINE || cutoff_kind == ANIHIP_CUTOFF_SMOOTH, "unknown cutoff_kind");
    ANIHIP_REQUIRE(kind >= ANIHIP_PAIR_XTB && kind <= ANIHIP_PAIR_COULOMB, "unknown pair potential kind");
    ANIHIP_REQUIRE(kind != ANIHIP_PAIR_ZBL || extra, "ZBL needs its 4 + 4 screening constants");
    if (hi == lo) return 0;
    int64_t blocks = (hi - lo + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    PairExtra x{};
    if (extra)
        for (int k = 0; k < 8; ++k) x.v[k] = extra[k];
    const int smooth = cutoff_kind == ANIHIP_CUTOFF_SMOOTH ? 1 : 0, push = (flags & ANIHIP_PAIR_PUSH) ? 1 : 0;
    const int clamp_r = (flags & ANIHIP_PAIR_NO_CLAMP) ? 0 : 1;
#define ANIHIP_LAUNCH_PAIR(K)                                                                                          \
    hipLaunchKernelGGL((k_pair<K>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, lo, hi, species, meta,   \
                       (const float4 *)ent, pair_table, x, cutoff, smooth, push, clamp_r, atomic_e, grad_coords, virial)
    switch (kind) {
        case ANIHIP_PAIR_XTB: ANIHIP_LAUNCH_PAIR(ANIHIP_PAIR_XTB); break;
        case ANIHIP_PAIR_ZBL: ANIHIP_LAUNCH_PAIR(ANIHIP_PAIR_ZBL); break;
        case ANIHIP_PAIR_LJ: ANIHIP_LAUNCH_PAIR(ANIHIP_PAIR_LJ); break;
        default: ANIHIP_LAUNCH_PAIR(ANIHIP_PAIR_COULOMB); break;
    }
#undef ANIHIP_LAUNCH_PAIR
    ANIHIP_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int anihip_pair_xtb_repulsion(void *stream, int64_t n_atoms, int64_t lo, int64_t hi, const int32_t *species,
                                         const uint32_t *meta, const float *ent, const float *pair_table, float cutoff,
                                         int32_t cutoff_kind, int32_t flags, float *atomic_e, float *grad_coords,
                                         double *virial)
{
    return anihip_pair_analytic(stream, ANIHIP_PAIR_XTB, n_atoms, lo, hi, species, meta, ent, pair_table, nullptr, cutoff,
                                cutoff_kind, flags, atomic_e, grad_coords, virial);
}

extern "C" int anihip_pair_d3(void *stream, int64_t n_atoms, int64_t lo, int64_t hi, const int32_t *species,
                              const uint32_t *meta, const float *ent, const float *c6_table,
                              const anihip_d3_params *params, float cutoff, int32_t cutoff_kind, float *cn, float *gcn,
                              float *atomic_e, float *grad_coords, double *virial)
{
    ANIHIP_REQUIRE(species && meta && ent && c6_table && params && cn && gcn, "null pointer argument");
    ANIHIP_REQUIRE(0 <= lo && lo <= hi && hi <= n_atoms, "central range outside 0..n_atoms");
    ANIHIP_REQUIRE(cutoff > 0.f, "cutoff must be positive (the rows hold pairs up to their own radial cutoff)");
    ANIHIP_REQUIRE(cutoff_kind == ANIHIP_CUTOFF_COSINE || cutoff_kind == ANIHIP_CUTOFF_SMOOTH, "unknown cutoff_kind");
    if (n_atoms == 0) return 0;
    D3P p;
    p.s6 = params->s6; p.s8 = params->s8; p.a1 = params->a1; p.a2 = params->a2;
    for (int k = 0; k < 8; ++k) { p.cov[k] = params->cov_radius_bohr[k]; p.sq[k] = params->sqrt_q[k]; }
    auto blocks_for = [](int64_t n) { int64_t b = (n + 3) / 4; return (unsigned)(b > 256 * 8 ? 256 * 8 : (b < 1 ? 1 : b)); };
    const int smooth = cutoff_kind == ANIHIP_CUTOFF_SMOOTH ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_d3_cn, dim3(blocks_for(n_atoms)), dim3(256), 0, st, n_atoms, species, meta,
                       (const float4 *)ent, p, cutoff, cn);
    hipLaunchKernelGGL(k_d3_pair, dim3(blocks_for(n_atoms)), dim3(256), 0, st, n_atoms, lo, hi, species, meta,
                       (const float4 *)ent, (const float4 *)c6_table, p, cutoff, smooth, cn, gcn, atomic_e,
                       grad_coords, virial);
    if (grad_coords && hi > lo)
        hipLaunchKernelGGL(k_d3_cngrad, dim3(blocks_for(hi - lo)), dim3(256), 0, st, lo, hi, species, meta,
                           (const float4 *)ent, p, cutoff, gcn, grad_coords, virial);
    ANIHIP_CHECK_HIP(hipGetLastError());
    return 0;
}

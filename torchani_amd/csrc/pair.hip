// Pair potentials evaluated on the neighbor rows of nbr.hip: the xTB repulsion term of ANI-2xr / ANI-2dr
// (reference: torchani/potentials/xtb.py:17-77 RepulsionXTB.pair_energies, with the cutoff envelope and the half-per-atom
// bookkeeping of potentials/core.py:155-207).
//
// One wave per central atom, lane = neighbor.  The rows are a FULL symmetric list, so atom i finishes everything that
// concerns itself from its own row: atomic energy sum_j e_ij / 2 (core.py:195-198) and gradient
// sum_j e'(d_ij) d r_ij / d r_i -- no atomics, deterministic.  (Rows from a LAMMPS full list are not symmetric: there the
// pair term is pushed to the neighbor with float atomics, flag ANIHIP_PAIR_PUSH.)
#include "anihip_common.h"

namespace anihip {

constexpr float A2B = 1.8897261258369282f;   // torchani/units.py:41

__global__ __launch_bounds__(256) void k_pair_xtb(int64_t lo, int64_t hi, const int32_t *__restrict__ species,
                                                  const uint32_t *__restrict__ meta, const float4 *__restrict__ ent,
                                                  const float *__restrict__ tab /* [8][8][4]: y, sqrt(alpha), k, - */,
                                                  float cutoff, int smooth, int push, float *__restrict__ atomic_e,
                                                  float *__restrict__ grad_coords, double *__restrict__ virial)
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
            const float r = fmaxf(r2 * inv, 1e-7f);   // (core.py:138-139 clamp)
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
            const float rb = r * A2B;                                   // Bohr
            const float pw = __builtin_amdgcn_exp2f(p.z * __builtin_amdgcn_logf(rb));   // rb^k
            const float ex = __expf(-p.y * pw);
            const float base = p.x / rb * ex;                           // y_ab / d * exp(-sqrt(alpha_ab) d^k)
            const float eij = base * fc;
            // d/dr [Angstrom]: base' = base (-1/rb - sqrt(alpha) k rb^(k-1)) A2B
            const float dbase = base * (-1.0f / rb - p.y * p.z * pw / rb) * A2B;
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

}  // namespace anihip

using namespace anihip;

extern "C" int anihip_pair_xtb_repulsion(void *stream, int64_t n_atoms, int64_t lo, int64_t hi, const int32_t *species,
                                         const uint32_t *meta, const float *ent, const float *pair_table, float cutoff,
                                         int32_t cutoff_kind, int32_t flags, float *atomic_e, float *grad_coords,
                                         double *virial)
{
    ANIHIP_REQUIRE(species && meta && ent && pair_table, "null pointer argument");
    ANIHIP_REQUIRE(0 <= lo && lo <= hi && hi <= n_atoms, "central range outside 0..n_atoms");
    ANIHIP_REQUIRE(cutoff > 0.f, "cutoff must be positive (the rows hold pairs up to their own radial cutoff)");
    ANIHIP_REQUIRE(cutoff_kind == ANIHIP_CUTOFF_COSINE || cutoff_kind == ANIHIP_CUTOFF_SMOOTH, "unknown cutoff_kind");
    if (hi == lo) return 0;
    int64_t blocks = (hi - lo + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(k_pair_xtb, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, lo, hi, species, meta,
                       (const float4 *)ent, pair_table, cutoff, cutoff_kind == ANIHIP_CUTOFF_SMOOTH ? 1 : 0,
                       (flags & ANIHIP_PAIR_PUSH) ? 1 : 0, atomic_e, grad_coords, virial);
    ANIHIP_CHECK_HIP(hipGetLastError());
    return 0;
}

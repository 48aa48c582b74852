/*
 * oracle/ani_oracle.c -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.
 *
 * A plain-C CPU restatement of the torchani ANI hot path (neighbor list -> radial/angular
 * AEV -> per-species MLP ensemble -> energies and forces).  It exists so that the HIP
 * engine in torchani_amd/csrc can be parity-checked on a GPU box where /root/reference
 * is not present.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library; nothing under torchani_amd/ imports, links or calls it.
 *
 * Parity status: PINNED.  the tests/golden npz fixtures were produced by importing the reference
 * itself (fp64, pyaev strategy) with tests/golden/gen_golden.py, and
 * tests/test_oracle_golden.py checks every function below against them.
 *
 * Each function cites the reference file:line it restates (paths relative to
 * /root/reference/torchani/).  Arithmetic is done in `real` (double by default; build
 * with -DREAL=float for the fp32 timing variant used by bench.py's cpu_baseline).
 * Following SURVEY section 0 item 7, the AEV constants and all network parameters are the
 * fp32-rounded values of the reference buffers, promoted to `real` by the caller.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef REAL
#define REAL double
#endif
typedef REAL real;

#define ANI_MAX_SHIFTS 64

/* AEV hyper-parameters: aev/_computer.py:550-600, aev/_terms.py:188-207,345-366 */
typedef struct {
    int S;          /* number of species */
    int nR, nA, nZ; /* radial shifts, angular shifts, angular sections */
    double Rcr, Rca;
    double EtaR, EtaA, Zeta;
    double ShfR[ANI_MAX_SHIFTS];
    double ShfA[ANI_MAX_SHIFTS];
    double ShfZ[ANI_MAX_SHIFTS];
    int cutoff_kind; /* 0 = CutoffCosine (cutoffs.py:71-81), 1 = CutoffSmooth order 2, eps 1e-10 (cutoffs.py:84-101) */
} ani_params;

/* Full (both directions) neighbor list, CSR by central atom over the flattened C*A atoms. */
typedef struct {
    int64_t n_atoms;
    int64_t n_entries;
    int64_t *start; /* [n_atoms+1] */
    int32_t *j;     /* flattened index of the neighbor atom */
    real *d;        /* [n_entries*3]  r_j + shift - r_i */
    real *r;        /* [n_entries] */
} ani_nbrs;

int ani_oracle_real_bytes(void) { return (int)sizeof(real); }

int ani_oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void ani_oracle_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------ */
/* small 3x3 helpers (cell rows are lattice vectors: shifts = idx @ cell, neighbors.py:196) */

static void inv3(const double *m, double *o)
{
    double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
    double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    double id = 1.0 / det;
    o[0] = (e * i - f * h) * id; o[1] = (c * h - b * i) * id; o[2] = (b * f - c * e) * id;
    o[3] = (f * g - d * i) * id; o[4] = (a * i - c * g) * id; o[5] = (c * d - a * f) * id;
    o[6] = (d * h - e * g) * id; o[7] = (b * g - a * h) * id; o[8] = (a * e - b * d) * id;
}

/* utils.py:237-255 map_to_central: frac = r @ inv(cell); frac -= floor(frac)*pbc; r = frac @ cell */
void ani_oracle_map_to_central(int64_t n, const real *coords, const double *cell, const int *pbc,
                               real *out)
{
    double inv[9];
    inv3(cell, inv);
    for (int64_t a = 0; a < n; ++a) {
        double x = coords[3 * a], y = coords[3 * a + 1], z = coords[3 * a + 2];
        double f[3];
        for (int k = 0; k < 3; ++k) {
            f[k] = x * inv[0 + k] + y * inv[3 + k] + z * inv[6 + k];
            if (pbc[k]) f[k] -= floor(f[k]);
        }
        for (int k = 0; k < 3; ++k)
            out[3 * a + k] = (real)(f[0] * cell[0 + k] + f[1] * cell[3 + k] + f[2] * cell[6 + k]);
    }
}

/* neighbors.py:250-275 _all_pairs_pbc_shifts: repeats_k = ceil(cutoff * |column k of inv(cell)|) */
static void pbc_repeats(const double *cell, const int *pbc, double cutoff, int *rep, double *height)
{
    double inv[9];
    inv3(cell, inv);
    for (int k = 0; k < 3; ++k) {
        double nrm = sqrt(inv[0 + k] * inv[0 + k] + inv[3 + k] * inv[3 + k] + inv[6 + k] * inv[6 + k]);
        height[k] = 1.0 / nrm; /* distance between the two cell faces perpendicular to axis k */
        rep[k] = pbc[k] ? (int)ceil(cutoff * nrm) : 0;
    }
}

void ani_oracle_free_nbrs(ani_nbrs *nb)
{
    if (!nb) return;
    free(nb->start); free(nb->j); free(nb->d); free(nb->r);
    free(nb);
}

static ani_nbrs *nbrs_from_counts(int64_t n_atoms, const int64_t *cnt)
{
    ani_nbrs *nb = (ani_nbrs *)calloc(1, sizeof(ani_nbrs));
    nb->n_atoms = n_atoms;
    nb->start = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_atoms + 1));
    nb->start[0] = 0;
    for (int64_t a = 0; a < n_atoms; ++a) nb->start[a + 1] = nb->start[a] + cnt[a];
    nb->n_entries = nb->start[n_atoms];
    size_t ne = (size_t)(nb->n_entries > 0 ? nb->n_entries : 1);
    nb->j = (int32_t *)malloc(sizeof(int32_t) * ne);
    nb->d = (real *)malloc(sizeof(real) * 3 * ne);
    nb->r = (real *)malloc(sizeof(real) * ne);
    return nb;
}

/*
 * Brute-force full neighbor list.  Restates neighbors.py:187-212 (all_pairs), :215-275 (PBC image
 * shifts) and :64-113 (narrow_down: drop dummy atoms, keep |d| <= cutoff).  The reference stores each
 * unordered pair once; here every ordered (central i, neighbor j+shift) is stored, which is the same
 * set seen from both ends.  An atom is its own neighbor only through a non-zero image shift.
 * coords must already be mapped to the central cell when pbc is used (all_pairs does this, :198).
 */
static void scan_atom_brute(int64_t i, int64_t mol0, int A, const int32_t *species, const real *coords,
                            const double *cell, const int *rep, double cutoff, int64_t *count,
                            ani_nbrs *nb, int64_t pos)
{
    const real xi = coords[3 * i], yi = coords[3 * i + 1], zi = coords[3 * i + 2];
    int64_t c = 0;
    for (int a = 0; a < A; ++a) {
        int64_t j = mol0 + a;
        if (species[j] < 0) continue;
        for (int n0 = -rep[0]; n0 <= rep[0]; ++n0)
            for (int n1 = -rep[1]; n1 <= rep[1]; ++n1)
                for (int n2 = -rep[2]; n2 <= rep[2]; ++n2) {
                    if (j == i && n0 == 0 && n1 == 0 && n2 == 0) continue;
                    real sx = 0, sy = 0, sz = 0;
                    if (cell) {
                        sx = (real)(n0 * cell[0] + n1 * cell[3] + n2 * cell[6]);
                        sy = (real)(n0 * cell[1] + n1 * cell[4] + n2 * cell[7]);
                        sz = (real)(n0 * cell[2] + n1 * cell[5] + n2 * cell[8]);
                    }
                    real dx = coords[3 * j] + sx - xi;
                    real dy = coords[3 * j + 1] + sy - yi;
                    real dz = coords[3 * j + 2] + sz - zi;
                    real r = (real)sqrt((double)(dx * dx + dy * dy + dz * dz));
                    if (r <= (real)cutoff) {
                        if (nb) {
                            nb->j[pos + c] = (int32_t)j;
                            nb->d[3 * (pos + c)] = dx;
                            nb->d[3 * (pos + c) + 1] = dy;
                            nb->d[3 * (pos + c) + 2] = dz;
                            nb->r[pos + c] = r;
                        }
                        ++c;
                    }
                }
    }
    if (count) *count = c;
}

ani_nbrs *ani_oracle_nbrs_brute(int C, int A, const int32_t *species, const real *coords,
                                const double *cell, const int *pbc, double cutoff)
{
    int64_t n = (int64_t)C * A;
    int rep[3] = {0, 0, 0};
    double h[3];
    if (cell && pbc) pbc_repeats(cell, pbc, cutoff, rep, h);
    int64_t *cnt = (int64_t *)calloc((size_t)n, sizeof(int64_t));
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t i = 0; i < n; ++i) {
        if (species[i] < 0) continue;
        scan_atom_brute(i, (i / A) * A, A, species, coords, (cell && pbc) ? cell : NULL, rep, cutoff,
                        &cnt[i], NULL, 0);
    }
    ani_nbrs *nb = nbrs_from_counts(n, cnt);
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t i = 0; i < n; ++i) {
        if (species[i] < 0) continue;
        scan_atom_brute(i, (i / A) * A, A, species, coords, (cell && pbc) ? cell : NULL, rep, cutoff,
                        NULL, nb, nb->start[i]);
    }
    free(cnt);
    return nb;
}

/*
 * Cell-binned full neighbor list for ONE large system (C == 1), the O(N) counterpart of the
 * reference's cell_list (neighbors.py:366-507: one bucket per cutoff, neighbor-bucket stencil, image
 * shifts under PBC; non-PBC uses the bounding box, :389-394).  Written independently of the brute
 * force path above; tests require both to give identical pair sets.
 */
typedef struct {
    int nb[3];
    int range[3];
    int pbc[3];
    double fmin[3], fscale[3];
    double cell[9], inv[9];
    int64_t *bin_start;
    int32_t *order;
} grid_t;

static void bin_of(const grid_t *g, const real *p, int *b)
{
    for (int k = 0; k < 3; ++k) {
        double f = p[0] * g->inv[0 + k] + p[1] * g->inv[3 + k] + p[2] * g->inv[6 + k];
        int v = (int)floor((f - g->fmin[k]) * g->fscale[k]);
        if (v < 0) v = 0;
        if (v >= g->nb[k]) v = g->nb[k] - 1;
        b[k] = v;
    }
}

static void scan_atom_grid(const grid_t *g, int64_t i, const int32_t *species, const real *coords,
                           double cutoff, int64_t *count, ani_nbrs *nb, int64_t pos)
{
    int bi[3];
    bin_of(g, &coords[3 * i], bi);
    const real xi = coords[3 * i], yi = coords[3 * i + 1], zi = coords[3 * i + 2];
    int64_t c = 0;
    for (int o0 = -g->range[0]; o0 <= g->range[0]; ++o0)
        for (int o1 = -g->range[1]; o1 <= g->range[1]; ++o1)
            for (int o2 = -g->range[2]; o2 <= g->range[2]; ++o2) {
                int o[3] = {o0, o1, o2}, b[3], s[3];
                int skip = 0;
                for (int k = 0; k < 3; ++k) {
                    int v = bi[k] + o[k];
                    if (g->pbc[k]) {
                        int q = (int)floor((double)v / g->nb[k]);
                        s[k] = q;
                        b[k] = v - q * g->nb[k];
                    } else {
                        s[k] = 0;
                        b[k] = v;
                        if (v < 0 || v >= g->nb[k]) skip = 1;
                    }
                }
                if (skip) continue;
                real sx = (real)(s[0] * g->cell[0] + s[1] * g->cell[3] + s[2] * g->cell[6]);
                real sy = (real)(s[0] * g->cell[1] + s[1] * g->cell[4] + s[2] * g->cell[7]);
                real sz = (real)(s[0] * g->cell[2] + s[1] * g->cell[5] + s[2] * g->cell[8]);
                int64_t bin = ((int64_t)b[0] * g->nb[1] + b[1]) * g->nb[2] + b[2];
                for (int64_t q = g->bin_start[bin]; q < g->bin_start[bin + 1]; ++q) {
                    int64_t j = g->order[q];
                    if (j == i && s[0] == 0 && s[1] == 0 && s[2] == 0) continue;
                    real dx = coords[3 * j] + sx - xi;
                    real dy = coords[3 * j + 1] + sy - yi;
                    real dz = coords[3 * j + 2] + sz - zi;
                    real r = (real)sqrt((double)(dx * dx + dy * dy + dz * dz));
                    if (r <= (real)cutoff) {
                        if (nb) {
                            nb->j[pos + c] = (int32_t)j;
                            nb->d[3 * (pos + c)] = dx;
                            nb->d[3 * (pos + c) + 1] = dy;
                            nb->d[3 * (pos + c) + 2] = dz;
                            nb->r[pos + c] = r;
                        }
                        ++c;
                    }
                }
            }
    if (count) *count = c;
}

ani_nbrs *ani_oracle_nbrs_cell(int64_t n, const int32_t *species, const real *coords,
                               const double *cell_in, const int *pbc_in, double cutoff)
{
    grid_t g;
    memset(&g, 0, sizeof(g));
    int use_pbc = (cell_in && pbc_in && (pbc_in[0] || pbc_in[1] || pbc_in[2]));
    if (cell_in && pbc_in) {
        memcpy(g.cell, cell_in, sizeof(double) * 9);
        for (int k = 0; k < 3; ++k) g.pbc[k] = pbc_in[k] ? 1 : 0;
    } else {
        g.cell[0] = g.cell[4] = g.cell[8] = 1.0;
    }
    (void)use_pbc;
    inv3(g.cell, g.inv);
    double height[3];
    for (int k = 0; k < 3; ++k) {
        double nrm = sqrt(g.inv[0 + k] * g.inv[0 + k] + g.inv[3 + k] * g.inv[3 + k] +
                          g.inv[6 + k] * g.inv[6 + k]);
        height[k] = 1.0 / nrm;
    }
    /* fractional extents of the real atoms along non-periodic axes */
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int64_t a = 0; a < n; ++a) {
        if (species[a] < 0) continue;
        for (int k = 0; k < 3; ++k) {
            double f = coords[3 * a] * g.inv[0 + k] + coords[3 * a + 1] * g.inv[3 + k] +
                       coords[3 * a + 2] * g.inv[6 + k];
            if (f < lo[k]) lo[k] = f;
            if (f > hi[k]) hi[k] = f;
        }
    }
    for (int k = 0; k < 3; ++k) {
        if (g.pbc[k]) {
            int nbk = (int)floor(height[k] / cutoff);
            if (nbk < 1) nbk = 1;
            g.nb[k] = nbk;
            g.fmin[k] = 0.0;
            g.fscale[k] = (double)nbk;
            g.range[k] = (int)ceil(cutoff / (height[k] / nbk));
        } else {
            double span = (hi[k] > lo[k]) ? (hi[k] - lo[k]) : 0.0;
            int nbk = (int)floor(span * height[k] / cutoff);
            if (nbk < 1) nbk = 1;
            g.nb[k] = nbk;
            g.fmin[k] = lo[k];
            g.fscale[k] = (span > 0.0) ? nbk / (span * (1.0 + 1e-12)) : 0.0;
            g.range[k] = 1;
        }
    }
    int64_t nbins = (int64_t)g.nb[0] * g.nb[1] * g.nb[2];
    g.bin_start = (int64_t *)calloc((size_t)(nbins + 1), sizeof(int64_t));
    g.order = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    int64_t *abin = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t a = 0; a < n; ++a) {
        if (species[a] < 0) { abin[a] = -1; continue; }
        int b[3];
        bin_of(&g, &coords[3 * a], b);
        abin[a] = ((int64_t)b[0] * g.nb[1] + b[1]) * g.nb[2] + b[2];
        g.bin_start[abin[a] + 1]++;
    }
    for (int64_t b = 0; b < nbins; ++b) g.bin_start[b + 1] += g.bin_start[b];
    int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nbins > 0 ? nbins : 1));
    memcpy(fill, g.bin_start, sizeof(int64_t) * (size_t)nbins);
    for (int64_t a = 0; a < n; ++a)
        if (abin[a] >= 0) g.order[fill[abin[a]]++] = (int32_t)a;
    free(fill);
    free(abin);

    int64_t *cnt = (int64_t *)calloc((size_t)(n > 0 ? n : 1), sizeof(int64_t));
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < n; ++i)
        if (species[i] >= 0) scan_atom_grid(&g, i, species, coords, cutoff, &cnt[i], NULL, 0);
    ani_nbrs *nb = nbrs_from_counts(n, cnt);
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < n; ++i)
        if (species[i] >= 0) scan_atom_grid(&g, i, species, coords, cutoff, NULL, nb, nb->start[i]);
    free(cnt);
    free(g.bin_start);
    free(g.order);
    return nb;
}

/* copy a list out for the tests (j, d, r may be NULL) */
int64_t ani_oracle_nbrs_size(const ani_nbrs *nb) { return nb->n_entries; }
void ani_oracle_nbrs_export(const ani_nbrs *nb, int64_t *start, int32_t *j, real *d, real *r)
{
    if (start) memcpy(start, nb->start, sizeof(int64_t) * (size_t)(nb->n_atoms + 1));
    if (j) memcpy(j, nb->j, sizeof(int32_t) * (size_t)nb->n_entries);
    if (d) memcpy(d, nb->d, sizeof(real) * 3 * (size_t)nb->n_entries);
    if (r) memcpy(r, nb->r, sizeof(real) * (size_t)nb->n_entries);
}

/* ------------------------------------------------------------------------------------ */
/* AEV terms */

/* cutoffs.py:80-81 CutoffCosine: 0.5*cos(r*pi/Rc)+0.5 ; cutoffs.py:98-100 CutoffSmooth (order 2, eps 1e-10):
 * exp(1 - 1/max(eps, 1 - (r/Rc)^2)); derivatives for the backward (the smooth one as in csrc/aev.cu:165-178:
 * zero where the clamp is active) */
#define ANI_SMOOTH_EPS 1.0e-10
static inline real fcut_k(int kind, real r, double Rc)
{
    if (kind == 1) {
        const double q = (double)r / Rc, m = fmax(ANI_SMOOTH_EPS, 1.0 - q * q);
        return (real)exp(1.0 - 1.0 / m);
    }
    return (real)(0.5 * cos((double)r * (M_PI / Rc)) + 0.5);
}
static inline real dfcut_k(int kind, real r, double Rc)
{
    if (kind == 1) {
        const double q = (double)r / Rc, pw = q * q, m = fmax(ANI_SMOOTH_EPS, 1.0 - pw);
        if (1.0 - pw - ANI_SMOOTH_EPS < 0.0) return (real)0;
        return (real)(-2.0 * pw * exp(1.0 - 1.0 / m) / ((double)r * m * m));
    }
    return (real)(-0.5 * (M_PI / Rc) * sin((double)r * (M_PI / Rc)));
}
#define fcut(r, Rc) fcut_k(p->cutoff_kind, (r), (Rc))
#define dfcut(r, Rc) dfcut_k(p->cutoff_kind, (r), (Rc))

/* aev/_computer.py:183-191 triu_index: row-major index into the upper triangle incl. diagonal */
static inline int triu_index(int S, int a, int b)
{
    if (a > b) { int t = a; a = b; b = t; }
    return a * S - a * (a - 1) / 2 + (b - a);
}

int ani_oracle_aev_dim(const ani_params *p)
{
    return p->S * p->nR + (p->S * (p->S + 1) / 2) * p->nA * p->nZ;
}

/*
 * AEV forward for every atom of the list.
 *   radial : aev/_terms.py:99-104,171-186 (0.25*exp(-EtaR (r-ShfR)^2) * fc) scattered by neighbor
 *            species, aev/_computer.py:337-350
 *   angular: aev/_terms.py:34-55 (cos = v1.v2/max(r1 r2,1e-10); term = radial x angular x fc1 fc2),
 *            :324-325 (exp(-EtaA((r1+r2)/2-ShfA)^2)), :339-343 (2*((1+cos(acos(.95cos)-ShfZ))/2)^Zeta)
 *            scattered by species pair, aev/_computer.py:302-333; layout [radial | angular] :298,
 *            angular sub-index a*nZ+z (outer product radial x angular, _terms.py:50).
 */
void ani_oracle_aev_forward(const ani_params *p, const ani_nbrs *nb, const int32_t *species, real *aev)
{
    const int L = ani_oracle_aev_dim(p);
    const int rad_len = p->S * p->nR;
    const int nAZ = p->nA * p->nZ;
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t i = 0; i < nb->n_atoms; ++i) {
        real *out = aev + (size_t)i * L;
        for (int q = 0; q < L; ++q) out[q] = 0;
        if (species[i] < 0) continue;
        const int64_t s0 = nb->start[i], s1 = nb->start[i + 1];
        for (int64_t e = s0; e < s1; ++e) {
            const real r = nb->r[e];
            const int sj = species[nb->j[e]];
            const real fc = fcut(r, p->Rcr);
            for (int s = 0; s < p->nR; ++s) {
                real dr = r - (real)p->ShfR[s];
                out[sj * p->nR + s] += (real)0.25 * (real)exp((double)(-(real)p->EtaR * dr * dr)) * fc;
            }
        }
        for (int64_t e1 = s0; e1 < s1; ++e1) {
            const real r1 = nb->r[e1];
            if (r1 > (real)p->Rca) continue;
            const real fc1 = fcut(r1, p->Rca);
            for (int64_t e2 = e1 + 1; e2 < s1; ++e2) {
                const real r2 = nb->r[e2];
                if (r2 > (real)p->Rca) continue;
                const real fc2 = fcut(r2, p->Rca);
                const real *d1 = &nb->d[3 * e1], *d2 = &nb->d[3 * e2];
                real den = r1 * r2;
                if (den < (real)1e-10) den = (real)1e-10;
                real cosang = (d1[0] * d2[0] + d1[1] * d2[1] + d1[2] * d2[2]) / den;
                real theta = (real)acos((double)((real)0.95 * cosang));
                real rm = (r1 + r2) / 2;
                int pidx = triu_index(p->S, species[nb->j[e1]], species[nb->j[e2]]);
                real *o = out + rad_len + pidx * nAZ;
                for (int a = 0; a < p->nA; ++a) {
                    real dr = rm - (real)p->ShfA[a];
                    real f2 = (real)exp((double)(-(real)p->EtaA * dr * dr));
                    for (int z = 0; z < p->nZ; ++z) {
                        real h = (1 + (real)cos((double)(theta - (real)p->ShfZ[z]))) / 2;
                        real f1 = 2 * (real)pow((double)h, p->Zeta);
                        o[a * p->nZ + z] += f2 * f1 * (fc1 * fc2);
                    }
                }
            }
        }
    }
}

/*
 * Analytic AEV backward: grad_coords[k] = sum_i sum_q grad_aev[i][q] * d aev[i][q] / d r_k.
 * The reference obtains this by autograd through the functions cited above (grad.py:57-64); the
 * closed forms are the chain rule on those same expressions.
 * d = r_j - r_i, so d/dr_j = +, d/dr_i = -.
 */
static void aev_backward_impl(const ani_params *p, const ani_nbrs *nb, const int32_t *species,
                              const real *grad_aev, real *grad_coords, double *virial)
{
    const int L = ani_oracle_aev_dim(p);
    const int rad_len = p->S * p->nR;
    const int nAZ = p->nA * p->nZ;
    const int64_t n = nb->n_atoms;
    for (int64_t q = 0; q < 3 * n; ++q) grad_coords[q] = 0;
    int nthreads = ani_oracle_num_threads();
    double *priv = (double *)calloc((size_t)nthreads * 3 * (size_t)n, sizeof(double));
    double *vpriv = (double *)calloc((size_t)nthreads * 9, sizeof(double));
#pragma omp parallel
    {
#ifdef _OPENMP
        int tid = omp_get_thread_num();
#else
        int tid = 0;
#endif
        double *g = priv + (size_t)tid * 3 * (size_t)n;
        double *vir = vpriv + (size_t)tid * 9;
#pragma omp for schedule(dynamic, 8)
        for (int64_t i = 0; i < n; ++i) {
            if (species[i] < 0) continue;
            const real *w = grad_aev + (size_t)i * L;
            const int64_t s0 = nb->start[i], s1 = nb->start[i + 1];
            /* radial */
            for (int64_t e = s0; e < s1; ++e) {
                const real r = nb->r[e];
                const int sj = species[nb->j[e]];
                const real fc = fcut(r, p->Rcr), dfc = dfcut(r, p->Rcr);
                real dR = 0;
                for (int s = 0; s < p->nR; ++s) {
                    real dr = r - (real)p->ShfR[s];
                    real ex = (real)0.25 * (real)exp((double)(-(real)p->EtaR * dr * dr));
                    real dex = -2 * (real)p->EtaR * dr * ex;
                    dR += w[sj * p->nR + s] * (dex * fc + ex * dfc);
                }
                const real *d = &nb->d[3 * e];
                for (int k = 0; k < 3; ++k) {
                    double v = (double)(dR * d[k] / r);
                    g[3 * nb->j[e] + k] += v;
                    g[3 * i + k] -= v;
                    for (int b = 0; b < 3; ++b) vir[3 * k + b] += v * (double)d[b];
                }
            }
            /* angular */
            for (int64_t e1 = s0; e1 < s1; ++e1) {
                const real r1 = nb->r[e1];
                if (r1 > (real)p->Rca) continue;
                const real fc1 = fcut(r1, p->Rca), dfc1 = dfcut(r1, p->Rca);
                for (int64_t e2 = e1 + 1; e2 < s1; ++e2) {
                    const real r2 = nb->r[e2];
                    if (r2 > (real)p->Rca) continue;
                    const real fc2 = fcut(r2, p->Rca), dfc2 = dfcut(r2, p->Rca);
                    const real *d1 = &nb->d[3 * e1], *d2 = &nb->d[3 * e2];
                    real u1[3], u2[3];
                    for (int k = 0; k < 3; ++k) { u1[k] = d1[k] / r1; u2[k] = d2[k] / r2; }
                    real c = u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2];
                    real ct = (real)0.95 * c;
                    real st = (real)sqrt((double)(1 - ct * ct));
                    real theta = (real)acos((double)ct);
                    real rm = (r1 + r2) / 2;
                    int pidx = triu_index(p->S, species[nb->j[e1]], species[nb->j[e2]]);
                    const real *ww = w + rad_len + pidx * nAZ;
                    real C0 = 0, Cth = 0, CR = 0; /* sum w f1 f2 ; sum w f1' f2 ; sum w f1 f2' */
                    for (int a = 0; a < p->nA; ++a) {
                        real dr = rm - (real)p->ShfA[a];
                        real f2 = (real)exp((double)(-(real)p->EtaA * dr * dr));
                        real df2 = -2 * (real)p->EtaA * dr * f2;
                        for (int z = 0; z < p->nZ; ++z) {
                            real dev = theta - (real)p->ShfZ[z];
                            real h = (1 + (real)cos((double)dev)) / 2;
                            real f1 = 2 * (real)pow((double)h, p->Zeta);
                            real df1 = -(real)p->Zeta * (real)pow((double)h, p->Zeta - 1) *
                                       (real)sin((double)dev);
                            real wz = ww[a * p->nZ + z];
                            C0 += wz * f1 * f2;
                            Cth += wz * df1 * f2;
                            CR += wz * f1 * df2;
                        }
                    }
                    real fcc = fc1 * fc2;
                    real kth = Cth * fcc * (-(real)0.95 / st); /* dE/dc */
                    real k1 = (real)0.5 * CR * fcc + C0 * dfc1 * fc2;
                    real k2 = (real)0.5 * CR * fcc + C0 * fc1 * dfc2;
                    for (int k = 0; k < 3; ++k) {
                        double g1 = (double)(kth * (u2[k] - c * u1[k]) / r1 + k1 * u1[k]);
                        double g2 = (double)(kth * (u1[k] - c * u2[k]) / r2 + k2 * u2[k]);
                        g[3 * nb->j[e1] + k] += g1;
                        g[3 * nb->j[e2] + k] += g2;
                        g[3 * i + k] -= g1 + g2;
                        for (int b = 0; b < 3; ++b) vir[3 * k + b] += g1 * (double)d1[b] + g2 * (double)d2[b];
                    }
                }
            }
        }
    }
    for (int t = 0; t < nthreads; ++t) {
        const double *g = priv + (size_t)t * 3 * (size_t)n;
        for (int64_t q = 0; q < 3 * n; ++q) grad_coords[q] += (real)g[q];
    }
    if (virial) {
        for (int q = 0; q < 9; ++q) virial[q] = 0;
        for (int t = 0; t < nthreads; ++t)
            for (int q = 0; q < 9; ++q) virial[q] += vpriv[(size_t)t * 9 + q];
    }
    free(priv);
    free(vpriv);
}

void ani_oracle_aev_backward(const ani_params *p, const ani_nbrs *nb, const int32_t *species,
                             const real *grad_aev, real *grad_coords)
{
    aev_backward_impl(p, nb, species, grad_aev, grad_coords, NULL);
}

/*
 * The same backward pass, also returning the virial  W[a][b] = sum over (central atom i, neighbor j) of
 * (d E_i / d d_ij)[a] * d_ij[b]  with d_ij the displacement stored in the neighbor list -- the "fdotr" stress of the
 * reference (ase.py:164-168: virial = dE/d(diff_vectors)^T @ diff_vectors; stress = virial / volume), which under
 * periodic boundary conditions equals d E / d strain (ase.py:170-173, the "scaling" stress).
 */
void ani_oracle_aev_backward_virial(const ani_params *p, const ani_nbrs *nb, const int32_t *species,
                                    const real *grad_aev, real *grad_coords, double *virial)
{
    aev_backward_impl(p, nb, species, grad_aev, grad_coords, virial);
}

/*
 * Forward-mode derivative of the AEVs: daev[i][q] = sum_k (d aev[i][q] / d r_k) . tang[k]  (J t, a Jacobian-vector
 * product).  This is what the reference's cuaev_double_backward returns for tang = the gradient arriving at the forces
 * (csrc/aev.cu:1986-2015 with the is_double_backward kernel variants, :474-766,837-967): the derivative of
 * grad_coords = J^T grad_aev with respect to grad_aev, contracted with that incoming gradient.
 * With d = r_j - r_i:  d' = t_j - t_i,  r' = u . d',  u' = (d' - u r') / r.
 */
void ani_oracle_aev_jvp(const ani_params *p, const ani_nbrs *nb, const int32_t *species, const real *tang,
                        real *daev)
{
    const int L = ani_oracle_aev_dim(p);
    const int rad_len = p->S * p->nR;
    const int nAZ = p->nA * p->nZ;
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t i = 0; i < nb->n_atoms; ++i) {
        real *out = daev + (size_t)i * L;
        for (int q = 0; q < L; ++q) out[q] = 0;
        if (species[i] < 0) continue;
        const int64_t s0 = nb->start[i], s1 = nb->start[i + 1];
        for (int64_t e = s0; e < s1; ++e) {
            const real r = nb->r[e];
            const int64_t j = nb->j[e];
            const int sj = species[j];
            const real *d = &nb->d[3 * e];
            real rdot = 0;
            for (int k = 0; k < 3; ++k) rdot += d[k] / r * (tang[3 * j + k] - tang[3 * i + k]);
            const real fc = fcut(r, p->Rcr), dfc = dfcut(r, p->Rcr);
            for (int s = 0; s < p->nR; ++s) {
                real dr = r - (real)p->ShfR[s];
                real ex = (real)0.25 * (real)exp((double)(-(real)p->EtaR * dr * dr));
                out[sj * p->nR + s] += (-2 * (real)p->EtaR * dr * ex * fc + ex * dfc) * rdot;
            }
        }
        for (int64_t e1 = s0; e1 < s1; ++e1) {
            const real r1 = nb->r[e1];
            if (r1 > (real)p->Rca) continue;
            const real fc1 = fcut(r1, p->Rca), dfc1 = dfcut(r1, p->Rca);
            const int64_t j1 = nb->j[e1];
            for (int64_t e2 = e1 + 1; e2 < s1; ++e2) {
                const real r2 = nb->r[e2];
                if (r2 > (real)p->Rca) continue;
                const real fc2 = fcut(r2, p->Rca), dfc2 = dfcut(r2, p->Rca);
                const int64_t j2 = nb->j[e2];
                const real *d1 = &nb->d[3 * e1], *d2 = &nb->d[3 * e2];
                real u1[3], u2[3], t1[3], t2[3], r1d = 0, r2d = 0;
                for (int k = 0; k < 3; ++k) {
                    u1[k] = d1[k] / r1; u2[k] = d2[k] / r2;
                    t1[k] = tang[3 * j1 + k] - tang[3 * i + k];
                    t2[k] = tang[3 * j2 + k] - tang[3 * i + k];
                    r1d += u1[k] * t1[k]; r2d += u2[k] * t2[k];
                }
                real c = 0, cdot = 0;
                for (int k = 0; k < 3; ++k) {
                    c += u1[k] * u2[k];
                    cdot += (t1[k] - u1[k] * r1d) / r1 * u2[k] + u1[k] * (t2[k] - u2[k] * r2d) / r2;
                }
                const real ct = (real)0.95 * c;
                const real theta = (real)acos((double)ct);
                const real thdot = -(real)0.95 * cdot / (real)sqrt((double)(1 - ct * ct));
                const real rm = (r1 + r2) / 2, rmdot = (r1d + r2d) / 2;
                const real fcc = fc1 * fc2, fccdot = dfc1 * r1d * fc2 + fc1 * dfc2 * r2d;
                int pidx = triu_index(p->S, species[j1], species[j2]);
                real *o = out + rad_len + pidx * nAZ;
                for (int a = 0; a < p->nA; ++a) {
                    real dr = rm - (real)p->ShfA[a];
                    real f2 = (real)exp((double)(-(real)p->EtaA * dr * dr));
                    real df2 = -2 * (real)p->EtaA * dr * f2;
                    for (int z = 0; z < p->nZ; ++z) {
                        real dev = theta - (real)p->ShfZ[z];
                        real h = (1 + (real)cos((double)dev)) / 2;
                        real f1 = 2 * (real)pow((double)h, p->Zeta);
                        real df1 = -(real)p->Zeta * (real)pow((double)h, p->Zeta - 1) * (real)sin((double)dev);
                        o[a * p->nZ + z] += df1 * thdot * f2 * fcc + f1 * df2 * rmdot * fcc + f1 * f2 * fccdot;
                    }
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* Networks */

/*
 * Packed parameters: for member m in 0..M-1, species s in 0..S-1, layer l in 0..nl-1 (nl = number of
 * Linear layers, last one is final_layer): weight[out,in] row-major then bias[out]
 * (torch.nn.Linear layout, nn/_core.py:131-137).  dims[s*(nl+1)+l] are the layer widths
 * (dims[..0] = AEV length).
 */
static inline real celu01(real x, real alpha)
{
    /* nn/_core.py:163-167 TightCELU = celu(x, 0.1) = max(0,x) + min(0, a*(exp(x/a)-1)) */
    return x > 0 ? x : alpha * ((real)exp((double)(x / alpha)) - 1);
}

static size_t net_param_count(const int *dims, int nl)
{
    size_t c = 0;
    for (int l = 0; l < nl; ++l) c += (size_t)dims[l + 1] * dims[l] + dims[l + 1];
    return c;
}

/*
 * Per-atom network energies and d(energy)/d(aev):
 *   nn/_core.py:146-149 AtomicNetwork.forward, nn/_containers.py:377-421 ANINetworks.forward (per-species
 *   dispatch, padding contributes nothing), :608-636 Ensemble.forward (mean over members).
 * atomic_e[i] = mean_m net_{m,species(i)}(aev_i);  grad_aev (may be NULL) = d atomic_e[i] / d aev_i.
 * member_e (may be NULL) = [M, n] per-member atomic energies (ensemble_values=True, :638-651).
 */
void ani_oracle_mlp(int64_t n, int S, int M, int nl, const int *dims, const real *params,
                    real celu_alpha, const int32_t *species, const real *aev, real *atomic_e,
                    real *grad_aev, real *member_e)
{
    const int L = dims[0];
    /* offsets of each (member, species) block */
    size_t *off = (size_t *)malloc(sizeof(size_t) * (size_t)(M * S + 1));
    off[0] = 0;
    int maxw = 0;
    for (int m = 0; m < M; ++m)
        for (int s = 0; s < S; ++s) {
            off[m * S + s + 1] = off[m * S + s] + net_param_count(dims + s * (nl + 1), nl);
            for (int l = 0; l <= nl; ++l)
                if (dims[s * (nl + 1) + l] > maxw) maxw = dims[s * (nl + 1) + l];
        }
#pragma omp parallel
    {
        real *act = (real *)malloc(sizeof(real) * (size_t)(nl + 1) * maxw);  /* activations */
        real *dact = (real *)malloc(sizeof(real) * (size_t)(nl + 1) * maxw); /* d celu / d pre */
        real *ga = (real *)malloc(sizeof(real) * (size_t)maxw);
        real *gb = (real *)malloc(sizeof(real) * (size_t)maxw);
#pragma omp for schedule(dynamic, 4)
        for (int64_t i = 0; i < n; ++i) {
            const int s = species[i];
            real *gout = grad_aev ? grad_aev + (size_t)i * L : NULL;
            if (gout) for (int q = 0; q < L; ++q) gout[q] = 0;
            atomic_e[i] = 0;
            if (s < 0) {
                if (member_e) for (int m = 0; m < M; ++m) member_e[(size_t)m * n + i] = 0;
                continue;
            }
            const int *d = dims + s * (nl + 1);
            real esum = 0;
            for (int m = 0; m < M; ++m) {
                const real *P = params + off[m * S + s];
                const real *x = aev + (size_t)i * L;
                const real *Pl = P;
                for (int l = 0; l < nl; ++l) {
                    const int in = d[l], out = d[l + 1];
                    const real *W = Pl, *b = Pl + (size_t)out * in;
                    real *y = act + (size_t)(l + 1) * maxw;
                    real *dy = dact + (size_t)(l + 1) * maxw;
                    for (int o = 0; o < out; ++o) {
                        real acc = b[o];
                        const real *wr = W + (size_t)o * in;
                        for (int k = 0; k < in; ++k) acc += wr[k] * x[k];
                        if (l < nl - 1) {
                            y[o] = celu01(acc, celu_alpha);
                            dy[o] = acc > 0 ? (real)1 : (real)exp((double)(acc / celu_alpha));
                        } else {
                            y[o] = acc;
                            dy[o] = 1;
                        }
                    }
                    x = y;
                    Pl += (size_t)out * in + out;
                }
                real e = act[(size_t)nl * maxw + 0];
                esum += e;
                if (member_e) member_e[(size_t)m * n + i] = e;
                if (gout) {
                    /* backward through the layers; output width of the last layer is 1 */
                    ga[0] = 1;
                    int cur = d[nl];
                    for (int l = nl - 1; l >= 0; --l) {
                        const int in = d[l], out = d[l + 1];
                        const real *Wl = P;
                        for (int q = 0; q < l; ++q) Wl += (size_t)d[q + 1] * d[q] + d[q + 1];
                        const real *dy = dact + (size_t)(l + 1) * maxw;
                        for (int k = 0; k < in; ++k) gb[k] = 0;
                        for (int o = 0; o < out; ++o) {
                            real go = ga[o] * dy[o];
                            const real *wr = Wl + (size_t)o * in;
                            for (int k = 0; k < in; ++k) gb[k] += go * wr[k];
                        }
                        real *t = ga; ga = gb; gb = t;
                        cur = in;
                    }
                    (void)cur;
                    for (int q = 0; q < L; ++q) gout[q] += ga[q] / M;
                }
            }
            atomic_e[i] = esum / M;
        }
        free(act); free(dact); free(ga); free(gb);
    }
    free(off);
}

/*
 * Gradients of  Loss = sum_i g_atom[i] * atomic_e[i]  (atomic_e = ensemble mean, as in ani_oracle_mlp) with respect
 * to every weight and bias: what autograd produces for the reference's containers when a loss on the energies is
 * back-propagated (training loop of tools/training-aev-benchmark.py:120-135; nn/_core.py:146-149,
 * nn/_containers.py:377-421,608-636).  grad_params has the layout of params and is overwritten.
 * One (member, species) network per task, so no two threads touch the same gradient block.
 */
void ani_oracle_mlp_weight_grads(int64_t n, int S, int M, int nl, const int *dims, const real *params,
                                 real celu_alpha, const int32_t *species, const real *aev,
                                 const real *g_atom, real *grad_params)
{
    const int L = dims[0];
    size_t *off = (size_t *)malloc(sizeof(size_t) * (size_t)(M * S + 1));
    off[0] = 0;
    int maxw = 0;
    for (int m = 0; m < M; ++m)
        for (int s = 0; s < S; ++s) {
            off[m * S + s + 1] = off[m * S + s] + net_param_count(dims + s * (nl + 1), nl);
            for (int l = 0; l <= nl; ++l)
                if (dims[s * (nl + 1) + l] > maxw) maxw = dims[s * (nl + 1) + l];
        }
    for (size_t q = 0; q < off[M * S]; ++q) grad_params[q] = 0;
#pragma omp parallel
    {
        real *act = (real *)malloc(sizeof(real) * (size_t)(nl + 1) * maxw);
        real *dact = (real *)malloc(sizeof(real) * (size_t)(nl + 1) * maxw);
        real *ga = (real *)malloc(sizeof(real) * (size_t)maxw);
        real *gb = (real *)malloc(sizeof(real) * (size_t)maxw);
#pragma omp for schedule(dynamic, 1)
        for (int task = 0; task < M * S; ++task) {
            const int s = task % S;
            const int *d = dims + s * (nl + 1);
            const real *P = params + off[task];
            real *G = grad_params + off[task];
            for (int64_t i = 0; i < n; ++i) {
                if (species[i] != s) continue;
                /* forward, keeping the input of every layer (act[l]) and celu' of its output (dact[l+1]) */
                const real *x = aev + (size_t)i * L;
                for (int k = 0; k < L; ++k) act[k] = x[k];
                const real *Pl = P;
                for (int l = 0; l < nl; ++l) {
                    const int in = d[l], out = d[l + 1];
                    const real *W = Pl, *b = Pl + (size_t)out * in;
                    const real *xin = act + (size_t)l * maxw;
                    real *y = act + (size_t)(l + 1) * maxw, *dy = dact + (size_t)(l + 1) * maxw;
                    for (int o = 0; o < out; ++o) {
                        real acc = b[o];
                        const real *wr = W + (size_t)o * in;
                        for (int k = 0; k < in; ++k) acc += wr[k] * xin[k];
                        if (l < nl - 1) {
                            y[o] = celu01(acc, celu_alpha);
                            dy[o] = acc > 0 ? (real)1 : (real)exp((double)(acc / celu_alpha));
                        } else {
                            y[o] = acc;
                            dy[o] = 1;
                        }
                    }
                    Pl += (size_t)out * in + out;
                }
                /* backward: d Loss / d e_member = g_atom / M (mean over members) */
                for (int o = 0; o < d[nl]; ++o) ga[o] = g_atom[i] / M;
                for (int l = nl - 1; l >= 0; --l) {
                    const int in = d[l], out = d[l + 1];
                    size_t lo = 0;
                    for (int q = 0; q < l; ++q) lo += (size_t)d[q + 1] * d[q] + d[q + 1];
                    const real *Wl = P + lo;
                    real *GW = G + lo, *Gb = G + lo + (size_t)out * in;
                    const real *xin = act + (size_t)l * maxw, *dy = dact + (size_t)(l + 1) * maxw;
                    for (int k = 0; k < in; ++k) gb[k] = 0;
                    for (int o = 0; o < out; ++o) {
                        const real go = ga[o] * dy[o];
                        const real *wr = Wl + (size_t)o * in;
                        real *gr = GW + (size_t)o * in;
                        Gb[o] += go;
                        for (int k = 0; k < in; ++k) {
                            gr[k] += go * xin[k];
                            gb[k] += go * wr[k];
                        }
                    }
                    real *t = ga; ga = gb; gb = t;
                }
            }
        }
        free(act); free(dact); free(ga); free(gb);
    }
    free(off);
}

/*
 * Second-order pass for training on forces.  For tangents v_i (one AEV-shaped row per atom) let
 *     S = sum_i v_i . d atomic_e[i] / d aev_i          (atomic_e = ensemble mean)
 * -- with v = -J t this is  S = sum_k t_k . F_k  for the forces F = -dE/dr, i.e. the part of a force loss that autograd
 * back-propagates through torch.autograd.grad(E, coords, create_graph=True) (tools/training-aev-benchmark.py:136-150).
 * Returns dS/d(weights, biases) in the layout of params (grad_params, overwritten) and S itself.
 * Forward-over-reverse: activations a_l, their tangents adot_l (adot_0 = v), then the adjoints of both:
 *     mu_l = dS/d adot_l, nu_l = dS/d a_l;   p = mu c'(z),  q = mu c''(z) zdot + nu c'(z);
 *     dS/dW_l = p adot_{l-1}^T + q a_{l-1}^T,  dS/db_l = q,  mu_{l-1} = W_l^T p,  nu_{l-1} = W_l^T q.
 */
double ani_oracle_mlp_tangent_weight_grads(int64_t n, int S, int M, int nl, const int *dims, const real *params,
                                           real celu_alpha, const int32_t *species, const real *aev,
                                           const real *tangent, real *grad_params)
{
    const int L = dims[0];
    size_t *off = (size_t *)malloc(sizeof(size_t) * (size_t)(M * S + 1));
    off[0] = 0;
    int maxw = 0;
    for (int m = 0; m < M; ++m)
        for (int s = 0; s < S; ++s) {
            off[m * S + s + 1] = off[m * S + s] + net_param_count(dims + s * (nl + 1), nl);
            for (int l = 0; l <= nl; ++l)
                if (dims[s * (nl + 1) + l] > maxw) maxw = dims[s * (nl + 1) + l];
        }
    for (size_t q = 0; q < off[M * S]; ++q) grad_params[q] = 0;
    double total = 0;
#pragma omp parallel
    {
        const size_t W = (size_t)maxw, NL = (size_t)(nl + 1);
        real *act = (real *)malloc(sizeof(real) * NL * W), *adot = (real *)malloc(sizeof(real) * NL * W);
        real *c1 = (real *)malloc(sizeof(real) * NL * W), *c2 = (real *)malloc(sizeof(real) * NL * W);
        real *zdot = (real *)malloc(sizeof(real) * NL * W);
        real *mu = (real *)malloc(sizeof(real) * W), *nu = (real *)malloc(sizeof(real) * W);
        real *mu2 = (real *)malloc(sizeof(real) * W), *nu2 = (real *)malloc(sizeof(real) * W);
        double local = 0;
#pragma omp for schedule(dynamic, 1)
        for (int task = 0; task < M * S; ++task) {
            const int s = task % S;
            const int *d = dims + s * (nl + 1);
            const real *P = params + off[task];
            real *G = grad_params + off[task];
            for (int64_t i = 0; i < n; ++i) {
                if (species[i] != s) continue;
                for (int k = 0; k < L; ++k) { act[k] = aev[(size_t)i * L + k]; adot[k] = tangent[(size_t)i * L + k]; }
                const real *Pl = P;
                for (int l = 0; l < nl; ++l) {
                    const int in = d[l], out = d[l + 1];
                    const real *Wl = Pl, *b = Pl + (size_t)out * in;
                    const real *x = act + (size_t)l * W, *xd = adot + (size_t)l * W;
                    for (int o = 0; o < out; ++o) {
                        real z = b[o], zd = 0;
                        const real *wr = Wl + (size_t)o * in;
                        for (int k = 0; k < in; ++k) { z += wr[k] * x[k]; zd += wr[k] * xd[k]; }
                        real f, f1, f2;
                        if (l < nl - 1) {
                            const real e = (real)exp((double)(z / celu_alpha));
                            f = z > 0 ? z : celu_alpha * (e - 1);
                            f1 = z > 0 ? (real)1 : e;
                            f2 = z > 0 ? (real)0 : e / celu_alpha;
                        } else {
                            f = z; f1 = 1; f2 = 0;
                        }
                        act[(size_t)(l + 1) * W + o] = f;
                        c1[(size_t)(l + 1) * W + o] = f1;
                        c2[(size_t)(l + 1) * W + o] = f2;
                        zdot[(size_t)(l + 1) * W + o] = zd;
                        adot[(size_t)(l + 1) * W + o] = f1 * zd;
                    }
                    Pl += (size_t)out * in + out;
                }
                local += (double)adot[(size_t)nl * W + 0] / M;
                for (int o = 0; o < d[nl]; ++o) { mu[o] = (real)1 / M; nu[o] = 0; }
                for (int l = nl - 1; l >= 0; --l) {
                    const int in = d[l], out = d[l + 1];
                    size_t lo = 0;
                    for (int q = 0; q < l; ++q) lo += (size_t)d[q + 1] * d[q] + d[q + 1];
                    const real *Wl = P + lo;
                    real *GW = G + lo, *Gb = G + lo + (size_t)out * in;
                    const real *x = act + (size_t)l * W, *xd = adot + (size_t)l * W;
                    for (int k = 0; k < in; ++k) { mu2[k] = 0; nu2[k] = 0; }
                    for (int o = 0; o < out; ++o) {
                        const real pp = mu[o] * c1[(size_t)(l + 1) * W + o];
                        const real qq = mu[o] * c2[(size_t)(l + 1) * W + o] * zdot[(size_t)(l + 1) * W + o] +
                                        nu[o] * c1[(size_t)(l + 1) * W + o];
                        const real *wr = Wl + (size_t)o * in;
                        real *gr = GW + (size_t)o * in;
                        Gb[o] += qq;
                        for (int k = 0; k < in; ++k) {
                            gr[k] += pp * xd[k] + qq * x[k];
                            mu2[k] += pp * wr[k];
                            nu2[k] += qq * wr[k];
                        }
                    }
                    real *t = mu; mu = mu2; mu2 = t;
                    t = nu; nu = nu2; nu2 = t;
                }
            }
        }
#pragma omp atomic
        total += local;
        free(act); free(adot); free(c1); free(c2); free(zdot); free(mu); free(nu); free(mu2); free(nu2);
    }
    free(off);
    return total;
}

/*
 * Whole path: species/coords -> per-atom NN energies, molecular energies (NN + self energies), forces.
 *   arch.py:302-349 ANI.forward; sae.py:54-64 SelfEnergy (padding -> 0); grad.py:57-64 forces = -dE/dr.
 * mol_energy is accumulated in double regardless of `real`.
 * cell/pbc may be NULL (no PBC).  use_cell_list selects the O(N) neighbor search (C must be 1).
 * Any output pointer may be NULL.
 */
int ani_oracle_energy_forces(const ani_params *p, int C, int A, const int32_t *species,
                             const real *coords_in, const double *cell, const int *pbc,
                             int use_cell_list, int M, int nl, const int *dims, const real *params,
                             real celu_alpha, const double *sae, real *aev_out, real *atomic_e_out,
                             double *mol_energy_out, real *forces_out)
{
    const int64_t n = (int64_t)C * A;
    const int L = ani_oracle_aev_dim(p);
    if (dims[0] != L) return -1;
    if (use_cell_list && C != 1) return -2;
    real *coords = (real *)malloc(sizeof(real) * 3 * (size_t)n);
    int have_pbc = (cell && pbc && (pbc[0] || pbc[1] || pbc[2]));
    if (have_pbc) ani_oracle_map_to_central(n, coords_in, cell, pbc, coords);
    else memcpy(coords, coords_in, sizeof(real) * 3 * (size_t)n);
    ani_nbrs *nb = use_cell_list
                       ? ani_oracle_nbrs_cell(n, species, coords, have_pbc ? cell : NULL,
                                              have_pbc ? pbc : NULL, p->Rcr)
                       : ani_oracle_nbrs_brute(C, A, species, coords, have_pbc ? cell : NULL,
                                               have_pbc ? pbc : NULL, p->Rcr);
    real *aev = aev_out ? aev_out : (real *)malloc(sizeof(real) * (size_t)n * L);
    ani_oracle_aev_forward(p, nb, species, aev);
    real *ae = atomic_e_out ? atomic_e_out : (real *)malloc(sizeof(real) * (size_t)n);
    real *gaev = forces_out ? (real *)malloc(sizeof(real) * (size_t)n * L) : NULL;
    ani_oracle_mlp(n, p->S, M, nl, dims, params, celu_alpha, species, aev, ae, gaev, NULL);
    if (mol_energy_out) {
        for (int c = 0; c < C; ++c) {
            double e = 0;
            for (int a = 0; a < A; ++a) {
                int64_t i = (int64_t)c * A + a;
                if (species[i] < 0) continue;
                e += (double)ae[i];
                if (sae) e += sae[species[i]];
            }
            mol_energy_out[c] = e;
        }
    }
    if (forces_out) {
        ani_oracle_aev_backward(p, nb, species, gaev, forces_out);
        for (int64_t q = 0; q < 3 * n; ++q) forces_out[q] = -forces_out[q];
        free(gaev);
    }
    if (!aev_out) free(aev);
    if (!atomic_e_out) free(ae);
    ani_oracle_free_nbrs(nb);
    free(coords);
    return 0;
}

// anihip_mlp_pack: the ensemble's parameters from torch.nn.Linear tensors into the layouts of include/anihip.h, behind the
// C ABI (round 3; the packer used to be ~170 lines of torch code on the Python side, so a caller binding libanihip.so
// from the reference -- MNPNetworks / mnp::run, nn/_infer.py:141-161,361-372, csrc/mnp.cpp:238-280 -- could not use the
// network half of the ABI without importing torchani_amd).  A model is packed once, on the host: 13.7 M parameters are
// read (device sources are copied to the host first), laid out into ONE buffer, and the buffer is copied to the device.
#include "anihip_common.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

namespace {

constexpr size_t ALIGN = 256;
inline size_t pad32(size_t x) { return (x + 31) / 32 * 32; }
inline size_t up(size_t x) { return (x + ALIGN - 1) / ALIGN * ALIGN; }

struct Geometry {
    int M, S, nl, K0, K0p, R, Rpad, K0h;
    bool f16, fused;
    int in_dim[ANIHIP_MAX_SPECIES][ANIHIP_MAX_LAYERS], out_dim[ANIHIP_MAX_SPECIES][ANIHIP_MAX_LAYERS];   // unpadded
    int dims[ANIHIP_MAX_SPECIES][ANIHIP_MAX_LAYERS + 1];                                                   // padded
};

// byte offsets of one species' arrays inside the buffer (0 = absent)
struct Offsets {
    size_t w[ANIHIP_MAX_LAYERS], wt[ANIHIP_MAX_LAYERS], bias[ANIHIP_MAX_LAYERS], wh[ANIHIP_MAX_LAYERS],
        wth[ANIHIP_MAX_LAYERS], whf[ANIHIP_MAX_LAYERS], wthf[ANIHIP_MAX_LAYERS], bounds;
};

int geometry(const anihip_mlp_shape *sh, Geometry *g)
{
    ANIHIP_REQUIRE(sh, "null pointer argument");
    ANIHIP_REQUIRE(sh->n_members >= 1 && sh->num_species >= 1 && sh->num_species <= ANIHIP_MAX_SPECIES,
                   "n_members >= 1 and 1 <= num_species <= %d", ANIHIP_MAX_SPECIES);
    ANIHIP_REQUIRE(sh->n_layers >= 2 && sh->n_layers <= ANIHIP_MAX_LAYERS, "networks must have 2..%d Linear layers",
                   ANIHIP_MAX_LAYERS);
    ANIHIP_REQUIRE(sh->aev_len > 0 && sh->aev_len % 16 == 0, "AEV length must be a multiple of 16");
    ANIHIP_REQUIRE(sh->precision == ANIHIP_MLP_FP32 || sh->precision == ANIHIP_MLP_F16X3, "unknown precision");
    ANIHIP_REQUIRE(sh->activation == ANIHIP_ACT_CELU || sh->activation == ANIHIP_ACT_GELU, "unknown activation");
    g->M = sh->n_members; g->S = sh->num_species; g->nl = sh->n_layers; g->K0 = sh->aev_len;
    g->K0p = (int)pad32(g->K0);
    g->f16 = sh->precision == ANIHIP_MLP_F16X3;
    int R = sh->aev_radial_len;
    if (R < 0) R = (g->K0 == 16 * g->S + 16 * g->S * (g->S + 1)) ? 16 * g->S : 0;   // the ANI form 16 S + 32 S (S + 1) / 2
    if (!g->f16 || R <= 0 || (g->K0 - R) % 32 != 0) R = 0;
    g->R = R;
    g->Rpad = (int)pad32(R);
    g->K0h = R ? g->Rpad + (g->K0 - R) : g->K0p;
    g->fused = g->f16 && g->nl == 4;
    for (int s = 0; s < g->S; ++s) {
        g->dims[s][0] = g->K0;
        for (int l = 0; l < g->nl; ++l) {
            const int o = sh->out_dims[s][l];
            ANIHIP_REQUIRE(o >= 1, "out_dims[%d][%d] must be positive", s, l);
            g->in_dim[s][l] = l == 0 ? g->K0 : sh->out_dims[s][l - 1];
            g->out_dim[s][l] = o;
            g->dims[s][l + 1] = l == g->nl - 1 ? 1 : (int)pad32(o);
        }
        ANIHIP_REQUIRE(sh->out_dims[s][g->nl - 1] == 1, "final layer must have one output");
    }
    return 0;
}

size_t layout(const Geometry &g, Offsets *off)
{
    size_t pos = ALIGN;   // (offset 0 means "absent")
    auto take = [&](size_t bytes) { const size_t at = pos; pos = up(pos + bytes); return at; };
    for (int s = 0; s < g.S; ++s) {
        Offsets &o = off[s];
        memset(&o, 0, sizeof(o));
        for (int l = 0; l < g.nl; ++l) {
            const size_t kin = g.dims[s][l], kout = g.dims[s][l + 1], M = g.M;
            if (l == g.nl - 1) {
                o.w[l] = take(4 * M * kin);
                o.bias[l] = take(4 * M);
                continue;
            }
            o.w[l] = take(4 * M * kin * kout);
            o.wt[l] = take(4 * M * kout * (l == 0 ? (size_t)g.K0p : kin));
            o.bias[l] = take(4 * M * kout);
            if (g.f16) {
                const size_t kred = l == 0 ? (size_t)g.K0h : kin;   // reduction length of the forward planes
                o.wh[l] = take(2 * 2 * M * kout * kred);
                o.wth[l] = take(2 * 2 * M * kout * kred);
                o.whf[l] = take(2 * 2 * M * kout * kred);
                // (layer 0: the transposed planes per member, [M][K0h][H1p], for the layer-0 backward inside the fused kernel)
                if (l >= 1 || g.fused) o.wthf[l] = take(2 * 2 * M * kout * kred);
            }
        }
        if (g.fused) o.bounds = take(4 * 8 * (size_t)g.M);
    }
    return pos;
}

// planes [2][rows][cols] of hi + lo = scale * src (round to nearest even, like torch's .to(float16))
void split_planes(const float *src, size_t n, float scale, _Float16 *hi, _Float16 *lo)
{
    for (size_t i = 0; i < n; ++i) {
        const float x = src[i] * scale;
        const _Float16 h = (_Float16)x;
        hi[i] = h;
        lo[i] = (_Float16)(x - (float)h);
    }
}

// planes [2][M][N][K] -> MFMA fragment order [M][N/32][K/16][2][h][32 r][8]: element (m, nb, kb, plane, h, r, j) =
// planes[plane][m][nb * 32 + r][kb * 16 + h * 8 + j]
void to_fragments(const _Float16 *planes, int M, int N, int K, _Float16 *out)
{
    const size_t plane = (size_t)M * N * K;
    size_t q = 0;
    for (int m = 0; m < M; ++m)
        for (int nb = 0; nb < N / 32; ++nb)
            for (int kb = 0; kb < K / 16; ++kb)
                for (int pl = 0; pl < 2; ++pl)
                    for (int h = 0; h < 2; ++h)
                        for (int r = 0; r < 32; ++r)
                            for (int j = 0; j < 8; ++j)
                                out[q++] = planes[pl * plane + ((size_t)m * N + nb * 32 + r) * K + kb * 16 + h * 8 + j];
}

}  // namespace

extern "C" size_t anihip_mlp_pack_bytes(const anihip_mlp_shape *sh)
{
    Geometry g;
    if (geometry(sh, &g)) return 0;
    Offsets off[ANIHIP_MAX_SPECIES];
    return layout(g, off);
}

extern "C" int anihip_mlp_pack(void *stream, const anihip_mlp_shape *sh, const float *const *weights,
                               const float *const *biases, int32_t src_on_device, void *out_buffer, size_t out_bytes,
                               int32_t dst_on_device, anihip_mlp_desc *desc)
{
    Geometry g;
    if (int rc = geometry(sh, &g)) return rc;
    ANIHIP_REQUIRE(weights && biases && out_buffer && desc, "null pointer argument");
    Offsets off[ANIHIP_MAX_SPECIES];
    const size_t total = layout(g, off);
    ANIHIP_REQUIRE(out_bytes >= total, "out_buffer holds %zu bytes, anihip_mlp_pack_bytes asks for %zu", out_bytes, total);
    const int M = g.M, S = g.S, nl = g.nl;
    std::vector<unsigned char> host(dst_on_device ? total : 0);
    unsigned char *buf = dst_on_device ? host.data() : (unsigned char *)out_buffer;
    memset(buf, 0, total);

    // sources on the host: W[m][s][l] [out][in], b[m][s][l] [out]
    std::vector<std::vector<float>> stage;
    auto fetch = [&](const float *p, size_t n) -> const float * {
        if (!src_on_device) return p;
        stage.emplace_back(n);
        if (hipMemcpy(stage.back().data(), p, 4 * n, hipMemcpyDeviceToHost) != hipSuccess) return nullptr;
        return stage.back().data();
    };
    memset(desc, 0, sizeof(*desc));
    desc->num_species = S; desc->n_members = M; desc->aev_len = g.K0; desc->celu_alpha = sh->celu_alpha;
    desc->precision = sh->precision; desc->aev_radial_len = g.R; desc->flags = 0; desc->activation = sh->activation;
    const unsigned char *base = (const unsigned char *)out_buffer;   // what the descriptor points into
    for (int s = 0; s < S; ++s) {
        anihip_species_net &net = desc->net[s];
        const Offsets &o = off[s];
        net.n_layers = nl;
        for (int l = 0; l <= nl; ++l) net.dims[l] = g.dims[s][l];
        std::vector<const float *> Wsrc((size_t)M * nl), Bsrc((size_t)M * nl);
        for (int m = 0; m < M; ++m)
            for (int l = 0; l < nl; ++l) {
                const size_t idx = ((size_t)m * S + s) * nl + l;
                ANIHIP_REQUIRE(weights[idx] && biases[idx], "weights / biases [m = %d][s = %d][l = %d] is NULL", m, s, l);
                Wsrc[(size_t)m * nl + l] = fetch(weights[idx], (size_t)g.out_dim[s][l] * g.in_dim[s][l]);
                Bsrc[(size_t)m * nl + l] = fetch(biases[idx], (size_t)g.out_dim[s][l]);
                ANIHIP_REQUIRE(Wsrc[(size_t)m * nl + l] && Bsrc[(size_t)m * nl + l], "copying the parameters from the device failed");
            }
        for (int l = 0; l < nl; ++l) {
            const int kin = g.dims[s][l], kout = g.dims[s][l + 1], in_u = g.in_dim[s][l], out_u = g.out_dim[s][l];
            float *w = (float *)(buf + o.w[l]), *bias = (float *)(buf + o.bias[l]);
            net.w[l] = (const float *)(base + o.w[l]);
            net.bias[l] = (const float *)(base + o.bias[l]);
            if (l == nl - 1) {   // w [M][kin], bias [M]
                for (int m = 0; m < M; ++m) {
                    memcpy(w + (size_t)m * kin, Wsrc[(size_t)m * nl + l], 4 * (size_t)in_u);
                    bias[m] = Bsrc[(size_t)m * nl + l][0];
                }
                continue;
            }
            float *wt = (float *)(buf + o.wt[l]);
            net.wt[l] = (const float *)(base + o.wt[l]);
            float amax = 0.f;
            if (l == 0) {   // w [K0][M * H1p], wt [M * H1p][K0p], bias [M * H1p]
                const size_t ld = (size_t)M * kout;
                for (int m = 0; m < M; ++m) {
                    const float *W = Wsrc[(size_t)m * nl + l], *b = Bsrc[(size_t)m * nl + l];
                    for (int oo = 0; oo < out_u; ++oo) {
                        const size_t col = (size_t)m * kout + oo;
                        bias[col] = b[oo];
                        for (int k = 0; k < in_u; ++k) {
                            const float v = W[(size_t)oo * in_u + k];
                            w[(size_t)k * ld + col] = v;
                            wt[col * g.K0p + k] = v;
                            amax = fmaxf(amax, fabsf(v));
                        }
                    }
                }
            } else {   // w [M][kin][kout] = W^T, wt [M][kout][kin] = W, bias [M][kout]
                for (int m = 0; m < M; ++m) {
                    const float *W = Wsrc[(size_t)m * nl + l], *b = Bsrc[(size_t)m * nl + l];
                    for (int oo = 0; oo < out_u; ++oo) {
                        bias[(size_t)m * kout + oo] = b[oo];
                        for (int k = 0; k < in_u; ++k) {
                            const float v = W[(size_t)oo * in_u + k];
                            w[((size_t)m * kin + k) * kout + oo] = v;
                            wt[((size_t)m * kout + oo) * kin + k] = v;
                            amax = fmaxf(amax, fabsf(v));
                        }
                    }
                }
            }
            if (!g.f16) continue;
            // power-of-two scale putting the largest weight into [2^13, 2^14); planes {hi, lo}
            const float scale = amax > 0.f ? ldexpf(1.0f, 13 - (int)floorf(log2f(amax))) : 1.0f;
            net.wh_scale[l] = scale;
            const int kred = l == 0 ? g.K0h : kin;
            const size_t n_el = (size_t)M * kout * kred;
            std::vector<float> fwd(n_el, 0.f), bwd(n_el, 0.f);
            if (l == 0) {   // forward [M * H1p][K0h] with the columns in slab order; backward = its transpose [K0h][M * H1p]
                const size_t ld = (size_t)M * kout;
                for (size_t row = 0; row < ld; ++row)
                    for (int k = 0; k < g.K0; ++k) {
                        const int col = g.R ? (k < g.R ? k : g.Rpad + (k - g.R)) : k;
                        const float v = wt[row * g.K0p + k];
                        fwd[row * kred + col] = v;
                        bwd[(size_t)col * ld + row] = v;
                    }
            } else {
                memcpy(fwd.data(), wt, 4 * n_el);
                memcpy(bwd.data(), w, 4 * n_el);
            }
            _Float16 *wh = (_Float16 *)(buf + o.wh[l]), *wth = (_Float16 *)(buf + o.wth[l]);
            split_planes(fwd.data(), n_el, scale, wh, wh + n_el);
            split_planes(bwd.data(), n_el, scale, wth, wth + n_el);
            net.wh[l] = base + o.wh[l];
            net.wth[l] = base + o.wth[l];
            // fragment order: forward planes as [2][M][N = kout][K = kred], backward planes (l >= 1) [2][M][N = kin][K = kout]
            to_fragments(wh, M, kout, kred, (_Float16 *)(buf + o.whf[l]));
            net.whf[l] = base + o.whf[l];
            if (l >= 1) {
                to_fragments(wth, M, kin, kout, (_Float16 *)(buf + o.wthf[l]));
                net.wthf[l] = base + o.wthf[l];
            } else if (g.fused) {
                // W0 transposed, member by member: planes [2][M][N = K0h (slab order)][K = H1p] -> fragments, so that a
                // column block of the layer-0 backward is one AEV slab and its k steps run over the member's H1 columns
                std::vector<_Float16> tp(2 * n_el);
                const size_t ld = (size_t)M * kout;
                for (int pl = 0; pl < 2; ++pl)
                    for (int m = 0; m < M; ++m)
                        for (int c = 0; c < kred; ++c)
                            for (int h = 0; h < kout; ++h)
                                tp[pl * n_el + ((size_t)m * kred + c) * kout + h] = wth[pl * n_el + (size_t)c * ld + (size_t)m * kout + h];
                to_fragments(tp.data(), M, kred, kout, (_Float16 *)(buf + o.wthf[l]));
                net.wthf[l] = base + o.wthf[l];
            }
        }
        if (g.fused) {   // operand bounds of the fused kernel's inner GEMMs, per member (include/anihip.h)
            float *bd = (float *)(buf + o.bounds);
            net.fused_bounds = (const float *)(base + o.bounds);
            const float dmax = sh->activation == ANIHIP_ACT_GELU ? 1.13f : 1.0f;   // max gelu' = 1.1290
            const int H1 = g.out_dim[s][0], H2 = g.out_dim[s][1], H3 = g.out_dim[s][2];
            for (int m = 0; m < M; ++m) {
                const float *W1 = Wsrc[(size_t)m * nl + 1], *W2 = Wsrc[(size_t)m * nl + 2], *w3 = Wsrc[(size_t)m * nl + 3],
                            *b1 = Bsrc[(size_t)m * nl + 1];
                float row1 = 0.f, col1 = 0.f, col2 = 0.f, bmax = 0.f, w3max = 0.f;
                std::vector<float> c1(H1, 0.f), c2(H2, 0.f);
                for (int j = 0; j < H2; ++j) {
                    float rs = 0.f;
                    for (int k = 0; k < H1; ++k) {
                        const float v = fabsf(W1[(size_t)j * H1 + k]);
                        rs += v;
                        c1[k] += v;
                    }
                    row1 = fmaxf(row1, rs);
                    bmax = fmaxf(bmax, fabsf(b1[j]));
                }
                for (int j = 0; j < H3; ++j) {
                    for (int k = 0; k < H2; ++k) c2[k] += fabsf(W2[(size_t)j * H2 + k]);
                    w3max = fmaxf(w3max, fabsf(w3[j]));
                }
                for (int k = 0; k < H1; ++k) col1 = fmaxf(col1, c1[k]);
                for (int k = 0; k < H2; ++k) col2 = fmaxf(col2, c2[k]);
                const float g2 = w3max / (float)M * dmax, g3 = g2 * col2 * dmax, g4 = g3 * col1;
                float *q = bd + 8 * (size_t)m;
                q[0] = row1; q[1] = bmax; q[2] = g2; q[3] = g3; q[4] = g4; q[5] = q[6] = q[7] = 0.f;
            }
        }
        stage.clear();
    }
    if (dst_on_device) {
        ANIHIP_CHECK_HIP(hipMemcpyAsync(out_buffer, buf, total, hipMemcpyHostToDevice, (hipStream_t)stream));
        ANIHIP_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));   // (the staging buffer dies with this call)
    }
    return 0;
}

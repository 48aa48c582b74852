"""The torch restatement of the parameter layouts of include/anihip.h (it was the product packer until round 3; the product
now calls anihip_mlp_pack, torchani_amd/csrc/pack.hip).  Test infrastructure: tests compare the C packer's buffer with
these tensors bit for bit."""
import numpy as np
import torch


def _pad32(x):
    return (x + 31) // 32 * 32


def pack_reference(weights, biases, aev_len, precision="f16x3", radial_len=None, activation="celu"):
    """-> {(species, name, layer): tensor}, scales {(species, layer): float}, radial_len.  names: w, wt, bias, wh, wth, whf,
    wthf; (species, "bounds", 0)."""
    M, S = len(weights), len(weights[0])
    nl = len(weights[0][0])
    out, scales = {}, {}
    k0p = _pad32(aev_len)
    if radial_len is None:
        radial_len = 16 * S if aev_len == 16 * S + 16 * S * (S + 1) else 0
    if precision != "f16x3" or radial_len <= 0 or (aev_len - radial_len) % 32 != 0:
        radial_len = 0
    rpad = _pad32(radial_len)
    k0h = rpad + (aev_len - radial_len) if radial_len else k0p
    slab_cols = torch.cat([torch.arange(radial_len), rpad + torch.arange(aev_len - radial_len)]) \
        if radial_len else torch.arange(aev_len)
    f32 = dict(dtype=torch.float32)
    for s in range(S):
        dims = [aev_len] + [_pad32(weights[0][s][l].shape[0]) for l in range(nl - 1)] + [1]
        for l in range(nl):
            kin, kout = dims[l], dims[l + 1]
            Ws, Bs = [], []
            for m in range(M):
                W = weights[m][s][l].detach().to(**f32)
                b = biases[m][s][l].detach().to(**f32)
                Wp = torch.zeros((kout, kin), **f32)
                Wp[: W.shape[0], : W.shape[1]] = W
                bp = torch.zeros((kout,), **f32)
                bp[: b.shape[0]] = b
                Ws.append(Wp)
                Bs.append(bp)
            Wst = torch.stack(Ws)   # [M, out_p, in_p]
            bst = torch.stack(Bs)   # [M, out_p]
            if l == nl - 1:
                w = Wst.reshape(M, kin).contiguous()
                wt = None
                bias = bst.reshape(M).contiguous()
            elif l == 0:
                cat = Wst.reshape(M * kout, kin)              # row m*H1p+o
                w = cat.t().contiguous()                      # [K0, M*H1p]
                wt = torch.zeros((M * kout, k0p), **f32)      # [M*H1p, K0p]
                wt[:, :kin] = cat
                bias = bst.reshape(M * kout).contiguous()
            else:
                w = Wst.transpose(1, 2).contiguous()          # [M, in_p, out_p]
                wt = Wst.contiguous()                         # [M, out_p, in_p]
                bias = bst.contiguous()
            out[(s, "w", l)], out[(s, "bias", l)] = w, bias
            if wt is not None:
                out[(s, "wt", l)] = wt
            if precision == "f16x3" and wt is not None:
                amax = float(Wst.abs().max())
                scale = 2.0 ** (13 - int(np.floor(np.log2(amax)))) if amax > 0 else 1.0
                if l == 0:
                    fwd_src = torch.zeros((M * kout, k0h), **f32)   # wt, columns in slab order
                    fwd_src[:, slab_cols] = cat
                    bwd_src = fwd_src.t().contiguous()              # [K0h, M*H1p]
                else:
                    fwd_src, bwd_src = wt, w
                planes = []
                for src in (fwd_src, bwd_src):
                    x = src * scale
                    hi = x.to(torch.float16)
                    lo = (x - hi.to(torch.float32)).to(torch.float16)
                    planes.append(torch.stack([hi, lo]).contiguous())
                out[(s, "wh", l)], out[(s, "wth", l)] = planes
                scales[(s, l)] = scale
                frags = []
                for pl in (planes if l >= 1 else [planes[0].view(2, M, kout, k0h)]):
                    N_, K_ = pl.shape[2], pl.shape[3]
                    f = pl.view(2, M, N_ // 32, 32, K_ // 16, 2, 8).permute(1, 2, 4, 0, 5, 3, 6)
                    frags.append(f.contiguous())
                out[(s, "whf", l)] = frags[0]
                if l >= 1:
                    out[(s, "wthf", l)] = frags[1]
                elif nl == 4:
                    # layer 0, transposed, member by member: planes [2][M][N = K0h][K = H1p] in fragment order (the layer-0
                    # backward inside the fused kernel: a column block is one AEV slab, k runs over the member's hidden columns)
                    tp = planes[1].view(2, k0h, M, kout).permute(0, 2, 1, 3).contiguous()
                    f = tp.view(2, M, k0h // 32, 32, kout // 16, 2, 8).permute(1, 2, 4, 0, 5, 3, 6)
                    out[(s, "wthf", 0)] = f.contiguous()
        if precision == "f16x3" and nl == 4:
            W1 = torch.stack([weights[m][s][1].detach().to(**f32) for m in range(M)])   # [M, H2, H1]
            W2 = torch.stack([weights[m][s][2].detach().to(**f32) for m in range(M)])   # [M, H3, H2]
            b1 = torch.stack([biases[m][s][1].detach().to(**f32) for m in range(M)])
            w3 = torch.stack([weights[m][s][3].detach().to(**f32).reshape(-1) for m in range(M)])
            dmax = 1.13 if activation == "gelu" else 1.0
            g2 = w3.abs().amax(dim=1) / M * dmax
            g3 = g2 * W2.abs().sum(dim=1).amax(dim=1) * dmax
            g4 = g3 * W1.abs().sum(dim=1).amax(dim=1)
            zero = torch.zeros_like(g2)
            out[(s, "bounds", 0)] = torch.stack([W1.abs().sum(dim=2).amax(dim=1), b1.abs().amax(dim=1), g2, g3, g4,
                                                 zero, zero, zero], dim=1).contiguous()
    return out, scales, radial_len

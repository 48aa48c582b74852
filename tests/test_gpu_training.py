"""GPU parity of the training pass (weight / bias gradients, BASELINE config 5) against the CPU oracle, whose
gradients are pinned to the reference's autograd by tests/golden/wgrads_*.npz (tests/test_oracle_golden.py).

Tolerance: gradients are sums over atoms of fp32 products accumulated in fp32 (MFMA + float atomics); they are held
to 2e-5 of the largest gradient entry of the case (measured: ~1e-6).
"""
import os

import numpy as np
import pytest
import torch

from _util import (FGRAD_NAMES, WGRAD_NAMES, fgrad_direction, load_fgrads, load_golden, oracle_networks, oracle_params,
                   seeded_state, wgrad_upstream)
from test_gpu_parity import report

pytestmark = pytest.mark.gpu

WG_REL_TOL = 2e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from torchani_amd import _lib

    _lib.lib()  # fail loudly if the native library is missing
    return torch.device("cuda:0")


def fresh_model(kind, seed, dev, cutoff_fn="cosine"):
    from torchani_amd.models import ANI1x, ANI2x

    ctor = ANI2x if kind == "ani2x" else ANI1x
    return ctor(state_dict=seeded_state(kind, 8, seed), device=dev, periodic_table_index=False, cutoff_fn=cutoff_fn)


def flat_from_lists(gw, gb, M, S, nl):
    """Engine gradients (Linear layout lists) -> the oracle's packed layout (oracle.pack_networks)."""
    out = []
    for m in range(M):
        for s in range(S):
            for l in range(nl):
                out.append(gw[m][s][l].detach().cpu().numpy().astype(np.float64).reshape(-1))
                out.append(gb[m][s][l].detach().cpu().numpy().astype(np.float64).reshape(-1))
    return np.concatenate(out)


def flat_from_params(nets, symbols):
    out = []
    for member in nets.members:
        for sym in symbols:
            for lin in member.atomics[sym].linears():
                for p in (lin.weight, lin.bias):
                    g = p.grad if p.grad is not None else torch.zeros_like(p)
                    out.append(g.detach().cpu().numpy().astype(np.float64).reshape(-1))
    return np.concatenate(out)


@pytest.mark.parametrize("base", WGRAD_NAMES)
@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
def test_weight_grads_match_oracle(dev, oracle64, base, precision):
    g = load_golden(base)
    dims, flat, _ = oracle_networks(g["kind"], g["n_members"], g["seed"])
    p = oracle_params(g["kind"], g["cutoff_fn"])
    aev = oracle64.aev(p, g["species"], g["coords"].astype(np.float64), g["cell"], g["pbc"])
    a32 = aev.astype(np.float32)
    C, A = g["species"].shape
    up = wgrad_upstream(C, A).astype(np.float32)
    ref = oracle64.mlp_weight_grads(g["species"], a32.astype(np.float64), up.astype(np.float64), dims, flat,
                                    n_members=8)
    ae, ga, _ = oracle64.mlp(g["species"], a32.astype(np.float64), dims, flat, n_members=8)
    model = fresh_model(g["kind"], g["seed"], dev, g["cutoff_fn"])
    nets = model.neural_networks
    nets.mlp_precision = precision
    packed = nets._pack(dev)
    sp32 = torch.from_numpy(g["species"].astype(np.int32)).to(dev)
    at = torch.from_numpy(a32).to(dev).view(C * A, -1)
    gw, gb, e, gaev = packed.weight_grads(sp32, at, torch.from_numpy(up).to(dev), want_grad_aev=True)
    torch.cuda.synchronize()
    got = flat_from_lists(gw, gb, packed.M, packed.S, packed.nl)
    assert got.shape == ref.shape
    scale = np.abs(ref).max()
    err = np.abs(got - ref).max()
    e_err = np.abs(e.cpu().numpy() - ae).max()
    ga_ref = ga * up.reshape(-1, 1)
    ga_err = np.abs(gaev.cpu().numpy() - ga_ref).max()
    report(f"wgrad {base:22s} {precision:6s} max|dL/dw err| = {err:.2e} (max |dL/dw| {scale:.2e})  "
           f"|e_atom err| = {e_err:.2e}  |dL/daev err| = {ga_err:.2e}")
    assert err < WG_REL_TOL * scale
    assert e_err < 3e-7
    assert ga_err < 1e-6 + 1e-5 * np.abs(ga_ref).max()
    # chunked evaluation (gradients of the chunks are summed) gives the same result
    gw2, gb2, e2, _ = packed.weight_grads(sp32, at, torch.from_numpy(up).to(dev), chunk=max(7, (C * A) // 3))
    got2 = flat_from_lists(gw2, gb2, packed.M, packed.S, packed.nl)
    assert np.abs(got2 - ref).max() < WG_REL_TOL * scale
    assert np.abs(e2.cpu().numpy() - ae).max() < 3e-7


@pytest.mark.parametrize("base", WGRAD_NAMES)
def test_fast_training_pass_matches_oracle(dev, oracle64, base):
    """Round 5, the fast training path (include/anihip.h, anihip_mlp_weight_grads of an F16X3 pack): forward + unit-gradient
    backward in ONE launch of the fused kernel's TRAIN instantiation, weight gradients on bf16 x 3 MFMA with the upstream
    gradient as a row scale, bias gradients by column reduction -- against the fp64 oracle, at the tolerance of the
    exact-fp32 passes; the two halves (train_forward now, weight_grads later) and the one-call form; gradients written
    straight into a flat per-member buffer (member_stride / accumulate) equal the packed ones."""
    g = load_golden(base)
    dims, flat, _ = oracle_networks(g["kind"], g["n_members"], g["seed"])
    p = oracle_params(g["kind"], g["cutoff_fn"])
    aev = oracle64.aev(p, g["species"], g["coords"].astype(np.float64), g["cell"], g["pbc"])
    a32 = aev.astype(np.float32)
    C, A = g["species"].shape
    up = wgrad_upstream(C, A).astype(np.float32)
    ref = oracle64.mlp_weight_grads(g["species"], a32.astype(np.float64), up.astype(np.float64), dims, flat, n_members=8)
    ae, _, _ = oracle64.mlp(g["species"], a32.astype(np.float64), dims, flat, n_members=8)
    model = fresh_model(g["kind"], g["seed"], dev, g["cutoff_fn"])
    nets = model.neural_networks
    nets.requires_grad_(True)
    assert nets._fast_trainable()
    packed = nets._train_pack(dev, fast=True)
    assert packed.precision == "f16x3" and packed.fast_training()
    sp32 = torch.from_numpy(g["species"].astype(np.int32)).to(dev)
    at = torch.from_numpy(a32).to(dev).view(C * A, -1)
    upd = torch.from_numpy(up).to(dev)
    e_fwd, ws = packed.train_forward(sp32, at)
    gw, gb, e, _ = packed.weight_grads(sp32, at, upd, workspace=ws)
    torch.cuda.synchronize()
    got = flat_from_lists(gw, gb, packed.M, packed.S, packed.nl)
    scale = np.abs(ref).max()
    err = np.abs(got - ref).max()
    e_err = np.abs(e_fwd.cpu().numpy() - ae).max()
    report(f"fast train {base:22s} max|dL/dw err| = {err:.2e} (max |dL/dw| {scale:.2e})  |e_atom err| = {e_err:.2e}")
    assert err < WG_REL_TOL * scale
    assert e_err < 3e-7
    # one call (its own forward), in three chunks: the chunks' gradients are summed
    gw2, gb2, e2, _ = packed.weight_grads(sp32, at, upd, chunk=max(7, (C * A) // 3))
    got2 = flat_from_lists(gw2, gb2, packed.M, packed.S, packed.nl)
    assert np.abs(got2 - ref).max() < WG_REL_TOL * scale
    assert np.abs(e2.cpu().numpy() - ae).max() < 3e-7
    # the same into a flat buffer in torch's parameter order (member -> species -> layer -> weight, bias), ACCUMULATED: twice
    # (needs widths that are multiples of 32: ANI-2x; ANI-1x's 144- and 112-wide layers are padded in the engine)
    if any(packed.desc.net[s_].dims[l + 1] != packed.shapes[s_][l][0] for s_ in range(packed.S) for l in range(packed.nl - 1)):
        with pytest.raises(ValueError, match="multiples of 32"):
            packed.flat_grad_target([[0] * packed.nl] * packed.S, [[0] * packed.nl] * packed.S, 1)
        return
    params = [q for q in nets.parameters()]
    sizes = [q.numel() for q in params]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    flat_g = torch.zeros(int(offs[-1]), dtype=torch.float32, device=dev)
    M, S, nl = packed.M, packed.S, packed.nl
    per = 2 * S * nl
    base_ptr = flat_g.data_ptr()
    w_ptr = [[base_ptr + 4 * int(offs[(s_ * nl + l) * 2]) for l in range(nl)] for s_ in range(S)]
    b_ptr = [[base_ptr + 4 * int(offs[(s_ * nl + l) * 2 + 1]) for l in range(nl)] for s_ in range(S)]
    tgt = packed.flat_grad_target(w_ptr, b_ptr, int(offs[per]))
    for _ in range(2):
        _, ws = packed.train_forward(sp32, at)
        packed.weight_grads(sp32, at, upd, workspace=ws, target=tgt)
    torch.cuda.synchronize()
    fg = flat_g.cpu().numpy().astype(np.float64)
    # (flat_from_lists orders member -> species -> layer -> weight, bias as well)
    assert fg.shape == ref.shape
    assert np.abs(0.5 * fg - ref).max() < WG_REL_TOL * scale


def test_stale_layouts_are_refreshed_before_a_layer_by_layer_pass(dev):
    """A fused-only device refresh leaves the layer-by-layer layouts of a split-fp16 pack at the OLD parameters
    (PackedNetworks.stale_layouts).  A pass the library serves layer by layer -- here: weight gradients with d Loss / d aev --
    runs the full repack first, so its results are those of a pack built from the new parameters (round-5 advice)."""
    from torchani_amd.engine import PackedNetworks

    model = fresh_model("ani2x", 4, dev)
    nets = model.neural_networks
    nets.requires_grad_(True)
    packed = nets._train_pack(dev, fast=True)
    rs = torch.Generator(device="cpu").manual_seed(6)
    with torch.no_grad():
        for q in nets.parameters():
            q.mul_(1.0 + 0.3 * (torch.rand(q.shape, generator=rs) - 0.5).to(dev))   # (a change no tolerance below hides)
    assert nets._train_pack(dev, fast=True) is packed and packed.stale_layouts
    n = 300
    sp = torch.randint(0, 7, (n,), generator=rs).to(torch.int32).to(dev)
    aev = torch.rand((n, packed.aev_len), generator=rs).to(dev) * 0.3
    g = (torch.rand(n, generator=rs) - 0.5).to(dev)
    gw, gb, e, ga = packed.weight_grads(sp, aev, g, want_grad_aev=True)
    assert not packed.stale_layouts
    members = nets._member_networks()
    lins = [[m.atomics[s].linears() for s in nets.symbols] for m in members]
    weights = [[[lin.weight for lin in sl] for sl in ml] for ml in lins]
    biases = [[[lin.bias for lin in sl] for sl in ml] for ml in lins]
    fresh = PackedNetworks(weights, biases, packed.aev_len, 0.1, dev, "f16x3")
    gw2, gb2, e2, ga2 = fresh.weight_grads(sp, aev, g, want_grad_aev=True)
    assert float((e - e2).abs().max()) < 1e-6 * max(1.0, float(e2.abs().max()))
    assert float((ga - ga2).abs().max()) < 1e-6 * float(ga2.abs().max())
    assert float((gw[0][0][1] - gw2[0][0][1]).abs().max()) <= 1e-5 * float(gw2[0][0][1].abs().max())


def test_device_repack_of_a_split_fp16_pack_equals_the_host_packer(dev):
    """anihip_mlp_repack of an F16X3 descriptor (round 5) rewrites every layout on the device: after the parameters changed,
    the refreshed buffer equals, byte for byte outside the operand bounds, what anihip_mlp_pack builds on the host from the new
    values (same weight scales: the change keeps every layer's largest weight in its binade)."""
    from torchani_amd.engine import PackedNetworks

    model = fresh_model("ani2x", 3, dev)
    nets = model.neural_networks
    nets.requires_grad_(True)
    packed = nets._train_pack(dev, fast=True)
    rs = torch.Generator(device="cpu").manual_seed(5)
    with torch.no_grad():
        for q in nets.parameters():
            q.mul_(1.0 + 0.02 * (torch.rand(q.shape, generator=rs) - 0.5).to(dev))   # (in place: versions bump)
    again = nets._train_pack(dev, fast=True)
    assert again is packed and packed.stale_layouts   # refreshed in place (the fused kernel's layouts only), not rebuilt
    members = nets._member_networks()
    lins = [[m.atomics[s].linears() for s in nets.symbols] for m in members]
    weights = [[[lin.weight for lin in sl] for sl in ml] for ml in lins]
    biases = [[[lin.bias for lin in sl] for sl in ml] for ml in lins]
    packed.refresh(weights, biases)   # ... and now every layout
    torch.cuda.synchronize()
    assert not packed.stale_layouts and not packed.scale_overflowed()
    fresh = PackedNetworks(weights, biases, packed.aev_len, 0.1, dev, "f16x3")
    M, nl, K0 = packed.M, packed.nl, packed.aev_len
    K0p = (K0 + 31) // 32 * 32
    K0h = 128 + (K0 - 112)   # (ANI-2x: radial part padded to 128)
    compared = 0
    for s in range(packed.S):
        a, b = packed.desc.net[s], fresh.desc.net[s]
        bd_a = packed.array(a.fused_bounds, (8 * M,))
        bd_b = fresh.array(b.fused_bounds, (8 * M,))
        assert torch.all(bd_a >= bd_b) and torch.allclose(bd_a, bd_b, rtol=3e-4)   # (upper bounds, a hair above the host's)
        for l in range(nl):
            kin, kout = a.dims[l], a.dims[l + 1]
            if l == nl - 1:
                assert torch.equal(packed.array(a.w[l], (M * kin,)), fresh.array(b.w[l], (M * kin,)))
                assert torch.equal(packed.array(a.bias[l], (M,)), fresh.array(b.bias[l], (M,)))
                continue
            n_w, n_wt = M * kin * kout, M * kout * (K0p if l == 0 else kin)
            assert torch.equal(packed.array(a.w[l], (n_w,)), fresh.array(b.w[l], (n_w,)))
            assert torch.equal(packed.array(a.wt[l], (n_wt,)), fresh.array(b.wt[l], (n_wt,)))
            assert torch.equal(packed.array(a.bias[l], (M * kout,)), fresh.array(b.bias[l], (M * kout,)))
            if a.wh_scale[l] != b.wh_scale[l]:
                continue   # (the host packer chose a new scale for the changed weights: other planes, equally valid)
            n_h = 2 * M * kout * (K0h if l == 0 else kin)
            for name in ("wh", "wth", "whf", "wthf"):
                pa, pb = getattr(a, name)[l], getattr(b, name)[l]
                assert bool(pa) == bool(pb)
                if pa:
                    assert torch.equal(packed.array(pa, (n_h,), torch.float16), fresh.array(pb, (n_h,), torch.float16)), (s, l, name)
                    compared += 1
    assert compared >= 3 * 4 * packed.S // 2   # (most layers keep their scale under a 1 % change)


def test_split_fp16_pack_is_rebuilt_when_a_weight_outgrows_its_scale(dev):
    """The device refresh keeps the power-of-two weight scales of the pack (largest weight in [2^13, 2^14) of the fp16 range):
    a layer whose weights grew sixteen-fold no longer fits, the refresh says so in its status word (read one call late), and
    the container packs again on the host -- energies stay right throughout the calls that follow."""
    g = load_golden("rand_batch_ani2x")
    model = fresh_model("ani2x", g["seed"], dev)
    nets = model.neural_networks
    nets.requires_grad_(True)
    sp = torch.from_numpy(g["species"].astype(np.int64)).to(dev)
    x = torch.from_numpy(g["coords"]).to(dev)
    aev = model.aev_computer(sp, x).detach()
    first = nets._train_pack(dev, fast=True)
    with torch.no_grad():
        nets.members[2].atomics["C"].layers[1].weight.mul_(16.0)
    e_ref = None
    packs = []
    for _ in range(3):
        e = nets(sp, aev)
        torch.cuda.synchronize()
        # (round 6: the step that overflows is harmless too -- the device repack clamps to the fp16 range instead of writing
        # hi = inf, lo = -inf, whose sum is NaN in the forward, the gradients and, through the optimizer, the parameters)
        assert torch.isfinite(e).all(), "the overflowing step itself must stay finite"
        packs.append(nets._train_pack(dev, fast=True))
        with torch.no_grad():   # (bump a version so that the next call refreshes / polls again)
            nets.members[0].atomics["H"].layers[0].bias.add_(0.0)
    assert packs[-1] is not first, "the pack was not rebuilt after a weight left the fp16 range of its scale"
    nets.train_precision = "fp32"
    e_ref = nets(sp, aev)
    e, e_ref = e.detach(), e_ref.detach()
    assert torch.isfinite(e).all() and float((e - e_ref).abs().max()) < 1e-5 * max(1.0, float(e_ref.abs().max()))


@pytest.mark.parametrize("weight_decay", [0.0, 0.01])
def test_fused_adam_matches_torch_adam(dev, weight_decay):
    """torchani_amd.optim.Adam (anihip_adam_step: one launch over flat buffers, step count on the device) follows
    torch.optim.Adam to 1e-6 relative over 25 steps of random gradients; the parameters keep their values when they are
    re-homed into the flat buffer; step() leaves the gradients zeroed (zero_grad_in_step)."""
    from torchani_amd.optim import Adam

    gen = torch.Generator(device="cpu").manual_seed(3)
    shapes = [(256, 1008), (256,), (192, 256), (192,), (1, 160), (1,), (7, 5, 3)]
    init = [torch.randn(sh, generator=gen) * 0.3 for sh in shapes]
    pa = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    pb = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    ours = Adam(pa, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=weight_decay)
    theirs = torch.optim.Adam(pb, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=weight_decay)
    for a, t in zip(pa, init):
        assert torch.equal(a.detach().cpu(), t)
    flat = ours._flat[0]
    assert all(a.grad is v for a, v in zip(pa, flat.grad_views))
    for step in range(25):
        for i, (a, b) in enumerate(zip(pa, pb)):
            gr = (torch.randn(a.shape, generator=gen) * (10.0 ** ((step % 5) - 3))).to(dev)
            if step % 3 == 0 and i == 2:
                a.grad = gr.clone()           # a gradient that arrives as a tensor of its own (autograd's default route)
            else:
                a.grad.copy_(gr)
            b.grad = gr.clone()
        ours.step()
        theirs.step()
        assert all(a.grad is v for a, v in zip(pa, flat.grad_views))
        assert float(flat.grad.abs().max()) == 0.0
    torch.cuda.synchronize()
    assert int(flat.step.item()) == 25
    for a, b in zip(pa, pb):
        assert float((a.detach() - b.detach()).abs().max()) <= 1e-6 * float(b.detach().abs().max())
    # the state dict has torch.optim.Optimizer's layout: torch.optim.Adam's own loads into ours and the other way round
    sd = ours.state_dict()
    sd_t = theirs.state_dict()
    assert set(sd) == {"state", "param_groups"} and sd["param_groups"][0]["params"] == sd_t["param_groups"][0]["params"]
    for k in sd_t["state"]:
        assert float(sd["state"][k]["step"]) == float(sd_t["state"][k]["step"]) == 25.0
        ref = sd_t["state"][k]["exp_avg_sq"]
        assert float((sd["state"][k]["exp_avg_sq"] - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    ours.load_state_dict(sd_t)
    assert int(flat.step.item()) == 25
    back = ours.state_dict()
    assert all(torch.equal(back["state"][k]["exp_avg"], sd_t["state"][k]["exp_avg"]) for k in sd_t["state"])
    theirs.load_state_dict(sd)
    ours.load_state_dict(sd)
    # a parameter without a gradient counts as a zero gradient (documented) -- and is refused where that would decay it
    pa[3].grad = None
    if weight_decay > 0.0:
        with pytest.raises(RuntimeError, match="without a gradient"):
            ours.step()
    else:
        ours.step()
    # a parameter that left the flat buffer is noticed, not silently skipped
    pa[1].data = pa[1].data.clone()
    with pytest.raises(RuntimeError, match="flat buffer"):
        for _ in range(4):   # (all parameters are looked at every 4th step, the first and the last one every step)
            ours.step()


def test_parameter_hooks_take_the_autograd_route(dev):
    """With torchani_amd.optim.Adam the weight-gradient kernels add straight into the optimizer's flat buffer and autograd
    produces nothing for the parameters -- unless a parameter carries a tensor hook: then the gradients go through autograd
    (the hook fires, p.grad still ends up in the flat buffer's view at step())."""
    from torchani_amd.optim import Adam

    model = fresh_model("ani2x", 9, dev)
    nets = model.neural_networks
    nets.requires_grad_(True)
    opt = Adam(nets.parameters(), lr=1e-4)
    gen = torch.Generator(device="cpu").manual_seed(2)
    sp = torch.randint(0, 4, (4, 16), generator=gen).to(dev)
    aev = (torch.rand((4, 16, 1008), generator=gen) * 0.2).to(dev)
    nets(sp, aev).sum().backward()
    flat = opt._flat[0]
    g_flat = flat.grad.clone()
    assert float(g_flat.abs().max()) > 0 and nets.__dict__.get("_flat_target_cache") is not None
    opt.zero_grad()
    fired = []
    p0 = next(iter(nets.parameters()))
    p0.register_hook(lambda g: fired.append(float(g.abs().max())))
    nets.__dict__.pop("_flat_target_cache")   # (the hooks are looked at when the gradient table is built)
    nets(sp, aev).sum().backward()
    assert fired and fired[0] > 0
    flat.gather_stray_grads()
    assert float((flat.grad - g_flat).abs().max()) <= 2e-5 * float(g_flat.abs().max())


def test_training_step_with_the_flat_optimizer(dev, oracle64):
    """The reference's recipe (tools/training-aev-benchmark.py:88,120-135) with torchani_amd.optim.Adam in torch.optim.Adam's
    place: the containers find their parameters' gradients in ONE flat buffer and the weight-gradient kernels add straight
    into it (autograd accumulates nothing), .grad of every parameter equals the oracle's gradient, the losses of a short run
    follow those of torch.optim.Adam on the exact-fp32 passes, and nets.zero_grad(set_to_none=True) -- gradients back through
    autograd -- still trains."""
    from torchani_amd.optim import Adam

    g = load_golden("rand_batch_ani2x")
    sp = torch.from_numpy(g["species"].astype(np.int64)).to(dev)
    x = torch.from_numpy(g["coords"]).to(dev)
    C, A = g["species"].shape
    target = torch.from_numpy(np.linspace(-0.3, 0.4, C)).to(dev)
    runs = {}
    for kind in ("fused", "torch"):
        model = fresh_model("ani2x", g["seed"], dev)
        nets = model.neural_networks
        nets.requires_grad_(True)
        aev = model.aev_computer(sp, x).detach()
        if kind == "fused":
            opt = Adam(nets.parameters(), lr=1e-4)
        else:
            nets.train_precision = "fp32"
            opt = torch.optim.Adam(nets.parameters(), lr=1e-4)
        losses = []
        for it in range(6):
            opt.zero_grad()
            e = nets(sp, aev).double()
            loss = ((e - target) ** 2).sum()
            loss.backward()
            if it == 0 and kind == "fused":
                pk = nets._train_pack(dev, fast=True)
                assert pk.flat_target is not None, "the gradients did not take the direct route"
                dims, flat, _ = oracle_networks("ani2x", 8, g["seed"])
                a64 = aev.cpu().numpy().astype(np.float64).reshape(C * A, -1)
                ae, _, _ = oracle64.mlp(g["species"], a64, dims, flat, n_members=8, want_grad=False)
                e_ref = ae.reshape(C, A).sum(axis=1)
                up = np.repeat(2.0 * (e_ref - target.cpu().numpy())[:, None], A, axis=1)
                ref = oracle64.mlp_weight_grads(g["species"], a64, up, dims, flat, n_members=8)
                got = flat_from_params(nets, nets.symbols)
                scale = np.abs(ref).max()
                err = np.abs(got - ref).max()
                report(f"train flat-optimizer rand_batch_ani2x  max|grad err| = {err:.2e} (max |grad| {scale:.2e})")
                assert err < 5e-5 * scale
            opt.step()
            losses.append(float(loss.detach()))
        runs[kind] = losses
        if kind == "fused":
            # gradients set to None: they come back through autograd as tensors, step() gathers them into the flat buffer
            nets.zero_grad(set_to_none=True)
            e = nets(sp, aev).double()
            loss = ((e - target) ** 2).sum()
            loss.backward()
            assert all(q.grad is not None for q in nets.parameters())
            opt.step()
            e2 = nets(sp, aev).double()
            assert float(((e2 - target) ** 2).sum()) < float(loss)
    report("train flat-optimizer losses " + " ".join(f"{v:.5f}" for v in runs["fused"]) + "   torch.optim.Adam / fp32: "
           + " ".join(f"{v:.5f}" for v in runs["torch"]))
    assert runs["fused"][-1] < 0.7 * runs["fused"][0]
    for a, b in zip(runs["fused"], runs["torch"]):
        assert abs(a - b) < 2e-4 * max(1.0, abs(b))


def test_autograd_training_step(dev, oracle64):
    """loss.backward() through the containers fills .grad of every Linear parameter (what the reference's training
    loop relies on, tools/training-aev-benchmark.py:120-135), and a few Adam steps reduce the loss."""
    g = load_golden("rand_batch_ani2x")
    model = fresh_model("ani2x", g["seed"], dev)
    nets = model.neural_networks
    nets.requires_grad_(True)
    sp = torch.from_numpy(g["species"].astype(np.int64)).to(dev)
    x = torch.from_numpy(g["coords"]).to(dev)
    C, A = g["species"].shape
    target = torch.from_numpy(np.linspace(-0.3, 0.4, C)).to(dev)
    aev = model.aev_computer(sp, x).detach()
    e = nets(sp, aev).double()
    loss = ((e - target) ** 2).sum()
    loss.backward()
    torch.cuda.synchronize()
    # oracle: upstream per atom = 2 (E_c - target_c) on the fp32 AEVs the engine produced
    dims, flat, _ = oracle_networks("ani2x", 8, g["seed"])
    a64 = aev.cpu().numpy().astype(np.float64).reshape(C * A, -1)
    ae, _, _ = oracle64.mlp(g["species"], a64, dims, flat, n_members=8, want_grad=False)
    e_ref = ae.reshape(C, A).sum(axis=1)
    up = np.repeat(2.0 * (e_ref - target.cpu().numpy())[:, None], A, axis=1)
    ref = oracle64.mlp_weight_grads(g["species"], a64, up, dims, flat, n_members=8)
    got = flat_from_params(nets, model.symbols if hasattr(model, "symbols") else nets.symbols)
    scale = np.abs(ref).max()
    err = np.abs(got - ref).max()
    report(f"train autograd rand_batch_ani2x  max|grad err| = {err:.2e} (max |grad| {scale:.2e})")
    assert err < 5e-5 * scale
    # third order is not implemented: it must raise, not return silently wrong numbers
    aev_g = aev.clone().requires_grad_(True)
    e2 = nets(sp, aev_g).sum()
    (ga,) = torch.autograd.grad(e2, aev_g, create_graph=True)
    (gp,) = torch.autograd.grad(ga.pow(2).sum(), nets.members[0].atomics["H"].layers[0].weight, create_graph=True)
    with pytest.raises(RuntimeError):
        gp.sum().backward()
    nets.zero_grad()
    # a short optimisation run
    opt = torch.optim.Adam(nets.parameters(), lr=1e-4)
    losses = []
    for _ in range(6):
        opt.zero_grad()
        e = nets(sp, aev).double()
        loss = ((e - target) ** 2).sum()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    report("train adam losses " + " ".join(f"{v:.5f}" for v in losses))
    assert losses[-1] < 0.7 * losses[0]


def test_frozen_species_and_inference_unchanged(dev):
    """Parameters without requires_grad get no gradient; with everything frozen the containers take the inference
    path (no training pass, d/d aev from the fused kernel)."""
    g = load_golden("rand_batch_ani2x")
    model = fresh_model("ani2x", g["seed"], dev)
    nets = model.neural_networks
    sp = torch.from_numpy(g["species"].astype(np.int64)).to(dev)
    x = torch.from_numpy(g["coords"]).to(dev)
    aev = model.aev_computer(sp, x).detach().requires_grad_(True)
    e0 = nets(sp, aev).sum()
    (g0,) = torch.autograd.grad(e0, aev)
    nets.members[0].atomics["H"].requires_grad_(True)
    e1 = nets(sp, aev).sum()
    e1.backward()
    torch.cuda.synchronize()
    assert abs(float(e1) - float(e0)) < 1e-6
    # the training pass computes d/d aev in exact fp32, the inference path in split fp16: same to round-off
    assert (aev.grad - g0).abs().max().item() < 1e-6 + 1e-5 * g0.abs().max().item()
    got = [p.grad is not None for p in nets.members[0].atomics["H"].parameters()]
    assert all(got)
    assert all(p.grad is None for p in nets.members[1].parameters())
    assert all(p.grad is None for p in nets.members[0].atomics["C"].parameters())


@pytest.mark.parametrize("base", FGRAD_NAMES)
def test_aev_jvp_matches_reference(dev, oracle64, base):
    """anihip_aev_jvp (the reference's cuaev double backward: J t) against the reference's forward-mode derivative of
    its AEVComputer (fixture rows) and against the oracle on every row; and the adjoint identity
    <w, J t> = <J^T w, t> with the HIP backward kernel."""
    from torchani_amd.aev import AEVComputer
    from torchani_amd.weights import arch_spec

    g, f = load_golden(base), load_fgrads(base)
    consts = arch_spec(g["kind"])[1]._replace(cutoff_fn=g["cutoff_fn"])
    p = oracle_params(g["kind"], g["cutoff_fn"])
    t = fgrad_direction(g["species"])
    _, jt_ref = oracle64.aev_jvp(p, g["species"], g["coords"].astype(np.float64), t, g["cell"], g["pbc"])
    C, A = g["species"].shape
    sp32 = torch.from_numpy(g["species"].astype(np.int32)).to(dev)
    x = torch.from_numpy(g["coords"]).to(dev).contiguous()
    cell = None if g["cell"] is None else torch.from_numpy(g["cell"]).to(dev)
    pbc = None if g["pbc"] is None else tuple(bool(b) for b in g["pbc"])
    td = torch.from_numpy(t.astype(np.float32)).to(dev)
    modes = ["batch", "cell"] if C == 1 else ["batch"]
    for mode in modes:
        aevc = AEVComputer(consts, neighborlist=mode, row_capacity=256).to(dev)
        eng = aevc.engine()
        rows = aevc.neighbor_rows(sp32, x, cell, pbc)
        jt = eng.jvp(sp32, rows, td)
        torch.cuda.synchronize()
        got = jt.cpu().numpy().astype(np.float64)
        scale = np.abs(jt_ref).max()
        err = np.abs(got - jt_ref.reshape(C * A, -1)).max()
        err_fix = np.abs(got[g["aev_rows"]] - f["aev_jvp"]).max()
        report(f"jvp   {base:22s} {mode:5s} max|J t err| = {err:.2e} (max |J t| {scale:.2f}; fixture rows {err_fix:.2e})")
        assert err < 2e-5 * max(1.0, scale) and err_fix < 2e-5 * max(1.0, scale)
        w = torch.from_numpy(np.random.RandomState(5).uniform(-1, 1, (C * A, eng.L)).astype(np.float32)).to(dev)
        gc = eng.backward(sp32, rows, w)
        lhs = (w.double() * jt.double()).sum().item()
        rhs = (gc.double() * td.view(-1, 3).double()).sum().item()
        assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs))
        assert torch.all(jt.view(C, A, -1)[sp32 < 0] == 0)


@pytest.mark.parametrize("base", FGRAD_NAMES)
def test_tangent_weight_grads_match_oracle(dev, oracle64, base):
    """anihip_mlp_tangent_weight_grads: d/d params of S = sum_i v_i . d e_i / d aev_i against the oracle (pinned to the
    reference's create_graph autograd by tests/golden/fgrads_*.npz), with v = -J t from the HIP JVP kernel -- i.e. the
    whole second-order chain of a force loss on the GPU."""
    from torchani_amd.aev import AEVComputer

    g = load_golden(base)
    dims, flat, _ = oracle_networks(g["kind"], g["n_members"], g["seed"])
    p = oracle_params(g["kind"], g["cutoff_fn"])
    t = fgrad_direction(g["species"])
    aev_ref, jt_ref = oracle64.aev_jvp(p, g["species"], g["coords"].astype(np.float64), t, g["cell"], g["pbc"])
    val_ref, ref = oracle64.mlp_tangent_weight_grads(g["species"], aev_ref, -jt_ref, dims, flat, n_members=8)
    model = fresh_model(g["kind"], g["seed"], dev, g["cutoff_fn"])
    model.aev_computer.row_capacity = 256
    C, A = g["species"].shape
    sp = torch.from_numpy(g["species"].astype(np.int64)).to(dev)
    sp32 = sp.to(torch.int32)
    x = torch.from_numpy(g["coords"]).to(dev).contiguous()
    cell = None if g["cell"] is None else torch.from_numpy(g["cell"]).to(dev)
    pbc = None if g["pbc"] is None else tuple(bool(b) for b in g["pbc"])
    aevc = model.aev_computer
    rows = aevc.neighbor_rows(sp32, x, cell, pbc)
    eng = aevc.engine()
    aev = eng.forward(sp32, rows)
    jt = eng.jvp(sp32, rows, torch.from_numpy(t.astype(np.float32)).to(dev))
    packed = model.neural_networks._train_pack(dev)
    gw, gb, de = packed.tangent_weight_grads(sp32, aev, -jt)
    torch.cuda.synchronize()
    got = flat_from_lists(gw, gb, packed.M, packed.S, packed.nl)
    scale = np.abs(ref).max()
    err = np.abs(got - ref).max()
    s_err = abs(de.double().sum().item() - val_ref)
    report(f"fgrad {base:22s} max|d(t.F)/dw err| = {err:.2e} (max {scale:.2e})  |t.F err| = {s_err:.2e} (t.F = {val_ref:+.5f})")
    assert err < 5e-5 * scale
    assert s_err < 1e-5 * max(1.0, abs(val_ref))


@pytest.mark.parametrize("base", FGRAD_NAMES)
def test_force_training_autograd_matches_reference(dev, base):
    """The reference's force-training recipe (tools/training-aev-benchmark.py:136-150) on the HIP engine:
    forces = -autograd.grad(E, coords, create_graph=True); loss(forces).backward() -> .grad of every parameter, against
    the digest of the reference's own second-order autograd (tests/golden/fgrads_*.npz)."""
    from _util import wgrad_digest

    g, f = load_golden(base), load_fgrads(base)
    model = fresh_model(g["kind"], g["seed"], dev, g["cutoff_fn"])
    model.aev_computer.row_capacity = 256
    nets = model.neural_networks
    nets.requires_grad_(True)
    sp = torch.from_numpy(g["species"].astype(np.int64)).to(dev)
    x = torch.from_numpy(g["coords"]).to(dev).requires_grad_(True)
    cell = None if g["cell"] is None else torch.from_numpy(g["cell"]).to(dev)
    pbc = None if g["pbc"] is None else torch.tensor([bool(b) for b in g["pbc"]])
    t = torch.from_numpy(fgrad_direction(g["species"]).astype(np.float32)).to(dev)
    aev = model.aev_computer(sp, x, cell, pbc)
    e = nets(sp, aev).sum()
    (gx,) = torch.autograd.grad(e, x, create_graph=True)
    loss = -(gx * t).sum()                      # = sum_k t_k . F_k
    loss.backward()
    torch.cuda.synchronize()
    got = flat_from_params(nets, model.symbols)
    sums, dots, heads = wgrad_digest(got)
    scale = float(f["grad_abs_max"])
    err = max(np.abs(sums - f["block_sums"]).max(), np.abs(dots - f["block_dots"]).max(),
              np.abs(heads - f["block_heads"]).max())
    l_err = abs(float(loss.detach()) - float(f["loss"]))
    report(f"ftrain {base:22s} |loss err| = {l_err:.2e}  max digest err = {err:.2e} (max |grad| {scale:.2e})")
    assert l_err < 1e-5
    assert err < 2e-4 * scale     # block sums over 4096 fp32-accumulated entries
    assert abs(np.abs(got).max() - scale) < 1e-4 * scale


def test_force_training_through_grad_energies_and_forces(dev):
    """grad.energies_and_forces(..., create_graph=True) -- the reference's signature, grad.py:263-290 -- returns forces that
    can be trained on: the parameter gradients of a force loss equal those of the hand-written recipe above."""
    from torchani_amd.grad import energies_and_forces
    from torchani_amd.tuples import EnergiesForces

    base = FGRAD_NAMES[0]
    g = load_golden(base)
    sp = torch.from_numpy(g["species"].astype(np.int64)).to(dev)
    cell = None if g["cell"] is None else torch.from_numpy(g["cell"]).to(dev)
    pbc = None if g["pbc"] is None else torch.tensor([bool(b) for b in g["pbc"]])
    t = torch.from_numpy(fgrad_direction(g["species"]).astype(np.float32)).to(dev)
    got = []
    for route in ("helper", "manual"):
        model = fresh_model(g["kind"], g["seed"], dev, g["cutoff_fn"])
        model.aev_computer.row_capacity = 256
        nets = model.neural_networks
        nets.requires_grad_(True)
        x = torch.from_numpy(g["coords"]).to(dev)
        if route == "helper":
            out = energies_and_forces(model, sp, x, cell, pbc, create_graph=True)
            assert isinstance(out, EnergiesForces) and out.forces.requires_grad and not x.requires_grad
            energies, forces = out
            loss = (forces * t).sum()
        else:
            x.requires_grad_(True)
            e = model((sp, x), cell, pbc).energies.sum()
            (gx,) = torch.autograd.grad(e, x, create_graph=True)
            loss = -(gx * t).sum()
        loss.backward()
        got.append(flat_from_params(nets, model.symbols))
    scale = np.abs(got[1]).max()
    assert scale > 0 and np.abs(got[0] - got[1]).max() < 1e-6 * scale


@pytest.mark.parametrize("base", ["rand_batch_ani2x", "water_pbc_ani2x"])
def test_training_the_gelu_networks_of_the_2xr_family(dev, base):
    """Round 4: the training passes also serve the GELU / bias-free networks of ANI-2xr / 2dr (the reference trains them
    through plain autograd, arch.py:992-1066, nn/_core.py:146-167): they keep the pre-activations (GELU' cannot be
    recovered from x Phi(x)) and the bias slots of the engine stay empty.  Energy loss and force loss (create_graph=True)
    against the digests of the reference's own first- and second-order autograd (tests/golden/x2rtrain_*.npz)."""
    from _util import wgrad_digest, wgrad_upstream
    from torchani_amd.models import ANI2xr

    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"x2rtrain_{base}.npz")) as z:
        f = {k: z[k] for k in z.files}
    g = load_golden(base)
    model = ANI2xr(seed=int(f["seed"]), device=dev, periodic_table_index=False, row_capacity=256)
    nets = model.neural_networks
    nets.requires_grad_(True)
    symbols = [str(s) for s in f["symbols"]]
    sp = torch.from_numpy(f["species"]).to(dev)
    cell = None if g["cell"] is None else torch.from_numpy(g["cell"]).to(dev)
    pbc = None if g["pbc"] is None else torch.tensor([bool(b) for b in g["pbc"]])
    C, A = sp.shape

    def flat():
        out = []
        for member in nets.members:
            for sym in symbols:
                for lin in member.atomics[sym].linears():
                    assert lin.bias is None
                    gr = lin.weight.grad if lin.weight.grad is not None else torch.zeros_like(lin.weight)
                    out.append(gr.detach().cpu().numpy().astype(np.float64).reshape(-1))
        return np.concatenate(out)

    def check(tag, got, loss, loss_ref):
        sums, dots, heads = wgrad_digest(got)
        scale = float(f[f"{tag}_abs_max"])
        err = max(np.abs(sums - f[f"{tag}_sums"]).max(), np.abs(dots - f[f"{tag}_dots"]).max(),
                  np.abs(heads - f[f"{tag}_heads"]).max())
        report(f"x2rtrain {base:20s} {tag}: |loss err| = {abs(loss - loss_ref):.2e}  max digest err = {err:.2e} (max |grad| {scale:.2e})")
        assert abs(loss - loss_ref) < 1e-5
        assert err < 2e-4 * scale and abs(np.abs(got).max() - scale) < 1e-4 * scale

    # energy loss
    x = torch.from_numpy(g["coords"]).to(dev)
    aev = model.aev_computer(sp, x, cell, pbc)
    atomic = nets(sp, aev, atomic=True)
    loss = (atomic * torch.from_numpy(wgrad_upstream(C, A).astype(np.float32)).to(dev)).sum()
    loss.backward()
    torch.cuda.synchronize()
    assert f["n_params"] == sum(p.numel() for p in nets.parameters())
    check("e", flat(), float(loss.detach()), float(f["loss_e"]))
    nets.zero_grad(set_to_none=True)
    # force loss
    xx = torch.from_numpy(g["coords"]).to(dev).requires_grad_(True)
    t = torch.from_numpy(fgrad_direction(f["species"]).astype(np.float32)).to(dev)
    e = nets(sp, model.aev_computer(sp, xx, cell, pbc)).sum()
    (gx,) = torch.autograd.grad(e, xx, create_graph=True)
    lossf = -(gx * t).sum()
    lossf.backward()
    torch.cuda.synchronize()
    check("f", flat(), float(lossf.detach()), float(f["loss_f"]))

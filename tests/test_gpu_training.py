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

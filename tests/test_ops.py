"""The dispatcher registration of the C ABI (torchani_amd/ops.py), modelled on the reference's checks of its native
operators: scriptability (tests/test_cuaev.py:104-142) and whole-graph compilation (tests/test_pt2.py:44-60)."""
import numpy as np
import pytest
import torch

from _util import load_golden, seeded_state


def _engine():
    from torchani_amd.constants import aev_constants_2x
    from torchani_amd.engine import AevEngine
    from torchani_amd.ops import register_engine

    eng = AevEngine(aev_constants_2x())
    return eng, register_engine(eng)


def test_ops_are_registered_with_schemas():
    import torchani_amd.ops  # noqa: F401

    for name in ("nbr_rows", "aev_from_rows", "aev_backward", "mlp"):
        op = getattr(torch.ops.anihip, name)
        assert "anihip::" + name in str(op.default._schema)


def test_scripted_function_keeps_the_operator_node():
    """torch.jit.script(f).graph contains anihip::aev_from_rows (cf. 'cuaev::run' in tests/test_cuaev.py:127)."""
    import torchani_amd.ops  # noqa: F401

    def f(species, coords, meta, ent, status, engine: int):
        return torch.ops.anihip.aev_from_rows(species, coords, meta, ent, status, engine)

    g = str(torch.jit.script(f).graph)
    assert "anihip::aev_from_rows" in g


def test_fake_kernels_give_shapes_without_a_gpu():
    """Meta/fake propagation (what torch.compile traces with): no device code runs."""
    eng, h = _engine()
    C, A, cap = 3, 5, 128
    sp = torch.empty((C, A), dtype=torch.int32, device="meta")
    x = torch.empty((C, A, 3), dtype=torch.float32, device="meta")
    meta, ent, status = torch.ops.anihip.nbr_rows(sp, x, None, 0, 1, cap, h)
    assert meta.shape == (C * A, 6) and ent.shape == (C * A * cap, 4) and status.shape == (8,)
    a = torch.ops.anihip.aev_from_rows(sp, x, meta, ent, status, h)
    assert a.shape == (C, A, eng.L) and a.dtype == torch.float32
    g = torch.ops.anihip.aev_backward(a, sp, meta, ent, status, h)
    assert g.shape == (C, A, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["rand_batch_ani2x", "small_ani2x"])
def test_compiled_energy_fullgraph_matches_eager_and_reference(name):
    from torchani_amd.models import ANI2x
    from torchani_amd.ops import CompiledEnergy

    dev = torch.device("cuda", 0)
    g = load_golden(name)
    model = ANI2x(state_dict=seeded_state("ani2x", 8, g["seed"]), device=dev, periodic_table_index=False)
    sp = torch.from_numpy(g["species"].astype(np.int64)).to(dev)
    x = torch.from_numpy(g["coords"]).to(dev).requires_grad_(True)
    mod = CompiledEnergy(model).to(dev).bind(dev)
    e = mod(sp, x)
    (gx,) = torch.autograd.grad(e.sum(), x)
    assert np.abs(e.detach().cpu().numpy() - g["energies"]).max() < 1e-5 * max(1.0, np.sqrt((g["species"] >= 0).sum()))
    assert np.abs(-gx.cpu().numpy() - g["forces"]).max() < 1e-4
    cmod = torch.compile(mod, fullgraph=True)   # raises if the graph breaks
    e2 = cmod(sp, x)
    (gx2,) = torch.autograd.grad(e2.sum(), x)
    assert torch.allclose(e2, e, atol=1e-9, rtol=0) and torch.allclose(gx2, gx, atol=2e-6, rtol=0)
    # operator-level consistency checks of torch.library (schema, fake kernel, autograd registration)
    sp32 = sp.to(torch.int32)
    rows = torch.ops.anihip.nbr_rows(sp32, x.detach(), None, 0, 1, 128, mod.engine)
    torch.library.opcheck(torch.ops.anihip.aev_from_rows, (sp32, x.detach(), *rows, mod.engine),
                          test_utils=("test_schema", "test_faketensor"))

"""CPU-only checks: the C-ABI library builds/loads and exports everything include/anihip.h declares, the
host-side packing/partition logic, and the API error behaviour without a GPU (no compute calls here)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from torchani_amd import _lib

    _lib.build()
    return _lib.lib()


def test_library_exports_every_declared_symbol(lib):
    from torchani_amd import _lib

    hdr = open(os.path.join(ROOT, "include", "anihip.h")).read()
    declared = set(re.findall(r"\b(anihip_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no prototypes found in include/anihip.h"
    assert declared == set(_lib.EXPORTED_SYMBOLS)
    for sym in declared:
        assert getattr(lib, sym) is not None
    assert lib.anihip_abi_version() == _lib.ABI_VERSION == 12


def test_struct_layouts_match_header(lib):
    """ctypes mirrors of the POD structs must have the C sizes (checked through the workspace query, which
    dereferences an anihip_mlp_desc)."""
    from torchani_amd import _lib

    assert ctypes.sizeof(_lib.AevParams) == 11 * 4
    assert ctypes.sizeof(_lib.SpeciesNet) == 4 + 5 * 4 + 5 * 4 * 8 + 4 * 4 + 2 * 4 * 8 + 8
    assert ctypes.sizeof(_lib.MlpDesc) == 8 * 4 + 8 * ctypes.sizeof(_lib.SpeciesNet)   # 7 ints + padding
    assert _lib.MlpDesc.flags.offset == 24 and _lib.MlpDesc.net.offset == 32
    d = _lib.MlpDesc()
    d.num_species, d.n_members, d.aev_len, d.celu_alpha = 2, 8, 1008, 0.1
    for s in range(2):
        d.net[s].n_layers = 4
        for l, v in enumerate((1008, 256, 192, 160, 1)):
            d.net[s].dims[l] = v
    n = 1000
    need = lib.anihip_mlp_workspace_bytes(ctypes.byref(d), n)
    acts = 4 * 8 * (256 + 192 + 160) * (n + 1)
    tiles = (n + 63) // 64 + 8   # tile table of the fused kernel: 16-B entry + 64 atom rows per tile, and its cost-sorted copy
    d0_pad = 4 * 8 * 256 * 64 * 8   # layer 0 doubles as the tile-major d E/d act0 buffer: + 64 rows per species slot
    assert acts <= need <= acts + d0_pad + 4 * (n + 1) * (1 + 8) + 2 * (16 + 256) * tiles + 66 * 256
    # one forward_backward call of a descriptor the fused kernel cannot serve (no fp16 planes here): the same buffers
    assert lib.anihip_mlp_forward_backward_workspace_bytes(ctypes.byref(d), n, 1) == need
    # training pass: the activations are kept and every hidden layer gets a gradient buffer of the same size
    assert ctypes.sizeof(_lib.SpeciesGrads) == 2 * 4 * 8 + 8 + 8   # + member_stride, accumulate (+ padding)
    need_t = lib.anihip_mlp_train_workspace_bytes(ctypes.byref(d), n)
    assert need + acts <= need_t <= need + acts + 4 * 256


def test_error_reporting_without_gpu(lib):
    from torchani_amd import _lib

    # (any grid up to 32 radial shifts and 16 x 16 angular terms is accepted since the general kernels exist)
    p = _lib.AevParams(num_species=7, n_shf_r=16, n_shf_a=17, n_shf_z=4, Rcr=5.1, Rca=3.5, EtaR=19.7, EtaA=12.5,
                       Zeta=14.1)
    z = np.zeros(_lib.TABLE_FLOATS, dtype=np.float32)
    rc = lib.anihip_aev_table_pack(ctypes.byref(p), z.ctypes.data, z.ctypes.data, z.ctypes.data, z.ctypes.data)
    assert rc != 0 and b"n_shf_a <= 16" in lib.anihip_last_error()
    ok = _lib.AevParams(num_species=7, n_shf_r=16, n_shf_a=5, n_shf_z=4, Rcr=5.1, Rca=3.5, EtaR=19.7, EtaA=12.5, Zeta=14.1)
    assert lib.anihip_aev_table_pack(ctypes.byref(ok), z.ctypes.data, z.ctypes.data, z.ctypes.data, z.ctypes.data) == 0
    with pytest.raises(RuntimeError, match="libanihip"):
        _lib.check(rc)


def test_argument_validation_without_gpu(lib):
    """Entry points validate their arguments before touching the device: bad calls return non-zero and leave a
    message (the reference raises from TORCH_CHECK, csrc/aev.cu:1693-1710)."""
    from torchani_amd import _lib

    p = _lib.AevParams(num_species=7, n_shf_r=16, n_shf_a=8, n_shf_z=4, Rcr=5.1, Rca=3.5, EtaR=19.7, EtaA=12.5,
                       Zeta=14.1)
    rc = lib.anihip_nbr_from_half(None, ctypes.byref(p), 10, None, 0, None, None, 0, 10, None, 0, None, None, 0, None)
    assert rc != 0 and b"null pointer" in lib.anihip_last_error()
    buf = (ctypes.c_char * 4096)()
    addr = ctypes.addressof(buf)
    rc = lib.anihip_nbr_from_half(None, ctypes.byref(p), 10, addr, 5, None, None, 0, 10, addr, 4096, addr, addr, 1280,
                                  addr)
    assert rc != 0 and b"null neighbor list" in lib.anihip_last_error()
    rc = lib.anihip_nbr_from_half(None, ctypes.byref(p), 10, addr, 0, None, None, 4, 12, addr, 4096, addr, addr, 1280,
                                  addr)
    assert rc != 0 and b"central range" in lib.anihip_last_error()
    assert lib.anihip_nbr_half_workspace_bytes(1000) >= 4000
    rc = lib.anihip_aev_forward(None, ctypes.byref(p), None, 10, 0, 10, None, None, None, None, None, None)
    assert rc != 0 and b"null pointer" in lib.anihip_last_error()
    p.cutoff_kind = 7   # neither ANIHIP_CUTOFF_COSINE nor ANIHIP_CUTOFF_SMOOTH
    rc = lib.anihip_aev_forward(None, ctypes.byref(p), addr, 10, 0, 10, addr, addr, addr, addr, None, addr)
    assert rc != 0 and b"cutoff_kind" in lib.anihip_last_error()
    rc = lib.anihip_mlp_weight_grads(None, None, 10, 0, 10, addr, addr, addr, addr, 4096, None, addr, None, 0)
    assert rc != 0 and b"null descriptor" in lib.anihip_last_error()
    rc = lib.anihip_mlp_train_forward(None, None, 10, 0, 10, addr, addr, addr, 4096, addr)
    assert rc != 0 and b"null descriptor" in lib.anihip_last_error()
    rc = lib.anihip_mlp_repack(None, None, addr, addr, None, 0)
    assert rc != 0 and b"null descriptor" in lib.anihip_last_error()
    one = ctypes.c_double(0.5)
    rc = lib.anihip_adam_step(None, addr, addr, addr, None, 16, one, one, one, one, one, addr, 1)
    assert rc != 0 and b"null pointer" in lib.anihip_last_error()
    rc = lib.anihip_adam_step(None, addr, addr, addr, addr, 16, one, ctypes.c_double(1.5), one, one, one, addr, 1)
    assert rc != 0 and b"hyper-parameters" in lib.anihip_last_error()
    d = _lib.MlpDesc()
    d.num_species, d.n_members, d.aev_len, d.celu_alpha, d.precision = 9, 8, 1008, 0.1, _lib.MLP_F16X3
    rc = lib.anihip_mlp_forward_backward(None, ctypes.byref(d), 10, 0, 10, addr, addr, None, addr, 4096, addr, None,
                                         None)
    assert rc != 0 and b"num_species" in lib.anihip_last_error()


def test_aev_table_pack_matches_constants(lib):
    from torchani_amd.constants import aev_constants_2x
    from torchani_amd.engine import AevEngine

    c = aev_constants_2x()
    t = AevEngine(c).host_table()
    assert np.array_equal(t[:16], np.asarray(c.ShfR, dtype=np.float32))
    assert np.array_equal(t[32:40], np.asarray(c.ShfA, dtype=np.float32))
    z = np.asarray(c.ShfZ, dtype=np.float32).astype(np.float64)
    assert np.allclose(t[48:52], np.cos(z), atol=1e-7) and np.allclose(t[64:68], np.sin(z), atol=1e-7)
    assert c.out_dim == 1008 and c.radial_len == 112 and c.angular_len == 896


def test_aev_table_pack_recurrence_constants(lib):
    """ANIHIP_AEV_REC_BWD: equally spaced ShfR and ShfA give the backward kernel's recurrence constants (double precision on the
    host, free slots of the table: csrc/anihip_common.h TAB_RECR / TAB_RECA / TAB_RECAK); any unequal spacing clears the flag."""
    from torchani_amd.constants import aev_constants_1x, aev_constants_2x
    from torchani_amd.engine import AevEngine

    for c in (aev_constants_2x(), aev_constants_1x()):
        e = AevEngine(c)
        t = e.host_table().astype(np.float64)
        assert e.params.flags & 2 and e.params.flags & 1
        log2e = 1.4426950408889634
        shfr, shfa = (np.asarray(v, dtype=np.float32).astype(np.float64) for v in (c.ShfR, c.ShfA))
        qr, qa = np.sqrt(np.float64(np.float32(c.EtaR)) * np.float64(np.float32(log2e))), np.sqrt(
            np.float64(np.float32(c.EtaA)) * np.float64(np.float32(log2e)))
        DR, DA = qr * (shfr[-1] - shfr[0]) / 15, qa * (shfa[-1] - shfa[0]) / (len(shfa) - 1)
        rel = lambda a, b: abs(a - b) / abs(b)   # noqa: E731
        assert rel(t[16], 2 * DR) < 2e-7 and rel(t[104], 2 * DA) < 2e-7
        assert rel(t[17], 2.0 ** -(DR * DR)) < 2e-7 and rel(t[18], DR * 2.0 ** -(DR * DR)) < 2e-7
        assert rel(t[19], 2.0 ** -(4 * DR * DR)) < 2e-7 and rel(t[20], 2 * DR * 2.0 ** -(4 * DR * DR)) < 2e-7
        for k, ex in ((21, 8.0), (22, 16.0), (23, -8.0), (24, -16.0)):
            assert rel(t[k], 2.0 ** (ex * DR * DR)) < 2e-7
        assert rel(t[25], 120.0 - 16.0 * DR * DR) < 2e-7 and 2 * DR * max(abs(qr * c.Rcr - qr * shfr[9]), qr * shfr[9]) < t[25]
        for m in range(1, 5):
            K = 2.0 ** -((m * DA) ** 2)
            assert rel(t[136 + 2 * (m - 1)], K) < 2e-7 and rel(t[137 + 2 * (m - 1)], m * DA * K) < 2e-7
        # the published entries are where they were
        assert np.array_equal(e.host_table()[:16], np.asarray(c.ShfR, dtype=np.float32))
    c = aev_constants_2x()
    bent = c._replace(ShfR=tuple(v + (0.01 if k == 5 else 0.0) for k, v in enumerate(c.ShfR)))
    e = AevEngine(bent)
    e.host_table()
    assert not e.params.flags & 2 and e.params.flags & 1   # (the forward's flag only looks at ShfA)


def test_per_species_launch_rule_follows_the_measurements():
    """models.ANI._per_species_launches_pay prices "one fused launch per species, compile-time widths" against "one launch, tiles
    handed out by falling cost" in tiles per workgroup; the choices measured on the MI355X (water boxes 24 k ... 2.3 M atoms, the
    46 k-atom solvated protein: DESIGN.md section 6) are the ones it makes."""
    from torchani_amd.models import ANI2x

    m = ANI2x(seed=0)
    water = lambda n: [2 * n // 3, 0, 0, n // 3, 0, 0, 0]   # noqa: E731
    measured = {24000: True, 41472: True, 81000: False, 192000: True, 526848: True, 1119744: True, 2336064: True}
    for n, want in measured.items():
        assert m._per_species_launches_pay(water(n), 256) is want, n
    assert m._per_species_launches_pay([30734, 314, 77, 15231, 1, 0, 0], 256) is False   # solvated 1hz5: H C N O S


def test_constants_match_reference_values():
    """SURVEY section 0 table (values cross-checked there against the reference's .params files)."""
    from torchani_amd.constants import aev_constants_1x, aev_constants_2x

    c2, c1 = aev_constants_2x(), aev_constants_1x()
    assert (c2.Rcr, c2.Rca, c2.EtaR, c2.EtaA, c2.Zeta) == (5.1, 3.5, 19.7, 12.5, 14.1)
    assert abs(c2.ShfR[1] - 1.06875) < 1e-12 and abs(c2.ShfA[7] - 3.1625) < 1e-12
    assert abs(c2.ShfZ[0] - np.pi / 8) < 1e-15 and len(c2.ShfZ) == 4
    assert c1.out_dim == 384 and (c1.Rcr, c1.EtaR, c1.EtaA, c1.Zeta) == (5.2, 16.0, 8.0, 32.0)


def test_model_state_dict_keys_and_loading():
    """Reference key names and parameter count of the ANI-2x architecture."""
    from torchani_amd.models import ANI2x
    from torchani_amd.weights import random_state_dict

    sd = random_state_dict("ani2x", 8, 3)
    m = ANI2x(state_dict=sd)
    keys = set(m.state_dict().keys())
    for k in ("potentials.nnp.neural_networks.members.7.atomics.Cl.final_layer.weight",
              "potentials.nnp.aev_computer.angular.sections", "potentials.nnp.aev_computer.triu_index",
              "energy_shifter.self_energies", "species_converter.conv_tensor"):
        assert k in keys
    assert sum(p.numel() for p in m.parameters()) == 13705784
    w = m.state_dict()["potentials.nnp.neural_networks.members.2.atomics.N.layers.1.weight"]
    assert torch.equal(w, torch.from_numpy(sd["potentials.nnp.neural_networks.members.2.atomics.N.layers.1.weight"]))
    assert m.aev_computer.out_dim == 1008 and m.aev_computer.triu_index[2, 1] == m.aev_computer.triu_index[1, 2]
    assert not any(p.requires_grad for p in m.parameters())  # models.py:196


def test_cpu_tensors_are_rejected_loudly():
    """No CPU fallback: the product path refuses non-device tensors (aev/_computer.py:444-447 analogue)."""
    from torchani_amd.models import ANI2x

    m = ANI2x(seed=1)
    sp = torch.tensor([[1, 6, 1, 1, 1]])
    x = torch.zeros(1, 5, 3)
    with pytest.raises(ValueError, match="ROCm device"):
        m.energies_and_forces(sp, x)
    with pytest.raises(ValueError, match="ROCm device"):
        m((sp, x))
    with pytest.raises(ValueError):
        m.species_converter(torch.tensor([[1, 5]]))  # boron is not an ANI-2x element


def test_species_converter_and_self_energy():
    from torchani_amd.nn import SelfEnergy, SpeciesConverter

    conv = SpeciesConverter(("H", "C", "N", "O", "S", "F", "Cl"))
    z = torch.tensor([[1, 6, 7, 8, 16, 9, 17, -1]])
    assert conv(z).tolist() == [[0, 1, 2, 3, 4, 5, 6, -1]]
    sae = SelfEnergy(("H", "C"), (-0.5, -37.8))
    e = sae(torch.tensor([[0, 1, -1], [0, 0, 0]]))
    assert torch.allclose(e, torch.tensor([-38.3, -1.5]))  # padding contributes nothing (sae.py:61)


def _check_pack_against_reference(W, B, K0, precision="f16x3", activation="celu", radial_len=None):
    """anihip_mlp_pack (through PackedNetworks, host buffer) == the torch restatement of the layouts, bit for bit."""
    from _pack_reference import pack_reference
    from torchani_amd.engine import PackedNetworks

    pk = PackedNetworks(W, B, K0, 0.1, torch.device("cpu"), precision=precision, activation=activation,
                        radial_len=radial_len)
    ref, scales, rlen = pack_reference(W, B, K0, precision, radial_len, activation)
    d = pk.desc
    assert pk.radial_len == rlen == d.aev_radial_len
    names = {"w": "w", "wt": "wt", "bias": "bias", "wh": "wh", "wth": "wth", "whf": "whf", "wthf": "wthf"}
    seen = 0
    for (s_, name, l), t in ref.items():
        if name == "bounds":
            got = pk.array(d.net[s_].fused_bounds, t.shape)
            assert torch.allclose(got, t, rtol=2e-6, atol=0), (s_, name)   # (sums of |w| in another order)
            continue
        ptr = getattr(d.net[s_], names[name])[l]
        assert ptr, (s_, name, l)
        got = pk.array(ptr, t.shape, t.dtype)
        assert torch.equal(got, t), (s_, name, l)
        seen += 1
    for (s_, l), sc in scales.items():
        assert d.net[s_].wh_scale[l] == sc
    assert seen > 0
    return pk


def test_packed_network_layout_cpu():
    """The C-ABI packer (anihip_mlp_pack into a HOST buffer: no GPU involved) on networks whose widths need padding."""
    rs = np.random.RandomState(0)
    M, S, K0 = 2, 2, 32
    hid = [(40, 24), (33, 16)]  # widths that need padding to 64/32 and 64/32
    W = [[[torch.from_numpy(rs.randn(o, i).astype(np.float32)) for i, o in
           zip((K0,) + hid[s], hid[s] + (1,))] for s in range(S)] for m in range(M)]
    B = [[[torch.from_numpy(rs.randn(w.shape[0]).astype(np.float32)) for w in W[m][s]] for s in range(S)]
         for m in range(M)]
    pk = _check_pack_against_reference(W, B, K0)
    d = pk.desc
    assert [d.net[0].dims[l] for l in range(4)] == [32, 64, 32, 1]
    assert [d.net[1].dims[l] for l in range(4)] == [32, 64, 32, 1]
    # layer 0 of species 1: w [K0, M*H1p], column m*H1p+o = W[m][1][0][o, :]
    w0 = pk.array(d.net[1].w[0], (32, 2 * 64))
    assert torch.equal(w0[:, 64 + 5], W[1][1][0][5]) and torch.all(w0[:, 64 + 33:] == 0)
    # f16x3 planes: hi + lo reproduces scale * weight to ~2^-22
    wt1 = pk.array(d.net[0].wt[1], (2, 32, 64))
    wh1 = pk.array(d.net[0].wh[1], (2, 2, 32, 64), torch.float16)
    sc = d.net[0].wh_scale[1]
    assert 2 ** 13 <= sc * wt1.abs().max() < 2 ** 14
    rec = (wh1[0].float() + wh1[1].float()) / sc
    assert (rec - wt1).abs().max() <= 2.0 ** -21 * wt1.abs().max()
    _check_pack_against_reference(W, B, K0, precision="fp32")


def test_packed_network_slab_order_cpu():
    """Layer-0 fp16 planes in slab order (include/anihip.h): radial part padded to a multiple of 32, then
    the angular part, for the ANI-2x shape (S = 7: 112 -> 128, K0p = 1024); four-layer networks with fused bounds."""
    from torchani_amd.engine import PackedNetworks

    rs = np.random.RandomState(1)
    M, S, K0 = 1, 7, 1008
    W = [[[torch.from_numpy(rs.randn(o, i).astype(np.float32)) for i, o in ((K0, 32), (32, 32), (32, 1))]
          for s in range(S)] for m in range(M)]
    B = [[[torch.zeros(w.shape[0]) for w in W[m][s]] for s in range(S)] for m in range(M)]
    pk = _check_pack_against_reference(W, B, K0)
    d = pk.desc
    assert pk.radial_len == 112 and d.aev_radial_len == 112
    wh0 = pk.array(d.net[3].wh[0], (2, 32, 1024), torch.float16)
    wth0 = pk.array(d.net[3].wth[0], (2, 1024, 32), torch.float16)
    sc = d.net[3].wh_scale[0]
    rec = (wh0[0].float() + wh0[1].float()) / sc
    ref = W[0][3][0]
    tol = 2.0 ** -21 * ref.abs().max()
    assert (rec[:, :112] - ref[:, :112]).abs().max() <= tol
    assert torch.all(rec[:, 112:128] == 0)
    assert (rec[:, 128:] - ref[:, 112:]).abs().max() <= tol
    assert torch.equal(wth0[0], wh0[0].t())
    # fp32 packing and AEV lengths without the ANI structure keep the plain order
    assert PackedNetworks(W, B, K0, 0.1, torch.device("cpu"), precision="fp32").radial_len == 0
    W2 = [[[torch.from_numpy(rs.randn(o, i).astype(np.float32)) for i, o in ((64, 32), (32, 1))]
           for s in range(2)]]
    B2 = [[[torch.zeros(w.shape[0]) for w in W2[0][s]] for s in range(2)]]
    assert PackedNetworks(W2, B2, 64, 0.1, torch.device("cpu")).radial_len == 0
    # the ANI-2x shape itself (4 layers, 2 members, padded widths, GELU bound factors): every layout incl. fused_bounds
    hid = {0: (256, 192, 160), 1: (224, 192, 160), 2: (192, 160, 128), 3: (192, 160, 128), 4: (160, 128, 96),
           5: (160, 128, 96), 6: (160, 128, 96)}
    W4 = [[[torch.from_numpy((rs.randn(o, i) / np.sqrt(i)).astype(np.float32)) for i, o in
            zip((K0,) + hid[s], hid[s] + (1,))] for s in range(S)] for m in range(2)]
    B4 = [[[torch.from_numpy(rs.randn(w.shape[0]).astype(np.float32) * 0.1) for w in W4[m][s]] for s in range(S)]
          for m in range(2)]
    _check_pack_against_reference(W4, B4, K0)
    _check_pack_against_reference(W4, B4, K0, activation="gelu")


def test_mlp_pack_through_raw_ctypes(lib):
    """The network half of the ABI from ctypes alone (what a reference-side binding would do in place of mnp::run's
    Tensor lists, csrc/mnp.cpp:238-248): shape struct -> anihip_mlp_pack_bytes -> anihip_mlp_pack into a host buffer ->
    a descriptor whose pointers lie inside that buffer; bad shapes are refused with a message."""
    from torchani_amd import _lib

    rs = np.random.RandomState(3)
    M, S, nl, K0 = 2, 1, 3, 64
    outs = (48, 32, 1)
    sh = _lib.MlpShape()
    sh.n_members, sh.num_species, sh.n_layers, sh.aev_len, sh.aev_radial_len = M, S, nl, K0, -1
    sh.precision, sh.activation, sh.celu_alpha = _lib.MLP_F16X3, _lib.ACT_CELU, 0.1
    for l, o in enumerate(outs):
        sh.out_dims[0][l] = o
    ws = [rs.randn(o, i).astype(np.float32) for m in range(M) for i, o in zip((K0,) + outs[:-1], outs)]
    bs = [rs.randn(w.shape[0]).astype(np.float32) for w in ws]
    wp = (ctypes.c_void_p * len(ws))(*[w.ctypes.data for w in ws])
    bp = (ctypes.c_void_p * len(bs))(*[b.ctypes.data for b in bs])
    need = lib.anihip_mlp_pack_bytes(ctypes.byref(sh))
    assert need > 4 * sum(w.size for w in ws)
    buf = np.zeros(need, dtype=np.uint8)
    desc = _lib.MlpDesc()
    assert lib.anihip_mlp_pack(None, ctypes.byref(sh), wp, bp, 0, buf.ctypes.data, need, 0, ctypes.byref(desc)) == 0
    assert desc.n_members == M and desc.net[0].n_layers == nl and [desc.net[0].dims[l] for l in range(4)] == [64, 64, 32, 1]
    lo, hi = buf.ctypes.data, buf.ctypes.data + need
    for arr in (desc.net[0].w, desc.net[0].bias, desc.net[0].wh, desc.net[0].whf):
        assert lo <= arr[0] < hi
    w1 = np.frombuffer(buf, dtype=np.float32, count=M * 64 * 32, offset=desc.net[0].wt[1] - lo).reshape(M, 32, 64)
    assert np.array_equal(w1[1, :32, :48], ws[nl + 1])           # wt of layer 1, member 1 = its nn.Linear weight
    assert lib.anihip_mlp_pack(None, ctypes.byref(sh), wp, bp, 0, buf.ctypes.data, need - 1, 0, ctypes.byref(desc)) != 0
    assert b"anihip_mlp_pack_bytes" in lib.anihip_last_error()
    sh.out_dims[0][nl - 1] = 2
    assert lib.anihip_mlp_pack_bytes(ctypes.byref(sh)) == 0 and b"one output" in lib.anihip_last_error()


def test_shard_bounds():
    from torchani_amd.parallel import shard_bounds, shard_range

    for n in (0, 1, 7, 2336064):
        for w in (1, 2, 3, 8):
            b = shard_bounds(n, w)
            assert b[0] == 0 and b[-1] == n and all(0 <= b[i + 1] - b[i] <= n // w + 1 for i in range(w))
    assert shard_range(100, None) == (0, 100)
    assert shard_range(10, rank=1, world=4) == (3, 6)


def test_cutoff_fn_selection():
    """cutoff_fn travels AEVComputer -> AEVConstants -> anihip_aev_params.cutoff_kind; unknown names raise like the
    reference's _parse_cutoff_fn (cutoffs.py:104-121)."""
    from torchani_amd import _lib
    from torchani_amd.aev import AEVComputer
    from torchani_amd.engine import AevEngine

    assert AEVComputer.like_2x().constants().cutoff_fn == "cosine"
    aevc = AEVComputer.like_2x(cutoff_fn="smooth")
    assert aevc.constants().cutoff_fn == "smooth"
    assert AevEngine(aevc.constants()).params.cutoff_kind == _lib.CUTOFF_KINDS["smooth"] == 1
    assert AevEngine(AEVComputer.like_1x().constants()).params.cutoff_kind == 0
    with pytest.raises(ValueError, match="cutoff"):
        AEVComputer.like_2x(cutoff_fn="biweight")


def test_verlet_neighborlist_selection():
    """neighborlist names of the reference (neighbors.py:899-914); the Verlet skin must be positive (:767-768)."""
    from torchani_amd.aev import AEVComputer
    from torchani_amd.engine import VerletRows

    assert AEVComputer.like_2x(neighborlist="cell_list").verlet is None
    v = AEVComputer.like_2x(neighborlist="verlet_cell_list", skin=0.8)
    assert isinstance(v.verlet, VerletRows) and v.verlet.skin == 0.8 and v.neighbor_mode == "cell"
    with pytest.raises(ValueError, match="skin"):
        VerletRows(0.0)
    with pytest.raises(ValueError, match="neighborlist"):
        AEVComputer.like_2x(neighborlist="octree")


def test_utils_helpers():
    """torchani_amd.utils mirrors the helpers of torchani/utils.py that callers of the hot path use."""
    from torchani_amd import utils as u

    assert u.cumsum_from_zero(torch.tensor([3, 1, 4, 1])).tolist() == [0, 3, 4, 8]
    x = torch.arange(12).view(3, 4)
    mask = torch.tensor([True, False, True, True])
    assert torch.equal(u.fast_masked_select(x, mask, 1), x[:, mask])
    batches = [{"species": torch.tensor([[1, 6]]), "coordinates": torch.zeros(1, 2, 3), "energies": torch.tensor([1.0])},
               {"species": torch.tensor([[8, 1, 1]], dtype=torch.int32), "coordinates": torch.ones(1, 3, 3),
                "energies": torch.tensor([2.0])}]
    p = u.pad_atomic_properties(batches)
    assert p["species"].tolist() == [[1, 6, -1], [8, 1, 1]] and p["species"].dtype == torch.long
    assert p["coordinates"].shape == (2, 3, 3) and p["energies"].tolist() == [1.0, 2.0]
    p["species"][:, 2] = -1
    assert u.strip_redundant_padding(p)["coordinates"].shape == (2, 2, 3)
    cell = torch.tensor([[10.0, 0, 0], [2, 9, 0], [0, 0, 8]])
    xyz = torch.tensor([[-1.0, 20.0, 3.0], [3.0, 4.0, -30.0]])
    w = u.map_to_central(xyz, cell, torch.tensor([True, True, False]))
    shift = (w - xyz) @ torch.inverse(cell)
    assert torch.allclose(shift, shift.round(), atol=1e-5) and torch.all(shift[:, 2] == 0)   # whole lattice vectors
    frac = w @ torch.inverse(cell)
    assert torch.all(frac[:, :2] >= -1e-6) and torch.all(frac[:, :2] < 1 + 1e-6)
    assert u.linspace(0.8, 5.1, 16)[1] == 0.8 + (5.1 - 0.8) / 16


def test_builtin_factory_signature():
    """ANI2x(model_index, neighborlist, strategy, periodic_table_index, device, dtype): the reference's positional
    order (models.py:165-196); model_index picks one member (models.py:195)."""
    from torchani_amd.models import ANI1x, ANI2x
    from torchani_amd.nn import ANINetworks, Ensemble

    full = ANI2x(seed=3)
    one = ANI2x(2, "cell_list", "cuaev", True, None, None, seed=3)
    assert isinstance(full.neural_networks, Ensemble) and isinstance(one.neural_networks, ANINetworks)
    w_full = full.neural_networks.members[2].atomics["O"].layers[1].weight
    assert torch.equal(one.neural_networks.atomics["O"].layers[1].weight, w_full)
    assert one.aev_computer.neighbor_mode == "cell" and len(ANI1x(seed=1)) == 8
    with pytest.raises(ValueError, match="strategy"):
        ANI2x(strategy="numpy")
    with pytest.raises(ValueError, match="float32"):
        ANI2x(dtype=torch.float64)


@pytest.mark.skipif(not os.path.isdir("/root/reference/torchani"), reason="needs the reference tree (build container only)")
def test_integration_section_a_against_the_live_reference():
    """INTEGRATION.md section A, as far as it can run without a GPU: a LIVE reference model's state dict loads into
    torchani_amd.models.ANI2x strictly (every network / self-energy key found, none unexpected), the two hot components
    can be assigned into the reference model, and the reference's own call path then reaches our AEVComputer, which
    refuses CPU tensors loudly (there is no fallback)."""
    import sys
    import types

    class _Any:
        def __class_getitem__(cls, k):
            return cls

    for name in ("h5py", "zarr"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            for k in ("File", "Group", "Dataset", "Datatype"):
                setattr(m, k, type(k, (_Any,), {}))
            sys.modules[name] = m
    os.environ["TORCHANI_NO_WARN_EXTENSIONS"] = "1"
    sys.path.insert(0, "/root/reference")
    try:
        import torchani
        from torchani.arch import Assembler
        from torchani.utils import SYMBOLS_2X

        asm = Assembler()
        asm.set_symbols(SYMBOLS_2X)
        asm.set_global_cutoff_fn("cosine")
        asm.set_aev_computer(radial="ani2x", angular="ani2x", strategy="pyaev")
        asm.set_atomic_networks(ctor="ani2x")
        asm.set_neighborlist("all_pairs")
        asm.set_gsaes_as_self_energies("wb97x-631gd")
        ref = asm.assemble(8)
        import torchani_amd

        amd = torchani_amd.models.ANI2x(state_dict=ref.state_dict())
        sd_ref, sd_amd = ref.state_dict(), amd.state_dict()
        nn_keys = [k for k in sd_ref if "neural_networks" in k or "energy_shifter" in k]
        assert len(nn_keys) == 8 * 7 * 8 + 1
        for k in nn_keys:
            assert torch.equal(sd_ref[k].to(sd_amd[k].dtype), sd_amd[k]), k
        # swap the two hot components into the reference model (tests/test_neighbors.py:311-312 style)
        ref.potentials["nnp"].aev_computer = amd.aev_computer
        ref.potentials["nnp"].neural_networks = amd.neural_networks
        z = torch.tensor([[6, 1, 1, 1, 1]])
        x = torch.rand(1, 5, 3)
        with pytest.raises(ValueError, match="ROCm device"):
            torchani.grad.energies_and_forces(ref, z, x)
        # the autograd helpers have the reference's signatures (names, order, defaults; this package's extra keywords come
        # BEHIND them) and its result type
        import inspect

        for fn in ("forces", "grads", "forces_for_training", "energies_and_forces", "single_point"):
            pr = list(inspect.signature(getattr(torchani.grad, fn)).parameters.values())
            pa = list(inspect.signature(getattr(torchani_amd.grad, fn)).parameters.values())
            assert [(p.name, p.default) for p in pa[:len(pr)]] == [(p.name, p.default) for p in pr], fn
        assert torchani_amd.tuples.EnergiesForces._fields == torchani.tuples.EnergiesForces._fields
        assert set(torchani_amd.grad.__all__) <= set(torchani.grad.__all__)
    finally:
        sys.path.remove("/root/reference")


def test_d3_reference_data_and_class():
    """The D3 element data shipped with the package (extracted from the reference's resources by
    tests/golden/gen_golden_d3.py) and the host-side constructor checks of TwoBodyDispersionD3 (dftd3.py:139-214)."""
    import math

    from torchani_amd.potentials import TwoBodyDispersionD3, d3_reference_data

    d = d3_reference_data()
    assert d["c6"].shape == (19, 19, 5, 5) and d["cn_a"].shape == d["cn_b"].shape == d["c6"].shape
    assert abs(d["c6"][1, 1, 0, 0] - 3.0267) < 1e-4           # H-H, both hydrogens free atoms (Grimme et al. 2010)
    assert np.allclose(d["c6"], d["c6"].transpose(1, 0, 3, 2))   # C6_ab[ref_a][ref_b] = C6_ba[ref_b][ref_a]
    assert np.allclose(d["cn_a"], d["cn_b"].transpose(1, 0, 3, 2))
    assert d["functionals"]["wb97x"] == (1.0, 0.2641, 0.0, 5.4959)
    pot = TwoBodyDispersionD3.from_functional(("H", "C", "N", "O"), "wB97X", cutoff=8.0)
    assert pot.precalc_coeff6.shape == (4, 4, 5, 5) and pot.cutoff == 8.0 and pot.needs_all_rows
    assert math.isclose(float(pot.covalent_radii[0]), 0.32 * 1.8897261258369282, rel_tol=1e-6)
    t = pot.table(torch.device("cpu"))
    assert t.shape == (8, 8, 25, 4) and float(t[0, 3, 0, 0]) == float(pot.precalc_coeff6[0, 3, 0, 0])
    assert int(t[0, 0, 0, 3]) == 4 and int(t[0, 3, 0, 3]) == 6 and int(t[3, 3, 0, 3]) == 9   # H-H, H-O, O-O references
    assert float(t[5, 5, 0, 3]) == 0.0   # unused element slots: no references
    with pytest.raises(ValueError):
        TwoBodyDispersionD3.from_functional(("H", "Xe"), "wb97x")
    with pytest.raises(ValueError):
        TwoBodyDispersionD3.from_functional(("H", "O"), "no-such-functional")
    with pytest.raises(ValueError):
        TwoBodyDispersionD3(("H", "O"), 1.0, 1.0, 0.4, 4.0, sqrt_empirical_charge=(1.0,))


@pytest.mark.skipif(not os.path.exists("/root/reference/torchani/resources/c6.h5"), reason="reference tree not present")
def test_d3_data_file_is_the_reference_table():
    """torchani_amd/data/d3_refs.npz against the reference's resources (build container only): the HDF5 reader of
    tests/golden/gen_golden_d3.py finds the three datasets of c6.h5 and the shipped slice equals them."""
    import json
    import struct

    from torchani_amd.potentials import d3_reference_data

    b = open("/root/reference/torchani/resources/c6.h5", "rb").read()
    nbytes = 4 * 95 * 95 * 25
    addrs, i = [], b.find(struct.pack("<Q", nbytes))
    while i >= 0:
        if b[i - 10] == 3 and b[i - 9] == 1:
            addrs.append(struct.unpack("<Q", b[i - 8:i])[0])
        i = b.find(struct.pack("<Q", nbytes), i + 1)
    assert len(addrs) == 3
    tabs = [np.frombuffer(b, dtype="<f4", count=nbytes // 4, offset=a).reshape(95, 95, 5, 5) for a in sorted(addrs)]
    d = d3_reference_data()
    for ours, theirs in zip((d["c6"], d["cn_a"], d["cn_b"]), tabs):
        assert np.array_equal(ours, theirs[:19, :19])
    ac = json.load(open("/root/reference/torchani/resources/atomic_constants.json"))
    for z, s_ in enumerate(d["symbols"], start=1):
        assert abs(d["covalent_radius"][z] - ac[s_]["covalent_radius"]) < 1e-12
        assert abs(d["sqrt_empirical_charge"][z] - ac[s_]["sqrt_empirical_charge"]) < 1e-12


def test_charge_networks_pack_and_normalizer():
    """ANI-mbis host side (models.py:201-252): the two-output charge networks pack the SECOND row of their final layers
    (nn/_internal.py:69-93), and ChargeNormalizer reproduces the reference's normalized charges from its raw ones
    (electro.py:29-87; fixture from tests/golden/gen_golden_mbis.py)."""
    from torchani_amd.models import ChargeNormalizer
    from torchani_amd.nn import ANINetworksDiscardFirstScalar

    torch.manual_seed(0)
    nets = ANINetworksDiscardFirstScalar.build(("H", "C"), 32, {"H": (24, 16), "C": (24, 16)}, "gelu", False, out_dim=2)
    pk = nets._pack(torch.device("cpu"))
    d = pk.desc
    assert d.activation == 1 and [d.net[1].dims[l] for l in range(4)] == [32, 32, 32, 1]
    wf = pk.array(d.net[1].w[2], (1, 32))
    assert torch.equal(wf[0, :16], nets.atomics["C"].final_layer.weight[1].detach())
    assert set(nets.state_dict()) == {f"atomics.{s}.{n}.weight" for s in "HC" for n in ("layers.0", "layers.1", "final_layer")}

    ref = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mbis_rand_batch_ani2x.npz")))
    norm = ChargeNormalizer.from_electronegativity_and_hardness([str(s) for s in ref["symbols"]],
                                                                scale_weights_by_charges_squared=True)
    q = norm(torch.from_numpy(ref["species"]), torch.from_numpy(ref["raw_charges"]))
    assert np.abs(q.numpy() - ref["atomic_charges"]).max() < 1e-7     # (fp32 weights against the reference's fp64)
    assert q.sum(dim=1).abs().max() < 1e-12
    q1 = norm(torch.from_numpy(ref["species"]), torch.from_numpy(ref["raw_charges"]), charge=1)
    assert (q1.sum(dim=1) - 1).abs().max() < 1e-12
    plain = ChargeNormalizer(["H", "C"])
    qq = plain(torch.tensor([[0, 1, -1]]), torch.tensor([[0.3, 0.1, 0.0]]))
    assert torch.allclose(qq, torch.tensor([[0.1, -0.1, 0.0]]))


def test_layer0_tile_hint_from_composition():
    """ANI._tile_hint: 128-row layer-0 backward tiles between 16 384 and 24 000 atoms when four or more elements are
    present, the library's default otherwise; cached per species tensor (held, so its address is not recycled)."""
    import warnings

    from torchani_amd import _lib
    from torchani_amd.models import ANI2x

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = ANI2x(seed=0, n_members=1)
    rs = np.random.RandomState(0)
    organic = torch.from_numpy(rs.randint(0, 5, (1, 20000)))
    water = torch.from_numpy(rs.randint(0, 2, (1, 20000)))
    padded = torch.cat([organic[:, :9000], -torch.ones((1, 11000), dtype=torch.long)], dim=1)
    assert m._tile_hint(organic, organic, 20000) == _lib.MLP_FLAG_SMALL_TILES
    assert m._n_elem_cache[2] is organic
    assert m._tile_hint(water, water, 20000) == 0
    assert m._tile_hint(padded, padded, 20000) == _lib.MLP_FLAG_SMALL_TILES
    assert m._tile_hint(organic, organic, 16383) == 0 and m._tile_hint(organic, organic, 65536) == 0
    # round 6: from 24 000 atoms on the layer-0 backward runs inside the fused kernel; one launch per species with compile-time
    # network widths where that is cheaper than one launch with a tile queue (test_per_species_launch_rule_follows_the_measurements)
    big = lambda k, n: torch.from_numpy(rs.randint(0, k, (1, n)))   # noqa: E731
    o24, w24, w2m, o2m = big(5, 24000), big(2, 24000), big(2, 2_336_064), big(5, 2_336_064)
    assert m._tile_hint(o24, o24, 24000) == 0                                 # five launches of a third of a round each
    assert m._tile_hint(w24, w24, 24000) == _lib.MLP_FLAG_SHAPED              # two launches of 0.7 rounds
    assert m._tile_hint(w2m, w2m, 2_336_064) == _lib.MLP_FLAG_SHAPED          # 71 rounds per species
    assert m._tile_hint(o2m, o2m, 2_336_064) == _lib.MLP_FLAG_SHAPED          # 28 rounds per species


def test_overflow_check_is_skipped_only_where_rows_cannot_overflow():
    """ANI._overflow_impossible: the default check_overflow=True reads the builder's status word (a host synchronisation) unless
    no row CAN overflow -- molecules without periodic images whose A - 1 atoms fit every row."""
    import warnings

    from torchani_amd.models import ANI2x

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = ANI2x(seed=0, n_members=1, neighborlist="batch", row_capacity=64)
    small, wide = torch.zeros((256, 28), dtype=torch.long), torch.zeros((4, 200), dtype=torch.long)
    cell = torch.eye(3) * 30.0
    assert m._overflow_impossible(small, None, None)
    assert m._overflow_impossible(small, cell, (False, False, False))
    assert not m._overflow_impossible(small, cell, (True, True, True))      # periodic images add neighbors
    assert not m._overflow_impossible(wide, None, None)                      # 199 possible neighbors > 64 slots
    m.aev_computer.row_capacity = 256
    assert not m._overflow_impossible(wide, None, None)                      # ... and > the 128 angular slots
    assert m._overflow_impossible(torch.zeros((4, 129), dtype=torch.long), None, None)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mc = ANI2x(seed=0, n_members=1, neighborlist="cell")
    assert not mc._overflow_impossible(small[:1], None, None)                # (the grid of cell mode has a status of its own)


def test_cached_parameter_list_follows_the_modules():
    """_EngineContainer._param_list keeps the flat parameter list between calls (walking the module tree costs more than
    a small step) and rebuilds it when a parameter or submodule is registered anywhere, when the active members change,
    and when a conversion swaps the Parameter objects behind the hooks' back."""
    import warnings

    from torchani_amd.models import ANI2x

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        nets = ANI2x(seed=0, n_members=2).neural_networks
    _, p0 = nets._param_list()
    assert nets._param_list()[1] is p0 and len(p0) == 2 * 7 * 8
    lin = nets.members[1].atomics["C"].final_layer
    lin.weight = torch.nn.Parameter(torch.zeros_like(lin.weight))        # attribute assignment -> registration hook
    _, p1 = nets._param_list()
    assert p1 is not p0 and any(q is lin.weight for q in p1)
    nets.set_active_members([1])
    _, p2 = nets._param_list()
    assert len(p2) == 7 * 8 and any(q is lin.weight for q in p2)
    nets.set_active_members([0, 1])
    torch.__future__.set_overwrite_module_params_on_conversion(True)
    try:
        nets.double()
    finally:
        torch.__future__.set_overwrite_module_params_on_conversion(False)
    _, p3 = nets._param_list()
    assert p3[0] is next(iter(nets.members[0].parameters())) and p3[0].dtype == torch.float64


def test_simple_ani_builder_host_side():
    """models.simple_ani mirrors arch.py:992-1066: AEV constants from cover_linearly, per-element widths of the chosen
    recipe, self energies of the level of theory, pair potentials on request; what the kernels do not cover is refused."""
    import math
    import warnings

    from torchani_amd.constants import GSAES
    from torchani_amd.models import simple_ani

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = simple_ani(("H", "C", "N", "O"), "wB97X-631Gd", ensemble_size=2, seed=3)
    c = m.aev_computer.constants()
    assert np.allclose((c.Rcr, c.Rca, c.EtaR, c.EtaA, c.Zeta), (5.2, 3.5, 19.7, 12.5, 14.1), rtol=1e-7) and c.cutoff_fn == "smooth"
    assert len(c.ShfR) == 16 and abs(c.ShfR[1] - (0.9 + 4.3 / 16)) < 1e-6 and abs(c.ShfZ[0] - math.pi / 8) < 1e-6   # (fp32 buffers)
    assert c.out_dim == 4 * 16 + 10 * 32
    assert [tuple(p.shape) for p in m.neural_networks.members[1].atomics["C"].parameters()] == \
        [(224, 384), (192, 224), (160, 192), (1, 160)]
    assert list(m.potentials) == ["nnp", "repulsion_xtb"] and m.potentials["repulsion_xtb"].cutoff == 5.2
    assert np.allclose(m.energy_shifter.self_energies.numpy(), [GSAES["wb97x-631gd"][s] for s in "HCNO"])
    m2 = simple_ani(("H", "O"), "b973c-def2mtzvp", dispersion=True, repulsion_cutoff=False, seed=1)
    assert list(m2.potentials) == ["nnp", "repulsion_xtb", "dispersion_d3"] and math.isinf(m2.potentials["repulsion_xtb"].cutoff)
    same = simple_ani(("H", "O"), "b973c-def2mtzvp", seed=1).state_dict()
    again = simple_ani(("H", "O"), "b973c-def2mtzvp", seed=1).state_dict()
    assert all(torch.equal(same[k], again[k]) for k in same)
    for bad in (dict(sections=6), dict(radial_shifts=32), dict(container="SingleNN"), dict(activation="tanh")):
        with pytest.raises(ValueError):
            simple_ani(("H", "C"), "wb97x-631gd", **bad)
    with pytest.raises(KeyError):
        simple_ani(("H", "C"), "hf-sto3g")


def test_single_point_atomic_charges_host_logic():
    """grad.single_point(atomic_charges=True) hands out the charges of models whose forward returns them (grad.py:345-353) and
    says so when a model has none; checked with stand-in models on CPU (the real ones need the GPU)."""
    from torchani_amd.grad import single_point
    from torchani_amd.tuples import SpeciesEnergies, SpeciesEnergiesAtomicCharges

    sp = torch.tensor([[0, 1, -1]])
    x = torch.zeros(1, 3, 3)

    def with_q(sc, cell=None, pbc=None, atomic=False, ensemble_values=False):
        return SpeciesEnergiesAtomicCharges(sc[0], sc[1].sum(dim=(1, 2)), torch.tensor([[0.25, -0.25, 0.0]]))

    def without_q(sc, cell=None, pbc=None, atomic=False, ensemble_values=False):
        return SpeciesEnergies(sc[0], sc[1].sum(dim=(1, 2)))

    out = single_point(with_q, sp, x, atomic_charges=True)
    assert set(out) == {"energies", "atomic_charges"} and out["atomic_charges"].tolist() == [[0.25, -0.25, 0.0]]
    assert set(single_point(with_q, sp, x)) == {"energies"}
    with pytest.raises(ValueError, match="atomic charges"):
        single_point(without_q, sp, x, atomic_charges=True)
    with pytest.raises(NotImplementedError):
        single_point(with_q, sp, x, atomic_charges=True, atomic_charges_grad=True)


def test_package_namespace_follows_the_reference():
    """``import torchani_amd as torchani``: the submodules and convenience names of torchani/__init__.py that exist here
    resolve (lazily), unknown names raise AttributeError."""
    import torchani_amd as t

    for mod in ("nn", "aev", "utils", "models", "constants", "grad", "potentials"):
        assert getattr(t, mod).__name__ == f"torchani_amd.{mod}"
    for mod in ("arch", "electro", "io"):   # host-side conveniences outside the hot path live in torchani_amd.extras
        assert getattr(t, mod).__name__ == f"torchani_amd.extras.{mod}"
    for name in ("AEVComputer", "ANINetworks", "ANIModel", "Ensemble", "SpeciesConverter", "SelfEnergy", "single_point"):
        assert callable(getattr(t, name))
    assert t.ANIModel is t.ANINetworks
    with pytest.raises(AttributeError):
        t.neurochem


def test_pair_potential_standalone_host_logic():
    """Pair potentials called on their own (core.py:37-67): atomic numbers are mapped to the potential's element indices on
    the host, unknown elements and CPU tensors are refused (the evaluation itself is a GPU test)."""
    from torchani_amd.potentials import RepulsionXTB, TwoBodyDispersionD3

    pot = RepulsionXTB(("H", "C", "O"), cutoff=5.2)
    z = torch.tensor([[8, 1, 1, -1], [6, 1, 1, 1]])
    assert pot._to_elem_idxs(z, True).tolist() == [[2, 0, 0, -1], [1, 0, 0, 0]] and pot._to_elem_idxs(z, False) is z
    with pytest.raises(ValueError, match="Unsupported element"):
        pot._to_elem_idxs(torch.tensor([[2, 1]]), True)
    with pytest.raises(ValueError, match="ROCm device"):
        pot(z, torch.zeros(2, 4, 3))
    d3 = TwoBodyDispersionD3.from_functional(("H", "O"), "b973c", cutoff=8.0)
    assert d3._to_elem_idxs(torch.tensor([[8, 1, 1]]), True).tolist() == [[1, 0, 0]]


def test_model_strategy_and_infer_conversions_are_accepted():
    """ANI.set_strategy / to_infer_model (arch.py:145-148,208-217): the reference's calls go through (there is one native
    path), unknown strategies raise like the reference's."""
    import warnings

    from torchani_amd.models import ANI2x

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = ANI2x(seed=0, n_members=2)
    for s in ("pyaev", "cuaev", "cuaev-fused", "cuaev-interface", "auto"):
        m.set_strategy(s)
    nets = m.neural_networks
    assert m.to_infer_model(use_mnp=True) is m and m.neural_networks is nets and m.aev_computer.strategy == "hip"
    with pytest.raises(ValueError):
        m.set_strategy("tpu")


def test_member_models_keep_pair_potentials_and_graph_stamp():
    """model[idx] / model_index= keep every pair potential with its enabled flag (the reference deep-copies the whole model,
    arch.py:252-261); the external-neighbor entry points refuse enabled pair potentials instead of dropping them; the
    auto-graph key changes with the configuration."""
    from torchani_amd.models import ANI2dr

    full = ANI2dr(seed=0, n_members=2)
    member = full[1]
    assert set(member.potentials.keys()) == set(full.potentials.keys()) and len(full.potentials) == 3
    for k in full.potentials:
        if k != "nnp":
            assert member.potentials[k] is full.potentials[k]
    assert len(member) == 1 and member.neural_networks is full.neural_networks.members[1]
    full.set_enabled("dispersion_d3", False)
    assert not full[0].potentials["dispersion_d3"]._enabled
    full.set_enabled("dispersion_d3", True)
    assert set(ANI2dr(seed=0, n_members=2, model_index=0).potentials.keys()) == set(full.potentials.keys())
    sp = torch.zeros((1, 2), dtype=torch.long)
    x = torch.zeros((1, 2, 3))
    with pytest.raises(NotImplementedError, match="pair potentials"):
        full.compute_from_neighbors(sp, x, (torch.zeros((2, 1), dtype=torch.long), torch.ones(1), torch.ones(1, 3)))
    s0 = full._config_stamp()
    full.set_enabled("repulsion_xtb", False)
    s1 = full._config_stamp()
    full.aev_computer.row_capacity = 256
    assert s0 != s1 and s1 != full._config_stamp()
    # the D3 parameter block is built from host copies (no device reads: legal during stream capture) and cached
    d3 = full.potentials["dispersion_d3"]
    assert d3.params() is d3.params() and abs(d3.params().cov_radius_bohr[0] - float(d3.covalent_radii[0])) < 1e-6


def test_species_column_map():
    """engine.species_column_map: column c of the AEV layout written with relabelled species is column map[c] of the
    reference layout (aev/_computer.py: radial blocks by species, angular blocks by the triu index of the species pair)."""
    from torchani_amd.engine import species_column_map

    S, nr, nb = 7, 16, 32
    L = S * nr + nb * S * (S + 1) // 2

    def triu(a, b):
        return a * S - (a * (a - 1)) // 2 + (b - a)

    ref = {}
    for sp in range(S):
        for k in range(nr):
            ref[sp * nr + k] = ("r", (sp,), k)
    for a in range(S):
        for b in range(a, S):
            for t in range(nb):
                ref[S * nr + triu(a, b) * nb + t] = ("a", (a, b), t)
    for order in ((0, 3, 1, 2, 4, 5, 6), (6, 5, 4, 3, 2, 1, 0), tuple(range(S))):
        m = species_column_map(S, S * nr, L, order)
        assert sorted(m.tolist()) == list(range(L))
        for c in range(L):
            kind, sp, t = ref[c]
            assert ref[int(m[c])] == (kind, tuple(sorted(order[x] for x in sp)), t)


def test_no_inline_asm_valu_to_mfma_hazard():
    """gfx950: two wait states between a VALU write of a VGPR and an MFMA reading it.  The compiler counts them for its own
    instructions, not behind asm(): k_gemm_l0b's first MFMA of a k step used to read the lo fragment one wait state
    behind the v_fma_mixhi_f16 that wrote it (wrong dE/dAEV in the first flagged slab of 4-slab tiles).  Disassemble the
    library that was built and check every such pair (tools/isa_hazards.py)."""
    import importlib.util

    from torchani_amd import _lib

    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "isa_hazards.py")
    spec = importlib.util.spec_from_file_location("isa_hazards", tools)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not os.path.exists(os.path.join(mod.LLVM, "llvm-objdump")):
        pytest.skip("no llvm-objdump")
    checked, bad = mod.check(_lib.LIB_PATH)
    assert checked > 1000 and not bad, bad
    # the checker itself: every rule fires on a bare producer -> consumer pair and accepts the padded one
    cases = [
        ("v_fma_mixhi_f16 v7, v17, v144, -v3 op_sel:[0,0,1]", "v_mfma_f32_32x32x16_f16 v[18:33], v[4:7], v[46:49], v[18:33]", 2),
        ("v_exp_f32_e32 v5, v4", "v_mul_f32_e32 v6, v5, v5", 1),
        ("v_add_f32_e32 v9, v1, v2", "v_mov_b32_dpp v3, v9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", 2),
        ("v_add_f32_e32 v9, v1, v2", "v_readlane_b32 s4, v9, 16", 1),
        ("v_mov_b32_e32 v9, v1", "v_permlane32_swap_b32_e32 v9, v10", 2),
    ]
    for prod, cons, need in cases:
        for pad in range(need + 1):
            seq = [("k", prod)] + ([("k", f"s_nop {pad - 1}")] if pad else []) + [("k", cons)]
            n_, bad_ = mod.check_instructions(seq)
            assert n_ == 1 and (len(bad_) == 1) == (pad < need), (prod, cons, pad, bad_)
    # (f) a scalar register written by the vector ALU as the base of a vector-memory instruction: five wait states
    for pad in range(6):
        seq = [("k", "v_readlane_b32 s59, v254, 6")] + ([("k", f"s_nop {pad - 1}")] if pad else []) + \
              [("k", "global_atomic_add v75, v2, v74, s[58:59] sc0")]
        n_, bad_ = mod.check_instructions(seq)
        assert n_ == 1 and (len(bad_) == 1) == (pad < 5), (pad, bad_)
    # an unrelated register, or an instruction in between, is no hazard
    assert mod.check_instructions([("k", "v_exp_f32_e32 v5, v4"), ("k", "v_mul_f32_e32 v6, v7, v7")]) == (0, [])
    assert mod.check_instructions([("k", "v_exp_f32_e32 v5, v4"), ("k", "v_mov_b32_e32 v8, v1"), ("k", "v_mul_f32_e32 v6, v5, v5")])[1] == []

"""Network containers with the reference's API, evaluated by the HIP/MFMA ensemble engine.

Mirrors torchani/nn/_core.py:117-167 (AtomicNetwork, TightCELU), torchani/nn/_containers.py:377-421
(ANINetworks), :590-660 (Ensemble), :663-734 (SpeciesConverter) and torchani/sae.py:54-64 (SelfEnergy).
Module/parameter names equal the reference's so its state dicts load unchanged
(``members.{m}.atomics.{Sym}.layers.{l}.weight``, ``...final_layer.weight``).

Gradients flow to the AEVs (hence to coordinates) and -- when parameters have requires_grad -- to every weight
and bias through the engine's training pass (anihip_mlp_weight_grads; the reference's native MNP path has no
weight gradients, csrc/mnp.cpp:138-232, it trains through eager autograd).  First order only.
"""
from __future__ import annotations

import operator
import os
import typing as tp
import warnings

import torch
from torch import Tensor

from .constants import ATOMIC_NUMBER, CELU_ALPHA
from .engine import PackedNetworks
from .tuples import SpeciesEnergies


class TightCELU(torch.nn.Module):
    """CELU with alpha = 0.1 (nn/_core.py:163-167)."""

    def forward(self, x: Tensor) -> Tensor:
        return torch.nn.functional.celu(x, alpha=CELU_ALPHA)


class AtomicNetwork(torch.nn.Module):
    """MLP parameter holder ``in -> h1 -> ... -> 1`` (nn/_core.py:117-149).  The arithmetic is done by the
    engine for whole containers; calling one network directly is not part of the hot path."""

    def __init__(self, layer_dims: tp.Sequence[int], activation: str = "celu", bias: bool = True) -> None:
        super().__init__()
        if any(d <= 0 for d in layer_dims):
            raise ValueError("Layer dims must be strict positive integers")
        if activation not in ("celu", "gelu"):
            raise ValueError("the HIP ensemble kernels implement CELU(0.1) (ANI-1x/2x) and GELU (ANI-2xr/2dr) networks")
        dims = tuple(layer_dims)
        self.layers = torch.nn.ModuleList(
            [torch.nn.Linear(i, o, bias=bias) for i, o in zip(dims[:-2], dims[1:-1])])
        self.final_layer = torch.nn.Linear(dims[-2], dims[-1], bias=bias)
        self.activation = TightCELU() if activation == "celu" else torch.nn.GELU()
        self.activation_name = activation
        self.has_biases = bool(bias)

    def linears(self) -> tp.List[torch.nn.Linear]:
        return list(self.layers) + [self.final_layer]


# training batches up to this many atoms keep their activations between forward and backward (39 KB / atom for
# ANI-2x x 8); larger ones recompute the forward chunk by chunk inside the backward call
_TRAIN_SPLIT_MAX_ATOMS = 1 << 17


def _flat_param_grads(packed: PackedNetworks, gw, gb) -> tp.List[Tensor]:
    """The engine's gradients in the order of the parameters handed to the Functions: member -> species -> layer ->
    (weight, bias), without the biases a bias-free network does not have."""
    has_bias = getattr(packed, "has_bias", None)
    flat, k = [], 0
    for m in range(packed.M):
        for s in range(packed.S):
            for l in range(packed.nl):
                flat.append(gw[m][s][l])
                if has_bias is None or has_bias[k]:
                    flat.append(gb[m][s][l])
                k += 1
    return flat


class _MLPBackwardFunction(torch.autograd.Function):
    """(grad_out, params) -> grad_aev = grad_out * d e / d aev as a differentiable function of the parameters (and of
    grad_out): what torch builds under create_graph=True for the eager networks of the reference.  Its backward for
    an incoming v [C, A, L] is the second-order pass anihip_mlp_tangent_weight_grads: d/d params of
    sum_i grad_out_i v_i . d e_i / d aev_i.  The second-order term with respect to the AEVs is not propagated."""

    @staticmethod
    def forward(ctx, grad_out: Tensor, a32: Tensor, species32: Tensor, packed: PackedNetworks, shape,
                *params: Tensor) -> Tensor:
        infer = getattr(packed, "infer_pack", None)
        _, g_unit, _ = (infer() if infer is not None else packed).forward_backward(species32, a32, want_grad=True)
        go = grad_out.detach().to(torch.float32).reshape(-1, 1)
        ctx.saved = (a32, species32, packed, g_unit, go)
        ctx.shape, ctx.go_shape, ctx.go_dtype = shape, grad_out.shape, grad_out.dtype
        ctx.param_dtypes = [p.dtype for p in params]
        return (g_unit * go).view(shape)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, v: Tensor):
        a32, species32, packed, g_unit, go = ctx.saved
        v32 = v.to(torch.float32).reshape(g_unit.shape)
        d_go = (v32 * g_unit).sum(dim=-1).view(ctx.go_shape).to(ctx.go_dtype)
        gw, gb, _ = packed.tangent_weight_grads(species32, a32, (v32 * go).contiguous())
        flat = _flat_param_grads(packed, gw, gb)
        flat = [t.to(dt) for t, dt in zip(flat, ctx.param_dtypes)]
        return (d_go, None, None, None, None, *flat)


class _MLPFunction(torch.autograd.Function):
    """params = the Linear parameters in the order member -> species -> layer -> (weight, bias)."""

    @staticmethod
    def forward(ctx, aevs: Tensor, species32: Tensor, packed: PackedNetworks, want_members: bool,
                *params: Tensor) -> Tensor:
        C, A = species32.shape
        train = any(p.requires_grad for p in params)
        need_grad = aevs.requires_grad and not train
        a32 = aevs.detach().to(torch.float32).contiguous().view(C * A, -1)
        ws = None
        if train and not want_members and C * A <= _TRAIN_SPLIT_MAX_ATOMS:
            # first half of the training pass: exact-fp32 forward, activations stay in ws for backward
            ae, ws = packed.train_forward(species32, a32)
            g = me = None
        else:
            ae, g, me = packed.forward_backward(species32, a32, want_grad=need_grad, want_members=want_members)
        ctx.g = g
        ctx.train = train
        ctx.flat_target = getattr(packed, "flat_target", None) if train else None
        ctx.flat_verify = getattr(packed, "flat_verify", None) if ctx.flat_target is not None else None
        ctx.aev_grad = aevs.requires_grad
        ctx.saved = (a32, species32, packed, ws) if train else None
        ctx.params = params if train else ()
        ctx.param_dtypes = [p.dtype for p in params]
        ctx.in_dtype = aevs.dtype
        ctx.shape = aevs.shape
        ctx.want_members = want_members
        ctx.member_packs = getattr(packed, "member_packs", None) if (want_members and aevs.requires_grad) else None
        ctx.saved_members = (a32, species32) if ctx.member_packs is not None else None
        if want_members:
            return me.view(packed.M, C, A).to(aevs.dtype)
        return ae.view(C, A).to(aevs.dtype)

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        if ctx.want_members:
            # grad_out [M, C, A]: d/d aev of sum_m grad_out_m e_m -- the engine returns the input gradient of ONE scalar
            # per atom, so every member gets a pass of its own (inference only: parameters frozen)
            if ctx.train or ctx.member_packs is None:
                raise RuntimeError("ensemble_values=True is differentiable with respect to the AEVs only (frozen "
                                   "parameters, aevs.requires_grad)")
            a32, species32 = ctx.saved_members
            go = grad_out.to(torch.float32)
            total = None
            for m, pk in enumerate(ctx.member_packs()):
                _, gm, _ = pk.forward_backward(species32, a32, want_grad=True)
                term = gm.view(ctx.shape) * go[m].unsqueeze(-1)
                total = term if total is None else total + term
            return (total.to(ctx.in_dtype), None, None, None, *([None] * len(ctx.param_dtypes)))
        if ctx.train:
            # training pass: forward recomputed in exact fp32 with the activations kept, then d/d weights, d/d biases
            # (and d/d aev scaled by the upstream gradient) in one engine call
            a32, species32, packed, ws = ctx.saved
            # create_graph=True (training on forces): d E / d aev must stay differentiable in the parameters
            second_order = torch.is_grad_enabled() and ctx.aev_grad
            target = ctx.flat_target if not second_order else None
            if target is not None and ctx.flat_verify is not None:
                ps, views = ctx.flat_verify
                for i, p_ in enumerate(ps):
                    tag = getattr(p_, "_anihip_flat", None)
                    if tag is None or p_.grad is not views[tag[1]]:
                        raise RuntimeError("the parameters' .grad were detached from the optimizer's flat gradient buffer between "
                                           "forward and backward (zero_grad(set_to_none=True)?): call forward again")
            gw, gb, _, gaev = packed.weight_grads(species32, a32, grad_out.detach().contiguous(),
                                                  want_grad_aev=ctx.aev_grad and not second_order, workspace=ws,
                                                  target=target)
            if target is not None:
                # the gradients were ADDED to the flat buffer the parameters' .grad are views of (torchani_amd.optim.Adam):
                # nothing for autograd to accumulate
                return (None, None, None, None, *([None] * len(ctx.param_dtypes)))
            if second_order:
                gaev = _MLPBackwardFunction.apply(grad_out, a32, species32, packed, ctx.shape, *ctx.params)
            flat = _flat_param_grads(packed, gw, gb)
            flat = [t.to(dt) for t, dt in zip(flat, ctx.param_dtypes)]
            ga = gaev.view(ctx.shape).to(ctx.in_dtype) if gaev is not None else None
            return (ga, None, None, None, *flat)
        if ctx.g is None:
            raise RuntimeError("AEVs did not require grad in forward")
        if torch.is_grad_enabled() and grad_out.requires_grad:
            raise RuntimeError("second-order gradients need trainable parameters (requires_grad) in the HIP engine")
        g = ctx.g.view(ctx.shape) * grad_out.to(torch.float32).unsqueeze(-1)
        return (g.to(ctx.in_dtype), None, None, None, *([None] * len(ctx.param_dtypes)))


# bumped whenever a parameter or a submodule is registered on any torch.nn.Module (incl. attribute assignment): cached
# parameter lists of the containers below are valid while it stands still
_STRUCT_EPOCH = [0]
_FLAT_ROUTE_NOTE = [False]   # the one-time note of _EngineContainer._flat_target to multi-rank runs
_VERSION_OF = operator.attrgetter("_version")
_DATA_PTR_OF = operator.methodcaller("data_ptr")


def _bump_struct_epoch(*_args) -> None:
    _STRUCT_EPOCH[0] += 1


torch.nn.modules.module.register_module_parameter_registration_hook(_bump_struct_epoch)
torch.nn.modules.module.register_module_module_registration_hook(_bump_struct_epoch)


class _EngineContainer(torch.nn.Module):
    """Shared machinery of ANINetworks / Ensemble: parameter packing cache + engine call.

    ``mlp_precision`` ("f16x3" default, or "fp32"; env TORCHANI_AMD_MLP_PRECISION) selects the GEMM
    arithmetic of the hidden layers (include/anihip.h)."""

    symbols: tp.Tuple[str, ...]
    mlp_precision: tp.Optional[str] = None

    def _member_networks(self) -> tp.List["ANINetworks"]:
        raise NotImplementedError

    def _param_list(self) -> tp.Tuple[tp.List["ANINetworks"], tp.List[torch.nn.Parameter]]:
        """Active member containers and their parameters.  Walking the module tree costs ~0.4 ms for 8 x 7 networks, more
        than a whole step of a small batch, so the flat list is kept until the member selection changes or ANY module
        in the process registers a parameter / submodule (_STRUCT_EPOCH, bumped by torch's registration hooks)."""
        members = self._member_networks()
        stamp = (tuple(id(m) for m in members), _STRUCT_EPOCH[0])
        hit = self.__dict__.get("_plist")
        # (Module._apply with torch.__future__.set_overwrite_module_params_on_conversion(True) swaps the Parameter objects
        # through the _parameters dicts, past the hooks: it swaps all of them, so looking at the first one is enough)
        if hit is None or hit[0] != stamp or (hit[1] and next(iter(members[0].parameters()), None) is not hit[1][0]):
            hit = (stamp, [p for m in members for p in m.parameters()])
            self.__dict__["_plist"] = hit
        return members, hit[1]

    def _pack(self, device: torch.device, species_order: tp.Optional[tp.Tuple[int, ...]] = None) -> PackedNetworks:
        """species_order (models.ANI.compact_species): the pack for a system whose species the engine numbers
        ``new = species_order.index(old)`` -- network s' is the one of species_order[s'], with the input rows of its first
        layer permuted to the AEV columns of the relabelled species (engine.species_column_map)."""
        members, params = self._param_list()
        precision = getattr(self, "mlp_precision", None) or os.environ.get("TORCHANI_AMD_MLP_PRECISION", "f16x3")
        if species_order is not None and tuple(species_order) == tuple(range(len(self.symbols))):
            species_order = None
        # (in-place updates bump _version, .to() / .data = ... change data_ptr; ~0.1 ms for 448 tensors)
        key = (device, precision, tuple(map(id, members)), tuple(map(_VERSION_OF, params)), tuple(map(_DATA_PTR_OF, params)),
               species_order)
        cache = self.__dict__.setdefault("_packed_cache", {})   # several member subsets stay packed
        if key not in cache:
            # (a container may read ONE of several outputs of its final layers, nn/_internal.py:69-93: that row only)
            oi = [getattr(m, "out_index", 0) for m in members]
            weights = [[[(lin.weight if li < len(m.atomics[s].linears()) - 1 or lin.out_features == 1
                          else lin.weight[oi[k]:oi[k] + 1].contiguous())
                         for li, lin in enumerate(m.atomics[s].linears())] for s in self.symbols]
                       for k, m in enumerate(members)]
            # (bias-free networks, nn/_core.py:122: zeros)
            biases = [[[(torch.zeros(w.shape[0], device=w.device) if lin.bias is None
                         else (lin.bias if lin.bias.shape[0] == w.shape[0] else lin.bias[oi[k]:oi[k] + 1].contiguous()))
                        for lin, w in zip(m.atomics[s].linears(), weights[k][si])]
                       for si, s in enumerate(self.symbols)] for k, m in enumerate(members)]
            aev_len = weights[0][0][0].shape[1]
            if species_order is not None:
                from .engine import species_column_map

                S = len(self.symbols)
                if aev_len != 16 * S + 32 * (S * (S + 1) // 2):
                    raise ValueError("species_order needs the ANI layout of the AEV (16 radial, 32 angular terms per block)")
                cmap = torch.from_numpy(species_column_map(S, 16 * S, aev_len, species_order))
                weights = [[[(w[:, cmap.to(w.device)].contiguous() if li == 0 else w) for li, w in enumerate(wm[old])]
                            for old in species_order] for wm in weights]
                biases = [[bm[old] for old in species_order] for bm in biases]
            acts = {getattr(m.atomics[s], "activation_name", "celu") for m in members for s in self.symbols}
            if len(acts) != 1:
                raise ValueError(f"all atomic networks of a container must share one activation, got {sorted(acts)}")
            if len(cache) >= 12:   # (parameters updated in place leave stale entries behind)
                cache.clear()
            cache[key] = PackedNetworks(weights, biases, aev_len, CELU_ALPHA, device, precision, activation=acts.pop())
        return cache[key]

    # "f16x3" (default): CELU networks of the fused kernel's shape train on the fast path -- forward and backward in ONE
    # launch of the fused split-fp16 kernel, weight gradients on bf16 x 3 MFMA (include/anihip.h) -- whenever the AEVs need no
    # gradient (energy training; training on forces differentiates the AEVs and takes the exact-fp32 passes).  "fp32": always
    # the exact-fp32 layer-by-layer passes of rounds 1-4.
    train_precision = "f16x3"

    def _fast_trainable(self) -> bool:
        members = self._member_networks()
        for m in members:
            for sname in self.symbols:
                net = m.atomics[sname]
                lins = net.linears()
                if (getattr(net, "activation_name", "celu") != "celu" or len(lins) != 4 or lins[-1].out_features != 1
                        or any(lin.bias is None for lin in lins) or any(lin.out_features > 256 for lin in lins[:-1])
                        or lins[0].in_features > 1024 or lins[0].in_features % 16):
                    return False
        return True

    def _plan(self) -> tp.Dict[str, tp.Any]:
        """What a training step needs to know about the module tree -- the Linear layers member -> species -> layer, their
        parameters, which of them have biases, whether the fast path can serve them -- gathered ONCE per structure (walking
        8 x 7 x 4 modules costs ~0.5 ms per call, as much as the kernels of a small batch; _STRUCT_EPOCH is bumped by torch's
        registration hooks whenever any module gains a parameter or submodule)."""
        members = self._member_networks()
        stamp = (tuple(id(m) for m in members), _STRUCT_EPOCH[0])
        hit = self.__dict__.get("_plan_cache")
        if hit is not None and hit[0] == stamp:
            first = hit[1]["params"][0] if hit[1]["params"] else None
            # (Module._apply with set_overwrite_module_params_on_conversion swaps the Parameter objects past the hooks)
            if first is None or next(iter(members[0].parameters()), None) is first:
                return hit[1]
        lins = [[m.atomics[s].linears() for s in self.symbols] for m in members]
        flat = [lin for ml in lins for sl in ml for lin in sl]
        plan = {"lins": lins, "flat": flat,
                "weights": [[[lin.weight for lin in sl] for sl in ml] for ml in lins],
                "biases": [[[lin.bias for lin in sl] for sl in ml] for ml in lins],
                "checked_ptrs": None,   # data_ptr tuple for which dtype / layout / device of every parameter were verified
                "params": [p for lin in flat for p in (lin.weight, lin.bias) if p is not None],
                "has_bias": [lin.bias is not None for lin in flat],
                "acts": {getattr(m.atomics[s], "activation_name", "celu") for m in members for s in self.symbols},
                "fast_trainable": self._fast_trainable()}
        self.__dict__["_plan_cache"] = (stamp, plan)
        return plan

    def _train_pack(self, device: torch.device, fast: bool = False) -> PackedNetworks:
        """Pack read by the training pass (fp32, or -- fast -- split-fp16); built once per parameter set and refreshed in place
        on the device (anihip_mlp_repack) whenever an optimizer step changed the parameters."""
        plan = self._plan()
        acts = set(plan["acts"])
        if len(acts) != 1:
            raise ValueError(f"all atomic networks of a container must share one activation, got {sorted(acts)}")
        lins = plan["lins"]
        weights = plan["weights"]
        # (bias-free networks -- the GELU networks of the ANI-2xr family, nn/_core.py:122 -- train against zero biases that
        # live as long as the pack: the engine's passes return their "gradients", which nobody receives)
        zeros = self.__dict__.setdefault("_zero_biases", {})

        def zero_bias(lin):
            k = (id(lin), device)
            if k not in zeros or zeros[k].shape[0] != lin.weight.shape[0]:
                zeros[k] = torch.zeros(lin.weight.shape[0], dtype=torch.float32, device=device)
            return zeros[k]

        biases = plan["biases"]
        if not all(plan["has_bias"]):
            biases = [[[lin.bias if lin.bias is not None else zero_bias(lin) for lin in sl] for sl in ml] for ml in lins]
        params = plan["params"]
        precision = "f16x3" if fast else "fp32"
        key = (device, precision, tuple(map(_DATA_PTR_OF, params)), len(params))   # (shapes are fixed by the structure stamp)
        versions = tuple(map(_VERSION_OF, params))
        cache = self.__dict__.setdefault("_train_cache", {})
        if key in cache and cache[key][0].scale_overflowed():
            del cache[key]   # (a weight outgrew the fp16 range of its layer's scale: pack again, new scales)
        if key not in cache:
            for k in [k for k in cache if k[2:] != key[2:] or k[0] != device]:   # (other parameter sets; both precisions may stay)
                del cache[k]
            aev_len = weights[0][0][0].shape[1]
            cache[key] = [PackedNetworks(weights, biases, aev_len, CELU_ALPHA, device, precision, activation=acts.pop()), versions]
        elif cache[key][1] != versions:
            # (a fast pack serves the fused kernel and the fast training pass only: their layouts alone are refreshed)
            cache[key][0].refresh(weights, biases, fused_only=fast)
            cache[key][1] = versions
        return cache[key][0]

    def _flat_target(self, packed: PackedNetworks, params: tp.List[Tensor]):
        """The engine's gradient table when every parameter's .grad is (still) its view of ONE flat gradient buffer of a
        torchani_amd.optim.Adam -- the weight-gradient kernels then add straight into that buffer and autograd receives
        nothing -- else None (the gradients go through autograd as tensors)."""
        tag0 = getattr(params[0], "_anihip_flat", None)
        grp = tag0[0]() if tag0 is not None else None
        if grp is None:
            return None
        views = grp.grad_views
        for p in params:
            tag = getattr(p, "_anihip_flat", None)
            if tag is None or tag[0]() is not grp or p.grad is not views[tag[1]]:
                return None
        key = (id(grp), id(packed), len(params))
        hit = self.__dict__.get("_flat_target_cache")
        if hit is not None and hit[0] == key and hit[2] is grp:
            packed.flat_target_views = views
            return hit[1]
        M, S, nl = packed.M, packed.S, packed.nl
        per = 2 * S * nl
        if len(params) != M * per:
            return None
        # (round-5 advice) on this route autograd never produces a gradient for the parameters: tensor hooks and
        # post-accumulate hooks registered on them would not fire -- with such hooks present (looked at when the table is built,
        # not per call) the gradients take the autograd route.  DistributedDataParallel's reducer hooks the gradient
        # accumulators out of Python's sight: ranks that train with the flat optimizer average their gradients with
        # torchani_amd.parallel.all_reduce_gradients(optimizer) instead, and are told so once.
        if any(getattr(p, "_backward_hooks", None) or getattr(p, "_post_accumulate_grad_hooks", None) for p in params):
            return None
        if not _FLAT_ROUTE_NOTE[0]:
            _FLAT_ROUTE_NOTE[0] = True
            import torch.distributed as dist

            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                warnings.warn("torchani_amd: the weight gradients are written straight into torchani_amd.optim.Adam's flat buffer; "
                              "autograd hooks on the parameters (DistributedDataParallel's reducer) do not see them -- average "
                              "them across ranks with torchani_amd.parallel.all_reduce_gradients(optimizer)")
        base = [params[i].grad.data_ptr() for i in range(per)]
        stride = sum(params[i].numel() for i in range(per))
        if M > 1:
            sb = params[per].grad.data_ptr() - base[0]
            if sb <= 0 or sb % 4 or any(params[m * per + i].grad.data_ptr() - base[i] != m * sb
                                        for m in range(1, M) for i in range(per)):
                return None
            stride = sb // 4
        w_ptr = [[base[(s_ * nl + l) * 2] for l in range(nl)] for s_ in range(S)]
        b_ptr = [[base[(s_ * nl + l) * 2 + 1] for l in range(nl)] for s_ in range(S)]
        try:
            tgt = packed.flat_grad_target(w_ptr, b_ptr, stride)
        except ValueError:
            return None
        self.__dict__["_flat_target_cache"] = (key, tgt, grp)
        packed.flat_target_views = views
        return tgt

    def _run(self, elem_idxs: Tensor, aevs: Tensor, atomic: bool, ensemble_values: bool) -> Tensor:
        if not aevs.is_cuda:
            raise ValueError("torchani_amd's network containers need tensors on a ROCm device")
        species32 = elem_idxs.to(torch.int32).contiguous()
        params: tp.List[Tensor] = []
        plan = self._plan()
        if torch.is_grad_enabled() and any(p.requires_grad for p in plan["params"]):
            params = plan["params"]
        trainable_fast = False
        if params and not ensemble_values:
            # (dtype, layout and device of 448 tensors: verified once per set of storages -- a .to() or a re-homing optimizer
            # changes the addresses --, 0.3 ms per call otherwise)
            ptrs = (aevs.device, tuple(map(_DATA_PTR_OF, params)))
            if plan["checked_ptrs"] is None or plan["checked_ptrs"][0] != ptrs:
                ok = all(p.dtype == torch.float32 and p.is_contiguous() and p.device == aevs.device for p in params)
                plan["checked_ptrs"] = (ptrs, ok)
            trainable_fast = plan["checked_ptrs"][1]
        # (the fast training path returns no d Loss / d aev: training on forces keeps the exact-fp32 passes)
        fast = bool(trainable_fast and not aevs.requires_grad and self.train_precision == "f16x3" and plan["fast_trainable"])
        packed = self._train_pack(aevs.device, fast) if trainable_fast else self._pack(aevs.device)
        packed.flat_target = self._flat_target(packed, params) if fast else None
        # which (weight, bias) slots of the engine's member -> species -> layer order have a parameter behind them
        packed.has_bias = plan["has_bias"]
        if trainable_fast and packed.activation == "gelu":
            # (the input gradient that force training differentiates once more comes from the inference pack: an fp32 GELU
            # pack only serves the training passes)
            dev_ = aevs.device   # (only the device: a closure over `aevs` would keep the tensor and its graph alive)
            packed.infer_pack = lambda: self._pack(dev_)
        if ensemble_values and aevs.requires_grad and not params:
            # differentiable member energies (nn/_containers.py:638-651 is plain autograd in the reference): the backward
            # needs every member's own d e_m / d aev, i.e. one single-member pass each
            dev_m = aevs.device
            packed.member_packs = lambda: [m._pack(dev_m) for m in self._member_networks()]
        fn_params = params
        if packed.flat_target is not None:
            # the gradients go straight into the optimizer's flat buffer: autograd need not carry 448 parameters through the
            # graph (0.6 ms per forward, several ms per backward of engine work) -- ONE stand-in input makes the backward run
            hook = self.__dict__.get("_grad_hook")
            if hook is None or hook.device != aevs.device:
                hook = torch.zeros(1, device=aevs.device, requires_grad=True)
                self.__dict__["_grad_hook"] = hook
            fn_params = [hook]
            packed.flat_verify = (params, packed.flat_target_views)
        out = _MLPFunction.apply(aevs, species32, packed, ensemble_values, *fn_params)
        # [C, A] (or [M, C, A]); molecular energies are the sum over atoms (nn/_containers.py:417-421)
        return out if atomic else out.sum(dim=-1)

    def forward(self, elem_idxs, aevs: tp.Optional[Tensor] = None, atomic: bool = False,
                ensemble_values: bool = False):
        if isinstance(elem_idxs, tuple):  # legacy ``_, E = nets((idxs, aevs), cell, pbc)``
            warnings.warn("The tuple call signature is deprecated; use nets(elem_idxs, aevs)")
            idxs, a = elem_idxs
            return SpeciesEnergies(idxs, self._run(idxs, a, False, False))
        assert aevs is not None
        return self._run(elem_idxs, aevs, atomic, ensemble_values)


class ANINetworks(_EngineContainer):
    """Per-element networks: ``E = nets(elem_idxs, aevs, atomic=False)`` (nn/_containers.py:325-425)."""

    def __init__(self, modules: tp.Dict[str, AtomicNetwork]) -> None:
        super().__init__()
        self.symbols = tuple(modules.keys())
        self.atomics = torch.nn.ModuleDict(modules)
        self.num_species = len(self.symbols)
        self.atomic_numbers = torch.tensor([ATOMIC_NUMBER[s] for s in self.symbols], dtype=torch.long)
        self.total_members_num = 1
        self.active_members_idxs = [0]

    def __getitem__(self, sym: str) -> AtomicNetwork:
        return self.atomics[sym]

    def _member_networks(self) -> tp.List["ANINetworks"]:
        return [self]

    def forward(self, elem_idxs, aevs=None, atomic: bool = False, ensemble_values: bool = False):
        out = super().forward(elem_idxs, aevs, atomic, ensemble_values)
        return out

    @classmethod
    def build(cls, symbols: tp.Sequence[str], in_dim: int, hidden: tp.Dict[str, tp.Sequence[int]],
              activation: str = "celu", bias: bool = True, out_dim: int = 1):
        return cls({s: AtomicNetwork((in_dim,) + tuple(hidden[s]) + (out_dim,), activation, bias) for s in symbols})

    def to_infer_model(self, use_mnp: bool = False) -> "ANINetworks":
        return self  # already the fused native path (reference: nn/_containers.py:423-425)


class ANINetworksDiscardFirstScalar(ANINetworks):
    """Networks with two outputs of which the SECOND is the container's value (nn/_internal.py:69-93
    _ANINetworksDiscardFirstScalar: the charge networks of ANI-mbis).  The engine evaluates the selected output row of
    the final layers like a one-output network."""

    out_index = 1


class Ensemble(_EngineContainer):
    """Mean over member containers, all members evaluated in one grouped GEMM per layer
    (nn/_containers.py:590-660)."""

    def __init__(self, modules: tp.Sequence[ANINetworks]) -> None:
        super().__init__()
        self.members = torch.nn.ModuleList(modules)
        self.symbols = modules[0].symbols
        self.num_species = modules[0].num_species
        self.atomic_numbers = modules[0].atomic_numbers
        self.total_members_num = len(self.members)
        self.active_members_idxs = list(range(len(self.members)))

    def __len__(self) -> int:
        return len(self.members)

    def __getitem__(self, idx: int) -> ANINetworks:
        return self.members[idx]

    def set_active_members(self, idxs: tp.Sequence[int]) -> None:
        # nn/_core.py:99-110
        if not idxs or any(i < 0 or i >= self.total_members_num for i in idxs) or len(set(idxs)) != len(idxs):
            raise IndexError("Invalid member indices")
        self.active_members_idxs = list(idxs)

    def get_active_members_num(self) -> int:
        return len(self.active_members_idxs)

    def _member_networks(self) -> tp.List[ANINetworks]:
        return [self.members[i] for i in self.active_members_idxs]

    def to_infer_model(self, use_mnp: bool = False) -> "Ensemble":
        return self


class SpeciesConverter(torch.nn.Module):
    """Atomic numbers -> element indices (nn/_containers.py:663-734); unsupported elements raise
    ValueError."""

    def __init__(self, symbols: tp.Sequence[str]) -> None:
        super().__init__()
        if isinstance(symbols, str):
            raise ValueError("Please use SpeciesConverter(['H', 'C', 'N', 'O']) instead of a string")
        conv = torch.full((120,), -1, dtype=torch.long)
        for i, s in enumerate(symbols):
            conv[ATOMIC_NUMBER[s]] = i
        self.register_buffer("conv_tensor", conv)
        self.atomic_numbers = torch.tensor([ATOMIC_NUMBER[s] for s in symbols], dtype=torch.long)

    RECHECK_EVERY = 64   # calls with the same species tensor between two validity checks (one host read each)

    def forward(self, atomic_nums, nop: bool = False):
        if isinstance(atomic_nums, tuple):
            warnings.warn("The tuple call signature is deprecated; use idxs = converter(atomic_nums)")
            return (self.forward(atomic_nums[0]), atomic_nums[1])
        # The validity check reads the device (the reference syncs here too, nn/_containers.py:727-733): once per species
        # TENSOR -- identity and version, the entry keeps the tensor alive so that its address cannot be handed to another
        # one meanwhile -- so that an MD loop that passes the same species tensor step after step never waits for the device
        # (identity + version only see writes made THROUGH torch: a buffer shared with numpy -- torch.from_numpy, as ASE and MD
        # drivers hand them over -- or written through .data can change behind the same key, so every RECHECK_EVERY-th call
        # with the same tensor validates again; mutate species tensors with torch in-place operations, or pass a new tensor)
        key = (atomic_nums.data_ptr(), atomic_nums._version, tuple(atomic_nums.shape), atomic_nums.dtype, bool(nop))
        hit = self.__dict__.get("_checked")
        checked = hit is not None and hit[0] == key
        calls = self.__dict__.get("_checked_calls", 0) + 1 if checked else 0
        if calls >= self.RECHECK_EVERY:
            checked, calls = False, 0
        self.__dict__["_checked_calls"] = calls
        if nop:
            if not checked and atomic_nums.max() >= len(self.atomic_numbers):
                raise ValueError(f"Unsupported element idx in {atomic_nums}")
            self.__dict__["_checked"] = (key, atomic_nums)
            return atomic_nums
        elem_idxs = self.conv_tensor[atomic_nums.clamp(min=-1)]  # -1 indexes the last (unused, -1) slot
        if not checked and (elem_idxs[atomic_nums != -1] == -1).any():
            raise ValueError(f"Model doesn't support some elements in input. Input elements include: "
                             f"{torch.unique(atomic_nums)} Supported elements are: {self.atomic_numbers}")
        self.__dict__["_checked"] = (key, atomic_nums)
        return elem_idxs


class SelfEnergy(torch.nn.Module):
    """Per-element constant energies (sae.py:25-64); the buffer is float32 like the reference's."""

    def __init__(self, symbols: tp.Sequence[str], self_energies: tp.Sequence[float]) -> None:
        super().__init__()
        self.symbols = tuple(symbols)
        self.register_buffer("self_energies", torch.tensor(list(self_energies), dtype=torch.float))
        self._enabled = True

    def forward(self, elem_idxs: Tensor, atomic: bool = False) -> Tensor:
        e = self.self_energies[elem_idxs.clamp(min=0)]
        e = e.masked_fill(elem_idxs == -1, 0.0)
        return e if atomic else e.sum(dim=-1)

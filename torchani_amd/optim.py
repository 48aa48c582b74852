"""Adam for the ensemble's parameters as ONE kernel launch over flat buffers (anihip_adam_step).

The reference's training recipe is ``torch.optim.Adam(nets.parameters(), lr=1e-4)`` followed by ``loss.backward();
opt.step()`` (tools/training-aev-benchmark.py:88,120-150): over the 448 tensors of ANI-2x x 8 that is a dozen foreach
launches per step (3 ms on an MI355X) for 28 bytes of traffic per parameter (0.06 ms).  ``torchani_amd.optim.Adam`` is a
drop-in with the same constructor and update rule:

* at construction the parameters of a group are RE-HOMED into one flat fp32 buffer in the order given (``p.data`` becomes a
  view of it, values unchanged) and every ``p.grad`` becomes a view of one flat gradient buffer;
* ``step()`` is one launch over (parameters, gradients, exp_avg, exp_avg_sq); the step count lives on the device, so the whole
  training step can be captured into a HIP graph;
* ``zero_grad()`` is one fill (``set_to_none`` is ignored: the views stay) -- and not needed per step: ``step()`` zeroes the
  gradients behind the update (``zero_grad_in_step=True``), which is what lets a captured HIP graph of the step replay;
* the network containers (nn.ANINetworks / nn.Ensemble) notice that their parameters' gradients are views of one flat buffer
  and let the weight-gradient kernels ADD straight into it (anihip_species_grads.member_stride / accumulate): autograd then
  has nothing to accumulate for the parameters -- ``p.grad`` holds the result as usual, but ``torch.autograd.grad(loss,
  params)`` would see None (use torch.optim.Adam, or ``nets.train_precision = "fp32"``, for that).

A gradient that arrives as a separate tensor (``p.grad`` replaced by autograd or by the caller) is copied into the flat buffer
by ``step()`` first, so any model trains; only engine containers take the direct route.
"""
from __future__ import annotations

import ctypes as C
import typing as tp
import weakref

import torch
from torch import Tensor

from . import _lib


class _FlatGroup:
    """Flat storage of one parameter group."""

    def __init__(self, params: tp.List[torch.nn.Parameter]) -> None:
        dev = params[0].device
        if dev.type != "cuda":
            raise ValueError("torchani_amd.optim.Adam updates parameters on a ROCm device")
        for p in params:
            if p.dtype != torch.float32 or p.device != dev or p.is_sparse:
                raise ValueError("torchani_amd.optim.Adam needs dense fp32 parameters on one device")
        self.params = params
        self.sizes = [p.numel() for p in params]
        n = sum(self.sizes)
        self.n = n
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.step = torch.zeros(1, dtype=torch.int32, device=dev)
        self.grad_views: tp.List[Tensor] = []
        self.offsets: tp.List[int] = []
        off = 0
        ref = weakref.ref(self)
        with torch.no_grad():
            for i, p in enumerate(params):
                k = p.numel()
                view = self.flat[off:off + k].view(p.shape)
                view.copy_(p.detach())
                p.data = view
                g = self.grad[off:off + k].view(p.shape)
                if p.grad is not None:
                    g.copy_(p.grad)
                p.grad = g
                p._anihip_flat = (ref, i)
                self.grad_views.append(g)
                self.offsets.append(off)
                off += k

    def check_homes(self) -> None:
        """The update kernel writes the flat buffer: every parameter must still be its view of it (``model.to(...)``,
        ``p.data = ...`` or a dtype change move a parameter out -- the optimizer would then update memory nobody reads)."""
        base = self.flat.data_ptr()
        # (every 4th call all of them -- 448 pointer reads are 0.1 ms of host time, 4 % of a graphed ANI-2x x 8 step --, in between
        # the first and the last: a model moved as a whole moves those too.  A single parameter re-homed by hand therefore trains
        # against stale memory for at most three steps before this raises)
        self._homes_calls = getattr(self, "_homes_calls", -1) + 1
        idx = range(len(self.params)) if self._homes_calls % 4 == 0 else (0, len(self.params) - 1)
        for i in idx:
            p, off = self.params[i], self.offsets[i]
            if p.data_ptr() != base + 4 * off:
                raise RuntimeError("torchani_amd.optim.Adam: a parameter no longer lives in the optimizer's flat buffer (moved to "
                                   "another device / dtype, or its .data was replaced): build a new optimizer for the model")

    def gather_stray_grads(self) -> None:
        """Gradients that arrived as tensors of their own (autograd's default route): into the flat buffer, views restored."""
        for i, p in enumerate(self.params):
            g = p.grad
            if g is self.grad_views[i]:
                continue
            if g is not None:
                self.grad_views[i].copy_(g)
            else:
                self.grad_views[i].zero_()
                self.saw_none_grad = True
            p.grad = self.grad_views[i]


class Adam(torch.optim.Optimizer):
    """``torchani_amd.optim.Adam(params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0)``: torch.optim.Adam's update
    (amsgrad / maximize / foreach / capturable / differentiable are not options: one fused launch, always capturable).
    The hyper-parameters of a group are read at every ``step()`` (learning-rate schedulers work as usual) and passed by value:
    a CAPTURED step replays the values it was captured with -- capture again after a scheduler changed them.

    Differences from torch.optim.Adam, all consequences of the one flat update: (1) a parameter whose ``grad`` is None at
    ``step()`` counts as having a ZERO gradient -- its moments decay and a weight decay still applies -- where torch skips it
    (``step()`` refuses weight_decay > 0 in that situation instead of decaying silently); (2) ``zero_grad_in_step=True`` (default)
    clears the gradients inside ``step()``: read gradient norms BEFORE the step, or pass False; (3) one step counter per group."""

    def __init__(self, params, lr: float = 1e-3, betas: tp.Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, zero_grad_in_step: bool = True) -> None:
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        # the update kernel zeroes the gradients behind itself (the next backward accumulates into them): a training step
        # needs no zero_grad launch, and a captured HIP graph of the step replays correctly.  False: torch's behaviour (the
        # gradients stay until zero_grad()).
        self.zero_grad_in_step = bool(zero_grad_in_step)
        self._flat: tp.List[_FlatGroup] = []
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.requires_grad]
            if not ps:
                raise ValueError("a parameter group without trainable parameters")
            self._flat.append(_FlatGroup(ps))
        # (the parameters moved: whatever cached their addresses or versions must look again)
        torch.autograd.graph.increment_version([p for f in self._flat for p in f.params])

    def add_param_group(self, param_group) -> None:
        super().add_param_group(param_group)
        if hasattr(self, "_flat"):   # (called by the constructor before _flat exists)
            ps = [p for p in self.param_groups[-1]["params"] if p.requires_grad]
            if not ps:
                raise ValueError("a parameter group without trainable parameters")
            self._flat.append(_FlatGroup(ps))
            torch.autograd.graph.increment_version(ps)

    def zero_grad(self, set_to_none: bool = True) -> None:
        for f in self._flat:
            f.gather_stray_grads()   # (restores the views; what they held is zeroed next)
            f.grad.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        for group, f in zip(self.param_groups, self._flat):
            f.check_homes()
            f.saw_none_grad = False
            f.gather_stray_grads()
            if f.saw_none_grad and group["weight_decay"] > 0.0:
                raise RuntimeError("torchani_amd.optim.Adam: a parameter without a gradient in a group with weight_decay > 0 -- "
                                   "torch.optim.Adam would skip it, the flat update would decay it (class docstring)")
            b1, b2 = group["betas"]
            stream = torch.cuda.current_stream(f.flat.device).cuda_stream
            _lib.check(L.anihip_adam_step(stream, f.flat.data_ptr(), f.grad.data_ptr(), f.exp_avg.data_ptr(),
                                          f.exp_avg_sq.data_ptr(), f.n, C.c_double(group["lr"]), C.c_double(b1), C.c_double(b2),
                                          C.c_double(group["eps"]), C.c_double(group["weight_decay"]), f.step.data_ptr(),
                                          1 if self.zero_grad_in_step else 0))
            # the kernel wrote the parameters behind torch's back: bump their versions (packed copies are refreshed by it)
            torch.autograd.graph.increment_version(f.params)
        return loss

    def state_dict(self):
        """torch.optim.Optimizer's layout -- ``{"state": {index: {"step", "exp_avg", "exp_avg_sq"}}, "param_groups": [{...,
        "params": [indices]}]}`` with the parameters numbered through the groups in order -- so that generic checkpoint tooling
        reads it; the per-parameter tensors are copies of the slices of the flat moment buffers, ``step`` a float32 scalar tensor
        (the device counter of the group)."""
        state, groups, k = {}, [], 0
        for g, f in zip(self.param_groups, self._flat):
            ids = []
            for i, p in enumerate(f.params):
                off, n = f.offsets[i], f.sizes[i]
                state[k] = {"step": f.step.to(torch.float32).reshape(()).clone(),
                            "exp_avg": f.exp_avg[off:off + n].view(p.shape).clone(),
                            "exp_avg_sq": f.exp_avg_sq[off:off + n].view(p.shape).clone()}
                ids.append(k)
                k += 1
            groups.append({**{key: v for key, v in g.items() if key != "params"}, "params": ids})
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, state) -> None:
        """What ``state_dict()`` returns (the torch layout; also a torch.optim.Adam's own state dict for the same parameters in
        the same order), or the ``{"param_groups", "flat"}`` form this class wrote before round 6."""
        if "flat" in state:
            if len(state["flat"]) != len(self._flat):
                raise ValueError("state dict of another optimizer layout")
            for g, sg in zip(self.param_groups, state["param_groups"]):
                g.update(sg)
            for f, sf in zip(self._flat, state["flat"]):
                if list(sf["sizes"]) != list(f.sizes):
                    raise ValueError("state dict of another parameter layout")
                f.exp_avg.copy_(sf["exp_avg"])
                f.exp_avg_sq.copy_(sf["exp_avg_sq"])
                f.step.copy_(sf["step"])
            return
        if len(state["param_groups"]) != len(self._flat):
            raise ValueError("state dict of another optimizer layout")
        for g, sg, f in zip(self.param_groups, state["param_groups"], self._flat):
            ids = list(sg["params"])
            if len(ids) != len(f.params):
                raise ValueError("state dict of another parameter layout")
            g.update({key: v for key, v in sg.items() if key != "params"})
            steps = set()
            with torch.no_grad():
                for i, k in enumerate(ids):
                    st = state["state"].get(k)
                    off, n = f.offsets[i], f.sizes[i]
                    if st is None:   # (torch leaves out parameters that never had a gradient: fresh moments)
                        f.exp_avg[off:off + n].zero_()
                        f.exp_avg_sq[off:off + n].zero_()
                        continue
                    if st["exp_avg"].numel() != n:
                        raise ValueError("state dict of another parameter layout")
                    f.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
                    f.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                    steps.add(int(float(st["step"])))
                if len(steps) > 1:
                    raise ValueError("the parameters of a group carry different step counts: this optimizer keeps ONE counter per "
                                     "group (torch skips parameters without a gradient, see the class docstring)")
                f.step.fill_(steps.pop() if steps else 0)

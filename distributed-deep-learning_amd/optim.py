"""FusedAdam: Adam over a flat arena - one HIP kernel launch per step.

Reference: ``optim.Adam(model.parameters())`` with every default (single.py:108,305:
lr=1e-3, betas=(0.9,0.999), eps=1e-8, weight_decay=0, no amsgrad); torch dispatches
that to a ``foreach`` implementation over 364 tensors.  Here params,
grads, exp_avg and exp_avg_sq are four contiguous buffers with one layout
(``ddl.utils.arena.FlatArena``) and the step is a single elementwise kernel
(``ddl.ops.adam_step_flat`` -> ``adam_flat_kernel`` in csrc/elementwise.hip) that can also clear the gradients,
replacing ``zero_grad`` (K13).  The step counter lives on the device so the whole
train step can be captured into a hipGraph.

It subclasses ``torch.optim.Optimizer`` and keeps ``self.state[p] = {step, exp_avg,
exp_avg_sq}`` (views into the flat buffers) and Adam's ``param_groups`` keys, so
``state_dict()`` is indistinguishable from ``torch.optim.Adam``'s and checkpoints keep
the reference's layout.  ``load_state_dict`` copies *into* the views
(never rebinds them).

ZeRO-1 (``shard=(rank, world)``): only this rank's 1/world slice of every arena segment
is updated; the DP engine all-gathers the parameters afterwards.
"""
from __future__ import annotations

from typing import Any, Dict, Iterable, List, Optional, Tuple, Union

import torch

from .utils.arena import FlatArena


def _adam_torch(p, g, m, v, step: int, lr, b1, b2, eps, wd, zero_grad: bool) -> None:
    """Reference flat Adam in plain torch ops (CPU path and numerics oracle)."""
    grad_buf = g
    if wd != 0.0:
        g = g + wd * p
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))
    if zero_grad:
        grad_buf.zero_()


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params: Union[FlatArena, Iterable], lr: float = 1e-3,
                 betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, shard: Optional[Tuple[int, int]] = None) -> None:
        if isinstance(params, FlatArena):
            arena = params
        else:
            plist = list(params)
            if plist and isinstance(plist[0], dict):
                raise ValueError("FusedAdam takes a flat parameter list (one param group)")
            if plist and isinstance(plist[0], tuple):
                named = plist
            else:
                named = [(f"param{i}", p) for i, p in enumerate(plist)]
            arena = FlatArena(named)
        self.arena = arena
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False,
                        maximize=False, foreach=None, capturable=False, differentiable=False,
                        fused=None, decoupled_weight_decay=False)
        super().__init__(arena.params, defaults)
        self.exp_avg = arena.like(0.0)
        self.exp_avg_sq = arena.like(0.0)
        # device-resident step counter (float32 like torch's per-param ``step`` tensors)
        self.step_t = torch.zeros(1, dtype=torch.float32, device=arena.device)
        self.aux_t = torch.ones(2, dtype=torch.float32, device=arena.device)   # bias corrections (device)
        self._host_step = 0
        self.shard = shard
        for i, p in enumerate(arena.params):
            self.state[p] = {
                "step": self.step_t.view(()),         # shared: all parameters step together
                "exp_avg": arena.view_in(self.exp_avg, i),
                "exp_avg_sq": arena.view_in(self.exp_avg_sq, i),
            }

    # ------------------------------------------------------------------
    def _ranges(self) -> List[Tuple[int, int]]:
        if self.shard is None:
            return [(0, self.arena.numel)]
        r, w = self.shard
        out = []
        for seg in self.arena.segments:
            n = (seg.end - seg.start) // w
            out.append((seg.start + r * n, seg.start + (r + 1) * n))
        return out

    @torch.no_grad()
    def step(self, closure=None, zero_grad: bool = False):
        """One Adam update.  ``zero_grad=True`` also clears the gradient arena in the same pass."""
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        lr, (b1, b2), eps, wd = g["lr"], g["betas"], g["eps"], g["weight_decay"]
        a = self.arena
        self._host_step += 1
        if a.device.type == "cuda":
            from . import ops
            ops.adam_step_flat(a.flat_param, a.flat_grad, self.exp_avg, self.exp_avg_sq, self.step_t, self.aux_t,
                               lr, b1, b2, eps, wd, zero_grad, self._ranges())
        else:
            self.step_t += 1
            for s, e in self._ranges():
                _adam_torch(a.flat_param[s:e], a.flat_grad[s:e], self.exp_avg[s:e], self.exp_avg_sq[s:e],
                            self._host_step, lr, b1, b2, eps, wd, zero_grad)
            if zero_grad and self.shard is not None:
                a.flat_grad.zero_()
        return loss

    def zero_grad(self, set_to_none: bool = False) -> None:
        # grads are arena views: never set to None (that would detach them from the buckets)
        self.arena.zero_grad()

    # ------------------------------------------------------------------
    def state_dict(self) -> Dict[str, Any]:
        sd = super().state_dict()
        # per-parameter ``step`` must be independent tensors in the serialised form
        # (the base class hands out the live per-param dicts: build new ones, never edit those)
        sd["state"] = {k: {n: (t.detach().clone() if isinstance(t, torch.Tensor) else t)
                           for n, t in st.items()} for k, st in sd["state"].items()}
        return sd

    @torch.no_grad()
    def load_state_dict(self, state_dict: Dict[str, Any]) -> None:
        groups = state_dict["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.arena.params):
            raise ValueError("optimizer state does not match this arena")
        for k, v in groups[0].items():
            if k != "params":
                self.param_groups[0][k] = v
        step_val = None
        for idx, st in state_dict["state"].items():
            p = self.arena.params[int(idx)]
            mine = self.state[p]
            mine["exp_avg"].copy_(st["exp_avg"])
            mine["exp_avg_sq"].copy_(st["exp_avg_sq"])
            step_val = float(st["step"])
        if step_val is not None:
            self.step_t.fill_(step_val)
            self._host_step = int(step_val)

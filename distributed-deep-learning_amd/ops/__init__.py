"""Python face of the hand-written gfx950 kernels (csrc/*.hip -> ``_ddl_hip`` extension).

Rules:
  * on a CUDA(HIP) tensor every op here runs the HIP extension and **fails loudly** if the
    extension is not built (``python __graft_entry__.py`` / ``python -m ddl.build``) - there is
    no silent eager fallback on the GPU path;
  * on CPU tensors the same functions run a plain-PyTorch implementation, which is both the
    plumbing path for the gloo tests and the numerics oracle for the kernels.
"""
from __future__ import annotations

import importlib
import os
import sys
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from ..utils.knobs import knob

_EXT = None
_EXT_ERR: Optional[BaseException] = None


def ext_dir() -> str:
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")


def load_ext():
    """Import the in-tree built extension ``csrc/_ddl_hip*.so`` (built by ``ddl.build``)."""
    global _EXT, _EXT_ERR
    if _EXT is not None:
        return _EXT
    d = ext_dir()
    if d not in sys.path:
        sys.path.insert(0, d)
    try:
        _EXT = importlib.import_module("_ddl_hip")
    except BaseException as e:      # noqa: BLE001 - surfaced verbatim by require_ext
        _EXT_ERR = e
        _EXT = None
    return _EXT


def have_ext() -> bool:
    return load_ext() is not None


def require_ext():
    ext = load_ext()
    if ext is None:
        raise RuntimeError(
            "the gfx950 HIP extension (csrc/_ddl_hip*.so) is not built or failed to load: "
            f"{_EXT_ERR!r}.  Build it with `python __graft_entry__.py` (or `python -m ddl.build`). "
            "GPU ops never fall back to eager PyTorch.")
    return ext


# --------------------------------------------------------------------------------------
# K14: uint8 NCHW image batch -> fp32 (/255), NCHW or NHWC
# --------------------------------------------------------------------------------------
def normalize_input(x: torch.Tensor, channels_last: bool = False) -> torch.Tensor:
    """uint8 [B,3,H,W] -> fp32 /255.  ``channels_last`` returns a contiguous [B,H,W,3] tensor."""
    if x.is_cuda and x.dtype == torch.uint8:
        return require_ext().normalize_u8(x.contiguous(), channels_last)
    y = x.float().div(255.0) if x.dtype == torch.uint8 else x.float()
    return y.permute(0, 2, 3, 1).contiguous() if channels_last else y


# --------------------------------------------------------------------------------------
# K10/K11: fused softmax + cross-entropy (+ argmax), forward and backward in one launch
# --------------------------------------------------------------------------------------
class _FusedCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets):
        ext = require_ext()
        loss, dlogits, preds = ext.ce_fwd_bwd(logits.contiguous(), targets.contiguous())
        ctx.save_for_backward(dlogits)
        ctx.mark_non_differentiable(preds)
        return loss, preds

    @staticmethod
    def backward(ctx, gloss, _gpreds):
        (dlogits,) = ctx.saved_tensors
        return dlogits * gloss, None


def cross_entropy_with_preds(logits: torch.Tensor, targets: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Mean cross-entropy and argmax predictions (reference: F.cross_entropy + torch.argmax,
    single.py:139-141).  One kernel on the GPU computes loss, d(loss)/d(logits) and argmax."""
    if logits.is_cuda:
        return _FusedCE.apply(logits, targets)
    return F.cross_entropy(logits, targets), logits.detach().argmax(dim=1)


def copy_inputs(dst_x: torch.Tensor, x: torch.Tensor, dst_y: torch.Tensor, y: torch.Tensor) -> None:
    """``dst_x.copy_(x); dst_y.copy_(y)`` - on the GPU as ONE launch when the layout allows it (the two input copies
    in front of a graph replay; each eager operation between two graph launches is a queue round trip)."""
    if (knob("DDL_COPY_PAIR", "1") == "1" and x.is_cuda and y.is_cuda and dst_x.is_cuda and dst_y.is_cuda and have_ext()
            and x.device == dst_x.device == y.device == dst_y.device
            and x.dtype == dst_x.dtype and y.dtype == dst_y.dtype and x.shape == dst_x.shape and y.shape == dst_y.shape
            and x.is_contiguous() and y.is_contiguous() and dst_x.is_contiguous() and dst_y.is_contiguous()
            and (x.numel() * x.element_size()) % 16 == 0 and x.data_ptr() % 16 == 0 and dst_x.data_ptr() % 16 == 0):
        require_ext().copy_pair(dst_x, x, dst_y, y)
        return
    dst_x.copy_(x, non_blocking=True)
    dst_y.copy_(y, non_blocking=True)


def cross_entropy_backward(logits: torch.Tensor, targets: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """``loss = cross_entropy(logits, targets); loss.backward()`` in one go, returning ``(loss.detach(), preds)``.
    On the GPU the fused kernel's d(loss)/d(logits) is handed straight to ``logits.backward`` - no autograd node for
    the loss, so no ones_like fill and no ``dlogits * 1`` launch in the captured step (bit-identical gradients)."""
    if logits.is_cuda:
        loss, dlogits, preds = require_ext().ce_fwd_bwd(logits.detach().contiguous(), targets.contiguous())
        logits.backward(dlogits)
        return loss, preds
    loss, preds = cross_entropy_with_preds(logits, targets)
    loss.backward()
    return loss.detach(), preds


def cross_entropy(logits: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
    return cross_entropy_with_preds(logits, targets)[0]


# --------------------------------------------------------------------------------------
# K12/K13: flat fused Adam (+ zero_grad)
# --------------------------------------------------------------------------------------
def adam_step_flat(p, g, m, v, step_t, aux_t, lr, b1, b2, eps, wd, zero_grad: bool,
                   ranges: Sequence[Tuple[int, int]]) -> None:
    ext = require_ext()
    ext.adam_step(p, g, m, v, step_t, aux_t, float(lr), float(b1), float(b2), float(eps), float(wd),
                  bool(zero_grad), [int(s) for s, _ in ranges], [int(e) for _, e in ranges])

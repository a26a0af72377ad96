"""Model factory: the portable PyTorch definition or the native gfx950 engine."""
from __future__ import annotations

from typing import Sequence

import torch

from .densenet import (UNIT_NAMES, DenseNetStage, build_stage, default_split_points, densenet121,
                       stage_units)


def resolve_impl(impl: str, device) -> str:
    device = torch.device(device)
    if impl == "auto":
        return "native" if device.type == "cuda" else "torch"
    if impl == "native" and device.type != "cuda":
        raise ValueError("impl=native needs a GPU (the engine is HIP kernels); use impl=torch on CPU")
    return impl


def create_stage(units: Sequence[str], device, impl: str = "auto", num_classes: int = 5,
                 dtype: str = "fp32", pretrained: str = None, **native_kw) -> torch.nn.Module:
    """A contiguous run of DenseNet-121 units on ``device``.

    ``impl='torch'``: plain PyTorch ops (CPU plumbing path; numerics oracle).
    ``impl='native'``: the hand-written HIP engine (NHWC, flat arena, fused BN/ReLU/conv)."""
    impl = resolve_impl(impl, device)
    if impl == "torch":
        stage = DenseNetStage(units, num_classes).to(device)
    else:
        from .native_densenet import NativeDenseNetStage

        stage = NativeDenseNetStage(units, num_classes=num_classes, device=device, dtype=dtype, **native_kw)
    if pretrained:
        # reference: densenet121(weights=IMAGENET1K_V1) + a fresh Linear(1024, classes) (single.py:297-299)
        from .pretrained import load_pretrained

        load_pretrained(stage, pretrained)
    return stage

"""Model factory: the native gfx950 engine on a GPU, the portable PyTorch definition on the CPU."""
from __future__ import annotations

from typing import Sequence

import torch

from .densenet import (UNIT_NAMES, DenseNetStage, build_stage, default_split_points, densenet121,
                       stage_units)


def create_stage(units: Sequence[str], device, num_classes: int = 5, dtype: str = "fp32",
                 pretrained: str = None, **native_kw) -> torch.nn.Module:
    """A contiguous run of DenseNet-121 units on ``device``.

    The device decides the implementation - there is no user-selectable backend: on a GPU a stage IS the
    hand-written HIP engine (NHWC, flat arena, fused BN/ReLU/conv; ``NativeDenseNetStage``); on the CPU
    (gloo plumbing tests, checkpoint tooling) it is the plain-PyTorch definition, which is also the fp32
    numerics oracle the GPU tests instantiate directly (``DenseNetStage(...).to(device)``)."""
    device = torch.device(device)
    if device.type != "cuda":
        stage = DenseNetStage(units, num_classes).to(device)
    else:
        from .native_densenet import NativeDenseNetStage

        stage = NativeDenseNetStage(units, num_classes=num_classes, device=device, dtype=dtype, **native_kw)
    if pretrained:
        # reference: densenet121(weights=IMAGENET1K_V1) + a fresh Linear(1024, classes) (single.py:297-299)
        from .pretrained import load_pretrained

        load_pretrained(stage, pretrained)
    return stage

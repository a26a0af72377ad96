"""Loading torchvision DenseNet-121 weights from a local file.

The reference starts every run from ``models.densenet121(weights=IMAGENET1K_V1)`` and then
replaces the 1000-way classifier by ``nn.Linear(1024, 5)`` (single.py:297-299, repeated in all four
scripts).  There is no network here, so the weights come from a file (``--pretrained`` /
``DDL_PRETRAINED``): a torchvision state-dict saved with ``torch.save(model.state_dict(), path)``
or the ``densenet121-a639ec97.pth`` file of the torchvision hub cache.  The file is read with
``weights_only=True`` (nothing in it is executed).

Same semantics as the reference: every tensor whose name and shape match is copied; the
classifier is taken only if it has the right number of classes (the reference always discards the
ImageNet head).  Stages of a pipeline load just their own keys.  The native engine converts conv
weights to its own layout through ``ddl_import_tensor``.
"""
from __future__ import annotations

import re
from typing import Dict, List, Tuple

import torch

# torchvision checkpoints older than 0.3 name the dense-layer members 'norm.1', 'conv.2', ...
_LEGACY = re.compile(r"^(.*denselayer\d+\.(?:norm|relu|conv))\.((?:1|2)\.(?:weight|bias|running_mean|running_var|num_batches_tracked))$")


def normalise_keys(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in sd.items():
        if k.startswith("module."):            # a checkpoint written through DistributedDataParallel
            k = k[len("module."):]
        m = _LEGACY.match(k)
        if m:
            k = m.group(1) + m.group(2)
        out[k] = v
    return out


def read_state_dict(path: str) -> Dict[str, torch.Tensor]:
    obj = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(obj, dict) and "state_dict" in obj and isinstance(obj["state_dict"], dict):
        obj = obj["state_dict"]
    if not isinstance(obj, dict):
        raise ValueError(f"{path}: expected a state-dict")
    return normalise_keys({k: v for k, v in obj.items() if isinstance(v, torch.Tensor)})


@torch.no_grad()
def apply_state_dict(stage: torch.nn.Module, sd: Dict[str, torch.Tensor]) -> Tuple[List[str], List[str]]:
    """Copy matching tensors into ``stage`` (a ``DenseNetStage`` or the native engine's stage).
    Returns ``(loaded, skipped)`` key lists; ``skipped`` = in the stage but absent / mis-shaped in ``sd``."""
    conv = getattr(stage, "ddl_import_tensor", None)
    loaded, skipped = [], []
    targets = list(stage.named_parameters()) + list(stage.named_buffers())
    for name, t in targets:
        src = sd.get(name)
        if src is not None and conv is not None:
            src = conv(name, src)
        if src is None or tuple(src.shape) != tuple(t.shape):
            skipped.append(name)
            continue
        t.copy_(src.to(device=t.device, dtype=t.dtype))
        loaded.append(name)
    return loaded, skipped


def load_pretrained(stage: torch.nn.Module, path: str, verbose: bool = True) -> Tuple[List[str], List[str]]:
    loaded, skipped = apply_state_dict(stage, read_state_dict(path))
    if not loaded:
        raise ValueError(f"{path}: no tensor matched this model (not a DenseNet-121 state-dict?)")
    if verbose:
        head = [k for k in skipped if k.startswith("classifier.")]
        other = [k for k in skipped if not k.startswith("classifier.")]
        print(f"pretrained: {len(loaded)} tensors from {path}"
              + (f"; fresh classifier ({len(head)} tensors)" if head else "")
              + (f"; NOT found: {other[:4]}{'...' if len(other) > 4 else ''}" if other else ""), flush=True)
    return loaded, skipped

"""DenseNet-121 as an explicit, stage-able list of units (plain PyTorch ops).

This is the *portable* definition of the reference's model
(``torchvision.models.densenet121`` + ``nn.Linear(1024, 5)``, reference:
single.py:297-299): it runs on CPU for the plumbing configs (gloo tests) and is
the fp32 numerics oracle for the hand-written HIP engine in
``ddl.models.native_densenet``.  torchvision is not installed in this image, so the
architecture is spelled out here; parameter/buffer names are byte-identical to
torchvision's so checkpoints interoperate.

Instead of fx-tracing the model to cut pipeline stages (reference:
pp.py:380-386 ``pipeline(model, mb_args, split_spec)``), the model *is* a list of
units - ``stem, denseblock1, transition1, ..., denseblock4, head`` - and a stage is
a contiguous run of units.  Exactly one tensor crosses every unit boundary, which
is the property the reference found it needed (debug.py:9,20: splitting inside a
block or at block ENDs ships every skip tensor and was 3x slower).

A dense block may itself be cut between two of its layers: the unit ``denseblock3[8:24]`` is
layers 9..24 of block 3.  What crosses such a boundary is still ONE tensor - the concatenated
feature map of everything before it (in the native engine that is literally one NHWC buffer) -
so the fx tracer's "skip connections" problem (debug.py:9) does not arise, and pipeline stages
can be balanced at layer granularity (``models/split.py``), not just at the nine unit borders.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

GROWTH = 32
BN_SIZE = 4
BLOCK_CONFIG = (6, 12, 24, 16)
INIT_FEATURES = 64

UNIT_NAMES = ("stem", "denseblock1", "transition1", "denseblock2", "transition2",
              "denseblock3", "transition3", "denseblock4", "head")


def parse_unit(unit: str) -> Tuple[str, int, int]:
    """``(base, first layer, end layer)`` of a unit name; layers are only meaningful for dense blocks
    (``denseblock3`` = ``("denseblock3", 0, 24)``, ``denseblock3[8:24]`` = ``("denseblock3", 8, 24)``)."""
    base, l0, l1 = unit, 0, 0
    if "[" in unit:
        base, rng = unit[:-1].split("[")
        a, b = rng.split(":")
        l0, l1 = int(a), int(b)
    if base not in UNIT_NAMES:
        raise ValueError(f"unknown unit {unit!r}")
    if base.startswith("denseblock"):
        n = BLOCK_CONFIG[int(base[-1]) - 1]
        if "[" not in unit:
            l0, l1 = 0, n
        if not 0 <= l0 < l1 <= n:
            raise ValueError(f"bad layer range in {unit!r} (block has {n} layers)")
    elif "[" in unit:
        raise ValueError(f"only dense blocks have layer ranges: {unit!r}")
    return base, l0, l1


def unit_name(base: str, l0: int = 0, l1: int = 0) -> str:
    if base.startswith("denseblock") and (l0, l1) != (0, BLOCK_CONFIG[int(base[-1]) - 1]):
        return f"{base}[{l0}:{l1}]"
    return base


def model_atoms() -> List[Tuple[str, int]]:
    """The model as its smallest stage-able pieces, in order: ``(unit, layer)`` - one atom per dense layer, one
    per other unit (layer = 0).  A pipeline stage is a contiguous run of atoms."""
    out: List[Tuple[str, int]] = []
    for u in UNIT_NAMES:
        if u.startswith("denseblock"):
            out += [(u, l) for l in range(BLOCK_CONFIG[int(u[-1]) - 1])]
        else:
            out.append((u, 0))
    return out


def atoms_to_units(atoms: Sequence[Tuple[str, int]]) -> Tuple[str, ...]:
    """Merge a contiguous run of atoms into unit names (consecutive layers of a block become one unit)."""
    units: List[str] = []
    i = 0
    while i < len(atoms):
        base, l = atoms[i]
        j = i
        while j + 1 < len(atoms) and atoms[j + 1][0] == base and base.startswith("denseblock"):
            j += 1
        units.append(unit_name(base, l, atoms[j][1] + 1) if base.startswith("denseblock") else base)
        i = j + 1
    return tuple(units)


def unit_atoms(units: Sequence[str]) -> List[Tuple[str, int]]:
    out: List[Tuple[str, int]] = []
    for u in units:
        base, l0, l1 = parse_unit(u)
        out += [(base, l) for l in range(l0, l1)] if base.startswith("denseblock") else [(base, 0)]
    return out


def block_channels() -> List[Tuple[int, int]]:
    """(C_in, C_out) of each dense block."""
    out, c = [], INIT_FEATURES
    for i, n in enumerate(BLOCK_CONFIG):
        out.append((c, c + n * GROWTH))
        c = (c + n * GROWTH) // 2 if i < len(BLOCK_CONFIG) - 1 else c + n * GROWTH
    return out


class _DenseLayer(nn.Module):
    def __init__(self, c_in: int) -> None:
        super().__init__()
        self.norm1 = nn.BatchNorm2d(c_in)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(c_in, BN_SIZE * GROWTH, kernel_size=1, stride=1, bias=False)
        self.norm2 = nn.BatchNorm2d(BN_SIZE * GROWTH)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(BN_SIZE * GROWTH, GROWTH, kernel_size=3, stride=1, padding=1, bias=False)

    def forward(self, feats: List[torch.Tensor]) -> torch.Tensor:
        x = torch.cat(feats, 1)
        x = self.conv1(self.relu1(self.norm1(x)))
        return self.conv2(self.relu2(self.norm2(x)))


class _DenseBlock(nn.ModuleDict):
    """Layers ``first .. first+num_layers-1`` of a dense block (the whole block by default); ``c_in`` is the width
    of the block's input, the layer names keep their position in the full block."""

    def __init__(self, num_layers: int, c_in: int, first: int = 0) -> None:
        super().__init__()
        for i in range(first, first + num_layers):
            self[f"denselayer{i + 1}"] = _DenseLayer(c_in + i * GROWTH)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        feats = [x]
        for layer in self.values():
            feats.append(layer(feats))
        return torch.cat(feats, 1)


class _Transition(nn.Sequential):
    def __init__(self, c_in: int, c_out: int) -> None:
        super().__init__()
        self.norm = nn.BatchNorm2d(c_in)
        self.relu = nn.ReLU(inplace=True)
        self.conv = nn.Conv2d(c_in, c_out, kernel_size=1, stride=1, bias=False)
        self.pool = nn.AvgPool2d(kernel_size=2, stride=2)


def _init_weights(module: nn.Module) -> None:
    # torchvision's DenseNet init
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight)
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.Linear):
            nn.init.constant_(m.bias, 0)


class DenseNetStage(nn.Module):
    """A contiguous run of DenseNet-121 units.  ``units == UNIT_NAMES`` is the whole model.

    Parameter names are the torchvision FQNs (``features.denseblock3.denselayer1.conv1.weight``,
    ``classifier.bias`` ...), so the stages' state-dicts are a disjoint partition of the
    single-model key space.
    """

    def __init__(self, units: Sequence[str] = UNIT_NAMES, num_classes: int = 5) -> None:
        super().__init__()
        units = tuple(units)
        atoms, mine = model_atoms(), unit_atoms(units)
        start = atoms.index(mine[0]) if mine and mine[0] in atoms else -1
        if start < 0 or atoms[start:start + len(mine)] != mine or atoms_to_units(mine) != units:
            raise ValueError(f"units must be a contiguous run of {UNIT_NAMES} (dense blocks may be cut between "
                             f"layers: 'denseblock3[8:24]'), got {units}")
        self.units = units
        self.num_classes = num_classes
        chans = block_channels()
        feats = OrderedDict()
        for u in units:
            if u == "stem":
                feats["conv0"] = nn.Conv2d(3, INIT_FEATURES, kernel_size=7, stride=2, padding=3, bias=False)
                feats["norm0"] = nn.BatchNorm2d(INIT_FEATURES)
                feats["relu0"] = nn.ReLU(inplace=True)
                feats["pool0"] = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
            elif u.startswith("denseblock"):
                base, l0, l1 = parse_unit(u)
                b = int(base[-1]) - 1
                feats[base] = _DenseBlock(l1 - l0, chans[b][0], first=l0)
            elif u.startswith("transition"):
                b = int(u[-1]) - 1
                feats[u] = _Transition(chans[b][1], chans[b][1] // 2)
            elif u == "head":
                feats["norm5"] = nn.BatchNorm2d(chans[-1][1])
        self.features = nn.Sequential(feats)
        if "head" in units:
            self.classifier = nn.Linear(chans[-1][1], num_classes)
        _init_weights(self)

    @property
    def is_first(self) -> bool:
        return self.units[0] == "stem"

    @property
    def is_last(self) -> bool:
        return self.units[-1] == "head"

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dtype == torch.uint8:          # raw image batch: the reference's Normalize (single.py:38-42)
            x = x.float().div_(255.0)
        x = self.features(x)
        if self.is_last:
            x = F.relu(x, inplace=True)
            x = F.adaptive_avg_pool2d(x, (1, 1))
            x = torch.flatten(x, 1)
            x = self.classifier(x)
        return x

    # ---- shape algebra: lets a pipeline stage size its receive buffers without tracing ----
    @staticmethod
    def unit_output_shape(unit: str, in_shape: Tuple[int, ...]) -> Tuple[int, ...]:
        """NCHW output shape of one unit."""
        b = in_shape[0]
        chans = block_channels()
        if unit == "stem":
            h = (in_shape[2] + 2 * 3 - 7) // 2 + 1
            h = (h + 2 * 1 - 3) // 2 + 1
            return (b, INIT_FEATURES, h, h)
        if unit.startswith("denseblock"):
            base, _l0, l1 = parse_unit(unit)
            return (b, chans[int(base[-1]) - 1][0] + GROWTH * l1, in_shape[2], in_shape[3])
        if unit.startswith("transition"):
            return (b, chans[int(unit[-1]) - 1][1] // 2, in_shape[2] // 2, in_shape[3] // 2)
        if unit == "head":
            return (b, -1)
        raise ValueError(unit)

    def output_shape(self, in_shape: Tuple[int, ...]) -> Tuple[int, ...]:
        s = tuple(in_shape)
        for u in self.units:
            s = self.unit_output_shape(u, s)
        if self.is_last:
            s = (s[0], self.num_classes)
        return s


def densenet121(num_classes: int = 5) -> DenseNetStage:
    """Random-init DenseNet-121 with a ``num_classes`` head (6,958,981 params at 5 classes)."""
    return DenseNetStage(UNIT_NAMES, num_classes)


# --------------------------------------------------------------------------------------
# split specs
# --------------------------------------------------------------------------------------
def split_point_to_atom(name: str) -> Tuple[str, int]:
    """Map a reference-style split point - the module whose BEGINNING starts a new stage (pp.py:383-385,
    debug.py:36-38) - to the atom that starts the stage: a dense block, one of its layers
    (``features.denseblock3.denselayer9``), a transition, or the head (``features.norm5`` / ``classifier``)."""
    n = name
    if n.startswith("features."):
        n = n[len("features."):]
    layer = 0
    if ".denselayer" in n:
        n, lay = n.split(".denselayer", 1)
        if not lay.isdigit():
            raise ValueError(f"unsupported split point {name!r}: a stage may start at a dense LAYER, not inside one")
        layer = int(lay) - 1
    if n in ("norm5", "classifier"):
        n = "head"
    atom = (n, layer)
    if n in UNIT_NAMES and n != "stem" and atom in model_atoms() and atom != model_atoms()[0]:
        return atom
    raise ValueError(
        f"unsupported split point {name!r}: stages may start at a dense block, a dense layer, a transition or the "
        f"head (exactly one tensor crosses there)")


def split_point_to_unit(name: str) -> str:
    """Unit that contains the stage start named by a split point (kept for callers that think in units)."""
    return split_point_to_atom(name)[0]


def stage_units(split_points: Sequence[str]) -> List[Tuple[str, ...]]:
    atoms = model_atoms()
    starts = sorted({atoms.index(split_point_to_atom(s)) for s in split_points})
    bounds = [0] + starts + [len(atoms)]
    return [atoms_to_units(atoms[a:b]) for a, b in zip(bounds[:-1], bounds[1:])]


def default_split_points(num_stages: int) -> List[str]:
    """Reference splits: 2 stages at denseblock3 (pp.py:384); 4 stages at every block BEGINNING
    (debug.py:36-38, job pp-ghp6hmmmz9x2zc)."""
    table = {
        1: [],
        2: ["features.denseblock3.denselayer1"],
        3: ["features.denseblock2.denselayer1", "features.denseblock3.denselayer1"],
        4: ["features.denseblock2.denselayer1", "features.denseblock3.denselayer1",
            "features.denseblock4.denselayer1"],
    }
    if num_stages not in table:
        # up to 8 stages: also cut before transitions
        extra = ["features.transition1", "features.transition2", "features.transition3", "features.norm5"]
        pts = table[4] + extra[: num_stages - 4]
        if num_stages > 8:
            raise ValueError("DenseNet-121 has 9 units: at most 8 split points")
        return pts
    return table[num_stages]


def split_state_dict(full_sd: Dict[str, torch.Tensor], stage: nn.Module) -> Dict[str, torch.Tensor]:
    """The subset of a full-model state-dict that belongs to ``stage``."""
    keys = set(stage.state_dict().keys())
    return {k: v for k, v in full_sd.items() if k in keys}


def build_stage(stage_index: int, split_points: Sequence[str], num_classes: int = 5,
                full_state: Optional[Dict[str, torch.Tensor]] = None) -> DenseNetStage:
    units = stage_units(split_points)[stage_index]
    st = DenseNetStage(units, num_classes)
    if full_state is not None:
        st.load_state_dict(split_state_dict(full_state, st))
    return st

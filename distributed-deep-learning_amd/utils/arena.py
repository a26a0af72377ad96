"""Flat parameter / gradient arenas.

The reference leaves parameter storage to PyTorch (364 separate tensors) and lets
DDP's C++ Reducer copy gradients into 25 MiB buckets and back every step
(SURVEY 2.5 K15).  Here the parameters of a module (or pipeline stage) live in ONE
contiguous buffer and their gradients in a second one; every ``param.data`` /
``param.grad`` is a view.  Consequences:

  * the gradients *are* the communication buckets: a bucket is a contiguous arena
    range handed to RCCL reduce-scatter / all-gather with no copy-in / copy-out;
  * Adam is one kernel launch over the arena (``ddl.optim.FusedAdam``), and
    ``zero_grad`` is one memset;
  * 288 GB of HBM3E means no reason to be clever about sharing or freeing any of it.

Layout: parameters in the order given (callers pass *reverse registration order* so
that the first segment holds the last layers, whose gradients are ready first), each
parameter start aligned to ``align`` elements, cut into segments of about
``segment_numel`` elements, each segment padded to a multiple of ``segment_multiple``
(= dp_world * align, so a segment splits evenly for reduce-scatter).  ``break_before`` names
parameters that must open a new segment (the native engine aligns segments with the phases of
its phased backward graphs, so a phase's buckets are complete when its graph has run).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterable, List, Sequence, Tuple

import torch


@dataclass
class Slot:
    name: str
    offset: int
    numel: int
    shape: Tuple[int, ...]
    segment: int


@dataclass
class Segment:
    start: int
    end: int                     # padded end
    slots: List[int]             # indices into arena.slots


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class FlatArena:
    def __init__(self, named_params: Sequence[Tuple[str, torch.nn.Parameter]], device=None,
                 dtype=torch.float32, align: int = 64, segment_numel: int = 0,
                 segment_multiple: int = 64, break_before: Iterable[str] = ()) -> None:
        named_params = [(n, p) for n, p in named_params]
        if not named_params:
            raise ValueError("FlatArena needs at least one parameter")
        self.device = torch.device(device) if device is not None else named_params[0][1].device
        self.dtype = dtype
        self.align = align
        segment_multiple = _round_up(max(segment_multiple, align), align)
        break_before = set(break_before)      # parameter names that must start a new segment
        self.slots: List[Slot] = []
        self.segments: List[Segment] = []
        off = 0
        seg_start, seg_slots = 0, []
        for i, (name, p) in enumerate(named_params):
            n = p.numel()
            if seg_slots and ((segment_numel and (off + n - seg_start) > segment_numel) or name in break_before):
                end = _round_up(off, segment_multiple)
                self.segments.append(Segment(seg_start, end, seg_slots))
                off, seg_start, seg_slots = end, end, []
            self.slots.append(Slot(name, off, n, tuple(p.shape), len(self.segments)))
            seg_slots.append(i)
            off = _round_up(off + n, align)
        end = _round_up(off, segment_multiple)
        self.segments.append(Segment(seg_start, end, seg_slots))
        self.numel = end
        self.params: List[torch.nn.Parameter] = [p for _, p in named_params]
        self.names: List[str] = [n for n, _ in named_params]

        self.flat_param = torch.zeros(self.numel, dtype=dtype, device=self.device)
        self.flat_grad = torch.zeros(self.numel, dtype=dtype, device=self.device)
        with torch.no_grad():
            for s, p in zip(self.slots, self.params):
                view = self.flat_param[s.offset:s.offset + s.numel].view(s.shape)
                view.copy_(p.detach().to(self.device, dtype))
                p.data = view
        self.attach_grads()

    # ------------------------------------------------------------------
    def param_view(self, i: int) -> torch.Tensor:
        s = self.slots[i]
        return self.flat_param[s.offset:s.offset + s.numel].view(s.shape)

    def grad_view(self, i: int) -> torch.Tensor:
        s = self.slots[i]
        return self.flat_grad[s.offset:s.offset + s.numel].view(s.shape)

    def attach_grads(self) -> None:
        """(Re)bind every ``param.grad`` to its arena view; autograd then accumulates in place."""
        for i, p in enumerate(self.params):
            if p.requires_grad:
                p.grad = self.grad_view(i)

    def zero_grad(self) -> None:
        self.flat_grad.zero_()

    def segment_views(self, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        seg = self.segments[k]
        return self.flat_param[seg.start:seg.end], self.flat_grad[seg.start:seg.end]

    def slot_of(self, name: str) -> Slot:
        return self.slots[self.names.index(name)]

    def like(self, fill: float = 0.0) -> torch.Tensor:
        """A new flat buffer with this arena's layout (optimizer state)."""
        return torch.full((self.numel,), fill, dtype=self.dtype, device=self.device)

    def view_in(self, flat: torch.Tensor, i: int) -> torch.Tensor:
        s = self.slots[i]
        return flat[s.offset:s.offset + s.numel].view(s.shape)

    def total_param_numel(self) -> int:
        return sum(s.numel for s in self.slots)

    def __repr__(self) -> str:
        return (f"FlatArena({len(self.slots)} params, {self.total_param_numel()} elems in {self.numel} "
                f"({self.numel * 4 / 2**20:.1f} MiB fp32), {len(self.segments)} segments)")


def reverse_named_parameters(module: torch.nn.Module) -> List[Tuple[str, torch.nn.Parameter]]:
    """Trainable parameters in reverse registration order (~ the order backward produces grads)."""
    return [(n, p) for n, p in reversed(list(module.named_parameters())) if p.requires_grad]


class FlatBuffers:
    """Float buffers (BN running_mean / running_var) of a module in one flat tensor, plus the
    int64 ``num_batches_tracked`` counters in another - so that DDP's per-forward buffer
    broadcast (121 BN layers, 0.34 MB) is ONE collective on one range."""

    def __init__(self, module: torch.nn.Module, device=None) -> None:
        f, i = [], []
        for name, b in module.named_buffers():
            (f if b.dtype.is_floating_point else i).append((name, b))
        self.float_names = [n for n, _ in f]
        self.int_names = [n for n, _ in i]
        dev = device if device is not None else (f[0][1].device if f else torch.device("cpu"))
        self.flat_float = torch.zeros(sum(b.numel() for _, b in f), dtype=torch.float32, device=dev)
        self.flat_int = torch.zeros(sum(b.numel() for _, b in i), dtype=torch.int64, device=dev)
        off = 0
        self.float_slots = {}            # name -> (offset, numel, shape) inside flat_float
        with torch.no_grad():
            for n, b in f:
                v = self.flat_float[off:off + b.numel()].view(b.shape)
                v.copy_(b)
                b.data = v
                self.float_slots[n] = (off, b.numel(), tuple(b.shape))
                off += b.numel()
            off = 0
            for _, b in i:
                v = self.flat_int[off:off + b.numel()].view(b.shape)
                v.copy_(b)
                b.data = v
                off += b.numel()

"""Balanced pipeline split points (``--split-points auto``).

The reference hard-codes its 2-stage split at ``features.denseblock3.denselayer1`` (pp.py:380-386) and tried
others by hand in debug.py:20-40 (4 stages at block BEGINNINGs: 115 s/epoch, at block ENDs: 334 s/epoch).  With
that split stage 0 carries 73 % of the multiply-accumulates, so the pipeline runs at the speed of
its first stage.  Here the stage starts are chosen by an exact search over the unit boundaries (stages are
contiguous runs of the 9 units and exactly one tensor crosses a boundary - at most C(8, S-1) = 70 candidates)
that minimises the slowest stage's cost, where a unit's cost is

  * on a GPU: its MEASURED forward + backward time at the micro-batch size of the run.  Every unit forward and
    every unit backward is captured as its own hipGraph (what a stage really replays - eager timings would
    measure Python launch overhead, i.e. the number of kernels, not the kernels) and timed with events over a
    few replays; the vector is averaged over all ranks so that every rank derives the same cut;
  * on the CPU (gloo plumbing tests): the analytic multiply-accumulate count of the unit.
"""
from __future__ import annotations

from itertools import combinations
from typing import List, Sequence, Tuple

import torch

from .densenet import BLOCK_CONFIG, BN_SIZE, GROWTH, INIT_FEATURES, UNIT_NAMES, block_channels

# the reference-style name of the module whose BEGINNING starts a stage at unit i (pp.py:383-385)
_SPLIT_NAME = {u: (f"features.{u}.denselayer1" if u.startswith("denseblock") else
                   "features.norm5" if u == "head" else f"features.{u}") for u in UNIT_NAMES}


def analytic_unit_macs(image_size: int = 224) -> List[float]:
    """Forward multiply-accumulates per image of every unit (computed from the shapes)."""
    out = []
    h = (image_size + 2 * 3 - 7) // 2 + 1                 # conv0 output
    macs_stem = h * h * 64 * 147
    h = (h + 2 - 3) // 2 + 1                              # max-pool
    chans = block_channels()
    for u in UNIT_NAMES:
        if u == "stem":
            out.append(float(macs_stem))
        elif u.startswith("denseblock"):
            b = int(u[-1]) - 1
            m = 0
            for l in range(BLOCK_CONFIG[b]):
                c = chans[b][0] + GROWTH * l
                m += h * h * (c * BN_SIZE * GROWTH + 9 * BN_SIZE * GROWTH * GROWTH)
            out.append(float(m))
        elif u.startswith("transition"):
            c = chans[int(u[-1]) - 1][1]
            out.append(float(h * h * c * (c // 2)))
            h //= 2
        else:
            out.append(float(chans[-1][1] * 5 + h * h * chans[-1][1]))
    assert len(out) == len(UNIT_NAMES) and INIT_FEATURES == 64
    return out


def balanced_starts(costs: Sequence[float], num_stages: int) -> Tuple[int, ...]:
    """Unit indices that start stages 1..S-1: the partition of ``costs`` into ``num_stages`` contiguous runs with the
    smallest maximum (ties: the smallest sum of squares, then the earliest cuts - deterministic on every rank)."""
    n = len(costs)
    if not 1 <= num_stages <= n:
        raise ValueError(f"{n} units cannot make {num_stages} stages")
    best, best_key = (), None
    for cuts in combinations(range(1, n), num_stages - 1):
        bounds = (0,) + cuts + (n,)
        loads = [sum(costs[a:b]) for a, b in zip(bounds[:-1], bounds[1:])]
        key = (round(max(loads), 9), round(sum(x * x for x in loads), 9), cuts)
        if best_key is None or key < best_key:
            best, best_key = cuts, key
    return best


def starts_to_split_points(starts: Sequence[int]) -> List[str]:
    return [_SPLIT_NAME[UNIT_NAMES[i]] for i in starts]


def stage_loads(costs: Sequence[float], starts: Sequence[int]) -> List[float]:
    bounds = [0] + list(starts) + [len(costs)]
    return [sum(costs[a:b]) for a, b in zip(bounds[:-1], bounds[1:])]


@torch.no_grad()
def _rand_images(mb: int, size: int, device) -> torch.Tensor:
    return torch.randint(0, 256, (mb, 3, size, size), dtype=torch.uint8, device=device)


def measure_unit_times(device, micro_batch: int, image_size: int = 224, dtype: str = "fp32",
                       iters: int = 5, use_graphs: bool = True) -> List[float]:
    """Forward + backward milliseconds of every unit of the native engine at ``micro_batch`` images, each unit
    pass replayed as its own hipGraph (see the module docstring).  The model is a throw-away instance."""
    from .. import ops
    from .native_densenet import NativeDenseNetStage

    dev = torch.device(device)
    model = NativeDenseNetStage(UNIT_NAMES, device=dev, dtype=dtype)
    model.train()
    x = _rand_images(micro_batch, image_size, dev)
    y = torch.randint(0, 5, (micro_batch,), device=dev)
    n = len(model.unit_objs)

    def fwd_unit(i, inp, stats):
        return model.unit_objs[i](inp, stats)

    # one eager pass: allocator growth, hipFuncSetAttribute calls, lazy module state
    ops.cross_entropy(model(x), y).backward()
    torch.cuda.synchronize(dev)

    fwd, bwd = [None] * n, [None] * n
    leaves, outs = [None] * n, [None] * n
    pool = torch.cuda.graph_pool_handle() if use_graphs else None
    mode = dict(pool=pool, capture_error_mode="thread_local")

    def run(fn, slot, i):
        if not use_graphs:
            slot[i] = fn
            fn()
            return
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, **mode):
            fn()
        slot[i] = g.replay

    state = {"stats": None}
    inp = x
    for i in range(n):
        leaf = inp.detach()
        if leaf.is_floating_point():
            leaf.requires_grad_(True)
        leaves[i] = leaf

        def f(i=i, leaf=leaf):
            with torch.enable_grad():
                outs[i], state["stats"] = fwd_unit(i, leaf, state["stats"])
        run(f, fwd, i)
        inp = outs[i]

    def b_last():
        with torch.enable_grad():
            loss = ops.cross_entropy(outs[n - 1], y)
        loss.backward()
    run(b_last, bwd, n - 1)
    for i in range(n - 2, -1, -1):
        def b(i=i):
            outs[i].backward(leaves[i + 1].grad)
        run(b, bwd, i)
    torch.cuda.synchronize(dev)

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2 * n + 2)] for _ in range(iters)]
    for it in range(iters + 1):                   # replay 0 is a warm-up of the graphs
        e = ev[max(it - 1, 0)]
        k = 0
        for i in range(n):
            e[k].record()
            fwd[i]()
            k += 1
        e[k].record()
        k += 1
        for i in range(n - 1, -1, -1):
            e[k].record()
            bwd[i]()
            k += 1
        e[k].record()
    torch.cuda.synchronize(dev)
    times = [0.0] * n
    for e in ev:
        for i in range(n):
            times[i] += e[i].elapsed_time(e[i + 1])
        for j, i in enumerate(range(n - 1, -1, -1)):
            times[i] += e[n + 1 + j].elapsed_time(e[n + 2 + j])
    del fwd, bwd, leaves, outs, model
    return [t / iters for t in times]


def auto_split_points(num_stages: int, device, micro_batch: int, image_size: int = 224, dtype: str = "fp32",
                      group=None) -> Tuple[List[str], List[float]]:
    """``(split point names, per-unit costs)`` - identical on every rank of ``group`` (costs are averaged)."""
    from .. import comm

    dev = torch.device(device)
    if dev.type == "cuda":
        costs = measure_unit_times(dev, micro_batch, image_size, dtype)
    else:
        costs = [m * 1e-9 for m in analytic_unit_macs(image_size)]
    if comm.is_dist() and comm.get_world_size(group) > 1:
        t = torch.tensor(costs, dtype=torch.float64, device=dev)
        comm.all_reduce(t, "avg", group)
        costs = [float(v) for v in t.tolist()]
    starts = balanced_starts(costs, num_stages)
    return starts_to_split_points(starts), costs

"""Balanced pipeline split points (``--split-points auto``).

The reference hard-codes its 2-stage split at ``features.denseblock3.denselayer1`` (pp.py:380-386) and tried
others by hand in debug.py:20-40 (4 stages at block BEGINNINGs: 115 s/epoch, at block ENDs: 334 s/epoch).  With
that split stage 0 carries 73 % of the multiply-accumulates, so the pipeline runs at the speed of
its first stage.  Here the stage starts are chosen by an exact linear-partition search over the model's ATOMS -
every dense layer, transition, the stem and the head (63 pieces; a dense block may be cut between two layers,
one tensor still crosses: ``models/densenet.py``) - that minimises the slowest stage's cost.  A unit's cost is

  * on a GPU: its MEASURED forward + backward time at the micro-batch size of the run.  Every unit forward and
    every unit backward is captured as its own hipGraph (what a stage really replays - eager timings would
    measure Python launch overhead, i.e. the number of kernels, not the kernels) and timed with events over a
    few replays; the vector is averaged over all ranks so that every rank derives the same cut;
  * on the CPU (gloo plumbing tests): the analytic multiply-accumulate count of the unit.
A dense block's cost is spread over its layers in proportion to their multiply-accumulates plus a per-layer
constant (the small feature maps are latency-bound: every layer costs about the same there).
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import torch

from .densenet import BLOCK_CONFIG, BN_SIZE, GROWTH, INIT_FEATURES, UNIT_NAMES, block_channels, model_atoms


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


def layer_macs(image_size: int = 224) -> dict:
    """``{block: [forward multiply-accumulates per image of each layer]}``."""
    h = ((image_size + 2 * 3 - 7) // 2 + 1 + 2 - 3) // 2 + 1
    chans, out = block_channels(), {}
    for b, n in enumerate(BLOCK_CONFIG):
        out[f"denseblock{b + 1}"] = [float(h * h * ((chans[b][0] + GROWTH * l) * BN_SIZE * GROWTH + 9 * BN_SIZE * GROWTH * GROWTH))
                                     for l in range(n)]
        h //= 2
    return out


def atom_costs(unit_costs: Sequence[float], image_size: int = 224, latency_share: float = 0.5) -> List[float]:
    """Per-atom costs from per-unit costs: a block's cost goes to its layers, ``latency_share`` of it evenly (launch
    / latency floor of the small layers), the rest in proportion to the layers' multiply-accumulates."""
    lm = layer_macs(image_size)
    per_unit = dict(zip(UNIT_NAMES, unit_costs))
    out = []
    for u, l in model_atoms():
        if u.startswith("denseblock"):
            macs, n = lm[u], len(lm[u])
            out.append(per_unit[u] * (latency_share / n + (1.0 - latency_share) * macs[l] / sum(macs)))
        else:
            out.append(per_unit[u])
    return out


def balanced_starts(costs: Sequence[float], num_stages: int) -> Tuple[int, ...]:
    """Indices that start stages 1..S-1: the partition of ``costs`` into ``num_stages`` contiguous runs with the
    smallest maximum (classic linear-partition DP; ties resolve to the earliest cuts - deterministic on every rank)."""
    n = len(costs)
    if not 1 <= num_stages <= n:
        raise ValueError(f"{n} pieces cannot make {num_stages} stages")
    pre = [0.0]
    for c in costs:
        pre.append(pre[-1] + float(c))
    INF = float("inf")
    # best[k][i]: smallest max load splitting the first i pieces into k runs; cut[k][i]: start of the last run
    best = [[INF] * (n + 1) for _ in range(num_stages + 1)]
    cut = [[0] * (n + 1) for _ in range(num_stages + 1)]
    best[0][0] = 0.0
    for k in range(1, num_stages + 1):
        for i in range(k, n + 1):
            for j in range(k - 1, i):
                v = max(best[k - 1][j], pre[i] - pre[j])
                if v < best[k][i] - 1e-12:
                    best[k][i], cut[k][i] = v, j
    starts, i = [], n
    for k in range(num_stages, 1, -1):
        i = cut[k][i]
        starts.append(i)
    return tuple(reversed(starts))


def starts_to_split_points(starts: Sequence[int]) -> List[str]:
    """Atom indices -> the reference's split-point names (module whose BEGINNING starts the stage)."""
    atoms = model_atoms()
    out = []
    for i in starts:
        u, l = atoms[i]
        out.append(f"features.{u}.denselayer{l + 1}" if u.startswith("denseblock") else
                   "features.norm5" if u == "head" else f"features.{u}")
    return out


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


def measure_stage_times(split_points: Sequence[str], device, micro_batch: int, image_size: int = 224,
                        dtype: str = "fp32", iters: int = 5) -> List[float]:
    """Forward + backward milliseconds of every pipeline stage a split produces (native engine, one hipGraph per
    stage pass, ``micro_batch`` images) - what the balance tests and ``debug.py --measure`` report."""
    from .densenet import DenseNetStage, stage_units
    from .native_densenet import NativeDenseNetStage

    dev = torch.device(device)
    shape = (micro_batch, 3, image_size, image_size)
    out = []
    for units in stage_units(split_points):
        if os.environ.get("DDL_SPLIT_DEBUG"):
            print(f"[split] timing stage {units}", flush=True)
        st = NativeDenseNetStage(units, device=dev, dtype=dtype)
        st.train()
        if st.is_first:
            x = _rand_images(micro_batch, image_size, dev)
        else:
            x = torch.randn(shape[0], shape[2], shape[3], shape[1], device=dev)                          # NHWC

        def fwd_bwd(gy=None):
            # a fresh leaf per pass: an AccumulateGrad node lives on the stream that created it, and one made
            # by the eager pass (default stream) would pull the default stream into the capture
            leaf = x if st.is_first else x.detach().requires_grad_(True)
            y = st(leaf)
            y.backward(torch.ones_like(y) if gy is None else gy)
            return y

        gy = torch.randn_like(fwd_bwd())                 # eager pass: allocator, attributes
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            fwd_bwd(gy)
            st.assert_joined()
        g.replay()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            g.replay()
        e1.record()
        torch.cuda.synchronize(dev)
        out.append(e0.elapsed_time(e1) / iters)
        for u in units:
            shape = DenseNetStage.unit_output_shape(u, shape)
        del g, st
    return out


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
    starts = balanced_starts(atom_costs(costs, image_size), num_stages)
    return starts_to_split_points(starts), costs

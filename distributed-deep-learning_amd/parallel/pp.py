"""Pipeline parallelism: explicit stages, GPipe and 1F1B micro-batch schedules.

Reference (pp.py:138-150,171-201; torch.distributed.pipelining): the model is fx-traced and
split, ``build_stage`` infers receive shapes from trace metadata, ``ScheduleGPipe`` runs all
forwards then all backwards, and a second loss-less schedule exists for evaluation.  Call
contract kept here (hard part 6):

    first stage :  schedule.step(x)
    last stage  :  out = schedule.step(target=y, losses=losses)   # merged [B, C] logits
    middle      :  schedule.step()

What is different, MI355X-first:

  * stages are explicit modules (``DenseNetStage`` / the native engine's stages) - no tracing;
    receive shapes come from the model's shape algebra (``recv_shape_fn``);
  * **1F1B** is the default schedule (BASELINE.json): at most ``S - stage`` micro-batches of
    activations are alive instead of all ``M``; GPipe is kept for parity runs;
  * P2P is RCCL send/recv issued as *batched groups* per peer (forward-send fused with
    backward-recv toward the next stage, backward-send fused with forward-recv toward the
    previous one) - opposite-direction transfers between one pair must be one RCCL group or
    they deadlock (SURVEY hard part 5); receives land in per-micro-batch pre-allocated slots;
  * gradients are scaled by 1/M (torch 2.6, which the reference pins, summed un-scaled
    per-micro-batch mean-loss gradients = M x the batch-mean gradient; documented deviation);
  * with a ``DataParallelEngine`` attached (hybrid), micro-batches 0..M-2 run under
    ``no_sync`` and the reducer is armed for the last backward only - one gradient sync per
    step per stage, overlapped with that backward.

BatchNorm statistics are per micro-batch, exactly as in the reference.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .. import comm
from ..utils.trace import trace_range


def split_sizes(batch: int, n: int) -> List[int]:
    """``torch.tensor_split`` chunk sizes (reference splits with tensor_split on dim 0)."""
    base, rem = divmod(batch, n)
    return [base + (1 if i < rem else 0) for i in range(n)]


class _GraphNode:
    __slots__ = ("graph", "x", "t", "inp", "out", "loss")

    def __init__(self) -> None:
        self.graph = self.x = self.t = self.inp = self.out = self.loss = None


class _GraphSet:
    """The hipGraphs of one step shape: one per micro-batch forward and one per micro-batch backward,
    captured in schedule order into one private memory pool and replayed in that same order (the
    schedule is static, so the order never changes).  The stage's compute is ~770 launches per
    micro-batch forward+backward of a few microseconds each - eager, a 5-micro-batch step is bound
    by the host; captured, it is 2 launches per micro-batch.  P2P transfers stay outside the graphs
    and move the graphs' static input / output buffers."""

    def __init__(self, warmup: int) -> None:
        self.warmup = warmup
        self.steps = 0
        self.active = False
        self.fwd, self.bwd = {}, {}
        self._pool = None

    def begin_step(self) -> None:
        self.active = self.steps >= self.warmup
        self.steps += 1

    def pool(self):
        if self._pool is None:
            self._pool = torch.cuda.graph_pool_handle()
        return self._pool


class PipelineStage:
    """One stage of the pipeline on this rank.

    ``recv_shape_fn(mb)`` -> shape of the activation this stage receives for a micro-batch of
    ``mb`` samples; ``send`` shapes are whatever the module produces.
    """

    def __init__(self, module: torch.nn.Module, stage_index: int, num_stages: int, device,
                 group=None, ranks: Optional[Sequence[int]] = None,
                 recv_shape_fn: Optional[Callable[[int], Tuple[int, ...]]] = None,
                 dtype: torch.dtype = torch.float32, dp_engine=None, hip_graph: bool = False,
                 graph_warmup: int = 2) -> None:
        self.module = module
        self.stage_index = stage_index
        self.num_stages = num_stages
        self.device = torch.device(device)
        self.group = group
        self.ranks = list(ranks) if ranks is not None else list(range(num_stages))
        self.is_first = stage_index == 0
        self.is_last = stage_index == num_stages - 1
        self.prev_rank = None if self.is_first else self.ranks[stage_index - 1]
        self.next_rank = None if self.is_last else self.ranks[stage_index + 1]
        self.recv_shape_fn = recv_shape_fn
        self.dtype = dtype
        self.dp_engine = dp_engine
        self._act_slots = {}      # (mb_index, shape) -> recv buffer for forward activations
        self._grad_slots = {}     # (mb_index, shape) -> recv buffer for backward gradients
        # hipGraph capture of the per-micro-batch forward / backward (see ``_GraphSet``)
        self.hip_graph = bool(hip_graph) and self.device.type == "cuda"
        self.graph_warmup = graph_warmup
        self._graph_sets = {}     # signature of a step (micro-batch shapes, train/eval) -> _GraphSet
        self._gs: Optional[_GraphSet] = None
        self.reset()

    def reset(self) -> None:
        self.inputs: dict = {}
        self.outputs: dict = {}
        self.grads_in: dict = {}
        self.losses: dict = {}

    # ---- P2P op builders -----------------------------------------------------------------
    def _slot(self, table: dict, i: int, shape) -> torch.Tensor:
        key = (i, tuple(shape))
        buf = table.get(key)
        if buf is None:
            buf = torch.empty(tuple(shape), dtype=self.dtype, device=self.device)
            table[key] = buf
        return buf

    def fwd_recv_ops(self, i: int, mb: int) -> List[dist.P2POp]:
        if self.is_first:
            return []
        buf = self._slot(self._act_slots, i, self.recv_shape_fn(mb))
        self.inputs[i] = buf
        return [dist.P2POp(dist.irecv, buf, self.prev_rank, self.group)]

    def fwd_send_ops(self, i: int) -> List[dist.P2POp]:
        if self.is_last:
            return []
        return [dist.P2POp(dist.isend, self.outputs[i].detach(), self.next_rank, self.group)]

    def bwd_recv_ops(self, i: int) -> List[dist.P2POp]:
        if self.is_last:
            return []
        buf = self._slot(self._grad_slots, i, self.outputs[i].shape)
        self.grads_in[i] = buf
        return [dist.P2POp(dist.irecv, buf, self.next_rank, self.group)]

    def bwd_send_ops(self, i: int) -> List[dist.P2POp]:
        if self.is_first:
            return []
        g = self.inputs.pop(i)
        return [dist.P2POp(dist.isend, g, self.prev_rank, self.group)]

    # ---- hipGraph plumbing ---------------------------------------------------------------
    def begin_step(self, signature) -> None:
        """Called by the schedule before the first micro-batch: selects the graph set of this step
        shape.  The first ``graph_warmup`` steps of a shape run eagerly, the next one captures every
        forward / backward in schedule order (capture + replay in place), later steps only replay."""
        self._gs = None
        if not self.hip_graph:
            return
        gs = self._graph_sets.get(signature)
        if gs is None:
            gs = self._graph_sets[signature] = _GraphSet(self.graph_warmup)
        gs.begin_step()
        self._gs = gs if gs.active else None

    def forward(self, i: int, x: Optional[torch.Tensor], target: Optional[torch.Tensor],
                loss_fn, train: bool) -> None:
        with trace_range(f"Forward {i}"):          # the library's range names (schedules.py:757)
            self._forward(i, x, target, loss_fn, train)

    def backward(self, i: int, scale: float, last: bool) -> None:
        with trace_range(f"Backward {i}"):         # schedules.py:785
            self._backward(i, scale, last)

    def _forward(self, i: int, x: Optional[torch.Tensor], target: Optional[torch.Tensor],
                 loss_fn, train: bool) -> None:
        gs = self._gs
        if gs is None:
            return self._forward_eager(i, x, target, loss_fn, train)
        node = gs.fwd.get(i)
        if node is None:
            node = gs.fwd[i] = _GraphNode()
            node.x = x.clone() if x is not None else None
            node.t = target.clone() if target is not None else None
            inp_slot = None if self.is_first else self.inputs[i]     # persistent receive slot
            node.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(node.graph, pool=gs.pool(), capture_error_mode="thread_local"):
                if not self.is_first:
                    self.inputs[i] = inp_slot
                self._forward_eager(i, node.x, node.t, loss_fn, train)
            node.inp, node.out, node.loss = self.inputs.get(i), self.outputs[i], self.losses.get(i)
        else:
            if node.x is not None:
                node.x.copy_(x, non_blocking=True)
            if node.t is not None:
                node.t.copy_(target, non_blocking=True)
        node.graph.replay()
        if not self.is_first:
            self.inputs[i] = node.inp
        self.outputs[i] = node.out
        if node.loss is not None:
            self.losses[i] = node.loss

    def _backward(self, i: int, scale: float, last: bool) -> None:
        gs = self._gs
        if gs is None:
            return self._backward_eager(i, scale, last)
        dp = self.dp_engine
        node = gs.bwd.get(i)
        if node is None:
            node = gs.bwd[i] = _GraphNode()
            if dp is not None:
                dp.arm(False)                      # no collectives inside a capture
            grad_slot = self.grads_in.get(i)
            node.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(node.graph, pool=gs.pool(), capture_error_mode="thread_local"):
                if grad_slot is not None:
                    self.grads_in[i] = grad_slot
                self._backward_eager(i, scale, False, arm=False)
            node.inp = self.inputs.get(i)          # gradient w.r.t. the stage input (sent upstream)
            node.out = self.outputs.get(i)
        else:
            self.grads_in.pop(i, None)
            if self.is_last:
                self.outputs[i] = node.out
            else:
                self.outputs.pop(i, None)
        node.graph.replay()
        if not self.is_first:
            self.inputs[i] = node.inp
        else:
            self.inputs.pop(i, None)
        if dp is not None and last:
            dp.arm(True)                           # buckets are launched by finish_grad_sync()

    # ---- compute -------------------------------------------------------------------------
    def _forward_eager(self, i: int, x: Optional[torch.Tensor], target: Optional[torch.Tensor],
                       loss_fn, train: bool) -> None:
        if self.is_first:
            inp = x
        else:
            inp = self.inputs[i].detach()
            if train:
                inp.requires_grad_(True)
            self.inputs[i] = inp
        out = self.module(inp)
        self.outputs[i] = out
        if self.is_last and loss_fn is not None:
            self.losses[i] = loss_fn(out, target)

    def _backward_eager(self, i: int, scale: float, last: bool, arm: bool = True) -> None:
        dp = self.dp_engine
        if dp is not None and arm:
            dp.arm(last)
        if self.is_last:
            loss = self.losses[i]
            (loss * scale).backward()
            self.outputs[i] = self.outputs[i].detach()
        else:
            # the 1/M factor entered at the loss on the last stage and rides the chain rule back
            torch.autograd.backward(self.outputs.pop(i), grad_tensors=self.grads_in.pop(i))
        if not self.is_first:
            inp = self.inputs[i]
            self.inputs[i] = inp.grad
            inp.grad = None
        else:
            self.inputs.pop(i, None)


class _StagedRecv:
    """gloo (test transport) cannot move GPU tensors: receive into host memory, copy up on wait."""

    def __init__(self, work, host: torch.Tensor, dev: torch.Tensor) -> None:
        self.work, self.host, self.dev = work, host, dev

    def wait(self) -> None:
        self.work.wait()
        self.dev.copy_(self.host)


def _batch_p2p(ops: List[dist.P2POp]):
    if not ops:
        return []
    if ops[0].tensor.is_cuda and comm.backend_of(ops[0].group) == "gloo":
        works = []
        for op in ops:
            if op.op is dist.isend:
                works.append(dist.isend(op.tensor.detach().cpu(), op.peer, group=op.group))
            else:
                host = torch.empty(op.tensor.shape, dtype=op.tensor.dtype)
                works.append(_StagedRecv(dist.irecv(host, op.peer, group=op.group), host, op.tensor))
        return works
    return dist.batch_isend_irecv(ops)


def _wait(works) -> None:
    for w in works:
        w.wait()


class _ScheduleBase:
    def __init__(self, stage: PipelineStage, n_microbatches: int, loss_fn=None, profiler=None,
                 scale_grads: bool = True) -> None:
        self.stage = stage
        self.n = n_microbatches
        self.loss_fn = loss_fn
        self.profiler = profiler
        self.scale_grads = scale_grads
        self.order: List[str] = []          # executed F/B sequence of the last step (tests, tracing)
        self._last_batch = None

    # ---- public API (reference contract) -------------------------------------------------
    def step(self, x: Optional[torch.Tensor] = None, *, target: Optional[torch.Tensor] = None,
             losses: Optional[list] = None, batch_size: Optional[int] = None):
        st = self.stage
        if st.is_first and x is None:
            raise ValueError("first stage needs the input batch: schedule.step(x)")
        if batch_size is None:
            if x is not None:
                batch_size = x.shape[0]
            elif target is not None:
                batch_size = target.shape[0]
            else:
                batch_size = self._last_batch
        if batch_size is None:
            raise ValueError("a middle stage needs batch_size= on its first step")
        self._last_batch = batch_size
        sizes = [s for s in split_sizes(batch_size, self.n)]
        if any(s == 0 for s in sizes):
            raise ValueError(f"batch of {batch_size} cannot feed {self.n} micro-batches")
        xs = list(torch.tensor_split(x, self.n)) if (x is not None and st.is_first) else [None] * self.n
        ts = list(torch.tensor_split(target, self.n)) if target is not None else [None] * self.n
        train = self.loss_fn is not None and torch.is_grad_enabled()
        if self.loss_fn is not None and not torch.is_grad_enabled():
            # same guard as torch's schedule (schedules.py:624-629): that is why the reference
            # builds a second, loss-less schedule for evaluation (pp.py:146-150)
            raise RuntimeError("a schedule with a loss_fn cannot run under no_grad; "
                               "use a loss-less schedule for inference")
        st.reset()
        st.begin_step((tuple(sizes), train, self.loss_fn is not None,
                       tuple(x.shape[1:]) if x is not None else None))
        self.order = []
        self._run(xs, ts, sizes, train)
        out = None
        if st.is_last:
            out = torch.cat([st.outputs[i].detach() for i in range(self.n)], dim=0)
            if losses is not None and st.losses:
                losses.clear()
                losses.extend(st.losses[i].detach() for i in range(self.n))
        st.reset()
        return out

    # ---- helpers -----------------------------------------------------------------------
    def _F(self, i, xs, ts, train):
        self.order.append(f"F{i}")
        tok = self.profiler.begin("compute") if self.profiler else None
        self.stage.forward(i, xs[i], ts[i], self.loss_fn if train else None, train)
        if tok is not None:
            self.profiler.end(tok)

    def _B(self, i):
        self.order.append(f"B{i}")
        tok = self.profiler.begin("compute") if self.profiler else None
        scale = 1.0 / self.n if self.scale_grads else 1.0
        self.stage.backward(i, scale, last=(i == self.n - 1))
        if tok is not None:
            self.profiler.end(tok)

    def _comm(self, ops, wait: bool = True):
        if not ops:
            return []
        tok = self.profiler.begin("p2p_wait") if (self.profiler and wait) else None
        works = _batch_p2p(ops)
        if wait:
            _wait(works)
        if tok is not None:
            self.profiler.end(tok)
        return works

    def _run(self, xs, ts, sizes, train):
        raise NotImplementedError


class ScheduleGPipe(_ScheduleBase):
    """Fill-drain: F0..F(M-1) then B0..B(M-1) (the reference's schedule, pp.py:140-144)."""

    def _run(self, xs, ts, sizes, train):
        st = self.stage
        sends = []
        for i in range(self.n):
            self._comm(st.fwd_recv_ops(i, sizes[i]))
            self._F(i, xs, ts, train)
            sends += self._comm(st.fwd_send_ops(i), wait=False)
        _wait(sends)
        if not train:
            return
        sends = []
        for i in range(self.n):
            self._comm(st.bwd_recv_ops(i))
            self._B(i)
            sends += self._comm(st.bwd_send_ops(i), wait=False)
        _wait(sends)


class Schedule1F1B(_ScheduleBase):
    """One-forward-one-backward: ``min(M, S - stage)`` warm-up forwards, then alternate B/F,
    then drain.  Steady-state transfers toward one peer are fused into one RCCL group."""

    def _run(self, xs, ts, sizes, train):
        st = self.stage
        n = self.n
        if not train:
            sends = []
            for i in range(n):
                self._comm(st.fwd_recv_ops(i, sizes[i]))
                self._F(i, xs, ts, False)
                sends += self._comm(st.fwd_send_ops(i), wait=False)
            _wait(sends)
            return
        warmup = min(n, st.num_stages - st.stage_index)
        f = b = 0
        pending_send = []
        fwd_sends: List[dist.P2POp] = []
        for _ in range(warmup):
            self._comm(st.fwd_recv_ops(f, sizes[f]))
            self._F(f, xs, ts, True)
            _wait(pending_send)
            fwd_sends = st.fwd_send_ops(f)
            if f != warmup - 1:
                pending_send = self._comm(fwd_sends, wait=False)
                fwd_sends = []
            f += 1
        # steady state: 1B1F
        while True:
            # forward-send of the newest activation fused with the backward-recv (same peer: next)
            self._comm(fwd_sends + st.bwd_recv_ops(b))
            self._B(b)
            bwd_sends = st.bwd_send_ops(b)
            b += 1
            if f == n:
                break
            # backward-send fused with the next forward-recv (same peer: prev)
            self._comm(bwd_sends + st.fwd_recv_ops(f, sizes[f]))
            self._F(f, xs, ts, True)
            fwd_sends = st.fwd_send_ops(f)
            f += 1
        # cooldown: drain the remaining backwards
        pending_send = self._comm(bwd_sends, wait=False)
        while b < n:
            self._comm(st.bwd_recv_ops(b))
            self._B(b)
            _wait(pending_send)
            pending_send = self._comm(st.bwd_send_ops(b), wait=False)
            b += 1
        _wait(pending_send)


SCHEDULES = {"gpipe": ScheduleGPipe, "1f1b": Schedule1F1B}


def expected_order(schedule: str, stage_index: int, num_stages: int, n: int) -> List[str]:
    """The F/B sequence a stage must execute - used by the schedule-order tests."""
    if schedule == "gpipe":
        return [f"F{i}" for i in range(n)] + [f"B{i}" for i in range(n)]
    warmup = min(n, num_stages - stage_index)
    seq = [f"F{i}" for i in range(warmup)]
    f, b = warmup, 0
    while True:
        seq.append(f"B{b}")
        b += 1
        if f == n:
            break
        seq.append(f"F{f}")
        f += 1
    while b < n:
        seq.append(f"B{b}")
        b += 1
    return seq

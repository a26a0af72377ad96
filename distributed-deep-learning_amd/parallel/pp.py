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
  * P2P goes through ``P2PChannel`` (parallel/p2p.py -> csrc/p2p.cpp): transfer groups are issued from a
    dedicated P2P stream, sends are fenced behind the kernel that produced their payload, the receive of the
    NEXT micro-batch is posted before the current one is computed (per-micro-batch slots, so the bytes cross
    xGMI under compute), and the compute stream - never the host - waits per group (tickets).  Steady-state
    1F1B transfers toward one peer stay ONE RCCL group (forward-send with backward-recv toward the next
    stage, backward-send with forward-recv toward the previous one): issued as separate calls in different
    orders on the two sides they deadlock (SURVEY hard part 5);
  * every micro-batch forward / backward of a stage is its own hipGraph (``_GraphSet``);
  * gradients are scaled by 1/M (torch 2.6, which the reference pins, summed un-scaled
    per-micro-batch mean-loss gradients = M x the batch-mean gradient; documented deviation);
  * with a ``DataParallelEngine`` attached (hybrid), micro-batches 0..M-2 run under
    ``no_sync`` and the reducer is armed for the last backward only - one gradient sync per
    step per stage, overlapped with that backward (DDP arms its reducer for the last
    micro-batch, ddp_n_pp.py:139 + torch stage.py:632-646).  Under hipGraphs that last backward is
    captured as a CHAIN of graphs cut at the stage's backward phases (``forward_phased``), and the
    gradient buckets a phase completed go to the comm stream before the next graph is enqueued.

  * **lanes** (``n_lanes > 1``): the reference's shapes run micro-batches of 6 (pp.py) and 2 (ddp_n_pp.py), where
    every kernel of the late dense blocks is a ~10 us latency-bound launch that leaves most of the 256 CUs idle -
    one micro-batch after the other costs 23 ms per 30 images on one stage against 8.6 ms for one batch of 30.
    Micro-batches are independent except for three shared objects, and each gets an ordered hand-off instead of a
    serialised stage: BN running statistics change only in forwards -> ALL forwards replay in order on ONE forward
    stream; gradients accumulate -> backward ``i`` replays on backward stream ``i % n_lanes`` and accumulates into
    gradient arena ``i % n_lanes`` (lane 0 = the arena Adam / the reducer read; the others are added into it once
    per step, in lane order - fixed arithmetic, so the run stays bit-deterministic); saved activations -> every
    micro-batch graph keeps its own (separate graph pools per stream, autograd graphs retained: nothing two
    concurrently replaying graphs touch is ever aliased).  So in 1F1B steady state ``B_i`` runs next to
    ``F_{i+1}`` and ``B_{i+1}``; P2P groups are fenced against the stream that produced / consumes their
    payload.  With a data-parallel reducer the LAST micro-batch's backward runs after the lanes have been joined
    and summed, on the compute stream, so the reducer still overlaps exactly that backward (reference semantic).
    ``--debug-sync`` / eager steps use the same lane arithmetic on one stream: the serial oracle of the overlapped
    run (bit-identical parameters, ``tests/test_gpu_dist.py``).

BatchNorm statistics are per micro-batch, exactly as in the reference.
"""
from __future__ import annotations

from contextlib import nullcontext
from typing import Callable, List, Optional, Sequence, Tuple

import torch

from .p2p import P2PChannel, Transfer
from ..utils.knobs import knob
from ..utils.streams import concurrent_streams
from ..utils.trace import trace_range


def split_sizes(batch: int, n: int) -> List[int]:
    """``torch.tensor_split`` chunk sizes (reference splits with tensor_split on dim 0)."""
    base, rem = divmod(batch, n)
    return [base + (1 if i < rem else 0) for i in range(n)]


def phase_segments(module, arena) -> List[List[int]]:
    """Which arena segments (= gradient buckets) are complete after each backward phase of ``module``
    (execution order: last units first).  Segments are cut at phase boundaries by construction
    (``NativeDenseNetStage``: ``break_before``), so this is a static table."""
    pending = [sum(1 for i in seg.slots if arena.params[i].requires_grad) for seg in arena.segments]
    index = {n: i for i, n in enumerate(arena.names)}
    out = []
    for names in module.phase_param_names():
        done = []
        for n in names:
            k = arena.slots[index[n]].segment
            pending[k] -= 1
            if pending[k] == 0:
                done.append(k)
        out.append(done)
    return out


class _GraphNode:
    __slots__ = ("graph", "x", "t", "inp", "out", "loss", "tail", "segments")

    def __init__(self) -> None:
        self.graph = self.x = self.t = self.inp = self.out = self.loss = None
        self.tail, self.segments = [], []      # phased backward: later graphs + buckets completed per phase


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
        self._pools = {}

    def begin_step(self) -> None:
        self.active = self.steps >= self.warmup
        self.steps += 1

    def pool(self, key=None):
        """One pool for a serial stage; with lanes one per stream (graphs of one pool replay in capture order on
        one stream - graphs that replay concurrently must never share transient memory)."""
        if key not in self._pools:
            self._pools[key] = torch.cuda.graph_pool_handle()
        return self._pools[key]


class PipelineStage:
    """One stage of the pipeline on this rank.

    ``recv_shape_fn(mb)`` -> shape of the activation this stage receives for a micro-batch of
    ``mb`` samples; ``send`` shapes are whatever the module produces.
    """

    def __init__(self, module: torch.nn.Module, stage_index: int, num_stages: int, device,
                 group=None, ranks: Optional[Sequence[int]] = None,
                 recv_shape_fn: Optional[Callable[[int], Tuple[int, ...]]] = None,
                 dtype: torch.dtype = torch.float32, dp_engine=None, hip_graph: bool = False,
                 graph_warmup: int = 2, debug_sync: bool = False, profiler=None, n_lanes: int = 1,
                 n_fwd_lanes: int = 1, rccl: str = "auto") -> None:
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
        self.chan = P2PChannel(group, self.device, debug_sync=debug_sync, profiler=profiler,
                               neighbours=[r for r in (self.prev_rank, self.next_rank) if r is not None], rccl=rccl)
        self._act_slots = {}      # (mb_index, shape) -> recv buffer for forward activations
        self._grad_slots = {}     # (mb_index, shape) -> recv buffer for backward gradients
        # hipGraph capture of the per-micro-batch forward / backward (see ``_GraphSet``)
        self.hip_graph = bool(hip_graph) and self.device.type == "cuda"
        self.graph_warmup = graph_warmup
        self._graph_sets = {}     # signature of a step (micro-batch shapes, train/eval) -> _GraphSet
        self._gs: Optional[_GraphSet] = None
        # hybrid under hipGraphs: the last micro-batch's backward runs as a chain of graphs cut at the module's
        # backward phases, finished buckets go out between them (see the module docstring)
        self.phased = (self.hip_graph and dp_engine is not None and not dp_engine.single
                       and bool(getattr(module, "phase_starts", None)))
        self.launch_trace: List[Tuple[str, int]] = []     # ("graph" | "bucket", index) of the last phased backward
        self._cuts = {}
        # lanes (module docstring): gradient-lane ARITHMETIC whenever the module can do it (eager, --debug-sync and
        # graphed steps then compute the same bits), lane CONCURRENCY only when graphs replay on their own streams
        self.n_lanes = max(1, int(n_lanes)) if hasattr(module, "set_grad_lane") else 1
        if self.n_lanes > 1:
            module.ensure_grad_lanes(self.n_lanes)
        self.lanes_concurrent = self.hip_graph and self.n_lanes > 1 and not debug_sync
        # With lanes every micro-batch graph is ONE chain: the weight-gradient side branch inside a graph runs on
        # runtime-internal streams that all graphs share, which serialised B_i with B_{i+1} and F_{i+2} in the trace
        # (profiles/r3/lane_timeline_*.txt: 20.1 ms per step with the fork, 15.4 without) - the lanes supply the
        # concurrency.  DDL_LANE_WGRAD_FORK=1 (experiment switch) keeps the fork.
        if self.lanes_concurrent and knob("DDL_LANE_WGRAD_FORK", "0") == "0" and hasattr(module, "overlap_wgrad"):
            module.overlap_wgrad = False
        self.join_before_last = dp_engine is not None and not dp_engine.single
        # forward lanes: forward i on forward stream i % n_fwd, BN running statistics handed off in micro-batch order
        # through per-micro-batch scratch copies (native_densenet "buffer lanes"); arithmetic applies in eager steps too
        self.n_fwd = max(1, int(n_fwd_lanes)) if (self.n_lanes > 1 and hasattr(module, "ensure_buffer_lanes")) else 1
        # borrowed lanes: once the step's last forwards are out, the forward streams idle while the remaining backwards
        # queue up on the backward streams (lane timeline of round 3: B4 ran ALONE for 2.5 ms of a 13.9 ms step).  The
        # last `n_fwd` backwards therefore run on the forward stream of their own micro-batch (no later forward is ever
        # issued on it) and accumulate into gradient arenas of their own - same fixed lane arithmetic in eager steps.
        self.borrow = self.n_lanes > 1 and self.n_fwd > 1 and knob("DDL_LANE_BORROW", "1") == "1"
        if self.borrow:
            module.ensure_grad_lanes(self.n_lanes + self.n_fwd)
        self._n_mb = 0
        self.fwd_streams: List = []
        self.bwd_streams: List = []
        self._lanes_live = False          # True between begin_step and join_lanes of a graphed training step
        self._f_done = {}
        self.reset()

    def reset(self) -> None:
        self.inputs: dict = {}
        self.outputs: dict = {}
        self.grads_in: dict = {}
        self.losses: dict = {}
        self._cuts = {}
        self._lanes_dirty = False

    # ---- P2P transfer builders -----------------------------------------------------------
    def _slot(self, table: dict, i: int, shape) -> torch.Tensor:
        key = (i, tuple(shape))
        buf = table.get(key)
        if buf is None:
            buf = torch.empty(tuple(shape), dtype=self.dtype, device=self.device)
            table[key] = buf
        return buf

    def fwd_recv(self, i: int, mb: int) -> List[Transfer]:
        if self.is_first:
            return []
        buf = self._slot(self._act_slots, i, self.recv_shape_fn(mb))
        self.inputs[i] = buf
        return [(buf, self.prev_rank)]

    def fwd_send(self, i: int) -> List[Transfer]:
        if self.is_last:
            return []
        return [(self.outputs[i].detach(), self.next_rank)]

    def bwd_recv(self, i: int) -> List[Transfer]:
        if self.is_last:
            return []
        buf = self._slot(self._grad_slots, i, self.outputs[i].shape)
        self.grads_in[i] = buf
        return [(buf, self.next_rank)]

    def bwd_send(self, i: int) -> List[Transfer]:
        if self.is_first:
            return []
        g = self.inputs.pop(i)
        return [(g, self.prev_rank)]

    # ---- hipGraph plumbing ---------------------------------------------------------------
    def begin_step(self, signature) -> None:
        """Called by the schedule before the first micro-batch: selects the graph set of this step
        shape.  The first ``graph_warmup`` steps of a shape run eagerly, the next one captures every
        forward / backward in schedule order (capture + replay in place), later steps only replay."""
        self._gs = None
        self._lanes_live = False
        if not self.hip_graph:
            return
        gs = self._graph_sets.get(signature)
        if gs is None:
            gs = self._graph_sets[signature] = _GraphSet(self.graph_warmup)
        gs.begin_step()
        self._gs = gs if gs.active else None
        train = bool(signature[1]) if len(signature) > 1 else False
        if self._gs is not None and self.lanes_concurrent and train:
            self._open_lanes()

    def prepare_step(self, n_microbatches: int, train: bool) -> None:
        """Called by the schedule before ``begin_step`` (eager and graphed steps alike)."""
        self._handoff = self.n_fwd > 1 and train
        self._n_mb = int(n_microbatches)
        if self._handoff:
            self.module.ensure_buffer_lanes(n_microbatches)

    # ---- lanes ---------------------------------------------------------------------------
    def _open_lanes(self) -> None:
        """Fork the lane streams from the compute stream: they start behind everything enqueued so far (input
        copies, zero_grad, the previous step's Adam)."""
        main = torch.cuda.current_stream(self.device)
        if not self.fwd_streams:
            # streams on DISTINCT hardware queues (measured, utils/streams.py): two lanes that share a queue are
            # serialised on the device whatever the events say
            avoid = [main, getattr(self.module, "side_stream", None),
                     getattr(self.dp_engine, "comm_stream", None), *self.chan.streams]
            lanes = concurrent_streams(self.device, self.n_fwd + self.n_lanes, avoid=avoid)
            self.fwd_streams, self.bwd_streams = lanes[:self.n_fwd], lanes[self.n_fwd:]
        begin = torch.cuda.Event()
        begin.record(main)
        for st in self.fwd_streams + self.bwd_streams:
            st.wait_event(begin)
        self._f_done = {}
        self._lanes_live = True

    def stream_for(self, kind: str, i: int, last: bool = False):
        """Stream micro-batch ``i``'s forward (``"F"``) / backward (``"B"``) runs on; None = the compute stream."""
        if not self._lanes_live:
            return None
        if kind == "F":
            return self.fwd_streams[i % self.n_fwd]
        if last and self.join_before_last:
            return None
        if self._borrowed(i):
            return self.fwd_streams[i % self.n_fwd]
        return self.bwd_streams[i % self.n_lanes]

    def _borrowed(self, i: int) -> bool:
        """Backward ``i`` is one of the step's last ``n_fwd`` backwards: it runs on its micro-batch's forward stream."""
        return self.borrow and i >= self._n_mb - self.n_fwd

    def lane_ctx(self, stream):
        return torch.cuda.stream(stream) if stream is not None else nullcontext()

    def join_lanes(self) -> None:
        """The compute stream waits for every lane; the gradient lanes are added into the arena (lane order).
        Idempotent within a step."""
        if self._lanes_live:
            main = torch.cuda.current_stream(self.device)
            for st in self.fwd_streams + self.bwd_streams:
                done = torch.cuda.Event()
                done.record(st)
                main.wait_event(done)
            self._lanes_live = False
        if self.n_lanes > 1 and self._lanes_dirty:
            self.module.sum_grad_lanes()
            self._lanes_dirty = False
        if getattr(self, "_handoff", False):
            self.module.combine_buffer_lanes()       # running statistics advanced in micro-batch order

    def forward(self, i: int, x: Optional[torch.Tensor], target: Optional[torch.Tensor],
                loss_fn, train: bool, last: bool = False) -> None:
        with trace_range(f"Forward {i}"), self.lane_ctx(self.stream_for("F", i)):      # range names: schedules.py:757
            self._forward(i, x, target, loss_fn, train, last)
            if self._lanes_live:
                self._f_done[i] = torch.cuda.Event()
                self._f_done[i].record(self.fwd_streams[i % self.n_fwd])

    def _grad_lane(self, i: int, last: bool) -> int:
        """Gradient arena micro-batch ``i`` accumulates into."""
        if self.n_lanes == 1 or (last and self.join_before_last):
            return 0
        if self._borrowed(i):
            return self.n_lanes + i % self.n_fwd
        return i % self.n_lanes

    def backward(self, i: int, scale: float, last: bool) -> None:
        if last and self.join_before_last and self.n_lanes > 1:
            # the reducer is armed for this backward only (ddp_n_pp.py:139 + torch stage.py:632-646): everything the
            # earlier micro-batches accumulated must be in the arena before its first bucket leaves
            self.join_lanes()
        stream = self.stream_for("B", i, last)
        with trace_range(f"Backward {i}"), self.lane_ctx(stream):                    # schedules.py:785
            if stream is not None and i in self._f_done:
                stream.wait_event(self._f_done[i])
            self._backward(i, scale, last)

    def _forward(self, i: int, x: Optional[torch.Tensor], target: Optional[torch.Tensor],
                 loss_fn, train: bool, last: bool) -> None:
        gs = self._gs
        if gs is None:
            return self._forward_eager(i, x, target, loss_fn, train)
        phased = self.phased and train and last
        node = gs.fwd.get(i)
        if node is None:
            node = gs.fwd[i] = _GraphNode()
            node.x = x.clone() if x is not None else None
            node.t = target.clone() if target is not None else None
            inp_slot = None if self.is_first else self.inputs[i]     # persistent receive slot
            node.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(node.graph, pool=gs.pool(("F", i % self.n_fwd) if self._lanes_live else None),
                                  capture_error_mode="thread_local"):
                if not self.is_first:
                    self.inputs[i] = inp_slot
                self._forward_eager(i, node.x, node.t, loss_fn, train, phased)
            node.inp, node.out, node.loss = self.inputs.get(i), self.outputs[i], self.losses.get(i)
        else:
            if node.x is not None:
                node.x.copy_(x, non_blocking=True)
            if node.t is not None:
                node.t.copy_(target, non_blocking=True)
            if train and getattr(self, "_handoff", False):
                self.module.note_microbatch_forward(i)
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
        lane = self._grad_lane(i, last)
        # lanes: graphs of different backward streams replay concurrently -> one pool per stream, and the autograd
        # graph is retained so that the saved activations (forward pool) are never handed to a later forward
        # capture while this backward may still be reading them
        lanes = self.n_lanes > 1 and self.lanes_concurrent
        mode = dict(pool=gs.pool(("B", lane) if lanes else None), capture_error_mode="thread_local")
        if self.n_lanes > 1:
            self._lanes_dirty = self._lanes_dirty or lane > 0
        if node is None:
            node = gs.bwd[i] = _GraphNode()
            if dp is not None:
                dp.arm(False)                      # no collectives inside a capture
            if self.n_lanes > 1:
                self.module.set_grad_lane(lane)    # the captured kernels write this lane's gradient arena
            grad_slot = self.grads_in.get(i)
            cuts = self._cuts.pop(i, None)         # phased forward of the last micro-batch (hybrid)
            node.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(node.graph, **mode):
                if grad_slot is not None:
                    self.grads_in[i] = grad_slot
                self._backward_eager(i, scale, False, arm=False, finish_input=not cuts, retain=lanes, set_lane=False)
                self._check_joined()
            if cuts:
                for k, (y_mid, leaf) in enumerate(reversed(cuts)):        # one graph per earlier phase
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, **mode):
                        y_mid.backward(leaf.grad, retain_graph=lanes)
                        if k == len(cuts) - 1:
                            self._finish_input(i)
                        self._check_joined()
                    node.tail.append(g)
                node.segments = phase_segments(self.module, dp.arena)
            node.inp = self.inputs.get(i)          # gradient w.r.t. the stage input (sent upstream)
            node.out = self.outputs.get(i)
        else:
            self.grads_in.pop(i, None)
            if self.is_last:
                self.outputs[i] = node.out
            else:
                self.outputs.pop(i, None)
        if dp is not None and last:
            dp.arm(True)                           # the reducer counts from here: this is the step's last backward
        self.launch_trace = []
        node.graph.replay()
        self.launch_trace.append(("graph", 0))
        for k, g in enumerate(node.tail):
            for seg in node.segments[k]:           # buckets completed by the graph just enqueued
                dp.segment_ready(seg)
                self.launch_trace.append(("bucket", seg))
            with trace_range(f"Backward {i} phase {k + 1}"):
                g.replay()
            self.launch_trace.append(("graph", k + 1))
        if not self.is_first:
            self.inputs[i] = node.inp
        else:
            self.inputs.pop(i, None)
        # the buckets of the final phase (and everything, when the backward is one graph) are launched by
        # the strategy's finish_grad_sync()

    def _check_joined(self) -> None:
        check = getattr(self.module, "assert_joined", None)     # native stage: forked streams are back in the origin
        if check is not None:
            check()

    # ---- compute -------------------------------------------------------------------------
    def _forward_eager(self, i: int, x: Optional[torch.Tensor], target: Optional[torch.Tensor],
                       loss_fn, train: bool, phased: bool = False) -> None:
        if self.is_first:
            inp = x
        else:
            inp = self.inputs[i].detach()
            if train:
                inp.requires_grad_(True)
            self.inputs[i] = inp
        if train and getattr(self, "_handoff", False):
            self.module.begin_microbatch_buffers(i)      # BN running-statistic updates of this forward -> scratch i
        if phased:
            out, self._cuts[i] = self.module.forward_phased(inp)
        else:
            out = self.module(inp)
        self.outputs[i] = out
        if self.is_last and loss_fn is not None:
            self.losses[i] = loss_fn(out, target)

    def _finish_input(self, i: int) -> None:
        if not self.is_first:
            inp = self.inputs[i]
            self.inputs[i] = inp.grad
            inp.grad = None
        else:
            self.inputs.pop(i, None)

    def _backward_eager(self, i: int, scale: float, last: bool, arm: bool = True, finish_input: bool = True,
                        retain: bool = False, set_lane: bool = True) -> None:
        dp = self.dp_engine
        if self.n_lanes > 1 and set_lane:
            # eager / --debug-sync: the same lane arithmetic as the graphed step, on one stream
            if last and self.join_before_last:
                self.join_lanes()
            lane = self._grad_lane(i, last)
            self.module.set_grad_lane(lane)
            self._lanes_dirty = self._lanes_dirty or lane > 0
        if dp is not None and arm:
            dp.arm(last)
        if self.is_last:
            loss = self.losses[i]
            (loss * scale).backward(retain_graph=retain)
            self.outputs[i] = self.outputs[i].detach()
        else:
            # the 1/M factor entered at the loss on the last stage and rides the chain rule back
            torch.autograd.backward(self.outputs.pop(i), grad_tensors=self.grads_in.pop(i), retain_graph=retain)
        if finish_input:
            self._finish_input(i)


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
        st.prepare_step(self.n, train)
        st.begin_step((tuple(sizes), train, self.loss_fn is not None,
                       tuple(x.shape[1:]) if x is not None else None))
        self.order = []
        self._sends: List[Optional[int]] = []
        self._run(xs, ts, sizes, train)
        self._settle()
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
        on = self._sf(i)
        tok = self.profiler.begin("compute", on) if self.profiler else None
        self.stage.forward(i, xs[i], ts[i], self.loss_fn if train else None, train, last=(i == self.n - 1))
        if tok is not None:
            self.profiler.end(tok, on)

    def _B(self, i):
        self.order.append(f"B{i}")
        on = self._sb(i)
        tok = self.profiler.begin("compute", on) if self.profiler else None
        scale = 1.0 / self.n if self.scale_grads else 1.0
        self.stage.backward(i, scale, last=(i == self.n - 1))
        if tok is not None:
            self.profiler.end(tok, on)

    def _post(self, sends=(), recvs=(), on=None):
        """Issue one transfer group on the P2P channel; returns its ticket (None for an empty group).  ``on``: the
        stream that produced the payload of ``sends`` (lanes) - the group is fenced behind that stream; None = the
        compute stream."""
        with self.stage.lane_ctx(on):
            return self.stage.chan.post(sends, recvs)

    def _send(self, sends, on=None) -> None:
        """A pure send group: nobody on this rank waits for it before the end of the step."""
        self._sends.append(self._post(sends=sends, on=on))

    def _await(self, ticket, on=None) -> None:
        """The stream that is about to read a group's payload (``on``; None = the compute stream) waits for it
        ("p2p_wait" = how long it actually stalled there)."""
        if ticket is None:
            return
        with self.stage.lane_ctx(on):
            tok = self.profiler.begin("p2p_wait") if self.profiler else None
            self.stage.chan.wait(ticket)
            if tok is not None:
                self.profiler.end(tok)

    def _sf(self, i: int):
        return self.stage.stream_for("F", i)

    def _sb(self, i: int):
        return self.stage.stream_for("B", i, last=(i == self.n - 1))

    def _settle(self) -> None:
        """End of the step: the send buffers (graph outputs, gradient slots) are rewritten by the next step."""
        self.stage.join_lanes()          # compute stream behind every lane; gradient lanes summed into the arena
        for t in self._sends:
            self.stage.chan.wait(t)
        self._sends = []
        self.stage.chan.drain()
        self.stage.chan.fence()          # next step's pre-posted receives reuse this step's slots: behind their readers

    def _forward_phase(self, xs, ts, sizes, train, count: int, first: int = 0) -> None:
        """Forwards ``first .. first+count-1`` with the next activation's receive posted one micro-batch ahead
        (double buffering through the per-micro-batch slots); every forward's output is sent on."""
        st = self.stage
        tickets = {}
        if count > 0:
            tickets[first] = self._post(recvs=st.fwd_recv(first, sizes[first]))
        for i in range(first, first + count):
            if i + 1 < first + count:
                tickets[i + 1] = self._post(recvs=st.fwd_recv(i + 1, sizes[i + 1]))
            self._await(tickets.pop(i), on=self._sf(i))
            self._F(i, xs, ts, train)
            self._send(st.fwd_send(i), on=self._sf(i))

    def _backward_phase(self, first: int, count: int) -> None:
        st = self.stage
        tickets = {}
        if count > 0:
            tickets[first] = self._post(recvs=st.bwd_recv(first))
        for i in range(first, first + count):
            if i + 1 < first + count:
                tickets[i + 1] = self._post(recvs=st.bwd_recv(i + 1))
            self._await(tickets.pop(i), on=self._sb(i))
            self._B(i)
            self._send(st.bwd_send(i), on=self._sb(i))

    def _run(self, xs, ts, sizes, train):
        raise NotImplementedError


class ScheduleGPipe(_ScheduleBase):
    """Fill-drain: F0..F(M-1) then B0..B(M-1) (the reference's schedule, pp.py:140-144)."""

    def _run(self, xs, ts, sizes, train):
        self._forward_phase(xs, ts, sizes, train, self.n)
        if train:
            self._backward_phase(0, self.n)


class Schedule1F1B(_ScheduleBase):
    """One-forward-one-backward: ``min(M, S - stage)`` warm-up forwards, then alternate B/F,
    then drain.  Steady-state transfers toward one peer are fused into one RCCL group."""

    def _run(self, xs, ts, sizes, train):
        st = self.stage
        n = self.n
        if not train:
            self._forward_phase(xs, ts, sizes, False, n)
            return
        warmup = min(n, st.num_stages - st.stage_index)
        # warm-up: like the forward phase, but the LAST warm-up activation is held back - it travels in one
        # group with the first gradient receive (same peer, opposite directions)
        tickets = {0: self._post(recvs=st.fwd_recv(0, sizes[0]))}
        fwd_sends: List[Transfer] = []
        for f in range(warmup):
            if f + 1 < warmup:
                tickets[f + 1] = self._post(recvs=st.fwd_recv(f + 1, sizes[f + 1]))
            self._await(tickets.pop(f), on=self._sf(f))
            self._F(f, xs, ts, True)
            fwd_sends = st.fwd_send(f)
            if f != warmup - 1:
                self._send(fwd_sends, on=self._sf(f))
                fwd_sends = []
        f, b = warmup, 0
        # steady state: 1B1F
        while True:
            # forward-send of the newest activation fused with the backward-recv (same peer: next); with lanes the
            # payload comes from the forward stream and the backward's stream waits for the group
            self._await(self._post(sends=fwd_sends, recvs=st.bwd_recv(b), on=self._sf(f - 1)), on=self._sb(b))
            self._B(b)
            bwd_sends = st.bwd_send(b)
            b += 1
            if f == n:
                break
            # backward-send fused with the next forward-recv (same peer: prev)
            self._await(self._post(sends=bwd_sends, recvs=st.fwd_recv(f, sizes[f]), on=self._sb(b - 1)), on=self._sf(f))
            self._F(f, xs, ts, True)
            fwd_sends = st.fwd_send(f)
            f += 1
        # cooldown: drain the remaining backwards (next gradient's receive posted one ahead)
        self._send(bwd_sends, on=self._sb(b - 1))
        self._backward_phase(b, n - b)


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

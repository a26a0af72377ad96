"""Data-parallel engine: flat-arena gradients, bucketed reduce-scatter / all-gather on a
side HIP stream, overlapped with backward.

What it replaces (reference: ddp.py:127 ``DDP(model, device_ids=[local_rank])`` and
ddp_n_pp.py:139 ``DDP(stage_mod, process_group=dp)``):

  =====================================  ==================================================
  torch DDP (c10d Reducer, C++)          this engine
  =====================================  ==================================================
  rank-0 param/buffer broadcast at wrap  one flat broadcast of the param arena + buffer arena
  per-forward BN-buffer broadcast (C4)   one flat broadcast (policy: forward | epoch | off)
  25 MiB buckets, copy-in/copy-out       bucket == contiguous arena segment, zero copies
  all-reduce(SUM) then divide            reduce-scatter(AVG) + all-gather per bucket; RCCL on a
                                         fully-connected xGMI node runs both directly over all 7
                                         links, and the scatter half is what ZeRO-1 needs anyway
  NCCL-internal stream                   explicit comm stream + hipEvent fences (bucket-ready:
                                         compute->comm, grads-ready: comm->compute), which also lets
                                         the profiler time RCCL separately from compute
  no_sync() / prepare_for_backward       ``arm()`` - the pipeline schedule arms the reducer for the
                                         last micro-batch only (one sync per step per stage)
  =====================================  ==================================================

Per-bucket collective (``auto``: all-reduce for 2 ranks, rs_ag from 3 on): ``rs_ag`` = in-place reduce-scatter(AVG) + all-gather of the arena range - on a
fully connected xGMI node both halves run directly over all 7 links of a GPU; ``allreduce`` = one RCCL launch
per bucket.  Which one wins at a given world size is a measurement, not a belief: the form can be switched
between two steps (``set_grad_sync``; the collectives live outside the captured graphs) and ``bench.py`` at
W > 1 times both on the same captured step (``grad_sync_ab`` in its JSON) together with two more bucket sizes.

Bucket size: 8 MiB default (DenseNet-121 = 27.8 MB -> 4 buckets, cut at the backward-phase boundaries): small
enough that the first bucket leaves while most of the backward pass is still to run, large enough to amortise
the ~10-20 us launch + fence cost of a collective.  ``bench.py`` sweeps 2 / 8 / 32 MiB behind its timed region.

ZeRO-1 (``zero1=True``): stop after the reduce-scatter, let each rank run Adam on its 1/W
shard (``FusedAdam(shard=...)``), then all-gather the *parameters* - same bytes on the wire,
1/W of the optimizer work and state traffic.
"""
from __future__ import annotations

import os
import time
from contextlib import contextmanager
from typing import Callable, List

import torch
import torch.distributed as dist

from .. import comm, ops
from ..utils.arena import FlatArena, FlatBuffers, reverse_named_parameters
from ..utils.streams import new_stream, new_stream_beside


class DataParallelEngine:
    def __init__(self, module: torch.nn.Module, group=None, device=None, bucket_mb: float = 8.0,
                 grad_sync: str = "auto", sync_buffers: str = "forward", zero1: bool = False,
                 profiler=None, debug_sync: bool = False, register_hooks: bool = True, rccl: str = "auto",
                 test_hook: int = 0, hook_spin_us: float = 0.0, skip_fence: str = "") -> None:
        self.module = module
        self.group = group
        self.world = comm.get_world_size(group) if comm.is_dist() else 1
        self.rank = comm.get_rank(group) if comm.is_dist() else 0
        # DDL_FORCE_COLLECTIVES=1: issue every collective even in a world of one (a one-GPU box can then drive
        # the real RCCL path - comm stream, fences, phased graphs - end to end; tests/test_gpu_dist.py)
        self.single = self.world == 1 and not (comm.is_dist() and comm.force_collectives())
        self.device = torch.device(device) if device is not None else next(module.parameters()).device
        # ZeRO-1 needs the scattered form (each rank keeps only its shard of the reduced gradient)
        if grad_sync == "auto":
            grad_sync = "allreduce" if self.world <= 2 else "rs_ag"       # config.py: grad_sync
        if grad_sync not in ("rs_ag", "allreduce"):
            raise ValueError(f"grad_sync: {grad_sync!r}")
        self.grad_sync = "rs_ag" if zero1 else grad_sync
        self.sync_buffers_policy = sync_buffers
        self.zero1 = zero1
        self.profiler = profiler
        self.debug_sync = debug_sync
        self.src_global = dist.get_global_rank(group, 0) if (comm.is_dist() and group is not None) else 0

        bucket_elems = max(int(bucket_mb * 2 ** 20 / 4), 1)
        existing = getattr(module, "arena", None)
        if isinstance(existing, FlatArena):
            # native engine modules own their arena (built with the right segmenting by the factory)
            self.arena = existing
        else:
            self.arena = FlatArena(reverse_named_parameters(module), device=self.device,
                                   segment_numel=bucket_elems, segment_multiple=self.world * 64)
        for seg in self.arena.segments:
            if (seg.end - seg.start) % self.world:
                raise ValueError("arena segments must be divisible by the dp world size")
        self.buffers = getattr(module, "flat_buffers", None) or FlatBuffers(module, self.device)

        self.on_cuda = self.device.type == "cuda"
        self.comm_stream = None
        if self.on_cuda:
            avoid = [getattr(module, "side_stream", None)]
            # with real peers the communication stream must sit on another hardware queue than the compute stream
            # (utils/streams.py: streams that share a queue are serialised on the device whatever the events say)
            self.comm_stream = (new_stream(self.device, avoid=avoid) if self.single else
                                new_stream_beside(self.device, [torch.cuda.current_stream(self.device)], avoid=avoid))
        self._done_events: List = []
        self._armed = True
        self._pending = [0] * len(self.arena.segments)
        self._launched = [False] * len(self.arena.segments)
        self.comm_seconds = 0.0           # host-timed (CPU transport); CUDA timing goes through the profiler
        self._slot_segment = [s.segment for s in self.arena.slots]
        self._reset_counters()

        # Native bucket reducer (csrc/reducer.cpp): on a GPU with an RCCL group the bookkeeping + fences +
        # collectives of the gradient path run in C++; gloo (CPU tests, several ranks on one GPU) keeps
        # the Python path below, which is the same algorithm.
        # Fence tests (tests/test_gpu_fences.py): `test_hook` makes the communication stream's work visible in the
        # data (spin + "*= 0.5" between the two fences), `skip_fence` leaves ONE fence out - the negative control:
        # bucket_ready | grads_ready (gradient buckets), buffers_ready | buffers_done (deferred buffer broadcast).
        if skip_fence not in ("", "bucket_ready", "grads_ready", "buffers_ready", "buffers_done"):
            raise ValueError(f"skip_fence: {skip_fence!r}")
        self._test_hook, self._hook_spin_us, self._skip_fence = int(test_hook), float(hook_spin_us), skip_fence
        self._native = None
        self._direct = None            # owned RCCL communicator of the dp group (csrc/rccl_comm.cpp)
        self.transport = "none" if self.single else ("c10d" if comm.backend_of(group) == "nccl" else comm.backend_of(group))
        if (self.on_cuda and not self.single and comm.backend_of(group) == "nccl" and ops.have_ext()
                and os.environ.get("DDL_NATIVE_REDUCER", "1") == "1"):
            if comm.rccl_transport(rccl) == "direct":
                self._direct = comm.direct_comm_or_fallback(comm.group_ranks(group), self.device, "dp", group,
                                                            who="DataParallelEngine")
                self.transport = "direct" if self._direct is not None else "c10d (owned communicator unavailable)"
            segs = self.arena.segments
            pending = [sum(1 for i in seg.slots if self.arena.params[i].requires_grad) for seg in segs]
            cs = self.comm_stream
            self._native = ops.require_ext().BucketReducer(
                self.arena.flat_grad, [s.start for s in segs], [s.end for s in segs], list(self._slot_segment),
                pending, group if group is not None else dist.group.WORLD, self.rank, self.world, bool(zero1),
                self.grad_sync == "allreduce", bool(debug_sync), bool(profiler is not None and profiler.enabled),
                cs.stream_id, cs.device_index, cs.device_type, self._direct, self._test_hook, self._hook_spin_us,
                {"bucket_ready": 1, "grads_ready": 2}.get(skip_fence, 0))
            if profiler is not None:
                profiler.add_source("rccl", self._native.drain_ms, self._native.set_timing)

        # per-forward buffer broadcast behind the forward instead of in front of it (after_forward); --debug-sync
        # keeps the same position in the step and only moves it onto the compute stream (the serial oracle)
        self._defer_buffers = (self.on_cuda and not self.single and comm.backend_of(group) == "nccl")
        self.broadcast_state()
        if register_hooks:
            for i, p in enumerate(self.arena.params):
                if p.requires_grad:
                    p.register_post_accumulate_grad_hook(self._make_hook(i))
            if hasattr(module, "set_grad_ready_callback"):
                module.set_grad_ready_callback(self.param_ready)

    def set_grad_sync(self, mode: str) -> None:
        """Switch the per-bucket collective between two steps (``allreduce`` | ``rs_ag``).  Buckets are arena
        ranges and collectives run outside the captured graphs, so nothing is re-captured."""
        if mode not in ("allreduce", "rs_ag"):
            raise ValueError(mode)
        if self.zero1 and mode != "rs_ag":
            raise ValueError("ZeRO-1 needs the scattered form (rs_ag)")
        self.grad_sync = mode
        if self._native is not None:
            self._native.set_allreduce(mode == "allreduce")

    # ------------------------------------------------------------------ state sync
    def broadcast_state(self) -> None:
        """Rank-0 parameters and buffers win (DDP ctor semantics)."""
        if self.single:
            return
        comm.broadcast(self.arena.flat_param, self.src_global, self.group)
        self.broadcast_buffers(include_int=True)

    def broadcast_buffers(self, include_int: bool = False) -> None:
        """Rank 0's BN buffers win.  Runs on the CURRENT stream: through the owned communicator the RCCL kernel is
        enqueued right there, through c10d it runs on the group's internal stream fenced against the current one."""
        if self.single:
            return
        for flat in (self.buffers.flat_float, self.buffers.flat_int if include_int else None):
            if flat is None or not flat.numel():
                continue
            if self._direct is not None:
                self._direct.broadcast(flat, 0)                      # root = group rank 0
            else:
                comm.broadcast(flat, self.src_global, self.group)
        if self._test_hook and self.on_cuda and self.buffers.flat_float.numel():
            ext = ops.require_ext()
            if self._hook_spin_us > 0:
                ext.test_spin(self._hook_spin_us, self.device.index or 0)
            ext.test_scale(self.buffers.flat_float, 0.5)

    def pre_forward(self, training: bool = False) -> None:
        """Call before a forward pass (train or eval): DDP's ``broadcast_buffers=True``.
        Training steps on RCCL defer it to ``after_forward`` (same values, off the critical path)."""
        if self.sync_buffers_policy != "forward" or (training and self._defer_buffers):
            return
        self._timed(self.broadcast_buffers)

    def after_forward(self) -> None:
        """Training step, policy ``forward``: BN buffers only change in the forward pass, so "rank 0's buffers
        before forward k+1" (what DDP broadcasts) are final as soon as forward k is done.  They travel on the
        communication stream while backward k runs; the grads-ready fence of ``finish_grad_sync`` covers them,
        so the next forward (or an evaluation / checkpoint in between) sees synchronised buffers - without a
        blocking broadcast in front of every step."""
        if self.sync_buffers_policy != "forward" or not self._defer_buffers:
            return
        if self.debug_sync:
            self._timed(self.broadcast_buffers)          # same place in the step, on the compute stream
            return
        main = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(self.comm_stream):
            if self._skip_fence != "buffers_ready":
                self.comm_stream.wait_event(ready)       # buffers-ready: forward (compute) -> communication stream
            tok = self.profiler.begin("rccl", self.comm_stream) if self.profiler else None
            self.broadcast_buffers()
            if tok is not None:
                self.profiler.end(tok, self.comm_stream)
            done = torch.cuda.Event()
            done.record(self.comm_stream)
        if self._skip_fence != "buffers_done":
            self._done_events.append(done)               # waited for by finish_grad_sync (communication -> compute)

    def epoch_boundary(self) -> None:
        if self.sync_buffers_policy == "epoch":
            self.broadcast_buffers(include_int=True)

    # ------------------------------------------------------------------ gradient sync
    def _reset_counters(self) -> None:
        for k, seg in enumerate(self.arena.segments):
            self._pending[k] = sum(1 for i in seg.slots if self.arena.params[i].requires_grad)
            self._launched[k] = False

    def param_ready(self, slot_index: int) -> None:
        """The gradient of arena slot ``slot_index`` is final (autograd hook, or the native
        engine's per-unit callback for gradients it writes into the arena itself)."""
        if self._native is not None:
            self._native.param_ready(slot_index)
            return
        if not self._armed:
            return
        k = self._slot_segment[slot_index]
        self._pending[k] -= 1
        if self._pending[k] == 0:
            self.segment_ready(k)

    def _make_hook(self, slot_index: int) -> Callable:
        def hook(param):
            self.param_ready(slot_index)
        return hook

    @contextmanager
    def no_sync(self):
        """Accumulate locally (micro-batches 0..M-2 of a pipeline step)."""
        prev, self._armed = self._armed, False
        if self._native is not None:
            self._native.arm(False, False)
        try:
            yield
        finally:
            self._armed = prev
            if self._native is not None:
                self._native.arm(prev, False)

    def arm(self, armed: bool = True) -> None:
        self._armed = armed
        if self._native is not None:
            self._native.arm(armed, armed)
        if armed:
            self._reset_counters()

    def segment_ready(self, k: int) -> None:
        """All gradients of arena segment ``k`` are final: launch its collective(s) now.
        Called by the autograd hooks, or directly by the native engine's backward."""
        if self._native is not None:
            self._native.segment_ready(k)
            return
        if self._launched[k] or not self._armed:
            return
        self._launched[k] = True
        if self.single:
            return
        seg = self.arena.segments[k]
        grad = self.arena.flat_grad[seg.start:seg.end]
        n = (seg.end - seg.start) // self.world
        shard = grad[self.rank * n:(self.rank + 1) * n]

        def run():
            if self.grad_sync == "allreduce":
                comm.all_reduce(grad, "avg", self.group)
            else:
                comm.reduce_scatter_into(shard, grad, "avg", self.group)
                if not self.zero1:
                    comm.all_gather_into(grad, shard, self.group)

        if self.on_cuda and not self.debug_sync:
            main = torch.cuda.current_stream(self.device)
            ready = torch.cuda.Event()
            ready.record(main)                         # bucket-ready fence: compute -> comm
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ready)
                tok = self.profiler.begin("rccl", self.comm_stream) if self.profiler else None
                run()
                if tok is not None:
                    self.profiler.end(tok, self.comm_stream)
                done = torch.cuda.Event()
                done.record(self.comm_stream)          # grads-ready fence: comm -> compute
            self._done_events.append(done)
        else:
            self._timed(run)

    def _timed(self, fn) -> None:
        if self.on_cuda:
            tok = self.profiler.begin("rccl", None) if self.profiler else None
            fn()
            if tok is not None:
                self.profiler.end(tok, None)
        else:
            tok = self.profiler.begin("rccl", None) if self.profiler else None
            t0 = time.perf_counter()
            fn()
            self.comm_seconds += time.perf_counter() - t0
            if tok is not None:
                self.profiler.end(tok, None)

    def finish_grad_sync(self) -> None:
        """Make the compute stream wait for every outstanding bucket; launch stragglers
        (segments whose hooks did not all fire, e.g. unused parameters)."""
        if self._native is not None:
            self._native.finish()
            main = torch.cuda.current_stream(self.device)
            for ev in self._done_events:           # deferred buffer broadcast (after_forward)
                main.wait_event(ev)
            self._done_events.clear()
            return
        if self._armed:
            for k in range(len(self.arena.segments)):
                if not self._launched[k]:
                    self.segment_ready(k)
        if self.on_cuda:
            main = torch.cuda.current_stream(self.device)
            for ev in self._done_events:
                main.wait_event(ev)
        self._done_events.clear()
        self._reset_counters()

    def post_step(self) -> None:
        """ZeRO-1: all-gather the freshly updated parameter shards."""
        if not self.zero1 or self.single:
            return

        def run():
            for seg in self.arena.segments:
                full = self.arena.flat_param[seg.start:seg.end]
                n = (seg.end - seg.start) // self.world
                comm.all_gather_into(full, full[self.rank * n:(self.rank + 1) * n], self.group)
        self._timed(run)

    def gather_sharded(self, *flats: torch.Tensor) -> None:
        """ZeRO-1: every rank owns only its 1/W slice of each arena segment of a sharded buffer (Adam's
        ``exp_avg`` / ``exp_avg_sq``); all-gather the slices in place so that every rank holds the full buffer.
        Called before a snapshot (DCP de-duplicates replicated keys across DP ranks and would otherwise write
        one rank's mostly-zero copy) - the same pattern as ``post_step`` for the parameters."""
        if not self.zero1 or self.single:
            return
        for flat in flats:
            for seg in self.arena.segments:
                full = flat[seg.start:seg.end]
                n = (seg.end - seg.start) // self.world
                comm.all_gather_into(full, full[self.rank * n:(self.rank + 1) * n], self.group)

    # ------------------------------------------------------------------ metric helpers
    def gather(self, t: torch.Tensor) -> torch.Tensor:
        return comm.all_gather_flat(t, self.group) if self.world > 1 else t

    def mean(self, t: torch.Tensor) -> torch.Tensor:
        if self.world > 1:
            t = t.clone()
            comm.all_reduce(t, "avg", self.group)
        return t

"""hipGraph capture of the training step.

DenseNet-121 at the reference's batch sizes is launch-bound: ~800 kernels per step, many of
them 5-50 us.  Issued eagerly from Python the host cannot keep the GPU fed; captured once
into a hipGraph the whole step (gradient-arena clear, forward, fused softmax-CE, backward,
fused Adam) replays as ONE launch (the reference has nothing comparable - it runs eager
PyTorch; SURVEY 7.4 hard part 1 "few launches (HIP graphs ...)").

Protocol
  * the first ``warmup`` steps of every new input shape run eagerly (they are ordinary training
    steps on real data - they also let MIOpen / the caching allocator settle);
  * the next step is captured on a side stream into a private memory pool with static input
    buffers, then replayed; later steps copy the batch into the static buffers and replay;
  * multi-stream work inside the step (the native engine forks weight-gradient kernels onto a
    side stream) is captured as parallel graph branches;
  * data parallel (world > 1): RCCL collectives are kept OUT of the graphs, and the step is
    captured as a CHAIN of graphs - the backward pass is cut at unit boundaries ("phases":
    ``NativeDenseNetStage.forward_phased``).  Graph 0 = zero-grad + forward + loss + backward of the
    last units; graph i = backward of the next group of units.  After each graph is enqueued the
    gradient buckets it completed are handed to the comm stream (reduce-scatter + all-gather), so
    79 % of the gradient bytes (dense blocks 3-4) travel over xGMI while the FLOP-heavy early
    blocks are still in backward; only the last small bucket is exposed.  Adam runs eagerly after
    the grads-ready fence.
  * evaluation and any non-captured shape fall back to the eager path.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from .. import ops
from .trace import trace_range


class _Captured:
    __slots__ = ("graph", "x", "y", "loss", "out", "preds", "tail_graphs", "phase_segments")


def _step_takes_zero_grad(optimizer) -> bool:
    import inspect
    try:
        return "zero_grad" in inspect.signature(optimizer.step).parameters
    except (TypeError, ValueError):
        return False


class GraphedTrainStep:
    def __init__(self, model: torch.nn.Module, optimizer, dp=None, profiler=None, warmup: int = 2,
                 enabled: bool = True, fuse_zero_grad: bool = True) -> None:
        self.model, self.optimizer, self.dp, self.profiler = model, optimizer, dp, profiler
        self.warmup = warmup
        self.enabled = enabled
        # reference order: zero_grad -> forward -> backward -> step (single.py:137-154).  The fused Adam kernel can
        # clear the gradient arena in the pass that consumed it, which makes the 28 MB memset at the head of every
        # step redundant; off (explicit memset, gradients readable after the step) for --log-gradients
        self.fuse_zero_grad = (bool(fuse_zero_grad) and _step_takes_zero_grad(optimizer)
                               and getattr(optimizer, "shard", None) is None)      # ZeRO-1 steps touch their shard only
        self._grads_clean = False        # True once a fused step has left the arena zeroed
        self.distributed = dp is not None and not dp.single
        self._seen: Dict[Tuple, int] = {}
        self._graphs: Dict[Tuple, _Captured] = {}
        self._pool = None

    # ---- the step body ----------------------------------------------------------------------
    def _zero_grad(self) -> None:
        if not (self.fuse_zero_grad and self._grads_clean):
            self.optimizer.zero_grad()

    def _opt_step(self) -> None:
        if self.fuse_zero_grad:
            self.optimizer.step(zero_grad=True)
            self._grads_clean = True
        else:
            self.optimizer.step()

    def _fwd_bwd(self, x, y, after_forward=None):
        self._zero_grad()
        out = self.model(x)
        if after_forward is not None:
            after_forward()
        loss, preds = ops.cross_entropy_backward(out, y)
        return loss, out.detach(), preds

    def _eager(self, x, y):
        dp, prof = self.dp, self.profiler
        if dp is not None:
            dp.pre_forward(training=True)
            dp.arm(True)
        res = self._fwd_bwd(x, y, dp.after_forward if dp is not None else None)
        if dp is not None:
            tok = prof.begin("exposed_comm") if prof else None
            dp.finish_grad_sync()
            if prof:
                prof.end(tok)
        self._opt_step()
        if dp is not None:
            dp.post_step()
        return res

    def _phase_segments(self):
        """Which arena segments are complete after each backward phase (static table)."""
        from ..parallel.pp import phase_segments
        return phase_segments(self.model, self.dp.arena)

    def _check_joined(self) -> None:
        """Inside a capture, before it ends: every forked stream is back in the origin (native_densenet.fork)."""
        check = getattr(self.model, "assert_joined", None)
        if check is not None:
            check()

    def _capture(self, x, y) -> _Captured:
        c = _Captured()
        c.x, c.y = x.clone(), y.clone()
        c.tail_graphs, c.phase_segments = [], []
        dev = x.device
        if self.dp is not None:
            self.dp.arm(False)                 # no collectives while capturing
        torch.cuda.synchronize(dev)
        c.graph = torch.cuda.CUDAGraph()
        if self._pool is None:
            self._pool = torch.cuda.graph_pool_handle()
        phased = self.distributed and bool(getattr(self.model, "phase_starts", None))
        # thread-local capture mode: the RCCL watchdog thread polls events while we capture
        mode = dict(pool=self._pool, capture_error_mode="thread_local")
        if not phased:
            with torch.cuda.graph(c.graph, **mode):
                c.loss, c.out, c.preds = self._fwd_bwd(c.x, c.y)
                if not self.distributed:
                    self._opt_step()
                self._check_joined()
        else:
            with torch.cuda.graph(c.graph, **mode):
                self._zero_grad()
                out, cuts = self.model.forward_phased(c.x)
                c.loss, c.preds = ops.cross_entropy_backward(out, c.y)
                c.out = out.detach()
                self._check_joined()
            for y_mid, leaf in reversed(cuts):          # one graph per earlier phase
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, **mode):
                    y_mid.backward(leaf.grad)
                    self._check_joined()
                c.tail_graphs.append(g)
            c.phase_segments = self._phase_segments()
        if hasattr(self.optimizer, "_host_step") and not self.distributed:
            self.optimizer._host_step -= 1     # capture did not execute the step
        return c

    def _replay(self, c: _Captured, x, y):
        dp = self.dp
        ops.copy_inputs(c.x, x, c.y, y)
        if dp is not None:
            dp.pre_forward(training=True)
        if self.distributed:
            dp.arm(True)
        with trace_range("graph 0: forward + backward tail"):
            c.graph.replay()
        if dp is not None:
            dp.after_forward()                 # graph 0 holds the whole forward: buffers are final behind it
        for i, g in enumerate(c.tail_graphs):
            for k in c.phase_segments[i]:      # buckets completed by the graph just enqueued
                dp.segment_ready(k)
            with trace_range(f"graph {i + 1}: backward phase"):
                g.replay()
        if self.distributed:
            tok = self.profiler.begin("exposed_comm") if self.profiler else None
            with trace_range("gradient sync fence"):
                dp.finish_grad_sync()          # launches the remaining buckets, then fences
            if self.profiler:
                self.profiler.end(tok)
            with trace_range("adam"):
                self._opt_step()
            dp.post_step()
        elif hasattr(self.optimizer, "_host_step"):
            self.optimizer._host_step += 1
        return c.loss, c.out, c.preds

    # ---- public -----------------------------------------------------------------------------
    def __call__(self, x: torch.Tensor, y: torch.Tensor):
        prof = self.profiler
        tok = prof.begin("step") if prof else None
        use_graph = self.enabled and x.is_cuda and self.model.training
        if not use_graph:
            res = self._eager(x, y)
        else:
            key = (tuple(x.shape), x.dtype, tuple(y.shape))
            c = self._graphs.get(key)
            if c is None:
                n = self._seen.get(key, 0)
                self._seen[key] = n + 1
                if n < self.warmup:
                    res = self._eager(x, y)
                else:
                    c = self._capture(x, y)
                    self._graphs[key] = c
                    res = self._replay(c, x, y)
            else:
                res = self._replay(c, x, y)
        if prof:
            prof.end(tok)
            prof.step_done()
        return res


class GraphedEvalStep:
    """Inference forward of the native engine replayed as one hipGraph per input shape (the evaluation loop is
    as launch-bound as training: ~250 kernels per batch).  The first batch of a shape runs eagerly, the second is
    captured; the static output buffer is cloned on return (callers keep every batch's logits)."""

    def __init__(self, model: torch.nn.Module, enabled: bool = True) -> None:
        self.model, self.enabled = model, enabled
        self._seen: Dict[Tuple, int] = {}
        self._graphs: Dict[Tuple, Tuple] = {}
        self._pool = None

    @torch.no_grad()
    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if not (self.enabled and x.is_cuda and not self.model.training):
            return self.model(x)
        key = (tuple(x.shape), x.dtype)
        g = self._graphs.get(key)
        if g is None:
            n = self._seen.get(key, 0)
            self._seen[key] = n + 1
            if n < 1:
                return self.model(x)
            sx = x.clone()
            torch.cuda.synchronize(x.device)
            graph = torch.cuda.CUDAGraph()
            if self._pool is None:
                self._pool = torch.cuda.graph_pool_handle()
            with torch.cuda.graph(graph, pool=self._pool, capture_error_mode="thread_local"):
                out = self.model(sx)
            g = self._graphs[key] = (graph, sx, out)
        graph, sx, out = g
        sx.copy_(x, non_blocking=True)
        graph.replay()
        return out.clone()

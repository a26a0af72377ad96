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
  * data parallel (world > 1): RCCL collectives are kept OUT of the graph - the captured part is
    zero-grad + forward + backward, then the bucket reduce-scatter / all-gather run on the comm
    stream and Adam runs eagerly.  (Per-bucket overlap with backward is what the eager path
    does; see ``DataParallelEngine``.)
  * evaluation and any non-captured shape fall back to the eager path.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import torch

from .. import ops


class _Captured:
    __slots__ = ("graph", "x", "y", "loss", "out", "preds")


class GraphedTrainStep:
    def __init__(self, model: torch.nn.Module, optimizer, dp=None, profiler=None, warmup: int = 2,
                 enabled: bool = True) -> None:
        self.model, self.optimizer, self.dp, self.profiler = model, optimizer, dp, profiler
        self.warmup = warmup
        self.enabled = enabled
        self.distributed = dp is not None and dp.world > 1
        self._seen: Dict[Tuple, int] = {}
        self._graphs: Dict[Tuple, _Captured] = {}
        self._pool = None

    # ---- the step body ----------------------------------------------------------------------
    def _fwd_bwd(self, x, y):
        self.optimizer.zero_grad()
        out = self.model(x)
        loss, preds = ops.cross_entropy_with_preds(out, y)
        loss.backward()
        return loss.detach(), out.detach(), preds

    def _eager(self, x, y):
        dp, prof = self.dp, self.profiler
        if dp is not None:
            dp.pre_forward()
            dp.arm(True)
        res = self._fwd_bwd(x, y)
        if dp is not None:
            tok = prof.begin("exposed_comm") if prof else None
            dp.finish_grad_sync()
            if prof:
                prof.end(tok)
        self.optimizer.step()
        if dp is not None:
            dp.post_step()
        return res

    def _capture(self, x, y) -> _Captured:
        c = _Captured()
        c.x, c.y = x.clone(), y.clone()
        dev = x.device
        if self.dp is not None:
            self.dp.arm(False)                 # no collectives while capturing
        torch.cuda.synchronize(dev)
        c.graph = torch.cuda.CUDAGraph()
        if self._pool is None:
            self._pool = torch.cuda.graph_pool_handle()
        # thread-local capture mode: the RCCL watchdog thread polls events while we capture
        with torch.cuda.graph(c.graph, pool=self._pool, capture_error_mode="thread_local"):
            c.loss, c.out, c.preds = self._fwd_bwd(c.x, c.y)
            if not self.distributed:
                self.optimizer.step()
        if hasattr(self.optimizer, "_host_step") and not self.distributed:
            self.optimizer._host_step -= 1     # capture did not execute the step
        return c

    def _replay(self, c: _Captured, x, y):
        dp = self.dp
        c.x.copy_(x, non_blocking=True)
        c.y.copy_(y, non_blocking=True)
        if dp is not None:
            dp.pre_forward()
        c.graph.replay()
        if self.distributed:
            dp.arm(True)
            tok = self.profiler.begin("exposed_comm") if self.profiler else None
            dp.finish_grad_sync()              # launches every bucket (none was launched in the graph)
            if self.profiler:
                self.profiler.end(tok)
            self.optimizer.step()
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

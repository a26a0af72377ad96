"""Per-step compute vs communication breakdown + the communication micro-benchmarks.

The reference measures wall-clock per epoch (``perf_counter`` around ``_run_epoch`` with no
device sync, ddp.py:189-191) and a single P2P ping (communication_time.py:17-44); it has no
compute/comm split at all - the author infers comm cost from the ping.  The
rebuild's profiler is event based:

  * ``StepProfiler``: named spans bracketed by HIP events **on the stream that executes
    them** (compute stream, DP comm stream, P2P waits), accumulated per step and resolved once
    at ``collect()`` - no host sync in the hot loop.  ``rccl`` spans sit on the comm stream, so
    ``rccl_ms`` is RCCL busy time; ``exposed_comm_ms`` is how long the compute stream actually
    stalled on the grads-ready fence; ``step_ms`` is the whole step.  On CPU (gloo tests) the
    same API falls back to ``perf_counter``.
  * ``ping`` / ``sweep_p2p`` / ``sweep_allreduce``: the reference ping (4 MiB fp32 send + 1-float
    ack, event-timed, same CSV row ``job_id,i,ms``) generalised to a message-size sweep for
    send/recv and all-reduce (BASELINE.json config 5).
"""
from __future__ import annotations

import time
from collections import defaultdict
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from . import comm


class StepProfiler:
    def __init__(self, device, enabled: bool = True) -> None:
        self.device = torch.device(device)
        self.enabled = enabled
        self.cuda = self.device.type == "cuda"
        self._open: List[Tuple[str, object, object]] = []       # (name, start, end) events / times
        self.totals_ms: Dict[str, float] = defaultdict(float)
        self.counts: Dict[str, int] = defaultdict(int)
        self.steps = 0
        self._sources: List[Tuple[str, object]] = []             # (name, fn() -> ms since the last call)
        self._switches: List[object] = []

    def add_source(self, name: str, fn, set_enabled=None) -> None:
        """A span total measured elsewhere (the native bucket reducer times its RCCL calls itself);
        ``set_enabled(bool)`` switches that measurement together with this profiler."""
        if self.enabled:
            self._sources.append((name, fn))
            if set_enabled is not None:
                self._switches.append(set_enabled)

    def set_enabled(self, on: bool) -> None:
        """Pause / resume span recording (bench.py times its steps with the profiler paused and
        measures the compute / RCCL split on extra steps afterwards)."""
        self.collect()
        self.enabled = on
        for sw in self._switches:
            sw(on)

    # ---- spans -----------------------------------------------------------------------
    def begin(self, name: str, stream=None):
        if not self.enabled:
            return None
        if self.cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(stream if stream is not None else torch.cuda.current_stream(self.device))
            return (name, ev)
        return (name, time.perf_counter())

    def end(self, token, stream=None) -> None:
        if token is None:
            return
        name, start = token
        if self.cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(stream if stream is not None else torch.cuda.current_stream(self.device))
            self._open.append((name, start, ev))
        else:
            self.totals_ms[name] += (time.perf_counter() - start) * 1e3
            self.counts[name] += 1

    def span(self, name: str, stream=None):
        prof = self

        class _Span:
            def __enter__(self_inner):
                self_inner.tok = prof.begin(name, stream)

            def __exit__(self_inner, *exc):
                prof.end(self_inner.tok, stream)
        return _Span()

    def step_done(self) -> None:
        if self.enabled:
            self.steps += 1

    # ---- results ---------------------------------------------------------------------
    def collect(self) -> None:
        for name, fn in self._sources:
            ms = float(fn())
            if ms > 0:
                self.totals_ms[name] += ms
                self.counts[name] += 1
        if self.cuda and self._open:
            torch.cuda.synchronize(self.device)
            for name, s, e in self._open:
                self.totals_ms[name] += s.elapsed_time(e)
                self.counts[name] += 1
            self._open.clear()

    def summary(self) -> Dict[str, float]:
        self.collect()
        n = max(self.steps, 1)
        out = {f"{k}_ms_per_step": v / n for k, v in self.totals_ms.items()}
        step = self.totals_ms.get("step", 0.0)
        if step > 0:
            busy = self.totals_ms.get("rccl", 0.0) + self.totals_ms.get("p2p", 0.0)
            exposed = self.totals_ms.get("exposed_comm", 0.0) + self.totals_ms.get("p2p_wait", 0.0)
            out["comm_busy_fraction"] = busy / step
            out["comm_exposed_fraction"] = exposed / step
            out["compute_fraction"] = max(0.0, 1.0 - exposed / step)
        out["steps"] = self.steps
        return out

    def reset(self) -> None:
        self.collect()
        self.totals_ms.clear()
        self.counts.clear()
        self.steps = 0


# --------------------------------------------------------------------------------------
# communication micro-benchmarks
# --------------------------------------------------------------------------------------
def _timer(device):
    if torch.device(device).type == "cuda":
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def start():
            s.record()

        def stop() -> float:
            e.record()
            torch.cuda.synchronize()
            return s.elapsed_time(e)
        return start, stop
    t = [0.0]

    def start():
        t[0] = time.perf_counter()

    def stop() -> float:
        return (time.perf_counter() - t[0]) * 1e3
    return start, stop


def ping(device, numel: int = 1024 * 1024, peer_a: int = 0, peer_b: int = 1, group=None,
         payload: Optional[torch.Tensor] = None, ack: Optional[torch.Tensor] = None) -> Optional[float]:
    """One reference ping: rank a sends ``numel`` fp32 to rank b, b returns a 1-float ack.
    Returns milliseconds on rank a, None elsewhere (communication_time.py:17-44).  Unlike the
    reference the buffers can be passed in (it allocates + randn's inside the timed function)."""
    rank = comm.get_rank()
    if payload is None:
        payload = torch.randn(numel, device=device) if rank == peer_a else torch.zeros(numel, device=device)
    if ack is None:
        ack = torch.zeros(1, device=device)
    if rank == peer_a:
        start, stop = _timer(device)
        start()
        dist.send(payload, dst=peer_b, group=group)
        dist.recv(ack, src=peer_b, group=group)
        return stop()
    if rank == peer_b:
        dist.recv(payload, src=peer_a, group=group)
        dist.send(ack, dst=peer_a, group=group)
        if torch.device(device).type == "cuda":
            torch.cuda.synchronize()
    return None


def sweep_p2p(device, sizes_bytes: List[int], iters: int = 50, warmup: int = 5) -> List[Dict]:
    """Round-trip latency and one-way bandwidth vs message size between ranks 0 and 1."""
    rows = []
    rank = comm.get_rank()
    for nbytes in sizes_bytes:
        numel = max(nbytes // 4, 1)
        payload = torch.randn(numel, device=device) if rank == 0 else torch.zeros(numel, device=device)
        ack = torch.zeros(1, device=device)
        ts = []
        for i in range(warmup + iters):
            ms = ping(device, numel, payload=payload, ack=ack)
            if ms is not None and i >= warmup:
                ts.append(ms)
        if rank == 0:
            ts.sort()
            med = ts[len(ts) // 2]
            rows.append(dict(op="send_recv", bytes=numel * 4, ms_median=med, ms_min=ts[0],
                             gbps=numel * 4 / (med * 1e-3) / 1e9))
    return rows


def sweep_allreduce(device, sizes_bytes: List[int], iters: int = 30, warmup: int = 5, group=None) -> List[Dict]:
    """All-reduce (and reduce-scatter + all-gather) time and bus bandwidth vs message size."""
    rows = []
    w = comm.get_world_size(group)
    for nbytes in sizes_bytes:
        numel = max(nbytes // 4 // max(w * 64, 1), 1) * max(w * 64, 1)
        buf = torch.randn(numel, device=device)
        for op in ("all_reduce", "rs_ag"):
            ts = []
            for i in range(warmup + iters):
                comm.barrier(group)
                start, stop = _timer(device)
                start()
                if op == "all_reduce":
                    comm.all_reduce(buf, "sum", group)
                else:
                    n = numel // w
                    r = comm.get_rank(group)
                    shard = buf[r * n:(r + 1) * n]
                    comm.reduce_scatter_into(shard, buf, "sum", group)
                    comm.all_gather_into(buf, shard, group)
                ms = stop()
                if i >= warmup:
                    ts.append(ms)
                buf.mul_(1.0 / max(w, 1))          # keep magnitudes bounded
            ts.sort()
            med = ts[len(ts) // 2]
            busbw = (2 * (w - 1) / max(w, 1)) * numel * 4 / (med * 1e-3) / 1e9 if w > 1 else 0.0
            rows.append(dict(op=op, bytes=numel * 4, ms_median=med, ms_min=ts[0], busbw_gbps=busbw,
                             world=w))
    return rows

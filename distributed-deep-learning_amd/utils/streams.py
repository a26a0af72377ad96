"""Side streams that are really separate streams.

``torch.cuda.Stream()`` does not create a stream: it hands out the next of a per-device pool of 32, round-robin, and
``torch.cuda.graph`` captures on ONE class-level stream drawn from the same pool.  In a long-lived process (the test
suite is one) every 32nd stream drawn is therefore the capture stream itself, and two engines can be handed the same
stream.  Nothing breaks numerically when that happens - work just serialises - except a fork from a stream into
itself during capture.  ``new_stream`` draws until the stream differs from the capture stream (when torch has
drawn it already) and from every stream the caller names."""
from __future__ import annotations

from typing import Iterable, Optional

import torch


def capture_stream() -> Optional["torch.cuda.Stream"]:
    """The stream ``torch.cuda.graph`` captures on by default, if torch has drawn it already.  It is NOT created
    here: torch draws it on the device that is current at the first capture, and drawing it earlier - before a rank
    has selected its device - would pin every later capture to the wrong GPU.  A stage whose side stream turns out
    to be the capture stream drawn later redraws at its first fork (``NativeDenseNetStage.fork``)."""
    return getattr(torch.cuda.graph, "default_capture_stream", None)


def new_stream(device, avoid: Iterable = (), priority: int = 0) -> "torch.cuda.Stream":
    taken = [s for s in avoid if s is not None]
    cap = capture_stream()
    if cap is not None:
        taken.append(cap)
    for _ in range(64):
        s = torch.cuda.Stream(device, priority=priority)
        if all(s != t for t in taken):
            return s
    raise RuntimeError("no stream distinct from %d others could be drawn from the stream pool" % len(taken))


# ---- streams that really run side by side --------------------------------------------------------------------------
# A HIP process owns FEW hardware queues (GPU_MAX_HW_QUEUES, 4 by default - and raising it makes things worse: with 8 or
# 16 queues the pipes time-slice them and the lanes' step went from 19 to 37-54 ms, profiles/r3/lanes_hw_queues.txt), and
# the runtime maps every stream onto one of them.  Two streams that share a hardware queue are SERIALISED on the
# device, whatever the events between them say: the first lanes build had its forward stream and one backward
# stream on the same queue and showed no overlap at all in the kernel trace.  Which stream lands on which queue
# depends on what the process created before (RCCL, prefetchers, ...), so it is measured: a spin kernel on each
# of two streams takes one kernel time when the streams are concurrent and two when they share a queue.
_SPIN_CYCLES = None


def _spin_cycles(device) -> int:
    """Cycles of ``torch.cuda._sleep`` for a ~150 us spin on this device (calibrated once)."""
    global _SPIN_CYCLES
    if _SPIN_CYCLES is None:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.device(device):
            torch.cuda._sleep(100_000)
            torch.cuda.synchronize(device)
            s.record()
            torch.cuda._sleep(1_000_000)
            e.record()
            torch.cuda.synchronize(device)
        ms = max(s.elapsed_time(e), 1e-3)
        _SPIN_CYCLES = max(10_000, int(1_000_000 * 0.15 / ms))
    return _SPIN_CYCLES


def run_concurrently(a: "torch.cuda.Stream", b: "torch.cuda.Stream") -> bool:
    """True if kernels on ``a`` and ``b`` overlap on the device (distinct hardware queues)."""
    import time
    dev = a.device
    n = _spin_cycles(dev)

    def both(x, y) -> float:
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        with torch.cuda.stream(x):
            torch.cuda._sleep(n)
        with torch.cuda.stream(y):
            torch.cuda._sleep(n)
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0

    both(a, b)
    serial = min(both(a, a) for _ in range(2))          # two spins on ONE stream: the serialised reference
    pair = min(both(a, b) for _ in range(2))
    return pair < 0.75 * serial


def new_stream_beside(device, beside: Iterable, avoid: Iterable = (), tries: int = 8) -> "torch.cuda.Stream":
    """A stream whose kernels overlap with those of every stream in ``beside`` (a communication / P2P stream must
    not share a hardware queue with the compute stream, or its RCCL kernels queue up behind the compute kernels
    whatever the events say).  Falls back to the last candidate when none of ``tries`` qualifies."""
    beside = [b for b in beside if b is not None]
    taken = [s for s in avoid if s is not None] + beside
    cand = None
    for _ in range(max(1, tries)):
        cand = new_stream(device, avoid=taken)
        if all(run_concurrently(cand, b) for b in beside):
            return cand
        taken.append(cand)
    return cand


def concurrent_streams(device, n: int, avoid: Iterable = ()) -> "list[torch.cuda.Stream]":
    """``n`` streams that pairwise run side by side on the device, drawn from torch's pool (up to 24 candidates are
    probed).  Falls back to plain distinct streams for the ones that cannot be found (fewer free hardware queues
    than lanes): the result is then correct but partly serialised."""
    chosen: "list[torch.cuda.Stream]" = []
    taken = [s for s in avoid if s is not None]
    tried = 0
    while len(chosen) < n and tried < 24:
        s = new_stream(device, avoid=taken + chosen)
        tried += 1
        if all(run_concurrently(s, c) for c in chosen):
            chosen.append(s)
        else:
            taken.append(s)
    while len(chosen) < n:
        chosen.append(new_stream(device, avoid=chosen))
    return chosen

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

"""Timeline tracing: named ranges + an optional profiler capture.

The reference records wall-clock per epoch only; the pipelining library it uses emits
``record_function("Forward i") / ("Backward i")`` ranges that nothing collects (SURVEY 5.1,
torch: distributed/pipelining/schedules.py:757,785).  Here the same ranges exist for every schedule
step, for the phases of a data-parallel step and for epochs / evaluation, and they can be collected:

  * ``--trace-dir DIR`` (or ``DDL_TRACE_DIR``) records ``--trace-steps`` training steps of the first
    epoch (after ``--trace-skip`` warm-up steps) with ``torch.profiler`` - host ranges plus the GPU
    kernel timeline through roctracer on ROCm - and writes a Chrome trace ``trace_rank{r}.json`` and a
    per-kernel summary table ``kernels_rank{r}.txt``;
  * under ``rocprofv3 --marker-trace`` / ``--kernel-trace`` the ranges show up as roctx markers
    (``DDL_TRACE=1`` enables them without the profiler capture).

Ranges cost two Python calls when tracing is off.
"""
from __future__ import annotations

import os
from contextlib import contextmanager
from typing import Optional

import torch

_enabled = os.environ.get("DDL_TRACE", "0") == "1"


def enable(on: bool = True) -> None:
    global _enabled
    _enabled = on


def enabled() -> bool:
    return _enabled


@contextmanager
def trace_range(name: str):
    """Named range: a ``record_function`` span (torch.profiler) and, on a GPU, a roctx range."""
    if not _enabled:
        yield
        return
    gpu = torch.cuda.is_available()
    if gpu:
        try:
            torch.cuda.nvtx.range_push(name)        # roctx on ROCm builds
        except Exception:                            # noqa: BLE001 - marker library not present
            gpu = False
    try:
        with torch.profiler.record_function(name):
            yield
    finally:
        if gpu:
            torch.cuda.nvtx.range_pop()


class StepTrace:
    """Profiler capture of a window of training steps: ``tick()`` once per step."""

    def __init__(self, out_dir: Optional[str], rank: int = 0, skip: int = 3, steps: int = 5) -> None:
        self.out_dir, self.rank, self.skip, self.steps = out_dir, rank, skip, steps
        self._seen = 0
        self._prof = None
        self.done = out_dir is None

    def tick(self) -> None:
        if self.done:
            return
        if self._seen == self.skip:
            enable(True)
            acts = [torch.profiler.ProfilerActivity.CPU]
            if torch.cuda.is_available():
                acts.append(torch.profiler.ProfilerActivity.CUDA)
            try:
                self._prof = torch.profiler.profile(activities=acts, record_shapes=False)
                self._prof.__enter__()
            except Exception:                        # noqa: BLE001 - no GPU tracer in this build: host ranges only
                self._prof = torch.profiler.profile(activities=acts[:1], record_shapes=False)
                self._prof.__enter__()
        elif self._seen == self.skip + self.steps:
            self.close()
        self._seen += 1

    def close(self) -> None:
        if self._prof is None or self.done:
            self.done = True
            return
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self._prof.__exit__(None, None, None)
        os.makedirs(self.out_dir, exist_ok=True)
        self._prof.export_chrome_trace(os.path.join(self.out_dir, f"trace_rank{self.rank}.json"))
        sort = "self_cuda_time_total" if torch.cuda.is_available() else "self_cpu_time_total"
        with open(os.path.join(self.out_dir, f"kernels_rank{self.rank}.txt"), "w") as f:
            f.write(self._prof.key_averages().table(sort_by=sort, row_limit=40))
        self._prof = None
        self.done = True
        enable(os.environ.get("DDL_TRACE", "0") == "1")

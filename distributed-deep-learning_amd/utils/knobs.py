"""Experiment switches (``DDL_*`` environment variables of the engine).

The defaults are the measured optimum and every alternative has a number in README.md / ``profiles/`` - the
variables exist so that those A/B runs can be repeated, not to configure production runs.  They are therefore
honoured only under ``DDL_EXPERIMENTAL=1``: a stray ``DDL_FUSED_BWD=1`` in somebody's shell never changes the code
path (a one-line notice on stderr says it was ignored), and ``bench.py`` lists every ``DDL_*`` variable it saw in
its JSON (``experimental_env``).  The C++ side uses the same gate (``csrc/ddl_kernels.h: ddl_knob``).

Switches that produce WRONG results (kernel knock-outs) are not knobs at all: they live in
``bench/experiments/knockout.py`` and the production entrypoints refuse to start with ``DDL_KNOCKOUT`` set
(``ddl/diagnostics.py``).
"""
from __future__ import annotations

import os
import sys

_WARNED = set()


def experimental() -> bool:
    return os.environ.get("DDL_EXPERIMENTAL", "0") == "1"


def knob(name: str, default: str) -> str:
    """Value of the experiment switch ``name`` (a string, like ``os.environ.get``) - the default unless
    ``DDL_EXPERIMENTAL=1``."""
    v = os.environ.get(name)
    if v is None:
        return default
    if experimental():
        return v
    if name not in _WARNED and v != default:
        _WARNED.add(name)
        print(f"[ddl] {name}={v} ignored: experiment switches need DDL_EXPERIMENTAL=1 (default {default!r} in effect)",
              file=sys.stderr, flush=True)
    return default

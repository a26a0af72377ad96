#!/usr/bin/env python3
"""pipeline parallel, 1F1B/GPipe (reference: pp.py).

Thin entrypoint kept under the reference's file name; the implementation is `ddl.cli.main_pp`.
Run: `python pp.py [--flags]`, or under `torchrun` / the torchx-style component for multi-GPU.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from ddl.cli import main_pp  # noqa: E402

if __name__ == "__main__":
    main_pp()

#!/usr/bin/env python3
"""single-device baseline (reference: single.py).

Thin entrypoint kept under the reference's file name; the implementation is `ddl.cli.main_single`.
Run: `python single.py [--flags]`, or under `torchrun` / the torchx-style component for multi-GPU.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from ddl.cli import main_single  # noqa: E402

if __name__ == "__main__":
    main_single()

#!/usr/bin/env python3
"""data parallel (reference: ddp.py).

Thin entrypoint kept under the reference's file name; the implementation is `ddl.cli.main_ddp`.
Run: `python ddp.py [--flags]`, or under `torchrun` / the torchx-style component for multi-GPU.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from ddl.cli import main_ddp  # noqa: E402

if __name__ == "__main__":
    main_ddp()

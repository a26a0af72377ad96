#!/usr/bin/env python3
"""hybrid 2-D data x pipeline parallel (reference: ddp_n_pp.py).

Thin entrypoint kept under the reference's file name; the implementation is `ddl.cli.main_ddp_n_pp`.
Run: `python ddp_n_pp.py [--flags]`, or under `torchrun` / the torchx-style component for multi-GPU.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from ddl.cli import main_ddp_n_pp  # noqa: E402

if __name__ == "__main__":
    main_ddp_n_pp()

#!/usr/bin/env python3
"""process-group / 2-D mesh smoke script (reference: test.py - not a pytest file).

Thin entrypoint kept under the reference's file name; the implementation is `ddl.cli.main_mesh_smoke`.
Run: `python test.py [--flags]`, or under `torchrun` / the torchx-style component for multi-GPU.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from ddl.cli import main_mesh_smoke  # noqa: E402

if __name__ == "__main__":
    main_mesh_smoke()

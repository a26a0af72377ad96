#!/usr/bin/env python3
"""P2P ping + message-size sweep (reference: communication_time.py).

Thin entrypoint kept under the reference's file name; the implementation is `ddl.cli.main_communication_time`.
Run: `python communication_time.py [--flags]`, or under `torchrun` / the torchx-style component for multi-GPU.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from ddl.cli import main_communication_time  # noqa: E402

if __name__ == "__main__":
    main_communication_time()

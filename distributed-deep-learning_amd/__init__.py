"""MI355X-native distributed training harness (DenseNet-121 / APTOS workload).

Layers:
  config / data / metrics / checkpoint      host-side plumbing
  models/                                   DenseNet-121 (torch reference + native HIP engine)
  ops/  csrc/                               hand-written gfx950 HIP kernels and their wrappers
  comm / parallel/                          RCCL process groups, DP, PP (GPipe + 1F1B), 2-D hybrid
  optim / trainer / profiler                fused flat Adam, the one Trainer, compute/comm profiler
"""
__version__ = "0.1.0"

import os as _os

# Multi-process GPU work (RCCL, tensors shared across processes) needs dmabuf IPC handles on hosts whose
# driver has no legacy IPC; must be in the environment before the HIP runtime starts.  An explicit
# setting wins.
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

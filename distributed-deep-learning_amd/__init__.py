"""MI355X-native distributed training harness (DenseNet-121 / APTOS workload).

Layers:
  config / data / metrics / checkpoint      host-side plumbing
  models/                                   DenseNet-121 (torch reference + native HIP engine)
  ops/  csrc/                               hand-written gfx950 HIP kernels and their wrappers
  comm / parallel/                          RCCL process groups, DP, PP (GPipe + 1F1B), 2-D hybrid
  optim / trainer / profiler                fused flat Adam, the one Trainer, compute/comm profiler
"""
__version__ = "0.1.0"

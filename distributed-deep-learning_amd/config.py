"""One config object + argparse for every entrypoint.

The reference has no flag system at all: every knob is a literal inside
``main()`` (reference: single.py:272-309, ddp.py:329-368, pp.py:342-398,
ddp_n_pp.py:365-426, communication_time.py:18,57).  Here the same literals are
the *defaults* of a dataclass, each script overrides the handful that differ
(``SCRIPT_DEFAULTS``), and everything is overridable from the command line or
the environment (``DDL_DATA_DIR`` / ``DDL_LOG_DIR`` / ``DDL_CKPT_DIR``) because
``/mnt/dcornelius`` does not exist outside the reference's cluster.
"""
from __future__ import annotations

import argparse
import dataclasses
import os
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional

REFERENCE_ROOT = "/mnt/dcornelius"


def _env_path(var: str, default: str) -> str:
    return os.environ.get(var, default)


@dataclass
class TrainConfig:
    # ---- identity / strategy -------------------------------------------------
    strategy: str = "single"            # single | ddp | pp | ddpnpp
    # ---- data (reference: single.py:276-295) ---------------------------------
    data_dir: str = field(default_factory=lambda: _env_path(
        "DDL_DATA_DIR", f"{REFERENCE_ROOT}/preprocessed-aptos-224"))
    synthetic: str = "auto"             # auto | on | off  (auto: on when data_dir is missing)
    train_samples: int = 2400           # samples/epoch recovered from the notebook
    eval_samples: int = 360
    image_size: int = 224
    num_classes: int = 5
    batch_size: int = 30                # per-process batch (single 30, ddp 15, pp 30, ddpnpp 10)
    num_workers: int = 2                # decode threads of the native PNG loader / DataLoader worker processes
    native_loader: str = "auto"         # auto | off: C++ threaded PNG decode into pinned batches (csrc/png_loader.cpp)
    device_data: bool = False           # generate synthetic batches directly on the GPU (benchmarks)
    cache_dataset: str = "auto"         # auto | device | off: decode once, keep the uint8 dataset in HBM (data.ResidentLoader)
    cache_budget_gb: float = 16.0       # auto: largest dataset (GB of uint8) kept resident
    eval_all: bool = False              # reference drops the eval tail (ddp.py:353); keep for parity
    # ---- optimisation (reference: optim.Adam defaults, single.py:108,305) ----
    epochs: int = 30
    lr: float = 1e-3
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-8
    weight_decay: float = 0.0
    seed: Optional[int] = None          # reference defines set_seed but never calls it
    max_steps: int = 0                  # >0: stop each epoch after this many steps (smoke runs)
    # ---- model ---------------------------------------------------------------
    dtype: str = "fp32"                 # fp32 (reference parity) | bf16 (mixed: bf16 MFMA inputs)
    hip_graph: bool = True              # capture the native train step into a hipGraph
    pretrained: Optional[str] = field(default_factory=lambda: os.environ.get("DDL_PRETRAINED") or None)
    #   ^ torchvision densenet121 state-dict file (reference: weights=IMAGENET1K_V1, single.py:297)
    # ---- data parallel -------------------------------------------------------
    bucket_mb: float = 8.0              # gradient bucket size (MiB); sized for xGMI, not NVSwitch
    grad_sync: str = "auto"             # rs_ag (reduce-scatter + all-gather: both run directly over all 7 xGMI links of a
    #                                     fully connected node; the north-star form, also what --zero1 needs) | allreduce
    #                                     (one RCCL launch per bucket, what the reference's DDP does) | auto: allreduce
    #                                     for a dp group of 2 (two launches cannot beat one over a single link), rs_ag
    #                                     from 3 ranks on.  No multi-GPU measurement backs either choice yet ("parity
    #                                     unpinned"): bench.py at W>1 times both on the same step ("grad_sync_ab") and
    #                                     `--grad-sync tune` keeps the faster one.
    rccl: str = "auto"                  # auto | direct | c10d: owned RCCL communicators enqueued on the engines' own HIP
    #                                     streams (csrc/rccl_comm.cpp) vs. everything through c10d's process group
    zero1: bool = False                 # shard the Adam step over the dp group (RS grads, AG params)
    sync_buffers: str = "forward"       # forward (DDP parity) | epoch | off
    grad_phases: str = "denseblock3,denseblock2"   # units that open a new backward graph phase (DP overlap)
    # ---- pipeline parallel (reference: pp.py:378-386) ------------------------
    num_microbatches: int = 5
    schedule: str = "1f1b"              # 1f1b | gpipe
    pp_lanes: int = 2                   # concurrent micro-batch lanes per stage (parallel/pp.py "lanes"): backward of
    #                                     micro-batch i next to the forward of i+k on separate streams / graph pools
    pp_fwd_lanes: int = 2               # forward streams when pp_lanes > 1 (BN running statistics handed off in
    #                                     micro-batch order through per-micro-batch scratch copies)
    split_points: List[str] = field(default_factory=lambda: ["features.denseblock3.denselayer1"])
    # ---- 2-D mesh (reference: ddp_n_pp.py:32-33) -----------------------------
    dp_size: int = 0                    # 0: world_size // pp_size
    pp_size: int = 2
    # ---- logging / checkpoint ------------------------------------------------
    log_dir: str = field(default_factory=lambda: _env_path(
        "DDL_LOG_DIR", f"{REFERENCE_ROOT}/training_logs"))
    ckpt_dir: str = field(default_factory=lambda: _env_path(
        "DDL_CKPT_DIR", f"{REFERENCE_ROOT}/checkpoints"))
    snapshot_job_id: Optional[str] = None
    snapshot_epoch: Optional[int] = None
    save_best: bool = False             # save when validation QWK improves (ddp.py:292-295)
    save_every: int = 0                 # save every N epochs (0 = never)
    resume_latest: bool = False
    log_gradients: bool = False         # opt-in version of the dead _log_gradient (ddp.py:310-326)
    # ---- runtime -------------------------------------------------------------
    backend: str = "auto"               # auto | nccl | gloo
    device: str = "auto"                # auto | cuda | cpu
    timeout_s: int = 600
    profile: bool = False               # per-step compute vs communication breakdown
    debug_sync: bool = False            # serialise comm streams (race hunting)
    check_collectives: bool = False     # collective-order checker (hash all-gather per step)
    trace_dir: Optional[str] = field(default_factory=lambda: os.environ.get("DDL_TRACE_DIR") or None)
    #   ^ torch.profiler capture (Chrome trace + kernel table) of trace_steps steps after trace_skip (utils/trace.py)
    trace_skip: int = 3
    trace_steps: int = 5
    fault_inject: str = ""              # "rank:epoch:step" - raise there (failure-path tests: abort + --resume-latest)

    def replace(self, **kw: Any) -> "TrainConfig":
        return dataclasses.replace(self, **kw)

    @property
    def ckpt_subdir(self) -> str:
        # reference: CHECKPOINT_DIR = /mnt/dcornelius/checkpoints/{single|ddp|pp|ddpnpp}
        return os.path.join(self.ckpt_dir, self.strategy)


# Per-script literals of the reference.
SCRIPT_DEFAULTS: Dict[str, Dict[str, Any]] = {
    "single": dict(strategy="single", batch_size=30),                        # single.py:286
    "ddp": dict(strategy="ddp", batch_size=15),                              # ddp.py:335
    "pp": dict(strategy="pp", batch_size=30, num_microbatches=5),            # pp.py:365,378
    "ddpnpp": dict(strategy="ddpnpp", batch_size=10, num_microbatches=5,     # ddp_n_pp.py:371,404
                   pp_size=2),
}


def _str2bool(v: str) -> bool:
    if isinstance(v, bool):
        return v
    if v.lower() in ("1", "true", "yes", "on", "y"):
        return True
    if v.lower() in ("0", "false", "no", "off", "n"):
        return False
    raise argparse.ArgumentTypeError(f"not a boolean: {v!r}")


def build_parser(script: str) -> argparse.ArgumentParser:
    base = TrainConfig(**SCRIPT_DEFAULTS[script])
    p = argparse.ArgumentParser(
        prog=f"{script}.py",
        description=f"MI355X-native {script} trainer (defaults = the reference's literals)",
    )
    for f in dataclasses.fields(TrainConfig):
        default = getattr(base, f.name)
        flag = "--" + f.name.replace("_", "-")
        if f.name == "split_points":
            p.add_argument(flag, nargs="+", default=default,
                           help="module names whose BEGINNING starts a new pipeline stage")
        elif isinstance(default, bool):
            p.add_argument(flag, type=_str2bool, nargs="?", const=True, default=default)
        elif f.name in ("seed", "snapshot_epoch"):
            p.add_argument(flag, type=int, default=default)
        elif f.name in ("snapshot_job_id", "pretrained", "trace_dir"):
            p.add_argument(flag, type=str, default=default)
        else:
            p.add_argument(flag, type=type(default), default=default)
    return p


def parse_config(script: str, argv: Optional[List[str]] = None) -> TrainConfig:
    ns = build_parser(script).parse_args(argv)
    return TrainConfig(**vars(ns))

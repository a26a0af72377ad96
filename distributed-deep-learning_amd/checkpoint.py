"""Checkpoint / resume in the reference's ``torch.distributed.checkpoint`` layout.

Reference (single.py:68-89,121-134; pp.py:77-99,154-169; ddp_n_pp.py:84-106): a
``Stateful`` ``AppState`` whose ``state_dict()`` is
``{"model{S}": ..., "optim{S}": ..., "epoch": tensor}`` (built with
``get_state_dict``: FQN-keyed optimizer state), saved with
``dcp.save({"app": AppState}, checkpoint_id=CHECKPOINT_DIR/{job_id}/epoch_{epoch})``.
Suffix ``S``: "" for single/ddp, the stage's rank for pp / ddp x pp.

On disk that flattens to::

    app.model{S}.<fqn>                              params and buffers
    app.optim{S}.state.<fqn>.{step,exp_avg,exp_avg_sq}
    app.optim{S}.param_groups.0.{lr,betas,...,params}
    app.epoch

This module produces exactly that tree *without* ``get_state_dict`` because the native
engine stores some weights in MFMA-friendly layouts inside a flat arena: tensors are
converted to the torch (torchvision) layout on the way out and copied back **into the
arena views** on the way in (views are never rebound - SURVEY 7.4 item 7).  A stage's
keys are the torchvision FQNs, so pp / hybrid checkpoints are a disjoint partition of
the single-model key space; ``merge_stage_checkpoints`` relies on that.

Fixed reference bugs: the saved epoch is the epoch being saved, not the
stale ``epochs_run``; the checkpoint directory itself is created.
"""
from __future__ import annotations

import json
import os
from typing import Any, Dict, List, Optional, Sequence

import torch
import torch.distributed.checkpoint as dcp
from torch.distributed.checkpoint.stateful import Stateful


def _export(model, fqn: str, t: torch.Tensor) -> torch.Tensor:
    f = getattr(model, "ddl_export_tensor", None)
    return f(fqn, t) if f is not None else t


def _import(model, fqn: str, t: torch.Tensor) -> torch.Tensor:
    f = getattr(model, "ddl_import_tensor", None)
    return f(fqn, t) if f is not None else t


def model_state(model: torch.nn.Module) -> Dict[str, torch.Tensor]:
    out = {}
    for fqn, p in model.named_parameters():
        out[fqn] = _export(model, fqn, p.detach()).clone()
    for fqn, b in model.named_buffers():
        out[fqn] = b.detach().clone()
    return out


def optim_state(model: torch.nn.Module, optimizer: torch.optim.Optimizer) -> Dict[str, Any]:
    """FQN-keyed optimizer state-dict (the ``get_state_dict`` form)."""
    fqn_of = {p: n for n, p in model.named_parameters()}
    state = {}
    for p, st in optimizer.state.items():
        fqn = fqn_of[p]
        state[fqn] = {k: (_export(model, fqn, v.detach()).clone() if isinstance(v, torch.Tensor) and v.dim() > 0
                          else (v.detach().clone() if isinstance(v, torch.Tensor) else v))
                      for k, v in st.items()}
    groups = []
    for g in optimizer.param_groups:
        gg = {k: v for k, v in g.items() if k != "params"}
        gg["params"] = [fqn_of[p] for p in g["params"]]
        groups.append(gg)
    return {"state": state, "param_groups": groups}


@torch.no_grad()
def load_model_state(model: torch.nn.Module, sd: Dict[str, torch.Tensor]) -> None:
    for fqn, p in model.named_parameters():
        p.data.copy_(_import(model, fqn, sd[fqn]))
    for fqn, b in model.named_buffers():
        b.data.copy_(sd[fqn])


@torch.no_grad()
def load_optim_state(model: torch.nn.Module, optimizer: torch.optim.Optimizer, sd: Dict[str, Any]) -> None:
    by_fqn = dict(model.named_parameters())
    for fqn, st in sd["state"].items():
        p = by_fqn[fqn]
        mine = optimizer.state[p]
        for k, v in st.items():
            if isinstance(v, torch.Tensor) and isinstance(mine.get(k), torch.Tensor):
                if v.dim() > 0:
                    v = _import(model, fqn, v)
                mine[k].copy_(v)
            else:
                mine[k] = v
    for g_mine, g_sd in zip(optimizer.param_groups, sd["param_groups"]):
        for k, v in g_sd.items():
            if k != "params":
                g_mine[k] = v
    if hasattr(optimizer, "_host_step") and sd["state"]:
        any_state = next(iter(sd["state"].values()))
        optimizer._host_step = int(float(any_state["step"]))


def _ensure_optim_state(optimizer: torch.optim.Optimizer) -> None:
    """torch.optim.Adam creates its state lazily; materialise it so a fresh optimizer has a
    complete template to load into (what DCP's ``_init_optim_state`` does with a fake step)."""
    if len(optimizer.state) > 0:
        return
    for g in optimizer.param_groups:
        for p in g["params"]:
            if p.requires_grad:
                optimizer.state[p] = {
                    "step": torch.zeros((), dtype=torch.float32, device="cpu"),
                    "exp_avg": torch.zeros_like(p),
                    "exp_avg_sq": torch.zeros_like(p),
                }


class AppState(Stateful):
    """Model + optimizer + epoch, in the reference's key layout."""

    def __init__(self, epoch: int, model: torch.nn.Module, optimizer: torch.optim.Optimizer,
                 suffix: str = "") -> None:
        self.epoch = epoch
        self.model = model
        self.optimizer = optimizer
        self.suffix = str(suffix)

    def state_dict(self) -> Dict[str, Any]:
        _ensure_optim_state(self.optimizer)
        return {
            f"model{self.suffix}": model_state(self.model),
            f"optim{self.suffix}": optim_state(self.model, self.optimizer),
            "epoch": torch.tensor(self.epoch),
        }

    def load_state_dict(self, state_dict: Dict[str, Any]) -> None:
        load_model_state(self.model, state_dict[f"model{self.suffix}"])
        load_optim_state(self.model, self.optimizer, state_dict[f"optim{self.suffix}"])
        self.epoch = int(state_dict["epoch"])


def snapshot_path(ckpt_dir: str, job_id: str, epoch: int) -> str:
    return os.path.join(ckpt_dir, job_id, f"epoch_{epoch}")


def save_snapshot(path: str, epoch: int, model, optimizer, suffix: str = "", process_group=None) -> None:
    os.makedirs(path, exist_ok=True)
    dcp.save({"app": AppState(epoch, model, optimizer, suffix)}, checkpoint_id=path,
             process_group=process_group)


def load_snapshot(path: str, model, optimizer, suffix: str = "", process_group=None) -> int:
    """Loads in place; returns the epoch stored in the snapshot."""
    app = AppState(0, model, optimizer, suffix)
    dcp.load({"app": app}, checkpoint_id=path, process_group=process_group)
    return app.epoch


PARTITION_FILE = "partition.json"


def save_partition(path: str, split_points: Sequence[str]) -> None:
    """Sidecar of a pipeline snapshot: the RESOLVED stage boundaries.  Snapshots are keyed ``model{stage}`` /
    ``optim{stage}`` like the reference's (pp.py:87-88), so a resumed run must rebuild exactly the stages that were
    saved - ``--split-points auto`` re-measures unit times and may move a cut by one dense layer otherwise."""
    tmp = os.path.join(path, PARTITION_FILE + ".tmp")
    with open(tmp, "w") as f:
        json.dump({"split_points": list(split_points), "num_stages": len(split_points) + 1}, f)
    os.replace(tmp, os.path.join(path, PARTITION_FILE))


def load_partition(path: Optional[str]) -> Optional[List[str]]:
    if not path:
        return None
    f = os.path.join(path, PARTITION_FILE)
    if not os.path.exists(f):
        return None
    with open(f) as fh:
        return list(json.load(fh)["split_points"])


def resume_path(ckpt_subdir: str, job_id: str, snapshot_job_id: Optional[str], snapshot_epoch: Optional[int],
                resume_latest: bool) -> Optional[str]:
    """The snapshot directory a run with these flags resumes from (None: fresh start).  One rule for the trainer
    (which loads it) and the pipeline strategy (which must know the saved partition BEFORE it builds its stage)."""
    job, epoch = snapshot_job_id, snapshot_epoch
    if resume_latest and job is None:
        latest = latest_snapshot(ckpt_subdir, job_id)
        if latest is not None:
            job, epoch = job_id, latest
    if job is None or epoch is None:
        return None
    path = snapshot_path(ckpt_subdir, job, epoch)
    return path if os.path.exists(path) else None                 # single.py:115-119


def latest_snapshot(ckpt_dir: str, job_id: str) -> Optional[int]:
    d = os.path.join(ckpt_dir, job_id)
    if not os.path.isdir(d):
        return None
    epochs = [int(n.split("_", 1)[1]) for n in os.listdir(d)
              if n.startswith("epoch_") and os.path.exists(os.path.join(d, n, ".metadata"))]
    return max(epochs) if epochs else None


def read_keys(path: str):
    """Flattened key names stored in a DCP checkpoint directory (format tests, tooling)."""
    from torch.distributed.checkpoint import FileSystemReader

    md = FileSystemReader(path).read_metadata()
    return sorted(md.state_dict_metadata.keys())

"""ONE trainer + four strategy objects (single | ddp | pp | ddp x pp).

The reference copy-pastes a ~230-line ``Trainer`` into each of its four scripts with small
diffs (single.py:92-269, ddp.py:102-326, pp.py:102-339, ddp_n_pp.py:109-362).
Here the epoch loop, bookkeeping, evaluation, CSV logging and snapshotting exist once; a
``Strategy`` says how a batch is stepped and which ranks hold inputs / targets / outputs.

Behaviour kept from the reference:
  * epoch_time covers the training loop only (eval excluded, ddp.py:189-191) - but is
    bracketed by a device synchronise, which the reference forgot;
  * per-epoch metrics ``epoch_time, loss, train_accuracy`` then eval metrics ``val_loss`` +
    the 8 sklearn-equivalent scores, one CSV per metric (ddp.py:207-218,260,283-290);
  * who logs: the process (single), global rank 0 (ddp), the last stage (pp), last stage of
    dp-rank 0 (hybrid) - and that rank broadcasts the save flag to everyone
    (ddp.py:223, pp.py:243, ddp_n_pp.py:275);
  * snapshot = DCP ``AppState`` under ``{ckpt}/{strategy}/{job_id}/epoch_{n}`` with the
    reference's key suffixes; resume via (snapshot_job_id, snapshot_epoch).
Deliberately not kept: O(n^2) ``torch.cat`` bookkeeping (pre-sized epoch buffers
instead), ``next(iter(loader))`` per epoch, all-ranks-decode-everything in pp, stale saved epoch.
"""
from __future__ import annotations

import os
import time
from typing import List, Optional, Tuple

import torch

from . import checkpoint as ckpt
from . import comm, metrics, ops
from .config import TrainConfig
from .data import DevicePrefetcher
from .models import create_stage
from .models.densenet import UNIT_NAMES, DenseNetStage, stage_units
from .optim import FusedAdam
from .parallel.dp import DataParallelEngine
from .parallel.pp import SCHEDULES, PipelineStage
from .profiler import StepProfiler
from .utils.arena import FlatArena, reverse_named_parameters
from .utils.graph import GraphedEvalStep, GraphedTrainStep
from .utils.trace import StepTrace, trace_range


def set_seed(seed: int) -> None:
    """reference: single.py:28-35 (defined there, never called; here behind --seed)."""
    import random

    import numpy as np

    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def job_id_from_env() -> str:
    return os.getenv("TORCHX_JOB_ID", "local").split("/")[-1]     # single.py:102


def _arena_for(module: torch.nn.Module, device, bucket_mb: float, world: int) -> FlatArena:
    existing = getattr(module, "arena", None)
    if isinstance(existing, FlatArena):
        return existing
    return FlatArena(reverse_named_parameters(module), device=device,
                     segment_numel=max(int(bucket_mb * 2 ** 20 / 4), 1), segment_multiple=world * 64)


# ======================================================================================
# strategies
# ======================================================================================
class Strategy:
    name = "base"
    needs_input = True
    needs_target = True
    has_output = True
    is_logger = True
    ckpt_suffix = ""
    dp: Optional[DataParallelEngine] = None

    def __init__(self, cfg: TrainConfig, ctx: comm.CommContext) -> None:
        self.cfg, self.ctx = cfg, ctx
        self.device = ctx.device
        self.global_rank, self.local_rank = ctx.rank, ctx.local_rank
        self.profiler = StepProfiler(self.device, enabled=cfg.profile)
        self.dp_rank, self.dp_size = 0, 1

    def _make_optimizer(self, arena: FlatArena, shard=None) -> FusedAdam:
        c = self.cfg
        return FusedAdam(arena, lr=c.lr, betas=(c.beta1, c.beta2), eps=c.eps,
                         weight_decay=c.weight_decay, shard=shard)

    # -- overridables ------------------------------------------------------------------
    def train_step(self, x, y, batch_size: int):
        raise NotImplementedError

    def eval_step(self, x, y, batch_size: int):
        raise NotImplementedError

    def gather(self, t: torch.Tensor) -> torch.Tensor:
        return self.dp.gather(t) if self.dp is not None else t

    def mean(self, t: torch.Tensor) -> torch.Tensor:
        return self.dp.mean(t) if self.dp is not None else t

    def broadcast_flag(self, flag: bool) -> bool:
        return flag

    def epoch_boundary(self) -> None:
        if self.dp is not None:
            self.dp.epoch_boundary()


class SingleStrategy(Strategy):
    """reference: single.py - one device, no process group."""
    name = "single"

    def __init__(self, cfg, ctx) -> None:
        super().__init__(cfg, ctx)
        self.model = create_stage(UNIT_NAMES, self.device, cfg.num_classes, cfg.dtype,
                                  pretrained=cfg.pretrained)
        self.arena = _arena_for(self.model, self.device, cfg.bucket_mb, 1)
        self.optimizer = self._make_optimizer(self.arena)
        self.global_rank = self.local_rank = 0          # single.py:269 hard-codes 0,0
        # zero_grad -> forward -> fused CE -> backward -> fused Adam (reference: single.py:136-154),
        # replayed as one hipGraph launch after two eager warm-up steps
        self._step = GraphedTrainStep(self.model, self.optimizer, None, self.profiler,
                                      enabled=cfg.hip_graph and self.device.type == "cuda",
                                      fuse_zero_grad=not cfg.log_gradients)
        self._eval = GraphedEvalStep(self.model, enabled=cfg.hip_graph and self.device.type == "cuda")

    def train_step(self, x, y, batch_size):
        return self._step(x, y)

    @torch.no_grad()
    def eval_step(self, x, y, batch_size):
        return self._eval(x)


class DDPStrategy(Strategy):
    """reference: ddp.py - replicated model, sharded data, gradient averaging."""
    name = "ddp"

    def __init__(self, cfg, ctx, group=None) -> None:
        super().__init__(cfg, ctx)
        self.model = create_stage(UNIT_NAMES, self.device, cfg.num_classes, cfg.dtype,
                                  pretrained=cfg.pretrained, **self._native_kw(cfg, comm.get_world_size(group)))
        self.dp = DataParallelEngine(self.model, group=group, device=self.device, bucket_mb=cfg.bucket_mb,
                                     grad_sync=cfg.grad_sync, sync_buffers=cfg.sync_buffers,
                                     zero1=cfg.zero1, profiler=self.profiler, debug_sync=cfg.debug_sync, rccl=cfg.rccl)
        self.arena = self.dp.arena
        shard = (self.dp.rank, self.dp.world) if (cfg.zero1 and self.dp.world > 1) else None
        self.optimizer = self._make_optimizer(self.arena, shard)
        self.dp_rank, self.dp_size = self.dp.rank, self.dp.world
        self.is_logger = ctx.rank == 0
        self._step = GraphedTrainStep(self.model, self.optimizer, self.dp, self.profiler,
                                      enabled=cfg.hip_graph and self.device.type == "cuda",
                                      fuse_zero_grad=not cfg.log_gradients)
        self._eval = GraphedEvalStep(self.model, enabled=cfg.hip_graph and self.device.type == "cuda")

    def _native_kw(self, cfg, world):
        if self.device.type != "cuda":
            return {}
        # phased backward graphs (utils/graph.py) only matter when there is something to overlap with
        phases = tuple(u for u in cfg.grad_phases.split(",") if u) if ((world > 1 or comm.force_collectives()) and cfg.hip_graph) else ()
        return dict(bucket_mb=cfg.bucket_mb, dp_world=world, grad_phases=phases)

    def train_step(self, x, y, batch_size):
        # eager form: buffer broadcast -> forward -> backward (bucket collectives launched by the
        # gradient-ready hooks, overlapped) -> wait -> Adam (reference: ddp.py:150-170)
        return self._step(x, y)

    @torch.no_grad()
    def eval_step(self, x, y, batch_size):
        self.dp.pre_forward()               # DDP broadcasts buffers in eval forwards too
        return self._eval(x)

    def broadcast_flag(self, flag):
        return comm.broadcast_flag(flag, 0, self.device)            # ddp.py:223


class PPStrategy(Strategy):
    """reference: pp.py (world == stages) and ddp_n_pp.py (2-D mesh) - one class, the
    hybrid is the same pipeline with a DataParallelEngine on the dp sub-group."""
    name = "pp"

    def __init__(self, cfg, ctx, mesh: Optional[comm.Mesh2D] = None) -> None:
        super().__init__(cfg, ctx)
        world = ctx.world_size
        if mesh is None:
            mesh = comm.Mesh2D(1, world)                              # pp.py: stage index == global rank
        self.mesh = mesh
        S = mesh.pp_size
        self.stage_index = mesh.pp_index
        self.split_points = list(cfg.split_points)
        self.unit_costs = None
        # A resumed run rebuilds exactly the stages that were saved: snapshots are keyed model{stage} / optim{stage}
        # (reference layout), so the resolved partition travels with the snapshot (checkpoint.save_partition)
        saved = ckpt.load_partition(ckpt.resume_path(cfg.ckpt_subdir, job_id_from_env(), cfg.snapshot_job_id,
                                                     cfg.snapshot_epoch, cfg.resume_latest))
        if saved is not None:
            if self.split_points == ["auto"]:
                self.split_points = list(saved)
                if ctx.rank == 0:
                    print(f"[pp] resuming: stage boundaries taken from the snapshot: {self.split_points}", flush=True)
            elif self.split_points != list(saved):
                raise ValueError(f"--split-points {self.split_points} differ from the partition the snapshot was saved "
                                 f"with ({saved}): its model{{stage}} / optim{{stage}} keys would not match the stages")
        if self.split_points == ["auto"]:
            # balance the stages on measured per-unit forward+backward time at this run's micro-batch size
            # (models/split.py); the reference's fixed split leaves 73 % of the work on stage 0
            from .models.split import auto_split_points
            self.split_points, self.unit_costs = auto_split_points(
                S, self.device, max(cfg.batch_size // max(cfg.num_microbatches, 1), 1), cfg.image_size, cfg.dtype)
            if ctx.rank == 0:
                print(f"[pp] auto split for {S} stages: {self.split_points} "
                      f"(unit costs {[round(c, 3) for c in self.unit_costs]})", flush=True)
        units_per_stage = stage_units(self.split_points)
        if len(units_per_stage) != S:
            raise ValueError(f"{len(self.split_points)} split points make {len(units_per_stage)} stages, "
                             f"but the pipeline group has {S} ranks")
        self.units = units_per_stage[self.stage_index]
        hybrid = mesh.dp_size > 1
        nat = {}
        if self.device.type == "cuda":
            # hybrid under hipGraphs: the last micro-batch's backward is cut at every dense block of the stage so
            # that finished gradient buckets leave while the earlier blocks are still in backward (pp.py docstring)
            phases = tuple(u for u in self.units if u.startswith("denseblock")) if (hybrid and cfg.hip_graph) else ()
            nat = dict(bucket_mb=cfg.bucket_mb, dp_world=mesh.dp_size, grad_phases=phases)
        self.model = create_stage(self.units, self.device, cfg.num_classes, cfg.dtype,
                                  pretrained=cfg.pretrained, **nat)
        self.name = "ddpnpp" if hybrid else "pp"
        if hybrid:
            self.dp = DataParallelEngine(self.model, group=mesh.get_group("dp"), device=self.device,
                                         bucket_mb=cfg.bucket_mb, grad_sync=cfg.grad_sync,
                                         sync_buffers=cfg.sync_buffers, zero1=cfg.zero1,
                                         profiler=self.profiler, debug_sync=cfg.debug_sync, rccl=cfg.rccl)
            self.arena = self.dp.arena
            self.dp_rank, self.dp_size = self.dp.rank, self.dp.world
        else:
            self.arena = _arena_for(self.model, self.device, cfg.bucket_mb, 1)
        shard = (self.dp.rank, self.dp.world) if (hybrid and cfg.zero1) else None
        self.optimizer = self._make_optimizer(self.arena, shard)

        self.is_first = self.stage_index == 0
        self.is_last = self.stage_index == S - 1
        self.needs_input = self.is_first
        self.needs_target = self.is_last
        self.has_output = self.is_last
        self.is_logger = self.is_last and mesh.dp_index == 0
        # reference suffixes: global rank (pp.py:87-88) / local rank == stage index (ddp_n_pp.py:94-95)
        self.ckpt_suffix = str(self.stage_index)
        self.flag_src = mesh.last_stage_rank_of_first_pipeline

        prior_units = [u for us in units_per_stage[: self.stage_index] for u in us]
        size = cfg.image_size
        native = not isinstance(self.model, DenseNetStage)

        def recv_shape(mb: int) -> Tuple[int, ...]:
            s = (mb, 3, size, size)
            for u in prior_units:
                s = DenseNetStage.unit_output_shape(u, s)
            return (s[0], s[2], s[3], s[1]) if native else s         # native stages exchange NHWC

        self.stage = PipelineStage(self.model, self.stage_index, S, self.device,
                                   group=mesh.get_group("pp"), ranks=mesh.pp_ranks,
                                   recv_shape_fn=recv_shape, dp_engine=self.dp,
                                   hip_graph=cfg.hip_graph and native and self.device.type == "cuda",
                                   debug_sync=cfg.debug_sync, profiler=self.profiler, n_lanes=cfg.pp_lanes,
                                   n_fwd_lanes=cfg.pp_fwd_lanes, rccl=cfg.rccl)
        sched = SCHEDULES[cfg.schedule]
        # two schedules over one stage, like the reference (pp.py:138-150): training (with loss)
        # and loss-less inference for evaluation under no_grad
        self.schedule = sched(self.stage, cfg.num_microbatches, loss_fn=ops.cross_entropy,
                              profiler=self.profiler)
        self.schedule_inference = sched(self.stage, cfg.num_microbatches, loss_fn=None,
                                        profiler=self.profiler)

    def train_step(self, x, y, batch_size):
        prof = self.profiler
        tok = prof.begin("step")
        self.optimizer.zero_grad()
        if self.dp is not None:
            # BN-buffer broadcast.  Deviation from the reference, documented (README "Deviations"): torch's
            # DDP forward inside each stage broadcasts the buffers on EVERY micro-batch forward (torch stage.py:560-568
            # under ddp_n_pp.py:139); here rank 0's buffers win ONCE per step.  Same fixed point (every replica starts
            # every step from rank 0's running statistics); what differs is that replicas r > 0 use their own
            # running statistics for micro-batches 1..M-1 of a step - values that only evaluation reads.
            # On RCCL the broadcast is deferred behind the step's last forward and rides the communication stream
            # (`after_forward`, below) instead of blocking the compute stream in front of every pipeline step.
            self.dp.pre_forward(training=True)
            self.dp.arm(False)
        out, loss = None, None
        if self.is_first and not self.is_last:
            self.schedule.step(x, batch_size=batch_size)                       # pp.py:174-175
        elif self.is_last:
            losses: List[torch.Tensor] = []
            out = self.schedule.step(x if self.is_first else None, target=y, losses=losses,
                                     batch_size=batch_size)                    # pp.py:176-178
            loss = torch.stack(losses).mean()
        else:
            self.schedule.step(batch_size=batch_size)                          # pp.py:190-191
        if self.dp is not None:
            self.dp.after_forward()          # the step's forwards are done (lanes joined): buffers travel beside the reduce
            tok2 = prof.begin("exposed_comm")
            self.dp.finish_grad_sync()
            prof.end(tok2)
        self.optimizer.step()
        if self.dp is not None:
            self.dp.post_step()
        prof.end(tok)
        prof.step_done()
        if out is None:
            return None
        return loss, out, out.argmax(dim=1)

    @torch.no_grad()
    def eval_step(self, x, y, batch_size):
        if self.dp is not None:
            self.dp.pre_forward()
        if self.is_first:
            return self.schedule_inference.step(x, batch_size=batch_size)
        return self.schedule_inference.step(batch_size=batch_size)

    def broadcast_flag(self, flag):
        return comm.broadcast_flag(flag, self.flag_src, self.device)           # pp.py:243 / ddp_n_pp.py:275


def build_strategy(cfg: TrainConfig, ctx: comm.CommContext) -> Strategy:
    if cfg.debug_sync:
        # race hunting: ONE stream for everything - eager launches (a captured graph would keep its
        # forked branches), collectives on the compute stream, weight gradients in line.  The native engine is
        # bit-deterministic, so a run that differs from the overlapped one points at a missing fence.
        cfg = cfg.replace(hip_graph=False)
    strat = _build_strategy(cfg, ctx)
    if cfg.debug_sync and hasattr(strat.model, "overlap_wgrad"):
        strat.model.overlap_wgrad = False
    return strat


def _build_strategy(cfg: TrainConfig, ctx: comm.CommContext) -> Strategy:
    if cfg.strategy == "single":
        return SingleStrategy(cfg, ctx)
    if cfg.strategy == "ddp":
        return DDPStrategy(cfg, ctx)
    if cfg.strategy == "pp":
        return PPStrategy(cfg, ctx, comm.Mesh2D(1, ctx.world_size))
    if cfg.strategy == "ddpnpp":
        dp, pp = comm.choose_mesh(ctx.world_size, cfg.pp_size, cfg.dp_size)
        return PPStrategy(cfg, ctx, comm.Mesh2D(dp, pp))
    raise ValueError(cfg.strategy)


# ======================================================================================
# trainer
# ======================================================================================
class Trainer:
    def __init__(self, strategy: Strategy, train_data, test_data, cfg: TrainConfig) -> None:
        self.s = strategy
        self.cfg = cfg
        self.train_data, self.test_data = train_data, test_data
        self.device = strategy.device
        self.job_id = job_id_from_env()
        self.epochs_run = 0
        self.best_qwk = -1.0
        self.snapshot_job_id = cfg.snapshot_job_id
        self.logger = metrics.CsvMetricLogger(
            cfg.log_dir, self.job_id, strategy.global_rank, strategy.local_rank,
            model_start_job_id=cfg.snapshot_job_id, enabled=strategy.is_logger)
        self.history: List[dict] = []
        self.last_epoch_time = 0.0
        self.samples_per_sec = 0.0
        self._fault = tuple(int(v) for v in cfg.fault_inject.split(":")) if cfg.fault_inject else None
        self._trace = StepTrace(cfg.trace_dir, strategy.global_rank, cfg.trace_skip, cfg.trace_steps)

        path = ckpt.resume_path(cfg.ckpt_subdir, self.job_id, cfg.snapshot_job_id, cfg.snapshot_epoch,
                                cfg.resume_latest)                  # single.py:115-119
        if path is not None:
            self._print("Loading snapshot")
            self._load_snapshot(path)

    # ---- plumbing ----------------------------------------------------------------------
    def _print(self, msg: str) -> None:
        print(f"[GPU{self.s.global_rank}] {msg}" if comm.is_dist() else msg, flush=True)

    def _sync(self) -> None:
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def _load_snapshot(self, path: str) -> None:
        epoch = ckpt.load_snapshot(path, self.s.model, self.s.optimizer, self.s.ckpt_suffix)
        self.epochs_run = epoch + 1
        self._print(f"Resuming training from snapshot at Epoch {self.epochs_run}")

    def _save_snapshot(self, epoch: int) -> None:
        path = ckpt.snapshot_path(self.cfg.ckpt_subdir, self.job_id, epoch)
        dp, opt = self.s.dp, self.s.optimizer
        if dp is not None and dp.zero1 and hasattr(opt, "exp_avg"):
            # ZeRO-1: each rank's Adam moments are valid on its 1/W shard only - make them whole before DCP
            # de-duplicates the (identically keyed) optimizer state across the DP replicas
            dp.gather_sharded(opt.exp_avg, opt.exp_avg_sq)
        ckpt.save_snapshot(path, epoch, self.s.model, self.s.optimizer, self.s.ckpt_suffix)
        if self.s.global_rank == 0 and getattr(self.s, "split_points", None) is not None:
            ckpt.save_partition(path, self.s.split_points)         # the resolved stage boundaries (resume with auto)
        self._print(f"Epoch {epoch} | Saved snapshot to {path}")

    # ---- training ----------------------------------------------------------------------
    def _run_epoch(self, epoch: int) -> Tuple[int, int]:
        s, cfg = self.s, self.cfg
        self.train_data.set_epoch(epoch)
        steps = len(self.train_data)
        if cfg.max_steps:
            steps = min(steps, cfg.max_steps)
        b_sz = self.train_data.batch_size
        self._print(f"Starting Epoch {epoch}")
        if s.has_output:
            cap = steps * b_sz
            self._preds = torch.empty(cap, dtype=torch.int64, device=self.device)
            self._targets = torch.empty(cap, dtype=torch.int64, device=self.device)
            self._losses = torch.zeros(steps, dtype=torch.float32, device=self.device)
        seen = 0
        for step, (x, y, n) in enumerate(DevicePrefetcher(self.train_data, self.device, s.needs_input,
                                                          s.needs_target)):
            if step >= steps:
                break
            if self._fault == (s.global_rank, epoch, step):
                raise RuntimeError(f"injected fault on rank {s.global_rank} at epoch {epoch} step {step} (--fault-inject)")
            self._trace.tick()
            with trace_range(f"train step {step}"):
                res = s.train_step(x, y, n)
            if res is not None:
                loss, _out, preds = res
                self._preds[seen:seen + n] = preds
                self._targets[seen:seen + n] = y
                self._losses[step] = loss
            seen += n
            if cfg.log_gradients and s.is_logger:
                self._log_gradient(epoch * steps + step)
        self._trace.close()                    # a window that runs past the epoch ends here
        self._print(f"Epoch {epoch} | Batchsize: {b_sz} | Steps: {steps}")
        return seen, steps

    def train(self, max_epochs: int) -> None:
        s = self.s
        for epoch in range(self.epochs_run, max_epochs):
            self._sync()
            t0 = time.perf_counter()
            seen, steps = self._run_epoch(epoch)
            self._sync()
            epoch_time = time.perf_counter() - t0
            self.last_epoch_time = epoch_time
            self.samples_per_sec = seen * s.dp_size / epoch_time if epoch_time > 0 else 0.0
            self._print(f"Epoch {epoch} | Time: {epoch_time:.2f}s")
            record = {"epoch": epoch, "epoch_time": epoch_time}

            if s.has_output:
                loss = s.mean(self._losses[:steps].mean())                         # ddp.py:211-212
                preds = s.gather(self._preds[:seen])                               # ddp.py:196-197
                targets = s.gather(self._targets[:seen])
                if s.is_logger:
                    train_acc = metrics.accuracy(targets, preds)
                    self._print(f"Epoch {epoch} | Loss: {loss.item()}")
                    self._print(f"Epoch {epoch} | Training Accuracy: {train_acc}")
                    self.logger.log("train_accuracy", train_acc, epoch)
                    self.logger.log("epoch_time", epoch_time, epoch)
                    self.logger.log("loss", loss.item(), epoch)
                    record.update(loss=loss.item(), train_accuracy=train_acc)

            s.epoch_boundary()
            s.model.eval()
            improved = self._evaluate(epoch, record)
            s.model.train()
            save = s.broadcast_flag(improved)
            if self.cfg.save_every and (epoch + 1) % self.cfg.save_every == 0:
                save = True
            if save:
                self._print(f"Saving snapshot at epoch {epoch}")
                self._save_snapshot(epoch)
            if self.cfg.profile:
                record["profile"] = s.profiler.summary()
                if s.is_logger:
                    self._print(f"Epoch {epoch} | Profile: {record['profile']}")
                s.profiler.reset()
            self.history.append(record)
        self.epochs_run = max_epochs

    # ---- evaluation --------------------------------------------------------------------
    @torch.no_grad()
    def _evaluate(self, epoch: int, record: dict) -> bool:
        s, cfg = self.s, self.cfg
        self.test_data.set_epoch(epoch)
        outs, tgts = [], []
        for step, (x, y, n) in enumerate(DevicePrefetcher(self.test_data, self.device, s.needs_input,
                                                          s.needs_target)):
            if cfg.max_steps and step >= cfg.max_steps:
                break
            out = s.eval_step(x, y, n)
            if s.has_output:
                outs.append(out)
                tgts.append(y)
        if not s.has_output or not outs:
            return False
        local_out = torch.cat(outs, 0)
        local_tgt = torch.cat(tgts, 0)
        all_out = s.gather(local_out)                                              # ddp.py:248-254
        all_tgt = s.gather(local_tgt)
        if not s.is_logger:
            return False
        loss = ops.cross_entropy(all_out.float(), all_tgt)                         # ddp.py:256
        self._print(f"Epoch {epoch} | Validation Loss: {loss.item()}")
        self.logger.log("val_loss", loss.item(), epoch)
        cm = metrics.confusion_matrix(all_tgt, all_out.argmax(dim=1), cfg.num_classes)
        m = metrics.metrics_from_confusion(cm)
        names = {"val_accuracy": "Accuracy", "macro_f1": "Macro F1", "weighted_f1": "Weighted F1",
                 "macro_precision": "Macro Precision", "weighted_precision": "Weighted Precision",
                 "macro_recall": "Macro Recall", "weighted_recall": "Weighted Recall", "qwk": "QWK"}
        for k in metrics.EVAL_METRICS:
            self._print(f"Epoch {epoch} | Validation {names[k]}: {m[k]}")
            self.logger.log(k, m[k], epoch)
        record.update(val_loss=loss.item(), **m)
        qwk = m["qwk"]
        if qwk == qwk and qwk > self.best_qwk:                                     # ddp.py:292-295
            self.best_qwk = qwk
            self._print(f"New Best Validation QWK: {self.best_qwk}")
            return bool(cfg.save_best)
        return False

    # ---- opt-in gradient dump (reference dead code: ddp.py:310-326) -------------------------
    def _log_gradient(self, step: int) -> None:
        stats = metrics.gradient_stats(self.s.model.named_parameters())
        fname = "gradient.csv" if self.s.name in ("single", "ddp") else f"gradients{self.s.global_rank}.csv"
        for name, st in stats.items():
            self.logger.log_row(fname, [self.job_id, step, name, st["min"], st["mean"], st["max"],
                                        st["p25"], st["median"], st["p75"], st["std"]])

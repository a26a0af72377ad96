"""``main()`` of every entrypoint script (single.py, ddp.py, pp.py, ddp_n_pp.py,
communication_time.py, test.py, debug.py).  The scripts themselves are three-line shims so
the reference's names keep working (`torchx run ... --script ddp.py`, command:5)."""
from __future__ import annotations

import argparse
import json
import os
import sys
from typing import List, Optional

import torch

from . import comm
from .config import parse_config
from .data import build_loaders
from .trainer import Trainer, build_strategy, job_id_from_env, set_seed


def run_training(script: str, argv: Optional[List[str]] = None) -> Trainer:
    cfg = parse_config(script, argv)
    ctx = comm.init(cfg.backend, cfg.device, cfg.timeout_s,
                    force=(script != "single" and "RANK" in os.environ))
    if script != "single" and not comm.is_dist():
        # plain `python ddp.py`: a world of one (the reference's "single with ddp" run, command:30-31)
        ctx = comm.init(cfg.backend, cfg.device, cfg.timeout_s, force=True)
    if cfg.seed is not None:
        set_seed(cfg.seed)
    if cfg.check_collectives:
        comm.enable_checker()
    strategy = build_strategy(cfg, ctx)
    train_loader, test_loader = build_loaders(cfg, dp_rank=strategy.dp_rank, dp_size=strategy.dp_size,
                                              device=ctx.device, needs_input=strategy.needs_input)
    trainer = Trainer(strategy, train_loader, test_loader, cfg)
    try:
        trainer.train(cfg.epochs)
    except BaseException as e:      # noqa: BLE001
        if not comm.is_dist() or isinstance(e, KeyboardInterrupt):
            raise
        # Failure path (the reference has none): a rank that dies inside the loop must not wait for
        # its peers in barrier / destroy_process_group - they are blocked in a collective it will never join.
        # Report, leave with a non-zero status at once; the launcher (torchrun / torchx) tears the job down, the
        # process-group timeout (--timeout-s) covers a peer that hangs instead of dying, and --resume-latest
        # restarts from the last complete snapshot.
        import traceback
        traceback.print_exc()
        print(f"[GPU{ctx.rank}] fatal: {type(e).__name__}: {e} - aborting the job", file=sys.stderr, flush=True)
        sys.stdout.flush()
        os._exit(1)
    if comm.is_dist():
        comm.barrier()
        comm.destroy()
    return trainer


def main_single(argv=None):
    return run_training("single", argv)


def main_ddp(argv=None):
    return run_training("ddp", argv)


def main_pp(argv=None):
    return run_training("pp", argv)


def main_ddp_n_pp(argv=None):
    return run_training("ddpnpp", argv)


# --------------------------------------------------------------------------------------
# communication_time.py
# --------------------------------------------------------------------------------------
def main_communication_time(argv=None):
    """Reference: rank 0 pings rank 1 1000x with 4 MiB fp32 + 1-float ack, event-timed, rows
    ``job_id,i,ms`` appended to training_logs/communication_time.csv
    (communication_time.py:48-62).  Added: ``--sweep`` message-size sweep of send/recv and
    all-reduce / reduce-scatter+all-gather (BASELINE.json config 5)."""
    from . import profiler as prof
    from .metrics import CsvMetricLogger

    p = argparse.ArgumentParser(prog="communication_time.py")
    p.add_argument("--iters", type=int, default=1000)
    p.add_argument("--numel", type=int, default=1024 * 1024)
    p.add_argument("--sweep", action="store_true")
    p.add_argument("--min-bytes", type=int, default=1024)
    p.add_argument("--max-bytes", type=int, default=256 * 2 ** 20)
    p.add_argument("--log-dir", default=os.environ.get("DDL_LOG_DIR", "/mnt/dcornelius/training_logs"))
    p.add_argument("--backend", default="auto")
    p.add_argument("--device", default="auto")
    p.add_argument("--json", default=None, help="write the sweep table to this file")
    a = p.parse_args(argv)

    ctx = comm.init(a.backend, a.device, force=True)
    job_id = job_id_from_env()
    dev = ctx.device
    print(f"job id: {job_id}")
    if dev.type == "cuda":
        print(f"GPU: {torch.cuda.get_device_name(dev)}")
    print(f"global rank: {ctx.rank}")
    logger = CsvMetricLogger(a.log_dir, job_id, ctx.rank, ctx.local_rank, enabled=ctx.rank == 0)
    result = {"world": ctx.world_size, "ping_ms": None, "sweep": []}

    if ctx.world_size >= 2:
        times = []
        for i in range(a.iters):
            ms = prof.ping(dev, a.numel)
            if ctx.rank == 0:
                logger.log_row("communication_time.csv", [job_id, i, ms])
                times.append(ms)
        if ctx.rank == 0 and len(times) > 1:
            steady = times[1:]                     # iteration 0 = communicator setup (notebook :731)
            result["ping_ms"] = sum(steady) / len(steady)
            result["first_iter_ms"] = times[0]
            print(f"ping {a.numel * 4} B + ack: mean {result['ping_ms']:.4f} ms over {len(steady)} iters "
                  f"(first {times[0]:.1f} ms)")
    else:
        print("world size 1: no peer to ping; running the local (W=1) collective sweep only")

    if a.sweep or ctx.world_size < 2:
        sizes, n = [], a.min_bytes
        while n <= a.max_bytes:
            sizes.append(n)
            n *= 4
        rows = []
        if ctx.world_size >= 2:
            rows += prof.sweep_p2p(dev, sizes)
        rows += prof.sweep_allreduce(dev, sizes)
        result["sweep"] = rows
        if ctx.rank == 0:
            for r in rows:
                print(json.dumps(r))
                logger.log_row("communication_sweep.csv", [job_id] + [r.get(k) for k in
                               ("op", "bytes", "ms_median", "ms_min", "gbps", "busbw_gbps", "world")])
    if a.json and ctx.rank == 0:
        os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
        with open(a.json, "w") as f:
            json.dump(result, f, indent=1)
    comm.barrier()
    comm.destroy()
    return result


# --------------------------------------------------------------------------------------
# test.py (mesh smoke) and debug.py (split explorer)
# --------------------------------------------------------------------------------------
def main_mesh_smoke(argv=None):
    """Reference test.py:8-30: init PG, build the (dp, pp) mesh, all-reduce [rank] over the
    pp sub-group and print it (1, 5, 9 for the three pairs of a 3x2 mesh)."""
    p = argparse.ArgumentParser(prog="test.py")
    p.add_argument("--pp-size", type=int, default=2)
    p.add_argument("--dp-size", type=int, default=0)
    p.add_argument("--backend", default="auto")
    p.add_argument("--device", default="auto")
    a = p.parse_args(argv)
    ctx = comm.init(a.backend, a.device, force=True)
    pp = a.pp_size if ctx.world_size % a.pp_size == 0 else 1
    dp, pp = comm.choose_mesh(ctx.world_size, pp, a.dp_size)
    mesh = comm.Mesh2D(dp, pp)
    print(f"[rank {ctx.rank}] {mesh} dp group ranks {mesh.dp_ranks} pp group ranks {mesh.pp_ranks}", flush=True)
    t = torch.tensor([ctx.rank], dtype=torch.int64, device=ctx.device)
    if mesh.get_group("pp") is not None:
        comm.all_reduce(t, "sum", mesh.get_group("pp"))
    print(f"[rank {ctx.rank}] all_reduce over pp group -> {t.item()}", flush=True)
    comm.barrier()
    comm.destroy()
    return int(t.item())


def main_split_explorer(argv=None):
    """Reference debug.py: print parameters per stage for candidate split specs."""
    from .models.densenet import DenseNetStage, default_split_points, densenet121, stage_units

    p = argparse.ArgumentParser(prog="debug.py")
    p.add_argument("--split-points", nargs="*", default=None)
    p.add_argument("--num-classes", type=int, default=1000)      # debug.py uses the stock 1000-class model
    p.add_argument("--auto-stages", type=int, nargs="*", default=[], help="also derive balanced splits for these stage "
                   "counts (models/split.py; on a GPU from measured unit times, and the stages are timed)")
    p.add_argument("--micro-batch", type=int, default=6)
    a = p.parse_args(argv)
    specs = [a.split_points] if a.split_points is not None else [default_split_points(n) for n in (2, 4)]
    if a.auto_stages:
        from .models import split
        dev = torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu")
        for n in a.auto_stages:
            pts, costs = split.auto_split_points(n, dev, a.micro_batch)
            print(f"auto split, {n} stages (unit costs {[round(c, 3) for c in costs]}): {pts}")
            if dev.type == "cuda":
                print(f"  measured stage forward+backward ms at micro-batch {a.micro_batch}: "
                      f"{[round(t, 3) for t in split.measure_stage_times(pts, dev, a.micro_batch)]}")
            specs.append(pts)
    total = sum(q.numel() for q in densenet121(a.num_classes).parameters())
    print(f"Model parameters: {total}")
    out = []
    for spec in specs:
        print(f"split spec: {spec}")
        sizes = []
        for i, units in enumerate(stage_units(spec)):
            n = sum(q.numel() for q in DenseNetStage(units, a.num_classes).parameters())
            sizes.append(n)
            print(f"Stage {i} num params: {n}   units: {units}")
        print(f"Total num params: {sum(sizes)}")
        out.append(sizes)
    return out

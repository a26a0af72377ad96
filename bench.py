#!/usr/bin/env python3
"""Headline benchmark: DenseNet-121 (5-class head) training throughput, samples/s, whole job.

Metric / config = BASELINE.json: the reference's workload (DenseNet-121, 3x224x224 inputs, Adam
defaults, fp32 like the reference, per-GPU batch 30 = single.py's batch) under data parallelism on
N MI355X of one node (N=1 is the single-device baseline).  Synthetic uint8 images generated in
HBM, random-init weights.  A timed step is the full training step: input normalisation, forward,
fused softmax-CE, backward, bucketed RCCL gradient all-reduce (or reduce-scatter + all-gather), fused Adam.

    python bench.py                       # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  Scaling is weak (per-GPU batch fixed).  ``--mode pp|ddpnpp``
benchmarks the pipeline / hybrid strategies instead of DDP.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch  # noqa: E402

# reference numbers, BASELINE.md section 2 (samples/s, derived from the published epoch times)
BASELINE = {"single": 26.4, "ddp": 39.3, "pp": 21.6, "ddpnpp": 17.7}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch-per-gpu", type=int, default=30)
    ap.add_argument("--mode", default="ddp", choices=["ddp", "pp", "ddpnpp"])
    ap.add_argument("--impl", default="auto", choices=["auto", "native", "torch"])
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--schedule", default="1f1b")
    ap.add_argument("--microbatches", type=int, default=5)
    ap.add_argument("--pp-size", type=int, default=2)
    ap.add_argument("--bucket-mb", type=float, default=8.0)
    ap.add_argument("--grad-sync", choices=["rs_ag", "allreduce"], default=None,
                    help="per-bucket collective (default: TrainConfig.grad_sync)")
    ap.add_argument("--zero1", action="store_true")
    ap.add_argument("--sync-buffers", default="forward", choices=["forward", "epoch", "off"],
                    help="BN buffer broadcast policy (forward = DDP parity, the default)")
    ap.add_argument("--grad-phases", default=None, help="units that open a new backward graph phase (DP overlap); "
                    "default = TrainConfig's, '' = one graph + all collectives behind it")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--backend", default="auto", help="auto = RCCL; gloo lets several ranks share one GPU (rehearsal only)")
    ap.add_argument("--profile", action="store_true", help="add the compute/RCCL breakdown to the JSON")
    a = ap.parse_args()

    from ddl import comm
    from ddl.config import TrainConfig
    from ddl.models.densenet import default_split_points
    from ddl.trainer import build_strategy

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        a.gpus = world
    if world > 1:
        a.profile = True      # event-based compute / RCCL / exposed-communication split (no host sync in the loop)
    ctx = comm.init(a.backend, "auto", force=True)
    dev = ctx.device
    B = a.batch_per_gpu
    strategy_name = a.mode
    pp_size = a.pp_size if a.mode != "ddp" else 1
    if a.mode == "pp":
        pp_size = world
    def make_strategy(hip_graph: bool):
        cfg = TrainConfig(strategy=strategy_name, batch_size=B, impl=a.impl, dtype=a.dtype,
                          bucket_mb=a.bucket_mb, zero1=a.zero1, hip_graph=hip_graph,
                          num_microbatches=a.microbatches, schedule=a.schedule, pp_size=pp_size,
                          split_points=default_split_points(max(pp_size, 1)), profile=a.profile,
                          sync_buffers=a.sync_buffers, num_workers=0,
                          **({} if a.grad_phases is None else {"grad_phases": a.grad_phases}),
                          **({} if a.grad_sync is None else {"grad_sync": a.grad_sync}))
        return build_strategy(cfg, ctx)

    strat = make_strategy(not a.no_graph)
    dp = strat.dp_size

    g = torch.Generator().manual_seed(1234 + ctx.rank)
    pool = []
    for _ in range(4):
        x = torch.randint(0, 256, (B, 3, 224, 224), dtype=torch.uint8, generator=g).to(dev)
        y = torch.randint(0, 5, (B,), dtype=torch.int64, generator=g).to(dev)
        pool.append((x if strat.needs_input else None, y if strat.needs_target else None))

    def sync():
        comm.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    # Priming (set-up, never timed, independent of --warmup): RCCL communicator creation, allocator growth
    # and the hipGraph capture (2 eager steps + the capturing step) must not land in the timed region even
    # when the driver asks for fewer warm-up steps than that.
    PRIME = 3
    strat.model.train()
    failure = None
    try:
        for i in range(PRIME + a.warmup):
            x, y = pool[i % len(pool)]
            strat.train_step(x, y, B)
        graph_mode = "off" if a.no_graph else "hipGraph"
    except Exception as e:      # noqa: BLE001 - e.g. a capture refused by the runtime: measure the eager path instead
        if a.no_graph:
            raise
        failure = e
    if world > 1 and not a.no_graph:          # every rank takes the same path (a capture failure is symmetric)
        flag = torch.tensor([0.0 if failure is None else 1.0], device=dev)
        comm.all_reduce(flag, "max")
        if flag.item() > 0 and failure is None:
            failure = RuntimeError("another rank failed in the hipGraph path")
    if failure is not None:
        e = failure
        print(f"[rank {ctx.rank}] hipGraph path failed during warm-up ({type(e).__name__}: {e}); "
              f"falling back to eager launches", file=sys.stderr, flush=True)
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        os.environ["DDL_NATIVE_REDUCER"] = "0"      # most conservative path: eager launches, Python bucket reducer
        strat = make_strategy(False)
        strat.model.train()
        for i in range(PRIME + a.warmup):
            x, y = pool[i % len(pool)]
            strat.train_step(x, y, B)
        graph_mode = "eager (fallback)"
    # The per-step compute / RCCL split costs ~1 % (timing events on every stream), so the profiler is paused for
    # the timed steps and the split is measured on extra steps behind them ("profile_steps" in the JSON).
    strat.profiler.reset()
    if a.profile:
        strat.profiler.set_enabled(False)
    sync()
    t0 = time.perf_counter()
    for i in range(a.steps):
        x, y = pool[i % len(pool)]
        strat.train_step(x, y, B)
    host_dt = time.perf_counter() - t0          # host time to ISSUE the steps (no device wait in the loop)
    sync()
    dt = time.perf_counter() - t0
    if a.profile:
        strat.profiler.set_enabled(True)
        for i in range(min(a.steps, 20)):
            x, y = pool[i % len(pool)]
            strat.train_step(x, y, B)
        sync()

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if comm.is_dist() and world > 1:
        comm.all_reduce(t, "max")
    dt = float(t.item())
    global_batch = B * dp
    value = global_batch * a.steps / dt
    base_key = "single" if (a.mode == "ddp" and world == 1) else a.mode
    impl = type(strat.model).__name__
    out = {
        "metric": "samples/sec (whole job), DenseNet-121 training",
        "value": round(value, 2),
        "unit": "samples/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": round(value / BASELINE[base_key], 2),
        "dtype": a.dtype,
        "data": "synthetic uint8 3x224x224 images generated on device, random-init weights",
        "config": {
            "model": "DenseNet-121 (5-class head, 6,958,981 params)",
            "global_batch": global_batch,
            "batch_per_gpu": B,
            "seq_len": None,
            "image": "3x224x224",
            "optimizer": "Adam(lr=1e-3) fused flat",
            "parallelism": (f"dp{dp}" if a.mode == "ddp" else
                            f"pp{pp_size}" if a.mode == "pp" else f"dp{dp}xpp{pp_size}"),
            "baseline": f"{base_key} {BASELINE[base_key]} samples/s (BASELINE.md sec. 2)",
            "impl": impl,
            "launch": graph_mode,
            "grad_sync": ("zero1 rs+ag(params)" if a.zero1 else
                          "bucketed all-reduce" if getattr(getattr(strat, "dp", None), "grad_sync", "") == "allreduce"
                          else "bucketed reduce-scatter + all-gather"),
            "reducer": ("native" if getattr(getattr(strat, "dp", None), "_native", None) is not None else "python"),
            "host_issue_ms_per_step": round(host_dt / a.steps * 1e3, 3),
            "peak_mem_gib": (round(torch.cuda.max_memory_reserved(dev) / 2 ** 30, 2) if dev.type == "cuda" else None),
        },
    }
    if a.profile:
        out["profile"] = {k: (round(v, 4) if isinstance(v, float) else v)
                          for k, v in strat.profiler.summary().items()}
        out["profile"]["profile_steps"] = out["profile"].pop("steps")     # extra steps behind the timed region
    if ctx.rank == 0:
        print(json.dumps(out), flush=True)
    sync()
    comm.destroy()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Headline benchmark: DenseNet-121 (5-class head) training throughput, samples/s, whole job.

Metric / config = BASELINE.json: the reference's workload (DenseNet-121, 3x224x224 inputs, Adam
defaults, fp32 like the reference, per-GPU batch 30 = single.py's batch) under data parallelism on
N MI355X of one node (N=1 is the single-device baseline).  Synthetic uint8 images generated in
HBM, random-init weights.  A timed step is the full training step: input normalisation, forward,
fused softmax-CE, backward, bucketed RCCL reduce-scatter + all-gather (or all-reduce), fused Adam.

    python bench.py                       # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  Scaling is weak by default (per-GPU batch 30 fixed); ``--scaling strong`` keeps the
reference's GLOBAL batch of 30 (ddp.py:335: 15 per rank at W=2).  ``--mode pp|ddpnpp`` benchmarks the pipeline /
hybrid strategies instead of DDP; ``--preset parity`` (default) keeps the reference's own sizes - pp: 30 = 5
micro-batches of 6 (pp.py:365,378), ddpnpp: 10 per replica = 5 micro-batches of 2 (ddp_n_pp.py:371,404), split at
denseblock3 - and ``--preset mi355x`` sizes them for the hardware (per-replica batch 240 = 8 micro-batches of 30,
stages balanced on measured unit times).

First contact: with more than one rank (or ``DDL_FORCE_COLLECTIVES=1``) a staged ladder runs before anything is
built - barrier, all-reduce, the native bucket reducer on a scratch arena (both collective forms, value-checked), a
grouped send/recv ring through the native P2P engine (pipeline modes) - then strategy construction, each eager
priming step and the hipGraph capture are stages too.  Every stage prints a progress line on stderr and has its own
watchdog: a hang names its stage and exits non-zero (``ddl/diagnostics.py``).  The JSON says what actually ran:
``backend``, ``world``, ``distinct_devices``, ``ranks_per_device``; ``n_gpus`` counts distinct physical GPUs.

Failure policy: a step that cannot be captured / run is a FAILED benchmark - the error is printed and the
process exits non-zero (no silent re-measurement on another path).  ``host_bound`` in the JSON (and a
warning on stderr) flags a run whose host needs more than 80 % of the step time to issue a step
(``host_issue_ms_per_step``: CPU time of one step issued into an idle GPU).

With more than one rank (or ``--ab``) extra steps BEHIND the timed region measure both per-bucket
collective forms on the same captured step (``grad_sync_ab``) and two more bucket sizes
(``bucket_mb_sweep``), and the compute / RCCL / exposed-communication split is reported as the maximum
over ranks (``profile``) - one multi-GPU run returns the tuning data.
"""
import argparse
import json
import os
import sys
import threading
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch  # noqa: E402

# reference numbers, BASELINE.md section 2 (samples/s, derived from the published epoch times)
BASELINE = {"single": 26.4, "ddp": 39.3, "pp": 21.6, "ddpnpp": 17.7}

# per-replica batch / micro-batches / split policy of the pipeline modes
PRESETS = {
    "parity": dict(batch=30, microbatches=5, split="reference"),     # pp.py:365,378-386
    "mi355x": dict(batch=240, microbatches=8, split="auto"),         # 8 micro-batches of 30: the single-GPU kernel regime
    "mi355x-large": dict(batch=1024, microbatches=8, split="auto"),  # 8 micro-batches of 128: where the kernels run at
}                                                                    #   ~4,700 samples/s; ~40 GB of the 288 GB per stage
# the hybrid's own reference shape: 10 images per DP replica = 5 micro-batches of 2 (ddp_n_pp.py:371,404)
PRESETS_BY_MODE = {("ddpnpp", "parity"): dict(batch=10, microbatches=5, split="reference")}


def preset_for(mode: str, name: str) -> dict:
    return PRESETS_BY_MODE.get((mode, name), PRESETS[name])


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch-per-gpu", type=int, default=None, help="per-replica batch (default: 30, or the preset's)")
    ap.add_argument("--mode", default="ddp", choices=["ddp", "pp", "ddpnpp"])
    ap.add_argument("--preset", default="parity", choices=sorted(PRESETS))
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16", "fp32x3"],
                    help="fp32 = the reference's precision on the fp32 MFMA (headline); bf16 / fp32x3 = optional modes")
    ap.add_argument("--schedule", default="1f1b")
    ap.add_argument("--microbatches", type=int, default=None)
    ap.add_argument("--pp-size", type=int, default=2)
    ap.add_argument("--split-points", nargs="+", default=None,
                    help="stage starts (reference module names) or 'auto' (balance on measured unit times)")
    ap.add_argument("--bucket-mb", type=float, default=8.0)
    ap.add_argument("--grad-sync", choices=["rs_ag", "allreduce", "auto", "tune"], default=None,
                    help="per-bucket collective (default: TrainConfig.grad_sync = auto: all-reduce for 2 ranks, rs_ag from "
                         "3 on; tune = time both on the captured step during priming and keep the faster one - all "
                         "ranks agree through a max-reduce)")
    ap.add_argument("--rccl", choices=["auto", "direct", "c10d"], default="auto",
                    help="RCCL transport of the bucket reducer and the pipeline P2P engine: owned communicators on the "
                         "engines' own streams (csrc/rccl_comm.cpp) or c10d's process group")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="ddp mode: weak = 30 images per GPU (default), strong = the reference's GLOBAL batch of 30 "
                         "split over the ranks (ddp.py:335: 15 per rank at W=2)")
    ap.add_argument("--image-size", type=int, default=224, help="input resolution (224 = the reference; anything else "
                                                                "is a plumbing run and says so in the JSON)")
    ap.add_argument("--ladder", default="auto", choices=["auto", "on", "off"],
                    help="first-contact ladder before anything is built (auto: when W>1 or DDL_FORCE_COLLECTIVES=1)")
    ap.add_argument("--stage-timeout", type=float, default=180.0,
                    help="seconds one first-contact / priming stage may take before the run names it and exits 4")
    ap.add_argument("--fwd-lanes", type=int, default=None, help="pipeline modes: forward streams (TrainConfig.pp_fwd_lanes)")
    ap.add_argument("--lanes", type=int, default=None, help="pipeline modes: concurrent micro-batch lanes per stage "
                                                            "(default: TrainConfig.pp_lanes)")
    ap.add_argument("--zero1", action="store_true")
    ap.add_argument("--sync-buffers", default="forward", choices=["forward", "epoch", "off"],
                    help="BN buffer broadcast policy (forward = DDP parity, the default)")
    ap.add_argument("--grad-phases", default=None, help="units that open a new backward graph phase (DP overlap); "
                    "default = TrainConfig's, '' = one graph + all collectives behind it")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--backend", default="auto", help="auto = RCCL; gloo lets several ranks share one GPU (rehearsal only)")
    ap.add_argument("--profile", action="store_true", help="add the compute/RCCL breakdown to the JSON")
    ap.add_argument("--ab", default="auto", choices=["auto", "on", "off"],
                    help="grad_sync A/B + bucket-size sweep on extra steps behind the timed region (auto: when W>1)")
    ap.add_argument("--ab-steps", type=int, default=12)
    ap.add_argument("--extras-timeout", type=float, default=150.0,
                    help="seconds the tuning extras behind the timed region may take before the headline is printed without them")
    ap.add_argument("--trace-dir", default=None, help="torch.profiler capture (Chrome trace + kernel table per rank) of "
                                                      "4 extra steps BEHIND the timed region (ddl/utils/trace.py)")
    a = ap.parse_args()

    # stdout carries exactly one line, the JSON of rank 0.  Native libraries write to file descriptor 1 as well (RCCL
    # prints a five-line version banner from every rank when its first communicator comes up), so fd 1 is pointed at
    # stderr for the whole run and the JSON goes to a private duplicate of the original stdout.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    from ddl import comm, diagnostics
    from ddl.config import TrainConfig
    from ddl.models.densenet import default_split_points
    from ddl.trainer import build_strategy

    diagnostics.refuse_wrong_result_switches("bench.py")          # DDL_KNOCKOUT & co: exit 5
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        a.gpus = world
    if world > 1:
        a.profile = True      # event-based compute / RCCL / exposed-communication split (no host sync in the loop)
    rank_env = int(os.environ.get("RANK", "0"))
    wd = diagnostics.StageWatchdog(rank_env, "bench.py", exit_code=4, quiet=(world == 1 and not comm.force_collectives()))
    with wd.stage("process group init (rendezvous on MASTER_ADDR)", a.stage_timeout):
        ctx = comm.init(a.backend, "auto", force=True)
    dev = ctx.device
    preset = preset_for(a.mode, a.preset)
    pipeline = a.mode != "ddp"
    if a.batch_per_gpu is not None:
        B = a.batch_per_gpu
    elif pipeline:
        B = preset["batch"]
    elif a.scaling == "strong":
        if 30 % world:
            print(f"bench.py: --scaling strong splits the reference's global batch of 30 over the ranks; "
                  f"{world} does not divide 30", file=sys.stderr, flush=True)
            os._exit(2)
        B = 30 // world                                           # ddp.py:335 (15 at W=2)
    else:
        B = 30
    n_mb = a.microbatches if a.microbatches is not None else preset["microbatches"]
    strategy_name = a.mode
    pp_size = a.pp_size if a.mode != "ddp" else 1
    if a.mode == "pp":
        pp_size = world
    if a.split_points is not None:
        split = a.split_points
    elif pipeline and preset["split"] == "auto":
        split = ["auto"]
    else:
        split = default_split_points(max(pp_size, 1))

    tune_grad_sync = a.grad_sync == "tune"

    def make_strategy(bucket_mb: float, grad_phases=None):
        gp = a.grad_phases if grad_phases is None else grad_phases
        cfg = TrainConfig(strategy=strategy_name, batch_size=B, dtype=a.dtype, image_size=a.image_size,
                          bucket_mb=bucket_mb, zero1=a.zero1, hip_graph=not a.no_graph,
                          num_microbatches=n_mb, schedule=a.schedule, pp_size=pp_size,
                          split_points=split, profile=a.profile,
                          sync_buffers=a.sync_buffers, num_workers=0, rccl=a.rccl,
                          **({} if gp is None else {"grad_phases": gp}),
                          **({} if a.lanes is None else {"pp_lanes": a.lanes}),
                          **({} if a.fwd_lanes is None else {"pp_fwd_lanes": a.fwd_lanes}),
                          **({} if (a.grad_sync is None or tune_grad_sync) else {"grad_sync": a.grad_sync}))
        return build_strategy(cfg, ctx)

    def fail(stage: str, e: BaseException) -> None:
        # A failed capture / step is a failed run.  Peers of a failing rank sit in collectives: leave at once
        # (the launcher tears the job down) instead of joining a barrier that can never complete.
        traceback.print_exc()
        print(f"[rank {ctx.rank}] bench.py FAILED during {stage}: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        sys.stdout.flush()
        os._exit(3)

    # ---- first contact (ddl/diagnostics.py): who is here, then one rung at a time -------------------------------
    try:
        with wd.stage("topology all-gather", a.stage_timeout):
            topo = diagnostics.topology(ctx)
        if ctx.rank == 0 and not wd.quiet:
            print(f"[rank 0] bench.py: world {topo['world']} over {topo['backend']}, {topo['distinct_devices']} distinct "
                  f"device(s), {topo['ranks_per_device']} rank(s) per device: {topo['device_ids']}", file=sys.stderr, flush=True)
        ladder_on = a.ladder == "on" or (a.ladder == "auto" and (world > 1 or comm.force_collectives()))
        ladder = diagnostics.first_contact_ladder(ctx, pipeline, wd, a.stage_timeout, rccl=a.rccl) if ladder_on else []
    except BaseException as e:      # noqa: BLE001
        fail("the first-contact ladder", e)

    try:
        with wd.stage("strategy construction (parameter broadcast, sub-groups, auto split)", max(a.stage_timeout, 600.0)):
            strat = make_strategy(a.bucket_mb)
    except BaseException as e:      # noqa: BLE001
        fail("strategy construction", e)
    dp = strat.dp_size

    g = torch.Generator().manual_seed(1234 + ctx.rank)
    pool = []
    for _ in range(4):
        x = torch.randint(0, 256, (B, 3, a.image_size, a.image_size), dtype=torch.uint8, generator=g).to(dev)
        y = torch.randint(0, 5, (B,), dtype=torch.int64, generator=g).to(dev)
        pool.append((x if strat.needs_input else None, y if strat.needs_target else None))

    def sync():
        comm.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    def run_steps(s, n, first=0):
        for i in range(first, first + n):
            x, y = pool[i % len(pool)]
            s.train_step(x, y, B)

    def timed(s, n):
        """(wall seconds incl. device drain, host seconds to issue) of n steps, barrier-bracketed."""
        sync()
        t0 = time.perf_counter()
        run_steps(s, n)
        host = time.perf_counter() - t0          # host time to ISSUE the steps (no device wait in the loop)
        sync()
        return time.perf_counter() - t0, host

    def max_over_ranks(v: float) -> float:
        t = torch.tensor([v], dtype=torch.float64, device=dev)
        if comm.is_dist() and world > 1:
            comm.all_reduce(t, "max")
        return float(t.item())

    # Priming (set-up, never timed, independent of --warmup): RCCL communicator creation, allocator growth
    # and the hipGraph capture (2 eager steps + the capturing step) must not land in the timed region even
    # when the driver asks for fewer warm-up steps than that.
    PRIME = 3
    strat.model.train()
    graph_mode = "off" if (a.no_graph or dev.type != "cuda") else "hipGraph"
    engine = getattr(strat, "dp", None)
    grad_sync_tuned = None
    try:
        for i in range(PRIME):
            # the graphed steps run 2 eager steps per new shape and capture on the third (utils/graph.py, pp._GraphSet)
            what = ("eager" if (i < 2 or graph_mode == "off") else "hipGraph capture + first replay")
            with wd.stage(f"priming step {i} ({what})", a.stage_timeout):
                run_steps(strat, 1, first=i)
                if dev.type == "cuda":
                    torch.cuda.synchronize(dev)
        if tune_grad_sync and engine is not None and not engine.single and not a.zero1:
            # per-bucket collective form chosen by measurement on THIS machine and world size: the forms can be
            # switched between two steps (collectives live outside the captured graphs)
            with wd.stage("grad-sync tuning (rs_ag vs allreduce on the captured step)", a.stage_timeout):
                grad_sync_tuned = {}
                for mode in ("rs_ag", "allreduce"):
                    engine.set_grad_sync(mode)
                    run_steps(strat, 2)
                    sync()
                    t0 = time.perf_counter()
                    run_steps(strat, 8)
                    sync()
                    grad_sync_tuned[mode] = round(max_over_ranks(time.perf_counter() - t0) / 8 * 1e3, 3)
                best = min(grad_sync_tuned, key=grad_sync_tuned.get)       # identical on every rank (max-reduced times)
                engine.set_grad_sync(best)
                grad_sync_tuned["chosen"] = best
        with wd.stage(f"{a.warmup} warm-up steps", max(a.stage_timeout, 2.0 * a.warmup)):
            run_steps(strat, a.warmup)
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
    except BaseException as e:      # noqa: BLE001
        fail("warm-up / hipGraph capture", e)

    # The per-step compute / RCCL split costs ~1 % (timing events on every stream), so the profiler is paused for
    # the timed steps and the split is measured on extra steps behind them ("profile_steps" in the JSON).
    strat.profiler.reset()
    if a.profile:
        strat.profiler.set_enabled(False)
    try:
        with wd.stage(f"the {a.steps} timed steps", max(a.stage_timeout, 2.0 * a.steps)):
            dt, host_dt = timed(strat, a.steps)
    except BaseException as e:      # noqa: BLE001
        fail("the timed steps", e)
    dt = max_over_ranks(dt)
    step_ms = dt / a.steps * 1e3
    # Host cost of a step.  Issuing steps back to back (the timed loop above) fills the queue to the GPU, and from
    # then on every launch call waits for queue space: that wall time (`host_issue_queued_ms_per_step`) is an upper
    # bound that mostly measures the GPU.  The CPU work of a step is measured behind the timed region by issuing
    # single steps into an IDLE GPU (device synchronised before each): `host_issue_ms_per_step`, median of 7.
    queued_ms = max_over_ranks(host_dt / a.steps * 1e3)
    idle = []
    for i in range(7 if dev.type == "cuda" else 2):
        sync()
        t0 = time.perf_counter()
        run_steps(strat, 1, first=i)
        idle.append((time.perf_counter() - t0) * 1e3)
    sync()
    host_ms = max_over_ranks(sorted(idle)[len(idle) // 2])
    host_bound = host_ms > 0.8 * step_ms

    profile = None
    if a.profile:
        strat.profiler.set_enabled(True)
        run_steps(strat, min(a.steps, 20))
        sync()
        profile = strat.profiler.summary()
        strat.profiler.set_enabled(False)
        n_prof = profile.pop("steps")
        # the slowest rank sets the step: report every span as the maximum over ranks
        keys = sorted(k for k in profile if k.endswith("_ms_per_step"))
        if comm.is_dist() and world > 1:
            t = torch.tensor([profile.get(k, 0.0) for k in keys], dtype=torch.float64, device=dev)
            comm.all_reduce(t, "max")
            profile = {k: float(v) for k, v in zip(keys, t.tolist())}
        step_p = profile.get("step_ms_per_step", 0.0)
        if step_p > 0:
            exposed = profile.get("exposed_comm_ms_per_step", 0.0) + profile.get("p2p_wait_ms_per_step", 0.0)
            busy = profile.get("rccl_ms_per_step", 0.0) + profile.get("p2p_ms_per_step", 0.0)
            profile["comm_busy_fraction"] = busy / step_p
            profile["comm_exposed_fraction"] = exposed / step_p
            profile["compute_fraction"] = max(0.0, 1.0 - exposed / step_p)
        profile["host_issue_ms_per_step"] = host_ms
        profile["profile_steps"] = n_prof
        profile["reduced_over_ranks"] = "max"

    if a.trace_dir:
        from ddl.utils.trace import StepTrace
        st = StepTrace(a.trace_dir, rank=ctx.rank, skip=1, steps=4)
        for i in range(6):
            st.tick()
            run_steps(strat, 1, first=i)
        st.tick()
        sync()

    global_batch = B * dp
    value = global_batch * a.steps / dt
    base_key = "single" if (a.mode == "ddp" and world == 1) else a.mode
    out = {
        "metric": "samples/sec (whole job), DenseNet-121 training",
        "value": round(value, 2),
        "unit": "samples/s",
        "n_gpus": topo["n_gpus"],
        "world": topo["world"],
        "backend": topo["backend"],
        "distinct_devices": topo["distinct_devices"],
        "ranks_per_device": topo["ranks_per_device"],
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(step_ms, 3),
        "higher_is_better": True,
        "scaling": a.scaling if a.mode == "ddp" else "weak",
        "vs_baseline": round(value / BASELINE[base_key], 2),
        "dtype": a.dtype,
        "data": f"synthetic uint8 3x{a.image_size}x{a.image_size} images generated on device, random-init weights",
        "host_issue_ms_per_step": round(host_ms, 3),
        "host_issue_queued_ms_per_step": round(queued_ms, 3),
        "host_bound": bool(host_bound),
        "config": {
            "model": "DenseNet-121 (5-class head, 6,958,981 params)",
            "global_batch": global_batch,
            "batch_per_gpu": B,
            "seq_len": None,
            "image": f"3x{a.image_size}x{a.image_size}",
            "optimizer": "Adam(lr=1e-3) fused flat",
            "parallelism": (f"dp{dp}" if a.mode == "ddp" else
                            f"pp{pp_size}" if a.mode == "pp" else f"dp{dp}xpp{pp_size}"),
            "baseline": f"{base_key} {BASELINE[base_key]} samples/s (BASELINE.md sec. 2)",
            "impl": type(strat.model).__name__,
            "launch": graph_mode,
            "grad_sync": ("zero1 rs+ag(params)" if a.zero1 else
                          "bucketed all-reduce" if getattr(engine, "grad_sync", "") == "allreduce"
                          else "bucketed reduce-scatter + all-gather"),
            "bucket_mb": a.bucket_mb,
            "reducer": ("native" if getattr(engine, "_native", None) is not None else "python"),
            "rccl_transport": {"reducer": getattr(engine, "transport", None),
                               "p2p": getattr(getattr(getattr(strat, "stage", None), "chan", None), "transport", None)},
            "host_issue_ms_per_step": round(host_ms, 3),
            "peak_mem_gib": (round(torch.cuda.max_memory_reserved(dev) / 2 ** 30, 2) if dev.type == "cuda" else None),
        },
    }
    if pipeline:
        out["config"].update(preset=a.preset, microbatches=n_mb, schedule=a.schedule,
                             split_points=list(getattr(strat, "split_points", split)),
                             lanes=getattr(getattr(strat, "stage", None), "n_lanes", 1),
                             fwd_lanes=getattr(getattr(strat, "stage", None), "n_fwd", 1))
    if ladder:
        out["first_contact_ladder"] = ladder          # rungs that completed (value-checked) before anything was built
    if grad_sync_tuned is not None:
        out["grad_sync_tuned"] = grad_sync_tuned
    exp_env = diagnostics.experimental_env()
    if exp_env:
        out["experimental_env"] = exp_env             # every DDL_* switch that was set: nothing changes the path silently
    if a.image_size != 224 or topo["n_gpus"] == 0 or topo["ranks_per_device"] > 1:
        out["plumbing_run"] = True                    # not a headline configuration (shared device / CPU / small images)
    if profile is not None:
        out["profile"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in profile.items()}

    def emit() -> None:
        if ctx.rank == 0:
            print(json.dumps(out), file=json_out, flush=True)

    # The headline is measured.  Everything behind this point is tuning data (multi-rank runs; `--ab on` to rehearse
    # with one rank): a hang or crash in it - a collective that never completes on a first multi-GPU run - must not
    # lose the number, so a watchdog prints the JSON without the extras and leaves.
    def bail() -> None:
        out["extras"] = f"abandoned after {a.extras_timeout} s (watchdog)"
        print(f"[rank {ctx.rank}] bench.py: tuning extras did not finish in {a.extras_timeout} s - reporting the "
              f"headline without them", file=sys.stderr, flush=True)
        emit()
        sys.stdout.flush()
        os._exit(0)

    watchdog = threading.Timer(a.extras_timeout, bail)
    watchdog.daemon = True
    watchdog.start()
    ab = a.ab == "on" or (a.ab == "auto" and world > 1)
    grad_sync_ab, bucket_sweep, phase_sweep = None, None, None
    if ab and engine is not None and not engine.single and not a.zero1:
        try:
            grad_sync_ab = {}
            chosen = engine.grad_sync
            for mode in ("rs_ag", "allreduce"):
                engine.set_grad_sync(mode)
                run_steps(strat, 3)
                t, _ = timed(strat, a.ab_steps)
                grad_sync_ab[mode] = round(max_over_ranks(t) / a.ab_steps * 1e3, 3)
            engine.set_grad_sync(chosen)
            if a.mode == "ddp":
                bucket_sweep = {str(a.bucket_mb): round(step_ms, 3)}
                for mb in (2.0, 32.0):
                    if mb == a.bucket_mb:
                        continue
                    s2 = make_strategy(mb)
                    s2.model.train()
                    s2.profiler.set_enabled(False)
                    run_steps(s2, PRIME + 3)
                    t, _ = timed(s2, a.ab_steps)
                    bucket_sweep[str(mb)] = round(max_over_ranks(t) / a.ab_steps * 1e3, 3)
                    del s2
                if not a.no_graph and dev.type == "cuda":
                    # where the backward graph chain is cut (= when finished buckets can leave): the default, no cut
                    # (every bucket behind the whole backward), one cut, a cut at every dense block
                    phase_sweep = {"default": round(step_ms, 3)}
                    for name, gp in (("none", ""), ("denseblock3", "denseblock3"),
                                     ("blocks 3,2,1", "denseblock3,denseblock2,denseblock1"),
                                     ("every block", "denseblock4,denseblock3,denseblock2,denseblock1")):
                        s2 = make_strategy(a.bucket_mb, gp)
                        s2.model.train()
                        s2.profiler.set_enabled(False)
                        run_steps(s2, PRIME + 3)
                        t, _ = timed(s2, a.ab_steps)
                        phase_sweep[name] = round(max_over_ranks(t) / a.ab_steps * 1e3, 3)
                        del s2
        except BaseException as e:      # noqa: BLE001 - the headline is measured; a failed extra must not lose it on THIS rank
            print(f"[rank {ctx.rank}] tuning extras failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
            grad_sync_ab = grad_sync_ab or {"error": f"{type(e).__name__}: {e}"}

    watchdog.cancel()
    if grad_sync_ab is not None:
        out["grad_sync_ab"] = grad_sync_ab          # ms/step of the same captured step with either collective form
    if bucket_sweep is not None:
        out["bucket_mb_sweep"] = bucket_sweep       # ms/step per bucket size (MiB)
    if phase_sweep is not None:
        out["grad_phase_sweep"] = phase_sweep       # ms/step per placement of the backward graph cuts
    if host_bound and ctx.rank == 0:
        print(f"WARNING: host-bound run - issuing a step took {host_ms:.2f} ms of the {step_ms:.2f} ms step "
              f"(> 80 %); the GPUs wait for the host", file=sys.stderr, flush=True)
    emit()
    # the number is out: a teardown that cannot complete (a peer that left early) must not turn into a time-out
    closer = threading.Timer(30.0, lambda: os._exit(0))
    closer.daemon = True
    closer.start()
    sync()
    comm.destroy()


if __name__ == "__main__":
    main()

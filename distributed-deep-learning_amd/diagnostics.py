"""First-contact diagnostics of a multi-rank run: who is here, on which devices, over which transport - and a
staged ladder that exercises every communication path a run is about to use, one rung at a time, each under
its own watchdog.

Reference: ``test.py:8-30`` is the reference's own first-contact script (init the process group, build the
(3, 2) mesh, all-reduce one element over a sub-group, print it).  It exists because the failure mode of a first
multi-GPU run is a HANG inside the communication library, which tells nobody anything.  Here the same idea is
part of every multi-rank ``bench.py`` run (and of ``test.py --ladder``):

    barrier -> all-reduce (value check) -> the native bucket reducer on a scratch arena: in-place
    reduce-scatter(AVG) + all-gather, and all-reduce(AVG) (value checks) -> a grouped send/recv ring through the
    native P2P engine (value check; pipeline modes only)

Every rung prints a progress line to stderr when it starts and when it is done; a rung that does not finish
within its time limit makes the process print WHICH rung hung (plus the Python stacks of all threads) and leave
with a non-zero status - a fresh exit, never a re-exec - so a launcher tears the job down and the log names the
stage.  A value mismatch raises ``LadderError`` naming the rung.

``topology`` answers "did RCCL really see N GPUs?": the distinct physical devices (uuid / PCI bus id,
all-gathered over the run's own process group), ranks per device, and the transport.
"""
from __future__ import annotations

import faulthandler
import os
import socket
import sys
import threading
import time
from typing import Callable, Dict, List, Optional

import torch
import torch.distributed as dist

from . import comm

# environment variables that configure paths / launch plumbing and are NOT experiment switches
_BENIGN_ENV = {
    "DDL_DATA_DIR", "DDL_LOG_DIR", "DDL_CKPT_DIR", "DDL_PRETRAINED", "DDL_TRACE_DIR",
    "DDL_TEST_GPU_PER_RANK",
}
# switches that knowingly produce WRONG results (timing experiments): production entrypoints refuse them
_REFUSED_ENV = ("DDL_KNOCKOUT",)


class LadderError(RuntimeError):
    pass


def experimental_env() -> List[str]:
    """Every ``DDL_*`` variable set in this process that can change the code path or the numerics
    (``NAME=value``), sorted - recorded in the benchmark JSON so a stray switch is never silent."""
    return sorted(f"{k}={v}" for k, v in os.environ.items()
                  if k.startswith("DDL_") and k not in _BENIGN_ENV)


def refuse_wrong_result_switches(who: str) -> None:
    """``DDL_KNOCKOUT`` and friends leave kernels out of the step (results are wrong by design).  They only exist
    in ``bench/experiments/``; a production entrypoint that finds one in its environment stops."""
    bad = [k for k in _REFUSED_ENV if os.environ.get(k)]
    if bad:
        print(f"{who}: refusing to run with {', '.join(f'{k}={os.environ[k]}' for k in bad)} - that switch "
              f"produces wrong results (timing experiments live in bench/experiments/knockout.py)",
              file=sys.stderr, flush=True)
        raise SystemExit(5)


# --------------------------------------------------------------------------------------
# who is here
# --------------------------------------------------------------------------------------
def device_identity(device: torch.device) -> str:
    """A string that is equal for two ranks iff they sit on the same physical device."""
    host = socket.gethostname()
    if device.type != "cuda":
        return f"{host}/cpu"
    try:
        p = torch.cuda.get_device_properties(device)
        uuid = getattr(p, "uuid", None)
        if uuid is not None and str(uuid).strip("0-") != "":
            return f"{host}/{uuid}"
        bus = [getattr(p, a, None) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id")]
        if any(b is not None for b in bus):
            return f"{host}/pci-" + ":".join("x" if b is None else f"{int(b):02x}" for b in bus)
    except Exception:      # noqa: BLE001 - identity is diagnostics, never fatal
        pass
    return f"{host}/cuda:{device.index}"


def topology(ctx: "comm.CommContext") -> Dict[str, object]:
    """``backend`` / ``world`` / ``distinct_devices`` / ``ranks_per_device`` / ``n_gpus`` of this run.
    ``n_gpus`` counts distinct physical GPUs (0 on CPU) - NOT the world size: two gloo ranks sharing one
    GPU report ``n_gpus: 1, ranks_per_device: 2``."""
    mine = device_identity(ctx.device)
    ids = [mine]
    if comm.is_dist() and comm.get_world_size() > 1:
        ids = [None] * comm.get_world_size()
        dist.all_gather_object(ids, mine)
    distinct = sorted(set(ids))
    on_gpu = ctx.device.type == "cuda"
    per = max(ids.count(d) for d in distinct)
    backend = comm.backend_of() if comm.is_dist() else "none"
    return {
        "backend": ("rccl" if backend == "nccl" else backend),
        "world": len(ids),
        "distinct_devices": len(distinct),
        "ranks_per_device": per,
        "n_gpus": len(distinct) if on_gpu else 0,
        "device_ids": distinct,
    }


# --------------------------------------------------------------------------------------
# watchdog
# --------------------------------------------------------------------------------------
class StageWatchdog:
    """``with wd.stage("name", seconds): ...`` - progress lines on stderr; on expiry the process names the stage,
    dumps every thread's Python stack and exits with ``exit_code`` (a fresh exit: the GPU is initialised, so
    nothing is ever re-executed in place)."""

    def __init__(self, rank: int, who: str = "bench.py", exit_code: int = 4, quiet: bool = False,
                 on_expire: Optional[Callable[[str], None]] = None) -> None:
        self.rank, self.who, self.exit_code, self.quiet = rank, who, exit_code, quiet
        self.on_expire = on_expire
        self.completed: List[str] = []

    def _expire(self, name: str, seconds: float) -> None:
        print(f"[rank {self.rank}] {self.who}: STAGE '{name}' did not complete within {seconds:.0f} s - "
              f"completed before it: {self.completed or 'nothing'}.  Leaving with status {self.exit_code}.",
              file=sys.stderr, flush=True)
        try:
            faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
        except Exception:      # noqa: BLE001
            pass
        if self.on_expire is not None:
            try:
                self.on_expire(name)
            except Exception:      # noqa: BLE001
                pass
        sys.stderr.flush()
        os._exit(self.exit_code)

    def stage(self, name: str, seconds: float):
        return _Stage(self, name, seconds)


class _Stage:
    def __init__(self, wd: StageWatchdog, name: str, seconds: float) -> None:
        self.wd, self.name, self.seconds = wd, name, seconds
        self.timer: Optional[threading.Timer] = None

    def __enter__(self):
        wd = self.wd
        if not wd.quiet:
            print(f"[rank {wd.rank}] {wd.who}: stage '{self.name}' ...", file=sys.stderr, flush=True)
        self.t0 = time.perf_counter()
        self.timer = threading.Timer(self.seconds, wd._expire, args=(self.name, self.seconds))
        self.timer.daemon = True
        self.timer.start()
        return self

    def __exit__(self, et, ev, tb):
        self.timer.cancel()
        wd = self.wd
        if et is None:
            wd.completed.append(self.name)
            if not wd.quiet:
                print(f"[rank {wd.rank}] {wd.who}: stage '{self.name}' ok "
                      f"({(time.perf_counter() - self.t0) * 1e3:.0f} ms)", file=sys.stderr, flush=True)
        return False


# --------------------------------------------------------------------------------------
# the ladder
# --------------------------------------------------------------------------------------
def _sync(dev: torch.device) -> None:
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)


def _check(name: str, got: torch.Tensor, want: torch.Tensor, rank: int) -> None:
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    if got.shape != want.shape or not torch.allclose(got, want, rtol=1e-6, atol=1e-6):
        bad = int((~torch.isclose(got, want, rtol=1e-6, atol=1e-6)).sum()) if got.shape == want.shape else -1
        raise LadderError(f"[rank {rank}] first-contact ladder: stage '{name}' returned wrong values "
                          f"({bad} of {want.numel()} elements differ; got[:4]={got.flatten()[:4].tolist()} "
                          f"want[:4]={want.flatten()[:4].tolist()})")


def first_contact_ladder(ctx: "comm.CommContext", pipeline: bool = False, wd: Optional[StageWatchdog] = None,
                         stage_seconds: float = 120.0, numel: int = 1 << 16, rccl: str = "auto") -> List[str]:
    """Run the rungs listed in the module docstring on the WORLD group; returns the names of the completed
    stages.  A world of one runs them only under ``DDL_FORCE_COLLECTIVES=1`` (one-GPU rehearsal; the P2P ring
    then also needs ``DDL_P2P_ALLOW_SELF=1``)."""
    from . import ops
    from .parallel.p2p import P2PChannel
    from .utils.streams import new_stream

    world, rank, dev = comm.get_world_size(), comm.get_rank(), ctx.device
    if not comm.is_dist() or (world == 1 and not comm.force_collectives()):
        return []
    wd = wd or StageWatchdog(rank)
    n = (numel // (world * 64)) * world * 64 or world * 64          # divisible like an arena segment
    ranks = torch.arange(world, dtype=torch.float32)

    with wd.stage("barrier", stage_seconds):
        comm.barrier()
        _sync(dev)

    with wd.stage("all-reduce", stage_seconds):
        t = torch.full((n,), float(rank + 1), dtype=torch.float32, device=dev)
        comm.all_reduce(t, "sum")
        _sync(dev)
        _check("all-reduce", t[:8], torch.full((8,), float((ranks + 1).sum())), rank)

    # rank r contributes (r + 1) * (1 + i / n) at position i: the average over ranks is known, and a shard that
    # ends up at the wrong offset (an in-place reduce-scatter gone wrong) is a value error, not a hang
    base = 1.0 + torch.arange(n, dtype=torch.float32, device=dev) / n
    want = base * float((ranks + 1).mean())
    native = (dev.type == "cuda" and comm.backend_of() == "nccl" and ops.have_ext()
              and os.environ.get("DDL_NATIVE_REDUCER", "1") == "1")
    direct = None
    if native and comm.rccl_transport(rccl) == "direct":
        # the communicator the data-parallel engine of a world-wide group will reuse (same tag, same members)
        with wd.stage("owned RCCL communicator: unique id through the store, ncclCommInitRank, all-reduce", stage_seconds):
            direct = comm.direct_comm(list(range(world)), dev, tag="dp")
            t = torch.full((n,), float(rank + 1), dtype=torch.float32, device=dev)
            direct.all_reduce(t, False)
            _sync(dev)
            direct.check_async()
            _check("owned RCCL communicator all-reduce", t[:8], torch.full((8,), float((ranks + 1).sum())), rank)
    for form in ("rs_ag", "allreduce"):
        name = f"bucket reducer ({('csrc/reducer.cpp, ' + ('owned communicator' if direct is not None else 'c10d')) if native else 'python'}): " + \
               ("in-place reduce-scatter(AVG) + all-gather" if form == "rs_ag" else "all-reduce(AVG)")
        with wd.stage(name, stage_seconds):
            g = base * float(rank + 1)
            half = n // 2 // (world * 64) * world * 64 or n
            bounds = [(0, half), (half, n)] if half < n else [(0, n)]
            if native:
                cs = new_stream(dev)
                red = ops.require_ext().BucketReducer(
                    g, [a for a, _ in bounds], [b for _, b in bounds], list(range(len(bounds))), [1] * len(bounds),
                    dist.group.WORLD, rank, world, False, form == "allreduce", False, False,
                    cs.stream_id, cs.device_index, cs.device_type, direct)
                red.arm(True, True)
                for k in range(len(bounds)):
                    red.segment_ready(k)
                red.finish()
            else:
                for a, b in bounds:
                    seg = g[a:b]
                    if form == "allreduce":
                        comm.all_reduce(seg, "avg")
                    else:
                        m = (b - a) // world
                        shard = seg[rank * m:(rank + 1) * m]
                        comm.reduce_scatter_into(shard, seg, "avg")
                        comm.all_gather_into(seg, shard)
            _sync(dev)
            _check(name, g, want, rank)

    if pipeline and (world > 1 or os.environ.get("DDL_P2P_ALLOW_SELF") == "1"):
        with wd.stage("grouped send/recv ring (P2P engine; one owned communicator + stream per neighbour unless --rccl c10d)",
                      stage_seconds):
            nxt, prv = (rank + 1) % world, (rank - 1) % world
            chan = P2PChannel(None, dev, neighbours=[nxt, prv], rccl=rccl)
            payload = torch.full((4096,), float(rank + 1), dtype=torch.float32, device=dev)
            inbox = torch.zeros(4096, dtype=torch.float32, device=dev)
            if world == 2 or world == 1:
                # two ranks: the send and the receive address the SAME peer - one group, like a 1F1B exchange
                ticket = chan.post(sends=[(payload, nxt)], recvs=[(inbox, prv)])
                chan.wait(ticket)
            else:
                # a ring over >2 ranks: even ranks send first, odd ranks receive first (separate groups toward
                # different peers are ordered per rank - the pipeline schedules' pattern)
                if rank % 2 == 0:
                    chan.wait(chan.post(sends=[(payload, nxt)]))
                    chan.wait(chan.post(recvs=[(inbox, prv)]))
                else:
                    chan.wait(chan.post(recvs=[(inbox, prv)]))
                    chan.wait(chan.post(sends=[(payload, nxt)]))
            chan.drain()
            _sync(dev)
            _check("grouped send/recv ring (P2P engine)", inbox[:8], torch.full((8,), float(prv + 1)), rank)
    return list(wd.completed)

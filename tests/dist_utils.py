"""Multi-process CPU (gloo) test harness: spawn ``world`` ranks, run ``fn(rank, world, *args)``
in each with an initialised process group on 127.0.0.1, propagate failures."""
import os
import socket
import sys
import traceback

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_device(rank: int) -> torch.device:
    """cuda:0 for every rank (ranks share the one GPU of the test box, gloo) - or cuda:<rank> when the test was
    started with ``backend="nccl"`` on a machine with a GPU per rank."""
    return torch.device("cuda", rank if os.environ.get("DDL_TEST_GPU_PER_RANK") == "1" else 0)


def _worker(rank, world, port, fn, args, errq, backend="gloo"):
    try:
        if ROOT not in sys.path:
            sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                          WORLD_SIZE=str(world), LOCAL_RANK=str(rank), OMP_NUM_THREADS="1")
        torch.set_num_threads(1)
        if backend == "nccl":                  # RCCL between distinct GPUs (machines with >= world GPUs only)
            os.environ["DDL_TEST_GPU_PER_RANK"] = "1"
            torch.cuda.set_device(rank)
        dist.init_process_group(backend, rank=rank, world_size=world)
        fn(rank, world, *args)
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        errq.put((rank, traceback.format_exc()))
        raise


def run_distributed(fn, world: int, *args, timeout: int = 240, backend: str = "gloo"):
    ctx = mp.get_context("spawn")
    errq = ctx.SimpleQueue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fn, args, errq, backend)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout)
    alive = [p for p in procs if p.is_alive()]
    for p in alive:
        p.terminate()
    errs = []
    while not errq.empty():
        errs.append(errq.get())
    if errs:
        raise AssertionError("\n".join(f"--- rank {r} ---\n{tb}" for r, tb in errs))
    if alive:
        raise AssertionError(f"{len(alive)} ranks hung (timeout {timeout}s)")
    bad = [p.exitcode for p in procs if p.exitcode != 0]
    if bad:
        raise AssertionError(f"rank exit codes {bad}")

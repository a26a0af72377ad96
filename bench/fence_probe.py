"""Fence tests that CAN fail.

On a one-GPU box every RCCL collective of a world of one is an identity, so a dropped bucket-ready / grads-ready /
data-ready / arrived event produces the same bits.  The engines therefore carry a test hook (constructor flags, never
environment variables): with it, the communication / P2P stream does visible work between its two fences - a bounded
spin, then "*= 0.5" on the data it just moved - and a second flag deliberately leaves ONE fence out.  This script runs
under ``torchrun --nproc-per-node 1`` with ``DDL_FORCE_COLLECTIVES=1 DDL_P2P_ALLOW_SELF=1`` and prints one line per
case:

    FENCE <engine> <transport> <case> got=<value> want=<value> <OK | DIFFERS>

Positive cases must print OK, negative controls must print DIFFERS; ``tests/test_gpu_fences.py`` asserts both.
Ordering guarantees under test: the reference's DDP makes ``optimizer.step`` wait for the reducer (ddp.py:166-170);
``batch_isend_irecv`` + ``wait`` order a transfer behind its payload and in front of its consumer (pp.py:138-150).

    part = reducer | p2p | buffers | steps
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from ddl import comm, ops
from ddl.parallel.p2p import P2PChannel
from ddl.utils.streams import new_stream, new_stream_beside, run_concurrently

part = sys.argv[1] if len(sys.argv) > 1 else "all"
ctx = comm.init("nccl", "auto", force=True)
dev = ctx.device
ext = ops.require_ext()
SPIN = 20000.0          # us: far longer than any launch gap, so a missing fence loses its race every time
N = 1 << 16


def beside(stream) -> bool:
    """True if `stream` really runs beside the compute stream.  Two streams that share a hardware queue are
    serialised on the device, so a missing fence between them can never show: a negative control on such a pair is
    INCONCLUSIVE, not a failure of the engine (utils/streams.py picks streams on other queues, but a box whose queues
    are all taken may not have one)."""
    return bool(run_concurrently(stream, torch.cuda.current_stream(dev)))


def report(engine, transport, case, got, want, expect_ok, concurrent=True):
    ok = got == want
    verdict = "OK" if ok else "DIFFERS"
    if not expect_ok and ok and not concurrent:
        verdict = "INCONCLUSIVE"
    print(f"FENCE {engine} {transport} {case} got={got} want={want} {verdict}"
          f"{'' if (ok == expect_ok or verdict == 'INCONCLUSIVE') else '  <-- UNEXPECTED'}", flush=True)


def reducer_case(transport, case, skip, delay, hook_spin):
    """compute stream: spin `delay`, THEN produce the gradient (fill 1.0).  comm stream (hook): spin `hook_spin`, THEN
    *= 0.5.  A fenced run gives 0.5 whatever the two delays are.
    bucket-ready missing (late producer): the hook halves the stale 3.0, the fill lands later -> 1.0
    grads-ready missing (slow comm stream): the reader runs before the hook -> 1.0"""
    g = torch.full((N,), 3.0, device=dev)
    direct = comm.direct_comm([0], dev, tag="dp") if transport == "direct" else None
    # a communication stream that really runs beside the compute stream: two streams that share a hardware queue are
    # serialised on the device and a missing fence could never show (utils/streams.py)
    cs = new_stream_beside(dev, [torch.cuda.current_stream(dev)])
    red = ext.BucketReducer(g, [0, N // 2], [N // 2, N], [0, 1], [1, 1], dist.group.WORLD, 0, 1, False, False, False,
                            False, cs.stream_id, cs.device_index, cs.device_type, direct, 1, hook_spin, skip)
    assert red.direct == (transport == "direct")
    torch.cuda.synchronize()
    red.arm(True, True)
    if delay:
        ext.test_spin(delay, dev.index)     # the producer is late ...
    g.fill_(1.0)                            # ... and this is the "backward" that writes the bucket
    red.param_ready(0)
    red.param_ready(1)
    red.finish()                            # grads-ready: compute waits for the comm stream
    out = g.clone()                         # the "optimizer" reads on the compute stream
    torch.cuda.synchronize()
    report("reducer", transport, case, (float(out.min()), float(out.max())), (0.5, 0.5), skip == 0, beside(cs))


def p2p_case(transport, case, skip, delay, hook_spin):
    """self transfer payload -> inbox; hook on the link stream: spin, inbox *= 0.5.
    data-ready missing -> the transfer copies the stale payload (3.0) -> 1.5;  arrived missing -> the reader sees the
    inbox before the hook ran;  slot fence missing -> the next step's pre-posted receive overwrites the slot before
    its (late) reader ran."""
    chan = P2PChannel(None, dev, neighbours=[0], rccl=transport, test_hook=1, hook_spin_us=hook_spin, skip_fence=skip)
    assert chan.native is not None and chan.transport == transport, (chan.transport, transport)
    payload = torch.full((N,), 3.0, device=dev)
    inbox = torch.zeros(N, device=dev)
    torch.cuda.synchronize()
    if skip != 3 and not case.endswith("slot"):
        if delay:
            ext.test_spin(delay, dev.index)
        payload.fill_(1.0)
        t = chan.post(sends=[(payload, 0)], recvs=[(inbox, 0)])
        chan.wait(t)
        out = inbox.clone()
        chan.drain()
        torch.cuda.synchronize()
        report("p2p", transport, case, (float(out.min()), float(out.max())), (0.5, 0.5), skip == 0,
               all(beside(st) for st in chan.streams[-1:]))
    else:
        # slot reuse across steps: step t's consumer is LATE (spin in front of it); step t+1's receive into the same
        # slot is pre-posted right after `fence()`
        inbox.fill_(7.0)
        torch.cuda.synchronize()
        ext.test_spin(delay, dev.index)
        seen = inbox.clone()                # step t's reader of the slot
        chan.fence()
        t = chan.post(sends=[(payload, 0)], recvs=[(inbox, 0)], after_compute=False)      # pre-posted receive
        chan.wait(t)
        chan.drain()
        torch.cuda.synchronize()
        report("p2p", transport, case, (float(seen.min()), float(seen.max())), (7.0, 7.0), skip == 0,
               all(beside(st) for st in chan.streams[-1:]))


def buffers_case(transport, case, skip, delay, hook_spin):
    """The deferred BN-buffer broadcast of the data-parallel engine (dp.after_forward): the "forward" writes the
    buffers (late, when `delay`); hook on the comm stream: spin, buffers *= 0.5; the next "forward" reads them."""
    from ddl.parallel.dp import DataParallelEngine
    m = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.BatchNorm1d(8)).to(dev)
    eng = DataParallelEngine(m, device=dev, rccl=transport, test_hook=1, hook_spin_us=hook_spin, skip_fence=skip,
                             register_hooks=False)
    assert eng.transport == transport and eng._defer_buffers, (eng.transport, eng._defer_buffers)
    ff = eng.buffers.flat_float
    ff.fill_(3.0)
    torch.cuda.synchronize()
    if delay:
        ext.test_spin(delay, dev.index)
    ff.fill_(1.0)                           # the forward's running-statistic update
    eng.after_forward()
    eng.arm(False)                          # no gradient buckets in this step: their grads-ready fence would cover for a
    eng.finish_grad_sync()                  #   missing buffers-done fence (same communication stream)
    out = ff.clone()                        # the next forward reads the buffers
    torch.cuda.synchronize()
    report("buffers", transport, case, (float(out.min()), float(out.max())), (0.5, 0.5), skip == "", beside(eng.comm_stream))


def steps_case():
    """Whole training steps of the native engine with the hook on: overlapped (phased hipGraphs, comm stream) against
    --debug-sync (everything on the compute stream, eager).  Every bucket is halved once per step in both, the engine
    is bit-deterministic -> identical parameters; then the same overlapped run with one fence left out."""
    from ddl.config import TrainConfig
    from ddl.trainer import build_strategy
    finals = {}
    # hook spin: a SLOW communication stream exposes a missing grads-ready fence (Adam runs ahead of the halving); a FAST
    # one exposes a missing bucket-ready fence (the halving runs ahead of the backward phase that writes the bucket -
    # phase 0's buckets are covered by the buffers-ready wait of the deferred broadcast on the same stream, the later
    # phases' are not).  The spin never changes the numerics of a fenced run.
    for name, debug, skip, hook_spin in (("overlapped", False, "", 300.0), ("debug-sync", True, "", 300.0),
                                         ("no-bucket-ready", False, "bucket_ready", 0.0),
                                         ("no-grads-ready", False, "grads_ready", 300.0)):
        torch.manual_seed(3)
        cfg = TrainConfig(strategy="ddp", batch_size=8, image_size=64, debug_sync=debug, bucket_mb=2.0)
        from ddl.parallel import dp as dp_mod
        orig = dp_mod.DataParallelEngine.__init__

        def patched(self, *a, _orig=orig, _skip=skip, _spin=hook_spin, **kw):
            _orig(self, *a, test_hook=1, hook_spin_us=_spin, skip_fence=_skip, **kw)
        dp_mod.DataParallelEngine.__init__ = patched
        try:
            strat = build_strategy(cfg, ctx)
        finally:
            dp_mod.DataParallelEngine.__init__ = orig
        strat.model.train()
        conc = beside(strat.dp.comm_stream)
        print(f"steps/{name}: comm stream beside the compute stream: {conc}", flush=True)
        g = torch.Generator().manual_seed(9)
        for step in range(5):
            x = torch.randint(0, 256, (8, 3, 64, 64), dtype=torch.uint8, generator=g).to(dev)
            y = torch.randint(0, 5, (8,), generator=g).to(dev)
            ext.test_spin(3000.0, dev.index)        # the step's compute starts late: an unfenced hook runs ahead of it
            strat.train_step(x, y, 8)
        torch.cuda.synchronize()
        finals[name] = (strat.arena.flat_param.clone(), strat.model.flat_buffers.flat_float.clone(), conc)
        del strat
    ref = finals["overlapped"]
    for name in ("debug-sync", "no-bucket-ready", "no-grads-ready"):
        same = bool(torch.equal(ref[0], finals[name][0]) and torch.equal(ref[1], finals[name][1]))
        verdict = "OK" if same else "DIFFERS"
        if same and name != "debug-sync" and not finals[name][2]:
            verdict = "INCONCLUSIVE"
        print(f"FENCE steps direct overlapped-vs-{name} got={'same' if same else 'different'} want=same {verdict}", flush=True)


if part in ("reducer", "all"):
    for transport in ("direct", "c10d"):
        reducer_case(transport, "fenced/late-producer", 0, SPIN, 0.0)
        reducer_case(transport, "fenced/slow-comm", 0, 0.0, SPIN)
        reducer_case(transport, "no-bucket-ready", 1, SPIN, 0.0)
        reducer_case(transport, "no-grads-ready", 2, 0.0, SPIN)
if part in ("p2p", "all"):
    for transport in ("direct", "c10d"):
        p2p_case(transport, "fenced/late-producer", 0, SPIN, 0.0)
        p2p_case(transport, "fenced/slow-link", 0, 0.0, SPIN)
        p2p_case(transport, "fenced/slot", 0, SPIN, 0.0)
        p2p_case(transport, "no-data-ready", 1, SPIN, 0.0)
        p2p_case(transport, "no-arrived", 2, 0.0, SPIN)
        p2p_case(transport, "no-slot-fence", 3, SPIN, 0.0)
if part in ("buffers", "all"):
    for transport in ("direct", "c10d"):
        buffers_case(transport, "fenced/late-forward", "", SPIN, 0.0)
        buffers_case(transport, "fenced/slow-comm", "", 0.0, SPIN)
        buffers_case(transport, "no-buffers-ready", "buffers_ready", SPIN, 0.0)
        buffers_case(transport, "no-buffers-done", "buffers_done", 0.0, SPIN)
if part in ("steps", "all"):
    steps_case()
print("FENCE PROBE DONE", flush=True)
comm.destroy()

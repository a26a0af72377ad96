"""Multi-process CPU (gloo) tests of the comm layer, the DP engine, the pipeline schedules and the
2-D hybrid - the coverage the reference never had (every distributed run there needs
real GPUs + NCCL)."""
import pytest
import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from dist_utils import run_distributed


# ------------------------------------------------------------------------------- comm / mesh
def _mesh_worker(rank, world):
    from ddl import comm
    mesh = comm.Mesh2D(world // 2, 2)
    assert (mesh.dp_index, mesh.pp_index) == divmod(rank, 2)
    t = torch.tensor([rank])
    comm.all_reduce(t, "sum", mesh.get_group("pp"))
    assert t.item() == 4 * mesh.dp_index + 1              # reference test.py: 1, 5, 9 on a 3x2 mesh
    t = torch.tensor([float(rank)])
    comm.all_reduce(t, "avg", mesh.get_group("dp"))
    assert t.item() == pytest.approx(mesh.pp_index + (world // 2 - 1))
    assert mesh.last_stage_rank_of_first_pipeline == 1    # ddp_n_pp.py:275 broadcast source
    # flat collectives (+ the gloo emulation of reduce-scatter)
    full = torch.arange(world * 3, dtype=torch.float32) + rank
    shard = full[rank * 3:(rank + 1) * 3]
    comm.reduce_scatter_into(shard, full, "avg")
    want = torch.arange(world * 3, dtype=torch.float32)[rank * 3:(rank + 1) * 3] + (world - 1) / 2
    torch.testing.assert_close(shard, want)
    comm.all_gather_into(full, shard)
    torch.testing.assert_close(full, torch.arange(world * 3, dtype=torch.float32) + (world - 1) / 2)
    g = comm.all_gather_flat(torch.full((2, 2), float(rank)))
    assert g.shape == (2 * world, 2) and g[2 * rank, 0] == rank
    assert comm.broadcast_flag(rank == 1, 1, "cpu") is True


@pytest.mark.parametrize("world", [4, 6])
def test_mesh_and_collectives(world):
    run_distributed(_mesh_worker, world)


def _checker_worker(rank, world):
    from ddl import comm
    chk = comm.enable_checker()
    comm.all_reduce(torch.ones(4), "sum")
    chk.verify()
    if rank == 1:
        chk.record("all_reduce", torch.ones(8), None)      # a rank that "issued" a different sequence
    try:
        chk.verify()
        raised = False
    except RuntimeError:
        raised = True
    assert raised
    comm.disable_checker()


def test_collective_order_checker():
    run_distributed(_checker_worker, 2)


# ------------------------------------------------------------------------------- DP
def _bn_free_net():
    torch.manual_seed(1)
    return nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.ReLU(), nn.Conv2d(8, 8, 3, padding=1), nn.ReLU(),
                         nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(8, 5))


def _dp_worker(rank, world, grad_sync, zero1):
    from ddl.optim import FusedAdam
    from ddl.parallel.dp import DataParallelEngine
    torch.manual_seed(100 + rank)                         # different init per rank: broadcast must fix it
    model = _bn_free_net() if rank == 0 else nn.Sequential(*[m for m in _bn_free_net()])
    if rank != 0:
        for p in model.parameters():
            p.data.add_(1.0)
    dp = DataParallelEngine(model, device="cpu", bucket_mb=0.0005, grad_sync=grad_sync, zero1=zero1)
    assert len(dp.arena.segments) > 1
    ref = _bn_free_net()
    for p, q in zip(model.parameters(), ref.parameters()):
        torch.testing.assert_close(p, q)                  # rank-0 params won
    opt = FusedAdam(dp.arena, shard=(rank, world) if zero1 else None)
    ref_opt = torch.optim.Adam(ref.parameters())
    g = torch.Generator().manual_seed(7)
    for step in range(3):
        X = torch.randn(4 * world, 3, 8, 8, generator=g)
        Y = torch.randint(0, 5, (4 * world,), generator=g)
        # single-process reference on the concatenated batch
        ref_opt.zero_grad()
        F.cross_entropy(ref(X), Y).backward()
        # data parallel on this rank's shard
        opt.zero_grad()
        dp.arm(True)
        F.cross_entropy(model(X[rank * 4:(rank + 1) * 4]), Y[rank * 4:(rank + 1) * 4]).backward()
        dp.finish_grad_sync()
        if not zero1:
            for p, q in zip(model.parameters(), ref.parameters()):
                torch.testing.assert_close(p.grad, q.grad, rtol=1e-4, atol=1e-6)
        ref_opt.step()
        opt.step()
        dp.post_step()
        for p, q in zip(model.parameters(), ref.parameters()):
            torch.testing.assert_close(p, q, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("grad_sync,zero1", [("rs_ag", False), ("allreduce", False), ("rs_ag", True)])
def test_dp_gradients_equal_single_process(grad_sync, zero1):
    """DP(W) grads == single-process grads on the concatenated batch."""
    run_distributed(_dp_worker, 2, grad_sync, zero1)


def _zero1_snapshot_worker(rank, world, path):
    """ZeRO-1 + snapshot: each rank's Adam moments are valid on its 1/W shard only; the snapshot must hold
    the whole state (DCP de-duplicates replicated keys -> (W-1)/W of the moments saved as zeros)."""
    from ddl import checkpoint as ck
    from ddl.optim import FusedAdam
    from ddl.parallel.dp import DataParallelEngine

    def make():
        model = _bn_free_net()
        dp = DataParallelEngine(model, device="cpu", bucket_mb=0.0005, zero1=True)
        return model, dp, FusedAdam(dp.arena, shard=(rank, world))

    def step(model, dp, opt, X, Y):
        opt.zero_grad()
        dp.arm(True)
        F.cross_entropy(model(X[rank * 4:(rank + 1) * 4]), Y[rank * 4:(rank + 1) * 4]).backward()
        dp.finish_grad_sync()
        opt.step()
        dp.post_step()

    g = torch.Generator().manual_seed(11)
    batches = [(torch.randn(4 * world, 3, 8, 8, generator=g), torch.randint(0, 5, (4 * world,), generator=g))
               for _ in range(4)]
    model, dp, opt = make()
    ref = _bn_free_net()                                   # unsharded single-process oracle
    ref_opt = torch.optim.Adam(ref.parameters())
    for X, Y in batches[:2]:
        step(model, dp, opt, X, Y)
        ref_opt.zero_grad()
        F.cross_entropy(ref(X), Y).backward()
        ref_opt.step()
    dp.gather_sharded(opt.exp_avg, opt.exp_avg_sq)         # what Trainer._save_snapshot does under --zero1
    for p, q in zip(model.parameters(), ref.parameters()):
        torch.testing.assert_close(opt.state[p]["exp_avg"], ref_opt.state[q]["exp_avg"], rtol=1e-4, atol=1e-7)
        torch.testing.assert_close(opt.state[p]["exp_avg_sq"], ref_opt.state[q]["exp_avg_sq"], rtol=1e-4, atol=1e-9)
    ck.save_snapshot(path, 0, model, opt)
    for X, Y in batches[2:]:
        step(model, dp, opt, X, Y)                         # the uninterrupted run
    # resume into a fresh, differently initialised replica set
    torch.manual_seed(999 + rank)
    model2, dp2, opt2 = make()
    for p in model2.parameters():
        p.data.add_(0.5)
    assert ck.load_snapshot(path, model2, opt2) == 0
    n_nonzero = int((opt2.exp_avg != 0).sum())
    assert n_nonzero > 0.9 * dp2.arena.total_param_numel()          # not the mostly-zero state of one rank
    for X, Y in batches[2:]:
        step(model2, dp2, opt2, X, Y)
    for p, q in zip(model.parameters(), model2.parameters()):
        torch.testing.assert_close(p, q, rtol=0, atol=0)           # resumed == uninterrupted, bit for bit


def test_zero1_snapshot_resume_equals_uninterrupted_run(tmp_path):
    run_distributed(_zero1_snapshot_worker, 2, str(tmp_path / "snap"))


def _dp_buffers_worker(rank, world):
    from ddl.parallel.dp import DataParallelEngine
    torch.manual_seed(rank)
    model = nn.Sequential(nn.Conv2d(3, 4, 1), nn.BatchNorm2d(4))
    dp = DataParallelEngine(model, device="cpu")
    model.train()
    model(torch.randn(8, 3, 4, 4) + rank)                 # ranks' running stats now differ
    dp.pre_forward()                                      # DDP broadcast_buffers semantics
    mine = model[1].running_mean.clone()
    got = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(got, mine)
    torch.testing.assert_close(got[0], got[1])


def test_dp_buffer_broadcast():
    run_distributed(_dp_buffers_worker, 2)


# ------------------------------------------------------------------------------- PP / hybrid
class _ToyStage(nn.Module):
    def __init__(self, idx, n, width=16):
        super().__init__()
        torch.manual_seed(50 + idx)
        self.first, self.last = idx == 0, idx == n - 1
        self.body = nn.Sequential(nn.Linear(12 if self.first else width, width), nn.Tanh())
        self.head = nn.Linear(width, 5) if self.last else None

    def forward(self, x):
        x = self.body(x)
        return self.head(x) if self.last else x


def _pp_worker(rank, world, schedule, n_mb, dp_size):
    from ddl import comm
    from ddl.optim import FusedAdam
    from ddl.parallel.dp import DataParallelEngine
    from ddl.parallel.pp import SCHEDULES, PipelineStage, expected_order
    from ddl.utils.arena import FlatArena, reverse_named_parameters
    S = world // dp_size
    mesh = comm.Mesh2D(dp_size, S)
    st_idx = mesh.pp_index
    module = _ToyStage(st_idx, S)
    dp = DataParallelEngine(module, group=mesh.get_group("dp"), device="cpu") if dp_size > 1 else None
    stage = PipelineStage(module, st_idx, S, "cpu", group=mesh.get_group("pp"), ranks=mesh.pp_ranks,
                          recv_shape_fn=lambda mb: (mb, 16), dp_engine=dp)
    sched = SCHEDULES[schedule](stage, n_mb, loss_fn=F.cross_entropy)
    infer = SCHEDULES[schedule](stage, n_mb)
    B = 2 * n_mb                                           # per-replica batch
    g = torch.Generator().manual_seed(3)
    X = torch.randn(B * dp_size, 12, generator=g)
    Y = torch.randint(0, 5, (B * dp_size,), generator=g)
    xs, ys = X[mesh.dp_index * B:(mesh.dp_index + 1) * B], Y[mesh.dp_index * B:(mesh.dp_index + 1) * B]

    # single-process reference: all stages chained, full batch, mean loss
    full = nn.Sequential(*[_ToyStage(i, S) for i in range(S)])
    F.cross_entropy(full(X), Y).backward()
    want = [p.grad for p in full[st_idx].parameters()]

    if dp is not None:
        dp.arena.zero_grad()
        dp.arm(False)
    losses = []
    if stage.is_first and not stage.is_last:
        out = sched.step(xs)
    elif stage.is_last:
        out = sched.step(target=ys, losses=losses)
    else:
        out = sched.step(batch_size=B)
    if dp is not None:
        dp.finish_grad_sync()
    assert sched.order == expected_order(schedule, st_idx, S, n_mb)
    for p, w in zip(module.parameters(), want):
        torch.testing.assert_close(p.grad, w, rtol=1e-4, atol=1e-6)      # PP(S,M) grads == single (1/M scaling)
    if stage.is_last:
        assert out.shape == (B, 5) and len(losses) == n_mb
        torch.testing.assert_close(out, full(xs), rtol=1e-5, atol=1e-6)  # merged logits of the whole batch
    else:
        assert out is None
    with torch.no_grad():
        if stage.is_first and not stage.is_last:
            infer.step(xs)
        elif stage.is_last:
            o = infer.step(batch_size=B)
            torch.testing.assert_close(o, full(xs), rtol=1e-5, atol=1e-6)
        else:
            infer.step(batch_size=B)
        try:
            sched.step(xs if stage.is_first else None, target=ys if stage.is_last else None, batch_size=B)
            raised = False
        except RuntimeError:
            raised = True                                  # loss schedule under no_grad is refused
        assert raised


@pytest.mark.parametrize("schedule", ["gpipe", "1f1b"])
@pytest.mark.parametrize("world,n_mb", [(2, 5), (4, 4), (4, 7), (3, 2)])
def test_pipeline_matches_single_process(schedule, world, n_mb):
    run_distributed(_pp_worker, world, schedule, n_mb, 1)


@pytest.mark.parametrize("schedule", ["gpipe", "1f1b"])
def test_hybrid_dp_x_pp_matches_single_process(schedule):
    run_distributed(_pp_worker, 4, schedule, 3, 2)        # 2-way DP x 2-stage PP


def test_expected_orders():
    from ddl.parallel.pp import expected_order
    assert expected_order("gpipe", 0, 2, 3) == ["F0", "F1", "F2", "B0", "B1", "B2"]
    assert expected_order("1f1b", 0, 2, 4) == ["F0", "F1", "B0", "F2", "B1", "F3", "B2", "B3"]
    assert expected_order("1f1b", 1, 2, 4) == ["F0", "B0", "F1", "B1", "F2", "B2", "F3", "B3"]
    assert expected_order("1f1b", 0, 4, 2) == ["F0", "F1", "B0", "B1"]


def test_link_creation_order_never_deadlocks():
    """parallel/p2p.py link_pairs: every rank creates its per-neighbour communicators in ascending (lower, higher)
    order.  Simulated with blocking semantics (a rank sits in its current init until the partner arrives at the SAME
    pair): pipelines of 2..8 stages and rings of 2..8 ranks (the first-contact ladder) must all run to completion."""
    from ddl.parallel.p2p import link_pairs

    def run(world, neighbours_of):
        todo = {r: list(link_pairs(r, neighbours_of(r))) for r in range(world)}
        for _ in range(10 * world):
            progressed = False
            for r in range(world):
                if not todo[r]:
                    continue
                a, b = todo[r][0]
                peer = b if a == r else a
                if peer == r:                                  # self link (one-GPU rehearsal)
                    todo[r].pop(0)
                    progressed = True
                elif todo[peer] and todo[peer][0] == (a, b):   # both members sit in the same init: it completes
                    todo[r].pop(0)
                    todo[peer].pop(0)
                    progressed = True
            if not any(todo.values()):
                return True
            if not progressed:
                return False
        return False

    for S in range(2, 9):
        assert run(S, lambda r, S=S: [x for x in (r - 1, r + 1) if 0 <= x < S]), f"pipeline of {S} stages deadlocks"
        assert run(S, lambda r, S=S: [(r + 1) % S, (r - 1) % S]), f"ring of {S} ranks deadlocks"
    assert link_pairs(0, [0]) == [(0, 0)]
    assert link_pairs(3, [2, 4]) == [(2, 3), (3, 4)] and link_pairs(0, [1, 7]) == [(0, 1), (0, 7)]


def _direct_comm_worker(rank, world):
    """comm.direct_comm / direct_comm_or_fallback rendezvous on CPU with a stand-in extension: member 0's unique id
    must reach every member through the c10d store byte for byte (NUL bytes included), repeated creations use fresh
    keys, the cache returns the same object, and a member that fails makes EVERY member fall back together."""
    import os
    from ddl import comm, ops

    class FakeComm:
        def __init__(self, uid, r, n, dev):
            if os.environ.get("FAKE_RCCL_FAIL_RANK") == str(r):
                raise RuntimeError("ncclCommInitRank failed (simulated)")
            self.uid, self.rank, self.nranks, self.device, self.aborted = bytes(uid), r, n, dev, False

        def abort(self):
            self.aborted = True

        def destroy(self):
            pass

    class FakeExt:
        RcclComm = FakeComm
        calls = 0

        @staticmethod
        def rccl_unique_id():
            FakeExt.calls += 1
            return bytes([0, 255, 0, FakeExt.calls]) + os.urandom(124)

    real = ops.require_ext
    ops.require_ext = lambda: FakeExt
    try:
        dev = torch.device("cpu")
        c1 = comm.direct_comm(list(range(world)), dev, tag="t")
        assert (c1.rank, c1.nranks) == (rank, world) and len(c1.uid) == 128 and c1.uid[:3] == b"\x00\xff\x00"
        ids = [None] * world
        dist.all_gather_object(ids, c1.uid)
        assert all(i == ids[0] for i in ids)                              # every member joined with member 0's id
        assert comm.direct_comm(list(range(world)), dev, tag="t") is c1     # cached
        c2 = comm.direct_comm(list(range(world)), dev, tag="t", fresh=True)
        ids2 = [None] * world
        dist.all_gather_object(ids2, c2.uid)
        assert all(i == ids2[0] for i in ids2) and ids2[0] != ids[0]        # a fresh key, a fresh id
        sub = comm.direct_comm([0, world - 1], dev, tag="pair") if rank in (0, world - 1) else None
        if sub is not None:
            assert sub.nranks == (2 if world > 1 else 1) and sub.rank == (0 if rank == 0 else 1)
        # collective fallback: rank 1 fails -> nobody keeps a communicator
        os.environ["FAKE_RCCL_FAIL_RANK"] = "1"
        got = comm.direct_comm_or_fallback(list(range(world)), dev, "failing", None, who="test")
        assert got is None and ("failing", tuple(range(world))) not in comm._DIRECT
        os.environ.pop("FAKE_RCCL_FAIL_RANK")
        ok = comm.direct_comm_or_fallback(list(range(world)), dev, "fine", None, who="test")
        assert ok is not None and ok.nranks == world
    finally:
        ops.require_ext = real
        comm._DIRECT.clear()


def test_owned_communicator_rendezvous_through_the_store():
    run_distributed(_direct_comm_worker, 3, timeout=120)

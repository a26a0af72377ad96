import copy

import pytest
import torch

from ddl.optim import FusedAdam
from ddl.utils.arena import FlatArena, FlatBuffers, reverse_named_parameters


def _mlp():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(7, 13), torch.nn.BatchNorm1d(13), torch.nn.ReLU(),
                               torch.nn.Linear(13, 5))


def test_arena_views_alias_params_and_grads():
    m = _mlp()
    ref = [p.detach().clone() for p in m.parameters()]
    a = FlatArena(reverse_named_parameters(m), segment_numel=64, segment_multiple=2 * 64)
    for p, r in zip(m.parameters(), ref):
        assert torch.equal(p, r)
    m(torch.randn(4, 7)).sum().backward()
    for i, p in enumerate(a.params):
        assert p.grad.data_ptr() == a.grad_view(i).data_ptr()        # autograd accumulated in place
    assert a.flat_grad.abs().sum() > 0
    assert len(a.segments) > 1 and all((s.end - s.start) % 128 == 0 for s in a.segments)
    assert a.segments[0].start == 0 and a.segments[-1].end == a.numel
    a.zero_grad()
    assert all(p.grad.abs().sum() == 0 for p in a.params)


def test_flat_buffers_alias():
    m = _mlp()
    fb = FlatBuffers(m)
    m.train()
    m(torch.randn(8, 7))
    assert fb.flat_float.abs().sum() > 0 and int(fb.flat_int.sum()) == 1


@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_fused_adam_matches_torch_adam(wd):
    """Same trajectory as optim.Adam defaults (single.py:108,305) over several steps."""
    def net():      # no BN: a bias in front of BN has a ~0 gradient, which Adam turns into noise
        torch.manual_seed(0)
        return torch.nn.Sequential(torch.nn.Linear(7, 13), torch.nn.Tanh(), torch.nn.Linear(13, 5))
    m1, m2 = net(), net()
    o1 = torch.optim.Adam(m1.parameters(), weight_decay=wd)
    o2 = FusedAdam(FlatArena(reverse_named_parameters(m2)), weight_decay=wd)
    for step in range(5):
        x = torch.randn(6, 7, generator=torch.Generator().manual_seed(step))
        for m, o in ((m1, o1), (m2, o2)):
            o.zero_grad()
            m(x).pow(2).mean().backward()
            o.step()
    for p1, p2 in zip(m1.parameters(), m2.parameters()):
        torch.testing.assert_close(p1, p2, rtol=1e-5, atol=1e-7)


def test_fused_adam_state_dict_roundtrip_keeps_views():
    m = _mlp()
    opt = FusedAdam(FlatArena(reverse_named_parameters(m)))
    m(torch.randn(4, 7)).sum().backward()
    opt.step()
    sd = copy.deepcopy(opt.state_dict())
    assert set(sd["param_groups"][0]) >= {"lr", "betas", "eps", "weight_decay", "amsgrad", "params"}
    st0 = sd["state"][0]
    assert set(st0) == {"step", "exp_avg", "exp_avg_sq"} and float(st0["step"]) == 1.0
    ptr = opt.exp_avg.data_ptr()
    opt.exp_avg.zero_()
    opt.load_state_dict(sd)
    assert opt.exp_avg.data_ptr() == ptr and opt.exp_avg.abs().sum() > 0
    assert opt._host_step == 1


def test_zero1_shard_updates_only_its_slice():
    m = _mlp()
    a = FlatArena(reverse_named_parameters(m), segment_numel=64, segment_multiple=2 * 64)
    before = a.flat_param.clone()
    a.flat_grad.fill_(1.0)
    FusedAdam(a, shard=(1, 2)).step()
    changed = (a.flat_param != before)
    for seg in a.segments:
        half = (seg.end - seg.start) // 2
        assert not changed[seg.start:seg.start + half].any()
        assert changed[seg.start + half:seg.end].all()


def test_arena_break_before_starts_a_new_segment():
    """Segments can be forced to start at named parameters (the native engine aligns gradient buckets
    with the phases of its chained backward graphs), and stay divisible for reduce-scatter."""
    import torch
    from ddl.utils.arena import FlatArena
    ps = [(f"p{i}", torch.nn.Parameter(torch.randn(100 + i))) for i in range(6)]
    a = FlatArena(ps, segment_numel=10_000, segment_multiple=4 * 64, break_before=["p2", "p5", "p0"])
    assert [[a.names[i] for i in seg.slots] for seg in a.segments] == [["p0", "p1"], ["p2", "p3", "p4"], ["p5"]]
    assert all((seg.end - seg.start) % (4 * 64) == 0 for seg in a.segments)
    assert [s.segment for s in a.slots] == [0, 0, 1, 1, 1, 2]
    for (n, p), s in zip(ps, a.slots):
        assert p.data_ptr() == a.flat_param[s.offset:].data_ptr()


@pytest.mark.parametrize("world", [1, 2, 3, 4, 6, 8, 16])
@pytest.mark.parametrize("bucket_mb", [2.0, 8.0, 32.0])
def test_segments_split_evenly_for_every_world_size(world, bucket_mb):
    """The reduce-scatter / all-gather buckets and the ZeRO-1 shards of the real DenseNet-121 parameter list: every
    segment must divide by the data-parallel world size (dp.py raises otherwise), the per-rank shard ranges must
    tile each segment exactly, and no parameter may straddle two segments.  Pure layout arithmetic - checked here
    for the world sizes the suite never spawns (the driver's 8-GPU node, 3/6 for non-powers of two)."""
    from ddl.models.densenet import UNIT_NAMES, DenseNetStage
    model = DenseNetStage(UNIT_NAMES)
    rev = reverse_named_parameters(model)
    # phase cuts of the native engine (native_densenet.py: a new segment at the first parameter, in backward order,
    # of the unit in front of every dense block)
    breaks = [next(n for n, _ in rev if n.startswith(f"features.{u}.")) for u in ("transition3", "transition2", "transition1", "norm0")]
    arena = FlatArena(rev, segment_numel=int(bucket_mb * 2 ** 20 / 4), segment_multiple=world * 64, break_before=breaks)
    assert all(any(arena.slot_of(b).offset == s.start for s in arena.segments) for b in breaks)
    assert arena.total_param_numel() == sum(p.numel() for p in model.parameters())
    covered = 0
    for seg in arena.segments:
        n = seg.end - seg.start
        assert n > 0 and n % world == 0 and seg.start % 64 == 0
        covered += n
    assert covered == arena.numel and arena.segments[0].start == 0
    assert all(a.end == b.start for a, b in zip(arena.segments[:-1], arena.segments[1:]))
    for i, p in enumerate(arena.params):                      # every parameter inside ONE segment
        v = arena.param_view(i)
        off = (v.data_ptr() - arena.flat_param.data_ptr()) // 4
        assert any(s.start <= off and off + p.numel() <= s.end for s in arena.segments)
    seen = torch.zeros(arena.numel, dtype=torch.int32)
    for r in range(world):
        opt = FusedAdam(arena, shard=(r, world))
        ranges = opt._ranges()
        assert len(ranges) == len(arena.segments)
        for s, e in ranges:
            seen[s:e] += 1
    assert bool((seen == 1).all())                             # shards tile the arena: no gap, no overlap


def test_fused_zero_grad_leaves_a_clean_arena_and_the_same_losses():
    """``GraphedTrainStep(fuse_zero_grad=True)``: the Adam pass clears the gradients it consumed, so no step starts
    with a memset (reference order zero_grad -> forward -> backward -> step, single.py:137-154, same result);
    ``--log-gradients`` keeps them readable after the step."""
    import torch
    from ddl import comm
    from ddl.config import TrainConfig
    from ddl.trainer import SingleStrategy
    dev = torch.device("cpu")
    g = torch.Generator().manual_seed(3)
    x = torch.randint(0, 256, (2, 3, 32, 32), dtype=torch.uint8, generator=g)
    y = torch.randint(0, 5, (2,), generator=g)
    runs = {}
    for log in (False, True):
        torch.manual_seed(0)
        strat = SingleStrategy(TrainConfig(strategy="single", batch_size=2, log_gradients=log),
                               comm.CommContext(0, 1, 0, dev, "none"))
        assert strat._step.fuse_zero_grad == (not log)
        losses = [float(strat.train_step(x, y, 2)[0]) for _ in range(3)]
        gnorm = float(strat.optimizer.arena.flat_grad.abs().sum())
        assert (gnorm == 0.0) == (not log)
        runs[log] = losses
    assert runs[False] == runs[True]

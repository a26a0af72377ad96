import torch
import torch.distributed.checkpoint as dcp
from torch.distributed.checkpoint.state_dict import get_state_dict

from ddl import checkpoint as ck
from ddl.models import densenet as dn
from ddl.optim import FusedAdam
from ddl.utils.arena import FlatArena, reverse_named_parameters


def _small():
    torch.manual_seed(0)
    return dn.DenseNetStage(("transition3",), 5)     # BN + conv: params and buffers, fast


def _step(model, opt):
    opt.zero_grad()
    model(torch.randn(2, 1024, 4, 4)).mean().backward()
    opt.step()


def test_keys_match_get_state_dict_layout():
    """AppState must produce the same tree as the reference's get_state_dict-based AppState
    (single.py:76-83): FQN-keyed optimizer state, Adam param_group fields."""
    m = _small()
    opt = torch.optim.Adam(m.parameters())
    _step(m, opt)
    ref_model, ref_optim = get_state_dict(m, opt)
    mine = ck.AppState(3, m, opt).state_dict()
    assert set(mine) == {"model", "optim", "epoch"} and int(mine["epoch"]) == 3
    assert set(mine["model"]) == set(ref_model)
    assert set(mine["optim"]["state"]) == set(ref_optim["state"])
    fq = next(iter(ref_optim["state"]))
    assert set(mine["optim"]["state"][fq]) == set(ref_optim["state"][fq])
    assert set(mine["optim"]["param_groups"][0]) == set(ref_optim["param_groups"][0])
    assert mine["optim"]["param_groups"][0]["params"] == ref_optim["param_groups"][0]["params"]
    torch.testing.assert_close(mine["optim"]["state"][fq]["exp_avg"], ref_optim["state"][fq]["exp_avg"])


def test_fused_adam_checkpoint_has_same_keys_as_torch_adam(tmp_path):
    m1, m2 = _small(), _small()
    o1 = torch.optim.Adam(m1.parameters())
    o2 = FusedAdam(FlatArena(reverse_named_parameters(m2)))
    _step(m1, o1)
    _step(m2, o2)
    ck.save_snapshot(str(tmp_path / "a"), 0, m1, o1)
    ck.save_snapshot(str(tmp_path / "b"), 0, m2, o2)
    ka, kb = ck.read_keys(str(tmp_path / "a")), ck.read_keys(str(tmp_path / "b"))
    assert ka == kb
    assert "app.epoch" in ka and "app.model.features.transition3.norm.running_mean" in ka
    assert "app.optim.state.features.transition3.conv.weight.exp_avg_sq" in ka
    assert any(k.startswith("app.optim.param_groups") for k in ka)
    assert (tmp_path / "b" / ".metadata").exists()


def test_roundtrip_resume_in_place(tmp_path):
    m = _small()
    opt = FusedAdam(FlatArena(reverse_named_parameters(m)))
    _step(m, opt)
    _step(m, opt)
    path = ck.snapshot_path(str(tmp_path), "job-x", 4)
    ck.save_snapshot(path, 4, m, opt, suffix="1")                 # pp-style suffix (pp.py:87-88)
    assert "app.model1.features.transition3.conv.weight" in ck.read_keys(path)
    want_p = {k: v.clone() for k, v in m.state_dict().items()}
    want_m = opt.exp_avg.clone()

    m2 = dn.DenseNetStage(("transition3",), 5)
    opt2 = FusedAdam(FlatArena(reverse_named_parameters(m2)))
    ptr = opt2.arena.flat_param.data_ptr()
    assert ck.load_snapshot(path, m2, opt2, suffix="1") == 4
    for k, v in m2.state_dict().items():
        torch.testing.assert_close(v, want_p[k])
    torch.testing.assert_close(opt2.exp_avg, want_m)
    assert opt2._host_step == 2 and float(opt2.step_t) == 2.0
    assert opt2.arena.flat_param.data_ptr() == ptr                # views survived the load
    assert next(m2.parameters()).data_ptr() >= ptr
    assert ck.latest_snapshot(str(tmp_path), "job-x") == 4


def test_stage_checkpoints_partition_the_single_key_space(tmp_path):
    full = {k for k in dn.densenet121(5).state_dict()}
    got = set()
    for i, units in enumerate(dn.stage_units(["features.denseblock3.denselayer1"])):
        got |= set(dn.DenseNetStage(units).state_dict())
    assert got == full

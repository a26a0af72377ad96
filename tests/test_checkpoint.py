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


# ---------------------------------------------------------------------------------------------------
# Interop with the reference's own AppState (single.py:68-89, pp.py:77-99): get_state_dict / set_state_dict +
# dcp.save / dcp.load.  "Same checkpoint format" means a snapshot written by either side loads on the other.
# ---------------------------------------------------------------------------------------------------
class _RefAppState(torch.distributed.checkpoint.stateful.Stateful):
    """The reference's Stateful wrapper, re-stated from its description: model/optim state through
    ``get_state_dict`` / ``set_state_dict``, keys ``model{S}`` / ``optim{S}`` / ``epoch``."""

    def __init__(self, epoch, model, optimizer, suffix=""):
        self.epoch, self.model, self.optimizer, self.suffix = epoch, model, optimizer, suffix

    def state_dict(self):
        from torch.distributed.checkpoint.state_dict import get_state_dict
        m, o = get_state_dict(self.model, self.optimizer)
        return {f"model{self.suffix}": m, f"optim{self.suffix}": o, "epoch": torch.tensor(self.epoch)}

    def load_state_dict(self, sd):
        from torch.distributed.checkpoint.state_dict import set_state_dict
        set_state_dict(self.model, self.optimizer, model_state_dict=sd[f"model{self.suffix}"],
                       optim_state_dict=sd[f"optim{self.suffix}"])
        self.epoch = int(sd["epoch"])


def _train_two_steps(model, opt, shape, seed=5):
    g = torch.Generator().manual_seed(seed)
    for _ in range(2):
        opt.zero_grad()
        (model(torch.randn(*shape, generator=g)).square().mean() * 1e4).backward()   # keep gradients far above Adam's eps
        opt.step()


def _assert_same_state(m_a, o_a, m_b, o_b):
    sa, sb = m_a.state_dict(), m_b.state_dict()
    assert set(sa) == set(sb)
    for k in sa:
        torch.testing.assert_close(sa[k], sb[k], rtol=0, atol=0)
    pa, pb = dict(m_a.named_parameters()), dict(m_b.named_parameters())
    for n in pa:
        for key in ("exp_avg", "exp_avg_sq"):
            torch.testing.assert_close(o_a.state[pa[n]][key], o_b.state[pb[n]][key], rtol=0, atol=0)
        assert float(o_a.state[pa[n]]["step"]) == float(o_b.state[pb[n]]["step"])


import pytest  # noqa: E402


@pytest.mark.parametrize("units,suffix,shape", [
    (dn.UNIT_NAMES, "", (2, 3, 64, 64)),                                      # single.py / ddp.py layout
    (dn.stage_units(["features.denseblock3.denselayer1"])[1], "1", (2, 256, 4, 4)),   # pp.py stage 1: model1 / optim1
])
def test_snapshot_written_here_loads_through_the_reference_appstate(tmp_path, units, suffix, shape):
    torch.manual_seed(0)
    m = dn.DenseNetStage(units, 5)
    opt = FusedAdam(FlatArena(reverse_named_parameters(m)))
    _train_two_steps(m, opt, shape)
    path = str(tmp_path / "mine")
    ck.save_snapshot(path, 7, m, opt, suffix=suffix)

    torch.manual_seed(1)
    m2 = dn.DenseNetStage(units, 5)
    opt2 = torch.optim.Adam(m2.parameters())
    app = _RefAppState(0, m2, opt2, suffix)
    dcp.load({"app": app}, checkpoint_id=path)                                # the reference's _load_snapshot
    assert app.epoch == 7
    _assert_same_state(m, opt, m2, opt2)
    # and training continues identically from the restored state on both sides
    _train_two_steps(m, opt, shape, seed=9)
    _train_two_steps(m2, opt2, shape, seed=9)
    for (n, p), (_, q) in zip(m.named_parameters(), m2.named_parameters()):
        torch.testing.assert_close(p, q, rtol=1e-4, atol=3e-4, msg=lambda s, n=n: f"{n}: {s}")   # two Adam implementations; lr = 1e-3 per step


@pytest.mark.parametrize("units,suffix,shape", [
    (dn.UNIT_NAMES, "", (2, 3, 64, 64)),
    (dn.stage_units(["features.denseblock3.denselayer1"])[0], "0", (2, 3, 64, 64)),   # pp.py stage 0
])
def test_snapshot_written_by_the_reference_appstate_loads_here(tmp_path, units, suffix, shape):
    torch.manual_seed(0)
    m = dn.DenseNetStage(units, 5)
    opt = torch.optim.Adam(m.parameters())
    _train_two_steps(m, opt, shape)
    path = str(tmp_path / "ref")
    dcp.save({"app": _RefAppState(3, m, opt, suffix)}, checkpoint_id=path)    # the reference's _save_snapshot

    torch.manual_seed(1)
    m2 = dn.DenseNetStage(units, 5)
    opt2 = FusedAdam(FlatArena(reverse_named_parameters(m2)))
    ptr = opt2.arena.flat_param.data_ptr()
    assert ck.load_snapshot(path, m2, opt2, suffix=suffix) == 3
    assert opt2.arena.flat_param.data_ptr() == ptr                            # loaded into the arena views
    _assert_same_state(m, opt, m2, opt2)
    assert opt2._host_step == 2
    _train_two_steps(m, opt, shape, seed=9)
    _train_two_steps(m2, opt2, shape, seed=9)
    for (n, p), (_, q) in zip(m.named_parameters(), m2.named_parameters()):
        torch.testing.assert_close(p, q, rtol=1e-4, atol=3e-4, msg=lambda s, n=n: f"{n}: {s}")   # two Adam implementations; lr = 1e-3 per step

import pytest
import torch

from ddl.models import densenet as dn


def test_param_and_buffer_counts():
    """6,958,981 params in 364 tensors, 121 BN layers -> 83,769 buffer elements."""
    m = dn.densenet121(5)
    assert sum(p.numel() for p in m.parameters()) == 6_958_981
    assert len(list(m.parameters())) == 364
    assert sum(b.numel() for b in m.buffers()) == 83_769
    assert sum(p.numel() for p in dn.densenet121(1000).parameters()) == 7_978_856   # torchvision's count


def test_torchvision_key_names():
    keys = list(dn.densenet121(5).state_dict())
    for k in ["features.conv0.weight", "features.norm0.running_mean",
              "features.denseblock1.denselayer1.norm1.weight",
              "features.denseblock3.denselayer24.conv2.weight", "features.transition2.conv.weight",
              "features.norm5.num_batches_tracked", "classifier.weight", "classifier.bias"]:
        assert k in keys


def test_reference_two_stage_split():
    """pp.py:383-385 split -> stage sizes 1,429,632 / 5,529,349, boundary [mb,256,14,14]."""
    units = dn.stage_units(["features.denseblock3.denselayer1"])
    sizes = [sum(p.numel() for p in dn.DenseNetStage(u).parameters()) for u in units]
    assert sizes == [1_429_632, 5_529_349]
    assert dn.DenseNetStage(units[0]).output_shape((6, 3, 224, 224)) == (6, 256, 14, 14)


def test_four_stage_split_partitions_keys():
    spec = dn.default_split_points(4)
    full = dn.densenet121(5)
    seen = []
    for u in dn.stage_units(spec):
        seen += list(dn.DenseNetStage(u).state_dict())
    assert sorted(seen) == sorted(full.state_dict())


def test_bad_split_points_rejected():
    for bad in (["features.denseblock3.denselayer22.conv1"],     # inside a dense layer: skip tensors would cross
                ["features.denseblock3.denselayer25"], ["features.conv0"], ["features.nonsense"]):
        with pytest.raises(ValueError):
            dn.stage_units(bad)


def test_mid_block_split_is_one_tensor_and_partitions_the_model():
    """A stage may start at any dense LAYER (the reference's fx split could not: debug.py:9 "skip connections"): the
    concatenated prefix is the one tensor that crosses.  Stages compose to the full model, keys partition."""
    spec = ["features.denseblock2.denselayer5", "features.denseblock3.denselayer9"]
    units = dn.stage_units(spec)
    assert units[0][-1] == "denseblock2[0:4]" and units[1][0] == "denseblock2[4:12]" and units[2][0] == "denseblock3[8:24]"
    torch.manual_seed(0)
    full = dn.densenet121(5).eval()
    sd = full.state_dict()
    x = torch.rand(2, 3, 64, 64)
    h, seen = x, []
    for i in range(3):
        st = dn.build_stage(i, spec, 5, sd).eval()
        assert st.output_shape(tuple(h.shape)) == tuple(st(h).shape)
        h = st(h)
        seen += list(st.state_dict())
    torch.testing.assert_close(h, full(x))
    assert sorted(seen) == sorted(sd)
    assert dn.DenseNetStage(units[0]).output_shape((6, 3, 224, 224)) == (6, 128 + 4 * 32, 28, 28)
    with pytest.raises(ValueError):
        dn.DenseNetStage(("denseblock2[4:12]", "denseblock3"))     # not contiguous: transition2 is missing


def test_balanced_split_points():
    """--split-points auto: linear partition over the 63 atoms; the analytic cost model (CPU path) puts the 2-stage cut
    inside denseblock2 (the reference's denseblock3 split leaves 73 % of the work on stage 0)."""
    from ddl.models import split
    costs = split.atom_costs([m * 1e-9 for m in split.analytic_unit_macs()])
    assert len(costs) == len(dn.model_atoms()) == 63
    for S, tol in ((2, 0.05), (3, 0.10), (4, 0.25), (8, 0.45)):
        starts = split.balanced_starts(costs, S)
        loads = split.stage_loads(costs, starts)
        assert len(loads) == S and max(loads) / (sum(loads) / S) < 1 + tol, (S, loads)
        assert len(dn.stage_units(split.starts_to_split_points(starts))) == S
    assert split.starts_to_split_points(split.balanced_starts(costs, 2))[0].startswith("features.denseblock2.denselayer")
    assert split.balanced_starts([1, 1, 1, 1], 2) == (2,) and split.balanced_starts([5, 1, 1, 1], 2) == (1,)


def test_stages_compose_to_full_model():
    torch.manual_seed(0)
    full = dn.densenet121(5).eval()
    sd = full.state_dict()
    x = torch.rand(2, 3, 64, 64)
    want = full(x)
    h = x
    for i in range(4):
        st = dn.build_stage(i, dn.default_split_points(4), 5, sd).eval()
        assert st.output_shape(tuple(h.shape)) == tuple(st(h).shape)
        h = st(h)
    torch.testing.assert_close(h, want)


def test_uint8_input_is_normalised():
    m = dn.densenet121(5).eval()
    x = torch.randint(0, 256, (1, 3, 64, 64), dtype=torch.uint8)
    torch.testing.assert_close(m(x), m(x.float() / 255.0))


def test_pretrained_file_loading(tmp_path):
    """--pretrained: a torchvision-style 1000-class state-dict (incl. legacy 'norm.1' key names) initialises
    features; the ImageNet classifier is discarded like in the reference (single.py:297-299); pipeline
    stages pick up only their own keys."""
    import torch
    from ddl.models import create_stage
    from ddl.models.densenet import UNIT_NAMES, DenseNetStage
    from ddl.models.pretrained import normalise_keys, read_state_dict
    torch.manual_seed(0)
    src = DenseNetStage(UNIT_NAMES, 1000)
    sd = {}
    for k, v in src.state_dict().items():
        k = k.replace("denselayer2.norm1.", "denselayer2.norm.1.").replace("denselayer2.conv2.", "denselayer2.conv.2.")
        sd["module." + k] = v
    path = str(tmp_path / "densenet121.pth")
    torch.save({"state_dict": sd}, path)
    assert "features.denseblock1.denselayer2.norm1.weight" in read_state_dict(path)
    assert normalise_keys({"a.denselayer3.conv.1.weight": torch.zeros(1)}).keys() == {"a.denselayer3.conv1.weight"}
    torch.manual_seed(1)
    full = create_stage(UNIT_NAMES, "cpu", 5, pretrained=path)
    ref = src.state_dict()
    for k, v in full.state_dict().items():
        if k.startswith("classifier."):
            assert tuple(v.shape) in ((5, 1024), (5,))
        else:
            torch.testing.assert_close(v, ref[k], rtol=0, atol=0)
    stage = create_stage(("denseblock3", "transition3"), "cpu", 5, pretrained=path)
    k = "features.denseblock3.denselayer7.conv1.weight"
    torch.testing.assert_close(stage.state_dict()[k], ref[k], rtol=0, atol=0)

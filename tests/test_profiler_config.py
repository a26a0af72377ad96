"""Small host-side pieces: the step profiler's span / source accounting and the flag system."""
import pytest

from ddl.config import SCRIPT_DEFAULTS, TrainConfig, parse_config
from ddl.profiler import StepProfiler


def test_profiler_spans_sources_and_fractions():
    prof = StepProfiler("cpu")
    calls = []

    def drain():                      # an externally timed span (the native bucket reducer reports RCCL time so)
        calls.append(1)
        return 2.0 if len(calls) == 1 else 0.0

    prof.add_source("rccl", drain)
    for _ in range(4):
        tok = prof.begin("step")
        with prof.span("exposed_comm"):
            pass
        prof.end(tok)
        prof.step_done()
    s = prof.summary()
    assert s["steps"] == 4 and s["rccl_ms_per_step"] == pytest.approx(0.5)
    assert 0.0 <= s["comm_exposed_fraction"] <= 1.0 and s["compute_fraction"] == pytest.approx(1.0 - s["comm_exposed_fraction"])
    prof.set_enabled(False)                        # paused: nothing is recorded or counted
    tok = prof.begin("step")
    prof.end(tok)
    prof.step_done()
    assert tok is None and prof.summary()["steps"] == 4
    prof.set_enabled(True)
    prof.reset()
    assert prof.summary()["steps"] == 0 and "rccl_ms_per_step" not in prof.summary()
    off = StepProfiler("cpu", enabled=False)
    off.add_source("rccl", drain)
    assert off.begin("step") is None and off.summary() == {"steps": 0}


def test_script_defaults_are_the_reference_literals():
    # single.py:286 batch 30, ddp.py:335 batch 15, pp.py:365,378 batch 30 / 5 micro-batches, ddp_n_pp.py:371,404
    assert parse_config("single", []).batch_size == 30
    assert parse_config("ddp", []).batch_size == 15
    pp = parse_config("pp", [])
    assert (pp.batch_size, pp.num_microbatches, pp.split_points) == (30, 5, ["features.denseblock3.denselayer1"])
    hy = parse_config("ddpnpp", [])
    assert (hy.batch_size, hy.num_microbatches, hy.pp_size) == (10, 5, 2)
    for cfg in (pp, hy):
        assert cfg.epochs == 30 and cfg.lr == 1e-3 and (cfg.beta1, cfg.beta2, cfg.eps) == (0.9, 0.999, 1e-8)
    assert set(SCRIPT_DEFAULTS) == {"single", "ddp", "pp", "ddpnpp"}


def test_every_field_is_a_flag_and_new_switches_parse():
    cfg = parse_config("ddp", ["--cache-dataset", "off", "--cache-budget-gb", "2.5", "--dtype", "bf16", "--zero1",
                               "--grad-phases", "denseblock4", "--hip-graph", "false", "--seed", "3",
                               "--split-points", "features.denseblock2.denselayer1", "features.denseblock4.denselayer1"])
    assert (cfg.cache_dataset, cfg.cache_budget_gb, cfg.dtype, cfg.zero1, cfg.hip_graph, cfg.seed) == \
        ("off", 2.5, "bf16", True, False, 3)
    assert cfg.grad_phases == "denseblock4" and len(cfg.split_points) == 2
    assert TrainConfig().replace(batch_size=7).batch_size == 7
    with pytest.raises(SystemExit):
        parse_config("single", ["--no-such-flag"])


def test_trace_ranges_and_profiler_capture(tmp_path):
    """utils/trace.py: ranges are no-ops when off; StepTrace records a window of steps into a Chrome trace and a table."""
    import json
    import torch
    from ddl.utils import trace
    assert not trace.enabled()
    with trace.trace_range("off"):
        pass
    st = trace.StepTrace(str(tmp_path / "tr"), rank=3, skip=1, steps=2)
    for i in range(5):
        st.tick()
        with trace.trace_range(f"train step {i}"):
            torch.ones(64, 64) @ torch.ones(64, 64)
    assert st.done and not trace.enabled()
    ev = json.load(open(tmp_path / "tr" / "trace_rank3.json"))["traceEvents"]
    names = {e.get("name") for e in ev}
    assert "train step 1" in names and "train step 2" in names and "train step 0" not in names and "train step 3" not in names
    assert "aten::mm" in open(tmp_path / "tr" / "kernels_rank3.txt").read()
    assert trace.StepTrace(None).done

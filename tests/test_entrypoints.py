"""Each of the reference's entry scripts runs one short epoch on synthetic data on CPU
(BASELINE.json configs 1-2)."""
import csv
import json
import os
import subprocess
import sys

import pytest

from dist_utils import ROOT, free_port

SMALL = ["--epochs", "1", "--max-steps", "2", "--train-samples", "64", "--eval-samples", "32",
         "--image-size", "64", "--num-workers", "0", "--device", "cpu"]


def _env(tmp_path):
    env = dict(os.environ, DDL_LOG_DIR=str(tmp_path / "logs"), DDL_CKPT_DIR=str(tmp_path / "ckpt"),
               DDL_DATA_DIR=str(tmp_path / "nodata"), OMP_NUM_THREADS="1", TORCHX_JOB_ID="k8s://ns/job-123")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    return env


def _run(cmd, env, timeout=420):
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-4000:]
    return r.stdout


def _torchrun(n, script, args, env):
    return _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
                 "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script, *args], env)


def _metric_files(tmp_path):
    d = tmp_path / "logs" / "by_job_id" / "job-123"
    return {p.stem: list(csv.reader(open(p))) for p in d.glob("*.csv")}


ALL_METRICS = {"epoch_time", "loss", "train_accuracy", "val_loss", "val_accuracy", "macro_f1", "weighted_f1",
               "macro_precision", "weighted_precision", "macro_recall", "weighted_recall", "qwk"}


def test_single_epoch_and_resume(tmp_path):
    env = _env(tmp_path)
    out = _run([sys.executable, "single.py", *SMALL, "--batch-size", "8", "--save-every", "1"], env)
    assert "Epoch 0 | Training Accuracy" in out and "Saved snapshot" in out
    files = _metric_files(tmp_path)
    assert set(files) == ALL_METRICS                               # the reference's 12 per-epoch metrics
    assert files["loss"][0][1:6] == ["job-123", "0", "0", "job-123", "0"]
    assert (tmp_path / "ckpt" / "single" / "job-123" / "epoch_0" / ".metadata").exists()
    out = _run([sys.executable, "single.py", *SMALL[2:], "--epochs", "2", "--batch-size", "8",
                "--snapshot-job-id", "job-123", "--snapshot-epoch", "0"], env)
    assert "Resuming training from snapshot at Epoch 1" in out and "Starting Epoch 1" in out
    assert "Starting Epoch 0" not in out


def test_ddp_world2(tmp_path):
    out = _torchrun(2, "ddp.py", [*SMALL, "--batch-size", "8", "--profile", "--check-collectives"], _env(tmp_path))
    assert "[GPU0] Epoch 0 | Validation QWK" in out
    files = _metric_files(tmp_path)
    assert set(files) == ALL_METRICS and all(r[2] == "0" for r in files["qwk"])   # rank 0 logs


@pytest.mark.parametrize("schedule", ["1f1b", "gpipe"])
def test_pp_two_stage(tmp_path, schedule):
    out = _torchrun(2, "pp.py", [*SMALL, "--batch-size", "10", "--schedule", schedule], _env(tmp_path))
    assert "[GPU1] Epoch 0 | Validation QWK" in out                # last stage logs (pp.py:222-238)
    assert all(r[2] == "1" for r in _metric_files(tmp_path)["loss"])


def test_pp_four_stage(tmp_path):
    from ddl.models.densenet import default_split_points
    out = _torchrun(4, "pp.py", [*SMALL, "--batch-size", "10", "--split-points", *default_split_points(4)],
                    _env(tmp_path))
    assert "[GPU3] Epoch 0 | Validation QWK" in out


def test_pp_auto_split_mid_block(tmp_path):
    """--split-points auto: the CPU path balances on analytic costs and cuts INSIDE a dense block; three stages
    then train / evaluate / log like the unit-boundary splits."""
    out = _torchrun(3, "pp.py", [*SMALL, "--batch-size", "10", "--split-points", "auto"], _env(tmp_path))
    assert "[pp] auto split for 3 stages" in out and "denselayer" in out
    assert "[GPU2] Epoch 0 | Validation QWK" in out


def test_hybrid_2x2_with_snapshot(tmp_path):
    out = _torchrun(4, "ddp_n_pp.py", [*SMALL, "--batch-size", "10", "--save-every", "1"], _env(tmp_path))
    assert "[GPU1] Epoch 0 | Validation QWK" in out                # last stage of dp-rank 0 (mesh[0][-1])
    from ddl import checkpoint as ck
    keys = ck.read_keys(str(tmp_path / "ckpt" / "ddpnpp" / "job-123" / "epoch_0"))
    assert "app.model0.features.conv0.weight" in keys              # stage-suffixed keys (ddp_n_pp.py:94-95)
    assert "app.model1.classifier.weight" in keys and "app.optim1.state.classifier.bias.exp_avg" in keys


def test_communication_time_ping_and_sweep(tmp_path):
    env = _env(tmp_path)
    out = _torchrun(2, "communication_time.py",
                    ["--iters", "5", "--numel", "4096", "--sweep", "--max-bytes", "16384", "--device", "cpu",
                     "--log-dir", str(tmp_path / "logs"), "--json", str(tmp_path / "sweep.json")], env)
    rows = list(csv.reader(open(tmp_path / "logs" / "communication_time.csv")))
    assert len(rows) == 5 and rows[0][0] == "job-123" and rows[4][1] == "4"        # job_id,i,ms
    res = json.load(open(tmp_path / "sweep.json"))
    assert {r["op"] for r in res["sweep"]} == {"send_recv", "all_reduce", "rs_ag"}


def test_mesh_smoke_and_split_explorer(tmp_path):
    out = _torchrun(4, "test.py", ["--device", "cpu"], _env(tmp_path))
    assert "all_reduce over pp group -> 1" in out and "all_reduce over pp group -> 5" in out
    out = _run([sys.executable, "debug.py"], _env(tmp_path))
    assert "Model parameters: 7978856" in out and "Total num params: 7978856" in out


def test_bench_json_contract_cpu(tmp_path):
    """bench.py prints ONE JSON line with the driver's fields; on CPU it runs the PyTorch stage over gloo (world 2)."""
    env = dict(os.environ, OMP_NUM_THREADS="2")
    env.pop("RANK", None), env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()), "bench.py", "--gpus", "2",
                        "--steps", "1", "--warmup", "0", "--batch-per-gpu", "2"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config"):
        assert k in rec, k
    assert rec["n_gpus"] == 2 and rec["steps"] == 1 and rec["warmup"] == 0 and rec["unit"] == "samples/s"
    assert rec["higher_is_better"] is True and rec["scaling"] == "weak" and rec["dtype"] == "fp32"
    assert rec["config"]["global_batch"] == 4 and rec["config"]["parallelism"] == "dp2"
    assert rec["value"] == pytest.approx(4 * 1 / (rec["ms_per_step"] / 1e3), rel=1e-3)
    assert "profile" in rec and rec["profile"]["profile_steps"] == 1 and rec["profile"]["step_ms_per_step"] > 0
    # multi-rank runs return the tuning data in the same line: both collective forms and the bucket sweep
    assert set(rec["grad_sync_ab"]) == {"rs_ag", "allreduce"} and set(rec["bucket_mb_sweep"]) == {"8.0", "2.0", "32.0"}
    assert "extras" not in rec


def test_bench_watchdog_keeps_the_headline_when_the_extras_hang(tmp_path):
    """The A/B extras run behind the timed region; if they cannot finish (here: a 0.2 s budget) every rank still
    leaves with status 0 and rank 0 prints the ONE JSON line, marked as without extras."""
    env = dict(os.environ, OMP_NUM_THREADS="2")
    env.pop("RANK", None), env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()), "bench.py", "--gpus", "2",
                        "--steps", "1", "--warmup", "0", "--batch-per-gpu", "2", "--extras-timeout", "0.2"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and "abandoned" in rec["extras"] and "grad_sync_ab" not in rec


def test_rank_failure_aborts_job_and_resume_latest_recovers(tmp_path):
    """Failure path (absent in the reference): a rank that dies mid-epoch takes the job down promptly
    with a non-zero status instead of leaving its peer blocked in a collective; --resume-latest then restarts from
    the last complete snapshot."""
    import time
    env = _env(tmp_path)
    args = [*SMALL[2:], "--epochs", "3", "--batch-size", "8", "--save-every", "1"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), "ddp.py", *args, "--fault-inject", "1:1:1"]
    t0 = time.time()
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=420)
    assert r.returncode != 0 and time.time() - t0 < 300
    assert "injected fault on rank 1 at epoch 1 step 1" in r.stdout and "aborting the job" in r.stdout
    assert "Epoch 0 | Saved snapshot" in r.stdout and "Epoch 1 | Saved snapshot" not in r.stdout
    out = _torchrun(2, "ddp.py", [*args, "--resume-latest"], env)
    assert "Resuming training from snapshot at Epoch 1" in out and "Starting Epoch 0" not in out
    assert "Epoch 2 | Saved snapshot" in out

"""Fence tests that can FAIL: the stream fences of the bucket reducer (csrc/reducer.cpp), the P2P
engine (csrc/p2p.cpp) and the deferred BN-buffer broadcast (parallel/dp.py), on the real RCCL with a world of one.

Every RCCL collective of a world of one is an identity, so the engines carry a test hook (constructor flags): the
communication stream halves the data it just moved, between its two fences, behind a bounded spin - and a second flag
leaves one fence out.  ``bench/fence_probe.py`` runs the cases under ``torchrun`` (one rank, forced collectives, self
transfers) for both transports (owned communicators / c10d); here every positive case must come out right and every
negative control must come out WRONG - a fence test that passes without its fence proves nothing.
Reference ordering guarantees: ddp.py:166-170 (optimizer.step waits for the reducer), pp.py:138-150 (batch_isend_irecv).
"""
import os
import re
import subprocess
import sys

import pytest

from dist_utils import ROOT, free_port

pytestmark = pytest.mark.gpu

LINE = re.compile(r"^FENCE (\S+) (\S+) (\S+) got=(.*) want=(.*) (OK|DIFFERS|INCONCLUSIVE)")


def _probe(part, tmp_path):
    env = dict(os.environ, DDL_FORCE_COLLECTIVES="1", DDL_P2P_ALLOW_SELF="1", OMP_NUM_THREADS="2",
               DDL_LOG_DIR=str(tmp_path / "logs"), DDL_CKPT_DIR=str(tmp_path / "ckpt"))
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()), "bench/fence_probe.py", part],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=420)
    assert r.returncode == 0 and "FENCE PROBE DONE" in r.stdout, r.stdout[-4000:]
    rows = [m.groups() for m in (LINE.match(l) for l in r.stdout.splitlines()) if m]
    print("\n".join(l for l in r.stdout.splitlines() if l.startswith("FENCE")))
    return rows


def _check(rows, n_pos, n_neg):
    pos = [r for r in rows if r[2].startswith(("fenced", "overlapped-vs-debug-sync"))]
    neg = [r for r in rows if r not in pos]
    assert len(pos) == n_pos and len(neg) == n_neg, rows
    for r in pos:
        assert r[5] == "OK", f"fenced run gave the wrong data: {r}"
    for r in neg:
        # INCONCLUSIVE = the probe found the two streams on ONE hardware queue (serialised on the device whatever the
        # events say): a negative control cannot differ there, and that is a property of the box, not of the engine
        assert r[5] in ("DIFFERS", "INCONCLUSIVE"), \
            f"negative control passed WITHOUT its fence - the test cannot see that fence: {r}"
    if neg and all(r[5] == "INCONCLUSIVE" for r in neg):
        print("WARNING: every negative control was inconclusive on this box (no stream beside the compute stream)")


def test_bucket_reducer_fences_hold_and_their_absence_shows(tmp_path):
    """bucket-ready (compute -> comm) and grads-ready (comm -> compute) of csrc/reducer.cpp, owned communicator and c10d."""
    _check(_probe("reducer", tmp_path), n_pos=4, n_neg=4)


def test_p2p_engine_fences_hold_and_their_absence_shows(tmp_path):
    """data-ready, arrived and the slot-reuse fence of csrc/p2p.cpp, per-neighbour links and c10d."""
    _check(_probe("p2p", tmp_path), n_pos=6, n_neg=6)


def test_deferred_buffer_broadcast_fences_hold_and_their_absence_shows(tmp_path):
    """parallel/dp.py after_forward: buffers-ready (forward -> comm) and buffers-done (comm -> next forward)."""
    _check(_probe("buffers", tmp_path), n_pos=4, n_neg=4)


def test_training_steps_overlapped_equal_debug_sync_with_the_hook_on_and_differ_without_a_fence(tmp_path):
    """Five real training steps of the native engine, every gradient bucket halved by the hook on the communication
    stream: phased hipGraphs + comm stream == --debug-sync (bit-identical parameters and BN buffers); the same
    overlapped run with the bucket-ready / grads-ready fence left out must NOT be identical."""
    _check(_probe("steps", tmp_path), n_pos=1, n_neg=2)

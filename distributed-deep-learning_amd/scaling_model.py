"""MODELLED multi-GPU step times - what the measured one-GPU numbers and the xGMI link rate imply for 2 / 4 / 8 GPUs.

No machine this framework was built on had more than one GPU (SURVEY 7.4 hard part 8): the scaling curve the
north star asks for can only be MEASURED by `bench.py` on an 8-GPU node (`python -m torch.distributed.run ...
bench.py --gpus N [--mode pp|ddpnpp]`, which also returns the compute / RCCL / exposed split per step).  Until such a
run exists this module states, with every assumption spelled out, what the measured single-GPU numbers predict - so
that a first real measurement has something to be compared with.  Every number it prints is labelled MODELLED.

Reference context: the reference publishes epoch times only (ddp 1.48x, pp 0.82x, ddp x pp
0.67x of its single-GPU rate, on a cluster whose inter-node fabric ran at ~0.22 GB/s).

Model
  * xGMI: 7 links per GPU, `link_gbps` per direction per link (default 64 GB/s = the ~153 GB/s bidirectional figure
    halved and derated to 84 % for protocol overhead); `lat_us` per collective / transfer launch.
  * data parallel: the step is the measured one-GPU step through the data-parallel machinery (`t_step_ms`, world of
    one, every collective issued).  Buckets are the arena segments; the backward graph chain releases bucket k when
    phase k ends, so a bucket's reduce-scatter + all-gather (direct algorithm on the fully connected node: every GPU
    exchanges 1/W of the bucket with each of the W-1 peers over its own link, both halves) overlaps everything behind
    its phase; only what is still running when the LAST phase ends is exposed.
  * pipeline: GPipe / 1F1B with S stages and M micro-batches take (M + S - 1) slots of the slowest stage's
    per-micro-batch time (measured on one stage, lanes on) + one activation hop per stage boundary on the critical path.

    python -m ddl.scaling_model            # prints the tables (also in profiles/r3/scaling_model.txt)
"""
from __future__ import annotations

import argparse
from dataclasses import dataclass
from typing import Dict, List, Sequence

PARAM_BYTES = 6_958_981 * 4          # DenseNet-121, 5-class head


@dataclass
class Link:
    link_gbps: float = 64.0          # per direction per link, GB/s
    lat_us: float = 25.0             # launch + synchronisation cost of one collective / transfer group


def rs_ag_ms(nbytes: float, world: int, link: Link) -> float:
    """In-place reduce-scatter + all-gather of `nbytes` over a fully connected group of `world` GPUs (direct
    algorithm: each phase moves nbytes / world to every peer, all W-1 links in parallel)."""
    if world <= 1:
        return 0.0
    per_phase = nbytes / world / (link.link_gbps * 1e9) * 1e3
    return 2 * (per_phase + link.lat_us * 1e-3)


def ring_allreduce_ms(nbytes: float, world: int, link: Link) -> float:
    """Ring all-reduce (one outgoing link per GPU): 2 (W-1)/W of the bytes over ONE link."""
    if world <= 1:
        return 0.0
    return 2 * (world - 1) / world * nbytes / (link.link_gbps * 1e9) * 1e3 + 2 * (world - 1) * link.lat_us * 1e-3 / 4


def ddp_step_ms(world: int, t_step_ms: float, bucket_bytes: Sequence[float], phase_end_ms: Sequence[float],
                adam_ms: float, link: Link, form: str = "rs_ag") -> Dict[str, float]:
    """`bucket_bytes[k]` leaves when backward phase k ends, `phase_end_ms[k]` after the start of the step; the comm
    stream runs the buckets in order; Adam starts when compute AND the last bucket are done."""
    if world <= 1:
        return {"step_ms": t_step_ms, "comm_busy_ms": 0.0, "comm_exposed_ms": 0.0}
    fn = rs_ag_ms if form == "rs_ag" else ring_allreduce_ms
    t_comm = 0.0
    busy = 0.0
    for nbytes, ready in zip(bucket_bytes, phase_end_ms):
        dur = fn(nbytes, world, link)
        t_comm = max(t_comm, ready) + dur
        busy += dur
    compute_end = t_step_ms - adam_ms
    exposed = max(0.0, t_comm - compute_end)
    return {"step_ms": t_step_ms + exposed, "comm_busy_ms": busy, "comm_exposed_ms": exposed}


def pipeline_step_ms(stages: int, microbatches: int, t_mb_stage_ms: float, act_bytes: float, link: Link) -> float:
    """Fill-drain / 1F1B makespan: (M + S - 1) slots of the slowest stage's forward+backward time per micro-batch,
    plus one activation and one gradient hop per stage boundary on the critical path."""
    hop = act_bytes / (link.link_gbps * 1e9) * 1e3 + link.lat_us * 1e-3
    return (microbatches + stages - 1) * t_mb_stage_ms + 2 * (stages - 1) * hop


def tables(t_single_ms: float = 8.56, t_ddp1_ms: float = 8.75, adam_ms: float = 0.05, link: Link = Link()) -> str:
    """The MODELLED tables for the configurations BASELINE.json names.  Inputs (all measured on ONE MI355X, fp32):
    t_single_ms - bench.py at batch 30; t_ddp1_ms - the same step through the data-parallel machinery (world of one,
    owned RCCL communicator, profiles/r3/bench_forced_ddp_direct.json); phase ends from the graph chain of the default
    `--grad-phases denseblock3,denseblock2` (backward of head+block4 | block3 | block2+block1+stem)."""
    out: List[str] = []
    out.append("ALL NUMBERS BELOW ARE MODELLED from one-GPU measurements + the xGMI link rate; none is a multi-GPU measurement.")
    out.append(f"assumptions: {link.link_gbps:.0f} GB/s per direction per xGMI link, {link.lat_us:.0f} us per collective launch, "
               f"direct RS+AG over all W-1 links, one-GPU DDP step {t_ddp1_ms} ms (batch 30 per GPU)")
    # arena segments of the full model cut at the default phases (backward order): head+block4 ~ 8.6 MB + part, ...
    # measured split of the 27.8 MB by backward phase: block4+head 8.7 MB, block3 13.1 MB, block2+block1+stem 6.0 MB
    buckets = [8.7e6, 8.0e6, 5.1e6, 6.0e6]
    # phase ends: forward ~2.9 ms, backward of head+block4 ~0.9 ms, block3 ~1.6 ms, rest ~3.1 ms (kernel trace, r2)
    phase_end = [3.8, 4.6, 5.4, t_ddp1_ms - adam_ms]
    out.append("\n# DDP, weak scaling (30 images per GPU) - MODELLED")
    out.append(f"{'GPUs':>5} {'form':>10} {'step ms':>9} {'samples/s':>11} {'efficiency':>11} {'comm busy ms':>13} {'comm exposed ms':>16}")
    base = 30 / t_single_ms * 1e3
    for w in (1, 2, 4, 8):
        for form in (("rs_ag", "allreduce") if w > 1 else ("-",)):
            r = ddp_step_ms(w, t_ddp1_ms if w > 1 else t_single_ms, buckets, phase_end, adam_ms, link,
                            "rs_ag" if form == "rs_ag" else "ring")
            sps = 30 * w / r["step_ms"] * 1e3
            out.append(f"{w:>5} {form:>10} {r['step_ms']:>9.2f} {sps:>11.0f} {sps / (base * w):>10.1%} "
                       f"{r['comm_busy_ms']:>13.3f} {r['comm_exposed_ms']:>16.3f}")
    out.append("\n# pipeline / hybrid at the reference's shapes and at the throughput preset - MODELLED")
    out.append(f"{'mode':>28} {'GPUs':>5} {'step ms':>9} {'samples/s':>11} {'vs 1 GPU batch 30':>18}")
    # measured one-stage numbers (profiles/r3): 5 x 6 parity shape 12.24 ms per 30 images on ONE stage holding the whole
    # model -> a balanced 2-stage pipeline runs ~half of that per micro-batch slot
    t_mb_parity = 12.24 / 5 / 2
    pp2 = pipeline_step_ms(2, 5, t_mb_parity, 6 * 14 * 14 * 256 * 4, link)
    out.append(f"{'pp 2 stages, 5 x 6 (parity)':>28} {2:>5} {pp2:>9.2f} {30 / pp2 * 1e3:>11.0f} {30 / pp2 * 1e3 / base:>17.2f}x")
    t_mb_mi = 49.5 / 8 / 2                                   # preset mi355x: 8 x 30 on one stage = 49.5 ms (r3, lanes)
    pp2m = pipeline_step_ms(2, 8, t_mb_mi, 30 * 14 * 14 * 256 * 4, link)
    out.append(f"{'pp 2 stages, 8 x 30 (mi355x)':>28} {2:>5} {pp2m:>9.2f} {240 / pp2m * 1e3:>11.0f} {240 / pp2m * 1e3 / base:>17.2f}x")
    t_mb_h = 8.99 / 5 / 2                                    # hybrid parity: 5 x 2 on one stage = 8.99 ms per 10 images
    hy = pipeline_step_ms(2, 5, t_mb_h, 2 * 14 * 14 * 256 * 4, link) + rs_ag_ms(PARAM_BYTES / 2 / 4, 4, link)
    out.append(f"{'dp4 x pp2, 5 x 2 (parity)':>28} {8:>5} {hy:>9.2f} {40 / hy * 1e3:>11.0f} {40 / hy * 1e3 / base:>17.2f}x")
    out.append("(pipeline rows: a coarse UPPER bound - they assume a stage holding half of the model runs a micro-batch in half of "
               "the one-stage time and keeps its lane concurrency; the DDP rows rest on firmer ground)")
    out.append("\n(the reference on its own hardware: ddp 1.48x, pp 0.82x, ddp x pp 0.67x of its single-GPU rate - BASELINE.md)")
    return "\n".join(out)


def main(argv=None) -> None:
    ap = argparse.ArgumentParser(prog="ddl.scaling_model")
    ap.add_argument("--single-ms", type=float, default=8.56)
    ap.add_argument("--ddp1-ms", type=float, default=8.75)
    ap.add_argument("--link-gbps", type=float, default=64.0)
    ap.add_argument("--lat-us", type=float, default=25.0)
    a = ap.parse_args(argv)
    print(tables(a.single_ms, a.ddp1_ms, link=Link(a.link_gbps, a.lat_us)))


if __name__ == "__main__":
    main()

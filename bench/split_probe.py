"""Probe for models/split.py: time the stages of a given split (optionally after the per-unit measurement)."""
import argparse
import faulthandler
import sys

import torch

sys.path.insert(0, ".")
from ddl.models import split  # noqa: E402

faulthandler.enable()
ap = argparse.ArgumentParser()
ap.add_argument("--points", nargs="*", default=["features.denseblock3.denselayer1"])
ap.add_argument("--auto", type=int, default=0)
ap.add_argument("--units-first", action="store_true")
ap.add_argument("--mb", type=int, default=30)
a = ap.parse_args()
dev = torch.device("cuda", 0)
pts = a.points
if a.units_first or a.auto:
    c = split.measure_unit_times(dev, a.mb)
    print("unit ms", [round(v, 3) for v in c], flush=True)
if a.auto:
    pts, _ = split.auto_split_points(a.auto, dev, a.mb)
print("points", pts, flush=True)
print("stage ms", [round(t, 3) for t in split.measure_stage_times(pts, dev, a.mb)], flush=True)

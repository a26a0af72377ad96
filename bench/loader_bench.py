#!/usr/bin/env python3
"""Input-pipeline throughput on the host: the C++ PNG thread pool (csrc/png_loader.cpp, `NativePngLoader`) against
the torch DataLoader + PIL path (`HostLoader`) on an APTOS-shaped dataset (224x224 RGB PNGs written to --dir).

    python bench/loader_bench.py [--images 512] [--batch 30] [--workers 1 2 4 8]

CPU only - no GPU needed.  Prints images/s per loader and worker count (one epoch each, after a warm-up epoch).
"""
import argparse
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=512)
    ap.add_argument("--batch", type=int, default=30)
    ap.add_argument("--workers", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--dir", default=None)
    a = ap.parse_args()

    from PIL import Image
    from ddl import data

    root = a.dir or tempfile.mkdtemp(prefix="ddl_loader_bench_")
    os.makedirs(os.path.join(root, "train_images"), exist_ok=True)
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:224, 0:224]
    with open(os.path.join(root, "train.csv"), "w") as f:
        f.write("new_id_code,diagnosis\n")
        for i in range(a.images):       # smooth structure + noise: compresses like a photograph, not like a flat field
            arr = (np.stack([(yy * 3 + xx + i) % 256, (yy + xx * 2) % 256, (yy * xx) % 256], -1) // 2
                   + rng.integers(0, 64, (224, 224, 3))).astype(np.uint8)
            Image.fromarray(arr).save(os.path.join(root, "train_images", f"im{i:05d}.png"))
            f.write(f"im{i:05d},{i % 5}\n")
    ds = data.AptosDataset(os.path.join(root, "train.csv"), os.path.join(root, "train_images"), "new_id_code", "diagnosis")
    size = sum(os.path.getsize(os.path.join(root, "train_images", n + ".png")) for n in ds.names) / a.images / 1024
    print(f"{a.images} PNGs 224x224 RGB, {size:.0f} KiB each, batch {a.batch}, {os.cpu_count()} CPUs")
    for w in a.workers:
        for name, cls in (("native C++ thread pool", data.NativePngLoader), ("torch DataLoader + PIL", data.HostLoader)):
            sampler = data.ShardedSampler(len(ds), 1, 0, shuffle=True, seed=0, drop_last=False)
            loader = cls(ds, a.batch, sampler, w, drop_last=False)
            for epoch in (0, 1):
                loader.set_epoch(epoch)
                t0 = time.perf_counter()
                n = sum(int(x.shape[0]) for x, _ in loader)
                dt = time.perf_counter() - t0
            assert n == a.images
            print(f"  workers {w:2d}  {name:24s} {n / dt:8.0f} images/s  ({dt / n * 1e3:.2f} ms/image)", flush=True)
            del loader
    torch.set_num_threads(1)


if __name__ == "__main__":
    main()

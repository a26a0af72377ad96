"""Datasets, samplers and loaders.

Reference behaviour (single.py:38-65, duplicated in all four trainers): a CSV of
``(id, diagnosis)`` rows, ``root/{id}.png`` decoded to uint8 CHW and divided by
255 on the CPU inside the DataLoader worker; the fp32 batch (9 MB at B=15) is
then copied host->device synchronously from pageable memory (ddp.py:181-182).

MI355X-first changes:
  * batches travel as **uint8** (4x fewer PCIe bytes) from pinned memory and are
    normalised (``/255``) and re-laid-out NCHW->NHWC by one HIP kernel on the GPU
    (``ddl.ops.normalize_input``); ``Normalize`` is still provided for the CPU path.
  * ``SyntheticAptos`` stands in for the (unshipped) APTOS-2019 images: same
    shape/dtype/label range, deterministic per index, class-dependent so that a
    model can actually learn it in smoke tests.
  * ``DeviceSyntheticLoader`` generates batches directly in HBM for benchmarks so
    that the measured number is the training step, not the host input pipeline
    (the reference is input-bound: SURVEY 6.2).
  * samplers are explicit: ``ShardedSampler`` (DDP / hybrid: per-dp-rank shard,
    epoch-seeded shuffle, drop_last) and the same class with ``num_replicas=1``
    and a fixed seed for pipeline stages, which makes the "every stage iterates
    the same permutation" contract of pp.py:363-371 explicit instead of relying
    on torch.Generator's default seed.
"""
from __future__ import annotations

import csv
import math
import os
from pathlib import Path
from typing import Iterator, List, Tuple

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset, Sampler

from .utils.streams import new_stream


class Normalize:
    """uint8 -> float32 in [0,1] (reference: single.py:38-42; no mean/std, no augmentation)."""

    def __call__(self, img: torch.Tensor) -> torch.Tensor:
        return (img / 255.0).float()


def read_png(path) -> torch.Tensor:
    """Decode a PNG to a uint8 CHW tensor (torchvision.io.read_image is not installed)."""
    from PIL import Image

    with Image.open(path) as im:
        arr = np.asarray(im.convert("RGB"), dtype=np.uint8)
    return torch.from_numpy(arr.copy()).permute(2, 0, 1).contiguous()


def write_png(path, chw_uint8: torch.Tensor) -> None:
    from PIL import Image

    Image.fromarray(chw_uint8.permute(1, 2, 0).contiguous().numpy(), mode="RGB").save(path)


class AptosDataset(Dataset):
    """CSV-indexed PNG folder (reference: single.py:45-65)."""

    def __init__(self, csv_file, root_dir, filename_col: str, label_col: str, transform=None) -> None:
        self.root_dir = Path(root_dir)
        self.transform = transform
        with open(csv_file, newline="") as f:
            rows = list(csv.DictReader(f))
        self.names: List[str] = [r[filename_col] for r in rows]
        self.labels: List[int] = [int(r[label_col]) for r in rows]

    def __len__(self) -> int:
        return len(self.names)

    def __getitem__(self, idx: int) -> Tuple[torch.Tensor, int]:
        img = read_png(self.root_dir / f"{self.names[idx]}.png")
        if self.transform is not None:
            img = self.transform(img)
        return img, self.labels[idx]


class SyntheticAptos(Dataset):
    """Deterministic stand-in for APTOS-2019: uint8 [3,S,S] images, labels in [0, C).

    Each class has a fixed low-frequency template; a sample is template + noise,
    so the task is learnable (train accuracy rises within an epoch) but not trivial.
    """

    def __init__(self, n: int, image_size: int = 224, num_classes: int = 5, seed: int = 0,
                 transform=None, noise: float = 48.0) -> None:
        self.n = n
        self.size = image_size
        self.num_classes = num_classes
        self.seed = seed
        self.transform = transform
        self.noise = noise
        g = torch.Generator().manual_seed(seed * 7919 + 13)
        coarse = torch.rand(num_classes, 3, 7, 7, generator=g)
        self.templates = torch.nn.functional.interpolate(
            coarse, size=(image_size, image_size), mode="bilinear", align_corners=False) * 255.0
        self.labels = torch.randint(0, num_classes, (n,), generator=g).tolist()

    def __len__(self) -> int:
        return self.n

    def __getitem__(self, idx: int) -> Tuple[torch.Tensor, int]:
        label = self.labels[idx]
        g = torch.Generator().manual_seed(self.seed * 1000003 + idx)
        img = self.templates[label] + self.noise * torch.randn(3, self.size, self.size, generator=g)
        img = img.clamp_(0, 255).to(torch.uint8)
        if self.transform is not None:
            img = self.transform(img)
        return img, label


class ShardedSampler(Sampler[int]):
    """Epoch-seeded shuffle, sharded over ``num_replicas`` (drop_last), explicit seed.

    Same contract as DistributedSampler(shuffle=True, drop_last=True) used by the
    reference (ddp.py:343,353): every replica sees ``len(dataset) // num_replicas``
    samples, disjoint across replicas, reshuffled by ``set_epoch``.
    """

    def __init__(self, n: int, num_replicas: int = 1, rank: int = 0, shuffle: bool = True,
                 seed: int = 0, drop_last: bool = True) -> None:
        if not 0 <= rank < num_replicas:
            raise ValueError(f"rank {rank} out of range for {num_replicas} replicas")
        self.n, self.num_replicas, self.rank = n, num_replicas, rank
        self.shuffle, self.seed, self.drop_last = shuffle, seed, drop_last
        self.epoch = 0
        if drop_last:
            self.num_samples = n // num_replicas
        else:
            self.num_samples = math.ceil(n / num_replicas)

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def __len__(self) -> int:
        return self.num_samples

    def __iter__(self) -> Iterator[int]:
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            order = torch.randperm(self.n, generator=g).tolist()
        else:
            order = list(range(self.n))
        total = self.num_samples * self.num_replicas
        if len(order) < total:      # pad (drop_last=False)
            order += order[: total - len(order)]
        order = order[:total]
        return iter(order[self.rank: total: self.num_replicas])


def _collate_uint8(batch):
    xs = torch.stack([b[0] for b in batch])
    ys = torch.tensor([int(b[1]) for b in batch], dtype=torch.int64)
    return xs, ys


class HostLoader:
    """DataLoader wrapper: known length/batch size up front (no ``next(iter())`` probe per
    epoch as in ddp.py:177), pinned uint8 batches, ``set_epoch`` passthrough."""

    def __init__(self, dataset: Dataset, batch_size: int, sampler: ShardedSampler,
                 num_workers: int = 2, drop_last: bool = True, pin_memory: bool = False) -> None:
        self.dataset, self.batch_size, self.sampler = dataset, batch_size, sampler
        self.drop_last = drop_last
        self.loader = DataLoader(dataset, batch_size=batch_size, sampler=sampler,
                                 num_workers=num_workers, drop_last=drop_last,
                                 collate_fn=_collate_uint8, pin_memory=pin_memory,
                                 persistent_workers=num_workers > 0)

    def set_epoch(self, epoch: int) -> None:
        self.sampler.set_epoch(epoch)

    def __len__(self) -> int:
        return len(self.loader)

    def __iter__(self):
        return iter(self.loader)

    def batch_sizes(self) -> List[int]:
        """Batch size of every step of an epoch - lets pipeline stages that load no data
        still know the shapes they will receive."""
        n = len(self.sampler)
        full, rem = divmod(n, self.batch_size)
        sizes = [self.batch_size] * full
        if rem and not self.drop_last:
            sizes.append(rem)
        return sizes


def native_png_ok(dataset, mode: str = "auto") -> bool:
    """True when ``dataset`` can be read by the C++ loader: an ``AptosDataset`` handing out raw uint8 images
    (no transform) and the extension built.  ``mode == 'off'`` keeps the torch DataLoader."""
    if mode == "off" or not isinstance(dataset, AptosDataset) or dataset.transform is not None or len(dataset) == 0:
        return False
    from .ops import have_ext
    return have_ext()


class NativePngLoader:
    """``HostLoader`` contract on the C++ thread pool (csrc/png_loader.cpp) instead of DataLoader worker processes.

    Every batch is decoded by all threads at once straight into its final (pinned) uint8 ``[m,3,H,W]`` buffer, two
    batches ahead of the consumer; the consumer blocks with the GIL released.  Files the native decoder does not
    cover (16-bit, interlaced) come back as slot numbers and are decoded here with PIL, so any PNG the reference's
    ``read_image`` accepts still loads."""

    def __init__(self, dataset: AptosDataset, batch_size: int, sampler, num_workers: int = 2,
                 drop_last: bool = True, pin_memory: bool = False, depth: int = 2) -> None:
        from .ops import load_ext
        self.dataset, self.batch_size, self.sampler, self.drop_last = dataset, batch_size, sampler, drop_last
        self.paths = [str(dataset.root_dir / f"{n}.png") for n in dataset.names]
        self.labels = torch.tensor(dataset.labels, dtype=torch.int64)
        ext = load_ext()
        w, h = ext.png_info(self.paths[0])[:2]
        self.threads = max(int(num_workers), 1)
        self._loader = ext.PngLoader(self.paths, h, w, self.threads, depth, bool(pin_memory))

    def set_epoch(self, epoch: int) -> None:
        if hasattr(self.sampler, "set_epoch"):
            self.sampler.set_epoch(epoch)

    def batch_sizes(self) -> List[int]:
        full, rem = divmod(len(self.sampler), self.batch_size)
        return [self.batch_size] * full + ([rem] if rem and not self.drop_last else [])

    def __len__(self) -> int:
        return len(self.batch_sizes())

    def __iter__(self):
        self._loader.start_epoch([int(i) for i in self.sampler], self.batch_size, self.drop_last)
        while self._loader.batches_left():
            x, idx, unsupported = self._loader.next()
            for slot in unsupported:
                # whatever the native decoder does not cover or refuses (16-bit, Adam7, a file libpng would still
                # accept): PIL, like the reference's torchvision.io.read_image; only a file neither can read is fatal
                path = self.paths[int(idx[slot])]
                try:
                    x[slot].copy_(read_png(path))
                except Exception as e:      # noqa: BLE001
                    raise RuntimeError(f"{path}: neither the native PNG decoder nor PIL can read this file "
                                       f"({type(e).__name__}: {e})") from e
            yield x, self.labels[idx]


class DevicePrefetcher:
    """Host -> device staging one batch ahead on a dedicated copy stream.

    The reference copies each fp32 batch synchronously from pageable memory right before it is used
    (ddp.py:181-182).  Here the loader hands out pinned uint8 batches; while step ``i`` computes,
    batch ``i + 1`` is already crossing PCIe on the copy stream, and the compute stream only waits on
    an event.  Stages that need neither images nor labels (middle pipeline stages) copy nothing.
    Batches that already live on the device (``DeviceSyntheticLoader``) pass straight through.
    """

    def __init__(self, loader, device, needs_input: bool = True, needs_target: bool = True) -> None:
        self.loader, self.device = loader, torch.device(device)
        self.needs_input, self.needs_target = needs_input, needs_target
        self.stream = new_stream(self.device) if self.device.type == "cuda" else None

    def __len__(self) -> int:
        return len(self.loader)

    def _stage(self, batch):
        x, y = batch
        n = int((x if x is not None else y).shape[0])
        if self.stream is None or (y.is_cuda and (x is None or x.is_cuda)):    # CPU run / device-resident batch
            return (x.to(self.device) if self.needs_input else None,
                    y.to(self.device) if self.needs_target else None, n, None)
        with torch.cuda.stream(self.stream):
            xd = x.to(self.device, non_blocking=True) if self.needs_input else None
            yd = y.to(self.device, non_blocking=True) if self.needs_target else None
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return xd, yd, n, ev

    def __iter__(self):
        """Yields ``(x_dev | None, y_dev | None, batch_size)``."""
        it = iter(self.loader)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while nxt is not None:
            xd, yd, n, ev = nxt
            try:
                nxt = self._stage(next(it))        # enqueue the next copy before handing out this batch
            except StopIteration:
                nxt = None
            if ev is not None:
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                for t in (xd, yd):                 # allocated on the copy stream, consumed on the compute stream
                    if t is not None and t.is_cuda:
                        t.record_stream(cur)
            yield xd, yd, n


class DeviceSyntheticLoader:
    """Synthetic uint8 batches generated directly on the device (benchmark input)."""

    def __init__(self, steps: int, batch_size: int, device, image_size: int = 224,
                 num_classes: int = 5, seed: int = 0, pool: int = 4) -> None:
        self.steps, self.batch_size, self.device = steps, batch_size, torch.device(device)
        g = torch.Generator(device="cpu").manual_seed(seed)
        self.pool = []
        for _ in range(pool):
            x = torch.randint(0, 256, (batch_size, 3, image_size, image_size), dtype=torch.uint8, generator=g)
            y = torch.randint(0, num_classes, (batch_size,), dtype=torch.int64, generator=g)
            self.pool.append((x.to(self.device), y.to(self.device)))

    def set_epoch(self, epoch: int) -> None:
        pass

    def __len__(self) -> int:
        return self.steps

    def __iter__(self):
        for i in range(self.steps):
            yield self.pool[i % len(self.pool)]

    def batch_sizes(self) -> List[int]:
        return [self.batch_size] * self.steps


class ResidentLoader:
    """The whole dataset decoded ONCE and kept in HBM as uint8; every later epoch is an index gather.

    The reference decodes every PNG again in every epoch (DataLoader workers -> ``read_image`` -> /255 ->
    fp32 -> pageable H2D copy, single.py:45-65,160-167).  APTOS at 224x224 is 3,662 x 150 KB = 0.55 GB as
    uint8 - 0.2 % of one MI355X's 288 GB - and the transform is deterministic (no augmentation), so the
    natural home of the dataset is the GPU: after the fill, an epoch involves no host I/O, no decode, no
    PCIe traffic and no DataLoader workers; a batch is ``index_select`` by the epoch's sampler order
    (same ``ShardedSampler`` contract: reshuffled by ``set_epoch``, sharded over the dp replicas).

    Fill: with a process group every rank decodes a contiguous 1/W slice and the slices are all-gathered
    (RCCL over xGMI; uint8), so start-up decode time also drops by W.
    """

    def __init__(self, dataset: Dataset, batch_size: int, sampler: ShardedSampler, device,
                 num_workers: int = 2, drop_last: bool = True, needs_input: bool = True,
                 shard_fill: bool = True, chunk: int = 256, native: str = "auto") -> None:
        from . import comm
        self.batch_size, self.sampler, self.drop_last = batch_size, sampler, drop_last
        self.device = torch.device(device)
        n = len(dataset)
        world = comm.get_world_size() if (shard_fill and comm.is_dist()) else 1
        rank = comm.get_rank() if world > 1 else 0
        per = -(-n // world)
        lo, hi = min(rank * per, n), min((rank + 1) * per, n)
        c, h, w = dataset[0][0].shape
        self.images = torch.empty((n, c, h, w), dtype=torch.uint8, device=self.device) if needs_input else None
        self.labels = torch.empty(n, dtype=torch.int64, device=self.device)
        chunk = min(chunk, per)
        pin = self.device.type == "cuda"
        if hi <= lo:
            it = iter(())
        elif native_png_ok(dataset, native):           # C++ thread pool, decoded straight into pinned chunks
            it = iter(NativePngLoader(dataset, chunk, range(lo, hi), max(num_workers, 1), drop_last=False,
                                      pin_memory=pin))
        else:
            it = iter(DataLoader(torch.utils.data.Subset(dataset, range(lo, hi)), batch_size=chunk, shuffle=False,
                                 num_workers=num_workers if hi - lo > chunk else 0, collate_fn=_collate_uint8,
                                 pin_memory=pin))
        self.fill_decoded = hi - lo
        for k in range(-(-per // chunk)):              # same number of rounds on every rank
            xb, yb = next(it, (None, None))
            m = 0 if xb is None else int(xb.shape[0])
            if m and xb.dtype != torch.uint8:
                raise TypeError("ResidentLoader caches raw uint8 images (dataset transform must be None)")
            if world == 1:
                if m:
                    if needs_input:
                        self.images[k * chunk:k * chunk + m].copy_(xb, non_blocking=True)
                    self.labels[k * chunk:k * chunk + m].copy_(yb, non_blocking=True)
                continue
            xs = torch.zeros((chunk, c, h, w), dtype=torch.uint8, device=self.device)
            ys = torch.zeros(chunk, dtype=torch.int64, device=self.device)
            if m:
                xs[:m].copy_(xb, non_blocking=True)
                ys[:m].copy_(yb, non_blocking=True)
            gx, gy = comm.all_gather_flat(xs), comm.all_gather_flat(ys)
            for r in range(world):                      # rank r's k-th chunk covers [r*per + k*chunk, ...)
                a = r * per + k * chunk
                b = min(a + chunk, (r + 1) * per, n)
                if b > a:
                    if needs_input:
                        self.images[a:b].copy_(gx[r * chunk:r * chunk + (b - a)])
                    self.labels[a:b].copy_(gy[r * chunk:r * chunk + (b - a)])
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    @property
    def nbytes(self) -> int:
        return (self.images.numel() if self.images is not None else 0) + 8 * self.labels.numel()

    def set_epoch(self, epoch: int) -> None:
        self.sampler.set_epoch(epoch)

    def batch_sizes(self) -> List[int]:
        full, rem = divmod(len(self.sampler), self.batch_size)
        return [self.batch_size] * full + ([rem] if rem and not self.drop_last else [])

    def __len__(self) -> int:
        return len(self.batch_sizes())

    def __iter__(self):
        order = torch.tensor(list(self.sampler), dtype=torch.int64).to(self.device, non_blocking=True)
        a = 0
        for m in self.batch_sizes():
            idx = order[a:a + m]
            a += m
            x = self.images.index_select(0, idx) if self.images is not None else None
            yield x, self.labels.index_select(0, idx)


def use_synthetic(cfg) -> bool:
    if cfg.synthetic == "on":
        return True
    if cfg.synthetic == "off":
        return False
    return not (Path(cfg.data_dir) / "train.csv").exists()


def resident_fits(cfg, n_samples: int, device) -> bool:
    """``--cache-dataset``: ``device`` forces the HBM-resident loader, ``off`` the host DataLoader; ``auto`` picks
    the resident loader on a GPU when the uint8 dataset fits ``--cache-budget-gb`` (default 16 of 288 GB)."""
    if cfg.cache_dataset == "off" or device is None:
        return False
    if cfg.cache_dataset == "device":
        return True
    if torch.device(device).type != "cuda":
        return False
    return n_samples * 3 * cfg.image_size * cfg.image_size <= cfg.cache_budget_gb * 2 ** 30


def build_datasets(cfg, transform=None):
    """(train, eval) datasets: APTOS from ``cfg.data_dir`` or the synthetic stand-in."""
    if use_synthetic(cfg):
        seed = 0 if cfg.seed is None else cfg.seed
        train = SyntheticAptos(cfg.train_samples, cfg.image_size, cfg.num_classes, seed=seed, transform=transform)
        test = SyntheticAptos(cfg.eval_samples, cfg.image_size, cfg.num_classes, seed=seed, transform=transform)
        # same templates, disjoint noise / labels
        test.seed = seed + 1
        g = torch.Generator().manual_seed(seed * 31 + 5)
        test.labels = torch.randint(0, cfg.num_classes, (cfg.eval_samples,), generator=g).tolist()
        return train, test
    d = Path(cfg.data_dir)
    train = AptosDataset(d / "train.csv", d / "train_images", "new_id_code", "diagnosis", transform)
    test = AptosDataset(d / "test.csv", d / "test_images", "id_code", "diagnosis", transform)
    return train, test


def build_loaders(cfg, dp_rank: int = 0, dp_size: int = 1, device=None, shared_order: bool = False,
                  needs_input: bool = True):
    """Train/eval loaders for one process.

    ``shared_order`` (pipeline stages): every rank iterates the *same* permutation
    (fixed seed), the explicit form of the reference's pp.py generator contract.
    """
    seed = 0 if cfg.seed is None else cfg.seed
    pin = device is not None and torch.device(device).type == "cuda"
    if cfg.device_data and pin:
        steps = (cfg.train_samples // dp_size) // cfg.batch_size
        esteps = (cfg.eval_samples // dp_size) // cfg.batch_size
        if cfg.max_steps:
            steps = min(steps, cfg.max_steps)
        return (DeviceSyntheticLoader(steps, cfg.batch_size, device, cfg.image_size, cfg.num_classes, seed + dp_rank),
                DeviceSyntheticLoader(max(esteps, 1), cfg.batch_size, device, cfg.image_size, cfg.num_classes,
                                      seed + 1000 + dp_rank))
    train_ds, test_ds = build_datasets(cfg)
    tr_s = ShardedSampler(len(train_ds), dp_size, dp_rank, shuffle=True, seed=seed, drop_last=True)
    # reference eval sampler: shuffle=True, drop_last=True (ddp.py:353); --eval-all keeps the tail
    te_s = ShardedSampler(len(test_ds), dp_size, dp_rank, shuffle=True, seed=seed + 1,
                          drop_last=not cfg.eval_all)
    # reference: train DataLoader drop_last only in pp/hybrid (pp.py:366-372); single/ddp keep the tail
    drop_train = cfg.strategy in ("pp", "ddpnpp")
    drop_test = (cfg.strategy in ("pp", "ddpnpp")) and not cfg.eval_all
    if resident_fits(cfg, len(train_ds) + len(test_ds), device):
        return (ResidentLoader(train_ds, cfg.batch_size, tr_s, device, cfg.num_workers, drop_train, needs_input,
                               native=cfg.native_loader),
                ResidentLoader(test_ds, cfg.batch_size, te_s, device, cfg.num_workers, drop_test, needs_input,
                               native=cfg.native_loader))
    # host path: the C++ PNG thread pool for the real dataset, torch DataLoader workers for anything else
    Loader = NativePngLoader if native_png_ok(train_ds, cfg.native_loader) else HostLoader
    train = Loader(train_ds, cfg.batch_size, tr_s, cfg.num_workers, drop_last=drop_train, pin_memory=pin)
    test = Loader(test_ds, cfg.batch_size, te_s, cfg.num_workers, drop_last=drop_test, pin_memory=pin)
    return train, test


def make_png_dataset(root, n_train: int, n_test: int, image_size: int = 32, num_classes: int = 5, seed: int = 0):
    """Write a tiny APTOS-layout dataset (CSV + PNG folders) - used by tests to exercise
    the real ``AptosDataset`` path without the private data."""
    root = Path(root)
    for split, n, col in (("train", n_train, "new_id_code"), ("test", n_test, "id_code")):
        ds = SyntheticAptos(n, image_size, num_classes, seed=seed)
        if split == "test":            # same class templates, disjoint noise / labels (as build_datasets does)
            ds.seed = seed + 1
            g = torch.Generator().manual_seed(seed * 31 + 5)
            ds.labels = torch.randint(0, num_classes, (n,), generator=g).tolist()
        os.makedirs(root / f"{split}_images", exist_ok=True)
        with open(root / f"{split}.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow([col, "diagnosis"])
            for i in range(n):
                img, label = ds[i]
                name = f"{split}_{i:05d}"
                write_png(root / f"{split}_images" / f"{name}.png", img)
                w.writerow([name, label])
    return root

import torch

from ddl import data
from ddl.config import TrainConfig


def test_synthetic_shape_and_determinism():
    ds = data.SyntheticAptos(10, 224, 5, seed=3)
    x, y = ds[4]
    assert x.dtype == torch.uint8 and tuple(x.shape) == (3, 224, 224) and 0 <= y < 5
    x2, y2 = data.SyntheticAptos(10, 224, 5, seed=3)[4]
    assert torch.equal(x, x2) and y == y2


def test_normalize_matches_reference():
    x = torch.randint(0, 256, (3, 8, 8), dtype=torch.uint8)
    out = data.Normalize()(x)
    assert out.dtype == torch.float32 and torch.allclose(out, x.float() / 255)


def test_sharded_sampler_partitions_and_reshuffles():
    n, w = 103, 4
    parts = [list(data.ShardedSampler(n, w, r, seed=5)) for r in range(w)]
    assert all(len(p) == n // w for p in parts)
    flat = sum(parts, [])
    assert len(set(flat)) == len(flat)                      # disjoint shards (drop_last)
    s = data.ShardedSampler(n, w, 0, seed=5)
    a = list(s)
    s.set_epoch(1)
    assert list(s) != a
    # pipeline contract: every stage iterates the same permutation
    assert list(data.ShardedSampler(n, 1, 0, seed=9)) == list(data.ShardedSampler(n, 1, 0, seed=9))


def test_png_csv_dataset_roundtrip(tmp_path):
    root = data.make_png_dataset(tmp_path, 6, 4, image_size=16)
    ds = data.AptosDataset(root / "train.csv", root / "train_images", "new_id_code", "diagnosis",
                           transform=data.Normalize())
    assert len(ds) == 6
    x, y = ds[2]
    ref_x, ref_y = data.SyntheticAptos(6, 16, 5, seed=0)[2]
    assert y == ref_y and torch.allclose(x, ref_x.float() / 255)
    te = data.AptosDataset(root / "test.csv", root / "test_images", "id_code", "diagnosis")
    assert len(te) == 4 and te[0][0].dtype == torch.uint8


def test_loaders_lengths_match_reference_epoch():
    cfg = TrainConfig(strategy="ddp", batch_size=15, synthetic="on", num_workers=0)
    tr, te = data.build_loaders(cfg, dp_rank=1, dp_size=2)
    assert len(tr) == 80 and sum(tr.batch_sizes()) == 1200      # 2400 / 2 ranks / 15
    assert len(te) == 12
    x, y = next(iter(tr))
    assert x.dtype == torch.uint8 and tuple(x.shape) == (15, 3, 224, 224) and y.dtype == torch.int64


def test_device_prefetcher_order_and_roles():
    """Prefetching one batch ahead must not reorder or drop batches; roles skip what a stage does not need."""
    import torch
    from ddl.data import DevicePrefetcher
    batches = [(torch.full((3, 2), i, dtype=torch.uint8), torch.tensor([i, i, i])) for i in range(5)]
    got = list(DevicePrefetcher(batches, "cpu"))
    assert [int(x[0, 0]) for x, _, _ in got] == list(range(5)) and all(n == 3 for _, _, n in got)
    got = list(DevicePrefetcher(batches, "cpu", needs_input=False))
    assert all(x is None and int(y[0]) == i for i, (x, y, _) in enumerate(got))
    assert list(DevicePrefetcher([], "cpu")) == []


def _resident_cfg(**kw):
    return TrainConfig(strategy="ddp", batch_size=4, synthetic="on", num_workers=0, image_size=16,
                       train_samples=37, eval_samples=10, **kw)


def test_resident_loader_equals_host_loader():
    """--cache-dataset device: same batches in the same order as the DataLoader path, epoch after epoch."""
    host = data.build_loaders(_resident_cfg(cache_dataset="off"), dp_rank=1, dp_size=2, device="cpu")
    res = data.build_loaders(_resident_cfg(cache_dataset="device"), dp_rank=1, dp_size=2, device="cpu")
    assert isinstance(res[0], data.ResidentLoader) and isinstance(host[0], data.HostLoader)
    assert not isinstance(data.build_loaders(_resident_cfg(), device="cpu")[0], data.ResidentLoader)   # auto: GPU only
    for h, r in zip(host, res):
        assert len(h) == len(r) and h.batch_sizes() == r.batch_sizes()
        for epoch in (0, 1):
            h.set_epoch(epoch), r.set_epoch(epoch)
            for (hx, hy), (rx, ry) in zip(h, r):
                assert torch.equal(hx, rx) and torch.equal(hy, ry)
    # a stage that needs no images caches labels only
    lab = data.build_loaders(_resident_cfg(cache_dataset="device"), device="cpu", needs_input=False)[0]
    assert lab.images is None and next(iter(lab))[0] is None and lab.nbytes == 8 * 37


def _resident_fill_worker(rank, world):
    from ddl import data
    cfg = _resident_cfg(cache_dataset="device")
    tr, _ = data.build_loaders(cfg, dp_rank=rank, dp_size=world, device="cpu")
    assert tr.fill_decoded == (13 if rank < 2 else 11)            # ceil(37 / 3) each, last rank the rest
    ds, _ = data.build_datasets(cfg)
    small = data.ResidentLoader(ds, 4, data.ShardedSampler(len(ds), world, rank), "cpu", num_workers=0, chunk=5)
    for i in range(len(ds)):                                      # several all-gather rounds, ragged last chunk
        x, y = ds[i]
        assert torch.equal(tr.images[i], x) and int(tr.labels[i]) == y
        assert torch.equal(small.images[i], x) and int(small.labels[i]) == y


def test_resident_loader_sharded_fill_gloo():
    """With a process group every rank decodes 1/W of the dataset and the slices are all-gathered."""
    from dist_utils import run_distributed
    run_distributed(_resident_fill_worker, 3)

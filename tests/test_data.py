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


# ---- native PNG loader (csrc/png_loader.cpp) -------------------------------------------------------------------
def _png_cfg(root, **kw):
    return TrainConfig(strategy="ddp", batch_size=4, synthetic="off", data_dir=str(root), num_workers=3,
                       image_size=16, **kw)


def test_native_png_decoder_matches_pil(tmp_path):
    """Every colour type / row filter of 8-bit non-interlaced PNGs decodes bit-exactly as PIL's convert('RGB')
    (the oracle; the reference's torchvision.io.read_image is not installed - parity with it is unpinned)."""
    import numpy as np
    import pytest
    from PIL import Image
    from ddl.ops import load_ext
    ext = load_ext()
    rng = np.random.default_rng(0)
    H, W = 37, 53
    yy, xx = np.mgrid[0:H, 0:W]
    smooth = np.stack([(yy * 3 + xx) % 256, (yy + xx * 2) % 256, (yy * xx) % 256], -1).astype(np.uint8)
    noise = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
    blocky = np.repeat(np.repeat(noise[::4, ::4], 4, 0), 4, 1)[:H, :W]
    images = {
        "rgb_noise": Image.fromarray(noise[..., :3].copy()), "rgb_smooth": Image.fromarray(smooth),
        "rgb_blocky": Image.fromarray(blocky[..., :3].copy()), "rgba": Image.fromarray(noise.copy()),
        "gray": Image.fromarray(smooth[..., 0].copy()), "gray_alpha": Image.fromarray(noise[..., :2].copy(), "LA"),
        "palette": Image.fromarray(smooth).quantize(64),
    }
    seen_filters = set()
    for name, im in images.items():
        p = tmp_path / f"{name}.png"
        im.save(p)
        got = ext.decode_png(str(p))
        assert got.dtype == torch.uint8 and tuple(got.shape) == (3, H, W), name
        assert torch.equal(got, data.read_png(p)), name
        assert ext.png_info(str(p))[:3] == (W, H, 8)
    # each of the five row filters, forced, on one image (PIL's encoder picks them adaptively above)
    import struct
    import zlib

    def write_png(path, rows, ftype):
        def paeth(a, b, c):
            pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
            return a if pa <= pb and pa <= pc else (b if pb <= pc else c)
        raw = bytearray()
        prev = [0] * (W * 3)
        for row in rows:
            cur = list(row)
            raw.append(ftype)
            for i, v in enumerate(cur):
                a = cur[i - 3] if i >= 3 else 0
                b = prev[i]
                c = prev[i - 3] if i >= 3 else 0
                pred = (0, a, b, (a + b) // 2, paeth(a, b, c))[ftype]
                raw.append((v - pred) & 255)
            prev = cur
            seen_filters.add(ftype)

        def chunk(t, body):
            return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body))
        comp = zlib.compress(bytes(raw))
        with open(path, "wb") as f:      # IDAT split in two: the decoder must concatenate them
            f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 2, 0, 0, 0))
                    + chunk(b"IDAT", comp[:len(comp) // 2]) + chunk(b"IDAT", comp[len(comp) // 2:]) + chunk(b"IEND", b""))
    rows = [smooth[y].reshape(-1).tolist() for y in range(H)]
    for ftype in range(5):
        p = tmp_path / f"f{ftype}.png"
        write_png(p, rows, ftype)
        assert torch.equal(ext.decode_png(str(p)), torch.from_numpy(smooth).permute(2, 0, 1)), ftype
        assert torch.equal(ext.decode_png(str(p)), data.read_png(p))
    assert seen_filters == {0, 1, 2, 3, 4}
    # failures are loud: a flipped byte (CRC), a truncated file, a non-PNG
    blob = bytearray((tmp_path / "rgb_noise.png").read_bytes())
    blob[len(blob) // 2] ^= 0x55
    (tmp_path / "bad_crc.png").write_bytes(bytes(blob))
    (tmp_path / "short.png").write_bytes(bytes(blob[:60]))
    (tmp_path / "text.png").write_bytes(b"not a png at all, just text that is long enough to pass the size check")
    for bad in ("bad_crc", "short", "text", "missing"):
        with pytest.raises(RuntimeError):
            ext.decode_png(str(tmp_path / f"{bad}.png"))


def test_native_png_loader_equals_dataloader(tmp_path):
    """The C++ thread-pool loader yields the same batches, in the same order, as the torch DataLoader path - over
    epochs, with and without the ragged tail - and `build_loaders` picks it for the real (PNG) dataset."""
    root = data.make_png_dataset(tmp_path, 23, 9, image_size=16)
    nat = data.build_loaders(_png_cfg(root, cache_dataset="off"), dp_rank=1, dp_size=2, device="cpu")
    ref = data.build_loaders(_png_cfg(root, cache_dataset="off", native_loader="off"), dp_rank=1, dp_size=2, device="cpu")
    assert isinstance(nat[0], data.NativePngLoader) and isinstance(ref[0], data.HostLoader)
    for n, r in zip(nat, ref):
        assert len(n) == len(r) and n.batch_sizes() == r.batch_sizes()
        for epoch in (0, 1, 2):
            n.set_epoch(epoch), r.set_epoch(epoch)
            got, want = list(n), list(r)
            assert len(got) == len(want) == len(r)
            for (nx, ny), (rx, ry) in zip(got, want):
                assert nx.dtype == torch.uint8 and torch.equal(nx, rx) and torch.equal(ny, ry)
    # an abandoned epoch (max_steps) does not leak into the next one
    it = iter(nat[0])
    next(it)
    del it
    nat[0].set_epoch(0), ref[0].set_epoch(0)
    assert all(torch.equal(a[0], b[0]) for a, b in zip(nat[0], ref[0]))
    # the resident (HBM cache) fill goes through the same thread pool
    res = data.build_loaders(_png_cfg(root, cache_dataset="device"), device="cpu")[0]
    ds = data.build_datasets(_png_cfg(root))[0]
    assert all(torch.equal(res.images[i], ds[i][0]) and int(res.labels[i]) == ds[i][1] for i in range(len(ds)))


def test_native_png_loader_falls_back_per_file(tmp_path):
    """Files the native decoder does not cover (16-bit, Adam7) are decoded by PIL in place; a corrupt file raises."""
    import numpy as np
    import pytest
    from PIL import Image
    root = data.make_png_dataset(tmp_path, 6, 2, image_size=16)
    ds = data.AptosDataset(root / "train.csv", root / "train_images", "new_id_code", "diagnosis")
    victim = ds.root_dir / f"{ds.names[3]}.png"
    arr = (np.arange(16 * 16, dtype=np.uint16).reshape(16, 16) * 257)
    Image.fromarray(arr, "I;16").save(victim)                      # 16-bit grey
    loader = data.NativePngLoader(ds, 4, range(len(ds)), num_workers=2, drop_last=False)
    batches = list(loader)
    assert [int(b[0].shape[0]) for b in batches] == [4, 2]
    assert torch.equal(batches[0][0][3], data.read_png(victim))
    assert torch.equal(batches[0][0][2], data.read_png(ds.root_dir / f"{ds.names[2]}.png"))
    broken = ds.root_dir / f"{ds.names[1]}.png"
    broken.write_bytes(broken.read_bytes()[:60])
    with pytest.raises(RuntimeError, match="neither the native PNG decoder nor PIL"):
        list(data.NativePngLoader(ds, 4, range(len(ds)), num_workers=2, drop_last=False))


def test_native_png_decoder_is_no_stricter_than_libpng(tmp_path):
    """a damaged ANCILLARY chunk (tEXt with a wrong CRC) must not abort a batch - libpng / PIL, which the
    reference's read_image sits on, ignore it; the native decoder checks the CRC of critical chunks only, and any
    decode failure of a slot goes to the per-slot PIL fallback instead of failing the epoch."""
    import struct
    import zlib
    root = data.make_png_dataset(tmp_path, 4, 2, image_size=16)
    ds = data.AptosDataset(root / "train.csv", root / "train_images", "new_id_code", "diagnosis")
    victim = ds.root_dir / f"{ds.names[1]}.png"
    want = data.read_png(victim)
    blob = victim.read_bytes()
    body = b"Comment\x00damaged ancillary chunk"
    bad = struct.pack(">I", len(body)) + b"tEXt" + body + struct.pack(">I", (zlib.crc32(b"tEXt" + body) ^ 0xFFFF) & 0xFFFFFFFF)
    victim.write_bytes(blob[:33] + bad + blob[33:])                 # right behind IHDR (8 + 25 bytes)
    from ddl import ops
    if ops.have_ext():
        assert torch.equal(ops.require_ext().decode_png(str(victim)), want)       # decoded natively, CRC of tEXt ignored
    batches = list(data.NativePngLoader(ds, 4, range(len(ds)), num_workers=2, drop_last=False))
    assert torch.equal(batches[0][0][1], want)

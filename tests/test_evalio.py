"""Host-side data formats next to the path, pinned by the reference-derived golden g4."""
import gzip
import struct

import numpy as np
import torch

from localdiffusion_hallucination_amd import evalio


def _write_idx(path, arr):
    hdr = struct.pack(">HBB", 0, 0x08, arr.ndim) + struct.pack(">" + "I" * arr.ndim, *arr.shape)
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "wb") as f:
        f.write(hdr + arr.astype(np.uint8).tobytes())


def test_idx_reader_plain_and_gz(tmp_path):
    rng = np.random.default_rng(0)
    imgs = rng.integers(0, 256, size=(5, 28, 28), dtype=np.uint8)
    labs = np.array([3, 1, 3, 8, 3], dtype=np.uint8)
    _write_idx(tmp_path / "im.idx3-ubyte.gz", imgs)
    _write_idx(tmp_path / "lb.idx1-ubyte", labs)
    a, b = evalio.read_idx(tmp_path / "im.idx3-ubyte.gz"), evalio.read_idx(tmp_path / "lb.idx1-ubyte")
    assert np.array_equal(a, imgs) and np.array_equal(b, labs)
    sel, sl = evalio.select_digits(a, b, 3, max_n=2)
    assert np.array_equal(sel, imgs[[0, 2]]) and list(sl) == [3, 3]
    (tmp_path / "bad").write_bytes(b"\\x01\\x02\\x03\\x04junk")
    try:
        evalio.read_idx(tmp_path / "bad")
        assert False
    except ValueError:
        pass


def test_lr_hr_transform_and_band_mask_match_reference_golden(golden):
    g = golden("g4_cfg1_mnist")                      # digits + the reference pipeline's cond / hr / mask
    hr, lr = evalio.mnist_pairs(g["digits"])
    assert np.array_equal(hr.numpy(), g["hr"])
    assert np.allclose(lr.numpy(), g["cond"], atol=1e-6)
    assert np.array_equal(evalio.band_mask(4, 28, 28, 7).numpy(), g["mask"])


def test_anomaly_map_mask():
    a = torch.linspace(30.0, 46.0, 28 * 28).reshape(1, 1, 28, 28)
    m, b = evalio.anomaly_map_to_mask(a, 41.7)
    assert float(m.min()) == 0.0 and abs(float(m.max()) - 1.0) < 1e-6
    assert torch.equal(b, (a > 41.7).float()) and bool((m[b == 1] == 1).all())

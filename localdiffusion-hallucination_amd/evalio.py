"""Data formats either side of the sampling path: what the reference's
``test.py`` does before and after ``sample()``, restated host-side so an end-to-end run is comparable
with the authors' script.

* MNIST idx(.gz) reader (the reference uses ``idx2numpy``; files under ``MNIST/raw``);
* ``MNIST.__getitem__`` LR/HR pair (/root/reference/data.py:808-829): HR = 2x/255; LR: the reference indexes the
  [1,1,28,28] image with ``img[:, ::2, ::2]`` -- dims 0, 1, 2 -- so only the ROWS are decimated (-> [1,1,14,28]),
  then bilinear resize back to 28x28 (align_corners=False: rows x2, columns unchanged), then 2*/255.  Pinned by
  tests/golden/g4_cfg1_mnist.npz, whose ``cond`` comes from the reference's own dataset class;
* the hand-drawn OOD mask of the released script (columns 0..6 = 1, /root/reference/test.py:379-381) and
  the soft mask derived from a thresholded anomaly map (/root/reference/test.py:259-262);
* the evaluation loop: one ``sample()`` per image (batch size 1, test.py:108,190,393), MSE of the last
  channel vs HR (:416), mean wall time (:445), ``hr_all / lr_all / pred_all / ad_masks .npy`` (:429-442).
"""
import gzip
import os
import struct
import time

import numpy as np
import torch
import torch.nn.functional as F

_IDX_DTYPES = {0x08: np.uint8, 0x09: np.int8, 0x0B: ">i2", 0x0C: ">i4", 0x0D: ">f4", 0x0E: ">f8"}


def read_idx(path):
    """Parse an idx file (optionally gzip-compressed): magic = 0, 0, dtype code, ndim; big-endian dims."""
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rb") as f:
        raw = f.read()
    zero, dcode, ndim = struct.unpack(">HBB", raw[:4])
    if zero != 0 or dcode not in _IDX_DTYPES:
        raise ValueError(f"{path}: not an idx file (magic {raw[:4].hex()})")
    dims = struct.unpack(">" + "I" * ndim, raw[4:4 + 4 * ndim])
    data = np.frombuffer(raw, dtype=_IDX_DTYPES[dcode], offset=4 + 4 * ndim)
    if data.size != int(np.prod(dims)):
        raise ValueError(f"{path}: payload has {data.size} items, header says {dims}")
    return data.reshape(dims)


def select_digits(images, labels, digits, max_n=None):
    """Images whose label is in ``digits`` in file order (MNIST.__init__, data.py:770-776)."""
    digits = [digits] if np.isscalar(digits) else list(digits)
    idx = np.nonzero(np.isin(labels, digits))[0]
    if max_n is not None:
        idx = idx[:max_n]
    return images[idx].copy(), labels[idx].copy()


def mnist_pairs(images_u8):
    """uint8 [N,28,28] -> (hr, lr) float32 [N,1,28,28] in [0, 2]  (data.py:808-829)."""
    x = torch.from_numpy(np.ascontiguousarray(images_u8).astype(np.float32))[:, None]
    # data.py:817 slices dims (0, 1, 2) of the [1,1,H,W] tensor: rows only
    lr = F.interpolate(x[:, :, ::2, :], size=(x.shape[-2], x.shape[-1]), mode="bilinear", align_corners=False)
    return 2.0 * (x / 255.0), 2.0 * (lr / 255.0)


def band_mask(n, h, w, ncols=7):
    """The released script's manual OOD mask: ones in the first ``ncols`` columns (test.py:379-381)."""
    m = torch.zeros(n, 1, h, w)
    m[:, :, :, :ncols] = 1.0
    return m


def anomaly_map_to_mask(anomaly_map, threshold):
    """Soft OOD mask from an anomaly map (test.py:259-262): -> (mask_pred in [0,1], binary_mask)."""
    a = anomaly_map.detach().cpu().float()
    binary = (a > threshold).float()
    m = torch.clip(a, min=threshold - float(a.std()), max=threshold)
    m = (m - m.min()) / (threshold - m.min())
    return m ** 2, binary


def evaluate(diffusion, hr, lr, masks, min_max_val, out_dir=None, device="cuda", batch_size=1):
    """test.py's loop: sample every LR image, compare with HR.  Returns a dict of metrics."""
    preds, losses, times = [], [], []
    n = hr.shape[0]
    for i in range(0, n, batch_size):
        sl = slice(i, min(n, i + batch_size))
        cond = lr[sl].to(device)
        mask = None if masks is None else masks[sl].to(device)
        if str(device).startswith("cuda"):
            torch.cuda.synchronize()
        t0 = time.time()
        out = diffusion.sample(cond, hr[sl].to(device), batch_size=cond.shape[0], mask=mask, min_max_val=min_max_val)
        if str(device).startswith("cuda"):
            torch.cuda.synchronize()
        times.append(time.time() - t0)
        if isinstance(out, list):
            out = torch.stack(out)
        out = out.detach().cpu()
        losses.append(float(torch.nn.functional.mse_loss(out[..., [-1], :, :] if out.dim() == 4 else out[-1][:, [-1]],
                                                         hr[sl][:, [-1]])))
        preds.append(out.numpy())
    res = {"test_loss": float(np.mean(losses)), "avg_sampling_time": float(np.mean(times)),
           "n": n, "pred": np.concatenate(preds) if preds[0].ndim == 4 else np.stack(preds)}
    if out_dir is not None:
        os.makedirs(out_dir, exist_ok=True)
        np.save(os.path.join(out_dir, "hr_all.npy"), hr.numpy())
        np.save(os.path.join(out_dir, "lr_all.npy"), lr.numpy())
        np.save(os.path.join(out_dir, "pred_all.npy"), res["pred"])
        if masks is not None:
            np.save(os.path.join(out_dir, "ad_masks.npy"), masks.numpy())
    return res

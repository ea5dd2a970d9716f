#!/usr/bin/env python3
"""test.py-equivalent driver for the MNIST configuration of the reference (config.yaml defaults):
reads MNIST idx files, builds LR/HR pairs and the band mask, samples with branch + fusion on the GPU,
prints "Test loss" / "Average sampling time" and writes hr_all / lr_all / pred_all / ad_masks .npy.

  python tools/run_eval.py --images MNIST/raw/t10k-images-idx3-ubyte.gz --labels MNIST/raw/t10k-labels-idx1-ubyte.gz \\
         [--checkpoint results/.../model-best2900.pt] [--digit 3] [--n 4] [--timesteps 100] [--out eval_out]
Without --checkpoint the procedural weights of the tests are used (no trained weights ship with the reference).
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import localdiffusion_hallucination_amd as ldh                                   # noqa: E402
from localdiffusion_hallucination_amd import checkpoint, evalio, weights        # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", required=True)
    ap.add_argument("--labels", required=True)
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--digit", type=int, default=3)          # config.yaml:14 anomaly_name: 3
    ap.add_argument("--n", type=int, default=4)
    ap.add_argument("--timesteps", type=int, default=100)
    ap.add_argument("--ddim", type=int, default=0, help="sampling_timesteps (0 = ancestral DDPM)")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16", "fp16"])
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    imgs, labs = evalio.select_digits(evalio.read_idx(a.images), evalio.read_idx(a.labels), a.digit, a.n)
    hr, lr = evalio.mnist_pairs(imgs)
    masks = evalio.band_mask(hr.shape[0], 28, 28, 7)
    config = dict(branch_out=True, start_intermediate=True, start_timestep=2, data="mnist", mask_x=True, mask_cond=False,
                  ood_AD=True, ood_confidence=False, classifier=False, use_gt=False, use_gt_timestep=100)
    net = ldh.Unet(dim=32, init_dim=32, dim_mults=(1, 2, 4), full_attn=(False, False, True), mode="mnist",
                   compute_dtype=a.dtype)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in weights.procedural_state_dict(net.cfg, 0).items()})
    gd = ldh.GaussianDiffusion(config, net, image_size=28, timesteps=a.timesteps, beta_schedule="sigmoid",
                               objective="pred_x0", sampling_timesteps=a.ddim or None)
    if a.checkpoint:
        print("checkpoint:", checkpoint.load_reference_checkpoint(a.checkpoint, gd))
    gd = gd.to("cuda")
    res = evalio.evaluate(gd, hr, lr, masks, (0.0, 2.0), out_dir=a.out)
    print("Test loss: {:.4f}".format(res["test_loss"]))
    print("Average sampling time: {:.4f}".format(res["avg_sampling_time"]))


if __name__ == "__main__":
    main()

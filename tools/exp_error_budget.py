#!/usr/bin/env python3
"""where does the 16-bit chain's distance to the reference come from?

CPU experiment (runs anywhere; ~2.5 minutes per variant on 8 cores).  cfg2 (one 1x128x128 patch, T = 1000, the
golden's inputs and noise stream) is run with the ORACLE's arithmetic (fp32, ATen) and the HIP path's ROUNDINGS
emulated one storage class at a time -- the same places where the 16-bit kernels round a value to bf16 / fp16:

  W      the packed convolution / qkv / to_out weights (MFMA A operands); init_conv (hi+lo split), final_conv, the
         time MLP and GroupNorm / RMSNorm parameters stay fp32 in the HIP path and here
  W0..W3 the same for the layers of ONE resolution level only (0 = full resolution ... 3 = 1/8 and the middle blocks);
         W0c3 / W0c1 / W0at: only that level's 3x3 convolutions / 1x1 convolutions / attention projections
  RAW    convolution outputs stored before their GroupNorm (raw1, raw2 of every ResnetBlock, the encoder's convs)
  ACT    the normalised + FiLM + SiLU tensor that enters the second convolution (rounded when the prologue packs it
         for the MFMA, or when gn_apply stores it)
  TRUNK  what the blocks hand on: block outputs, attention + residual outputs, resampling convolution outputs,
         init_conv's output, the conditioning features
  ATTN   attention internals: the RMS-normalised input of to_qkv, q / k / v, the softmax weights P and the context
         fold M_b as MFMA operands, the attention output before to_out

and reports max-abs / mean-abs against the golden at the stored states (t = 999, 500, 100).  The yardstick is fixture
G11 (the reference with ONLY its output rounded once per step).  "all" should land near what the HIP path measures
(tests/test_hip_lowp_chain.py); the single-class rows say which roundings carry the gap.

usage: python tools/exp_error_budget.py [--dtype bf16|fp16] [--until 100] [--classes all,W,RAW,...] [--threads 8]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from localdiffusion_hallucination_amd import rng, weights   # noqa: E402
from oracle import diffusion_ref, unet_ref                  # noqa: E402

CLASSES = ("W", "RAW", "ACT", "TRUNK", "ATTN")


class Emu:
    def __init__(self, sd, cfg, on, tdt):
        self.cfg, self.on, self.tdt = cfg, set(on), tdt
        self.sd = dict(sd)
        ns = len(cfg.dim_mults)

        def level(k):                              # resolution level of the layer a weight belongs to (0 = full resolution)
            if k.startswith("downs."):
                return int(k.split(".")[1])
            if k.startswith("ups."):
                return ns - 1 - int(k.split(".")[1])
            if k.startswith("final_res_block"):
                return 0
            if k.startswith("cond_model."):
                return {"residual_conv1": 0, "residual_conv2": 1, "residual_conv3": 2, "mid_conv": 3}[k.split(".")[1]]
            return ns - 1                          # mid blocks, mid attention, conv_fusion
        for k, v in sd.items():
            if v.dim() == 4 and not k.startswith(("init_conv", "final_conv")):
                # 3x3 convolutions / attention projections (to_qkv; to_out stays fp32 in the HIP path) / other 1x1 convolutions
                kind = "c3" if v.shape[-1] == 3 else ("at" if (".to_qkv" in k or ".to_out" in k) else "c1")
                if ".to_out" in k and "W" not in self.on:
                    continue
                if "W" in self.on or f"W{level(k)}" in self.on or f"W{level(k)}{kind}" in self.on:
                    self.sd[k] = v.to(tdt).float()

    def q(self, cls, t):
        return t.to(self.tdt).float() if cls in self.on else t

    def conv_gn_act(self, p, x, groups, film=None, act=F.silu):
        sd = self.sd
        y = self.q("RAW", F.conv2d(x, sd[p + ".proj.weight"], sd[p + ".proj.bias"], padding=1))
        y = F.group_norm(y, groups, sd[p + ".norm.weight"], sd[p + ".norm.bias"], eps=1e-5)
        if film is not None:
            y = y * (film[0] + 1) + film[1]
        return act(y)

    def resnet_block(self, p, x, temb, groups=8):
        sd = self.sd
        film = None
        if temb is not None:
            e = F.linear(F.silu(temb), sd[p + ".mlp.1.weight"], sd[p + ".mlp.1.bias"])[:, :, None, None]
            film = e.chunk(2, dim=1)
        h = self.q("ACT", self.conv_gn_act(p + ".block1", x, groups, film))
        h = self.conv_gn_act(p + ".block2", h, groups)          # the tail is computed in fp32 from raw2 and x
        if (p + ".res_conv.weight") in sd:
            x = F.conv2d(x, sd[p + ".res_conv.weight"], sd[p + ".res_conv.bias"])
        return self.q("TRUNK", h + x)

    def linear_attention(self, p, x, heads=4, dh=32):
        sd = self.sd
        b, c, hh, ww = x.shape
        n = hh * ww
        y = self.q("ATTN", unet_ref.rms_norm(x, sd[p + ".norm.g"]))
        qkv = F.conv2d(y, sd[p + ".to_qkv.weight"])
        q, k, v = [t.reshape(b, heads, dh, n) for t in qkv.chunk(3, dim=1)]
        q = self.q("ATTN", q.softmax(dim=-2) * (dh ** -0.5))
        k = k - k.amax(dim=-1, keepdim=True)
        pk = self.q("ATTN", k.exp())                            # P = exp(k - m) as an MFMA operand; Z in fp32
        z = pk.sum(dim=-1, keepdim=True)
        ctx = torch.einsum("bhdn,bhen->bhde", pk, self.q("ATTN", v)) / z
        wout = sd[p + ".to_out.0.weight"].reshape(c, heads, dh)                 # the fold M_b = W_out . ctx^T is a 16-bit MFMA operand
        mb = self.q("ATTN", torch.einsum("che,bhde->bchd", wout, ctx))
        o = torch.einsum("bchd,bhdn->bcn", mb, q).reshape(b, c, hh, ww) + sd[p + ".to_out.0.bias"][None, :, None, None]
        return unet_ref.rms_norm(o, sd[p + ".to_out.1.g"])

    def full_attention(self, p, x, heads=4, dh=32):
        sd = self.sd
        b, c, hh, ww = x.shape
        n = hh * ww
        y = self.q("ATTN", unet_ref.rms_norm(x, sd[p + ".norm.g"]))
        qkv = self.q("ATTN", F.conv2d(y, sd[p + ".to_qkv.weight"]))
        q, k, v = [t.reshape(b, heads, dh, n).transpose(-1, -2) for t in qkv.chunk(3, dim=1)]
        sim = torch.einsum("bhid,bhjd->bhij", q, k) * (dh ** -0.5)
        sim = sim - sim.amax(dim=-1, keepdim=True)
        pe = self.q("ATTN", sim.exp())
        out = torch.einsum("bhij,bhjd->bhid", pe, v) / pe.sum(dim=-1, keepdim=True)
        out = self.q("ATTN", out.transpose(-1, -2).reshape(b, heads * dh, hh, ww))
        return F.conv2d(out, sd[p + ".to_out.weight"], sd[p + ".to_out.bias"])

    def basic_block(self, p, x, groups=16):
        sd = self.sd
        y = self.q("RAW", F.conv2d(x, sd[p + ".convblock.0.weight"], sd[p + ".convblock.0.bias"], padding=1))
        y = self.q("ACT", F.relu(F.group_norm(y, groups, sd[p + ".convblock.1.weight"], sd[p + ".convblock.1.bias"])))
        y = self.q("RAW", F.conv2d(y, sd[p + ".convblock.3.weight"], sd[p + ".convblock.3.bias"], padding=1))
        y = F.group_norm(y, groups, sd[p + ".convblock.4.weight"], sd[p + ".convblock.4.bias"])
        idn = self.q("RAW", F.conv2d(x, sd[p + ".identity.0.weight"], sd[p + ".identity.0.bias"], padding=1))
        idn = F.group_norm(idn, groups, sd[p + ".identity.1.weight"], sd[p + ".identity.1.bias"])
        return self.q("TRUNK", F.relu(y + idn))

    def cond_encoder(self, cond):
        x = F.max_pool2d(self.basic_block("cond_model.residual_conv1.0", cond), 2)
        x = F.max_pool2d(self.basic_block("cond_model.residual_conv2.0", x), 2)
        x = F.max_pool2d(self.basic_block("cond_model.residual_conv3.0", x), 2)
        return self.basic_block("cond_model.mid_conv.0", x)

    def forward(self, x, cond_feat, time):
        sd, cfg = self.sd, self.cfg
        g, ns, fa = cfg.resnet_block_groups, len(cfg.dim_mults), tuple(cfg.full_attn)
        attn = lambda p, t, full: (self.full_attention if full else self.linear_attention)(p, t)
        x = self.q("TRUNK", F.conv2d(x, sd["init_conv.weight"], sd["init_conv.bias"], padding=3))
        r = x
        temb = unet_ref.time_embedding(sd, time, cfg.dim)
        skips = []
        for i in range(ns):
            p = f"downs.{i}"
            x = self.resnet_block(p + ".0", x, temb, g)
            skips.append(x)
            x = self.resnet_block(p + ".1", x, temb, g)
            x = self.q("TRUNK", attn(p + ".2", x, fa[i]) + x)
            skips.append(x)
            if i < ns - 1:
                x = self.q("TRUNK", unet_ref.pixel_unshuffle_conv(sd, p + ".3", x))
            else:
                x = self.q("TRUNK", F.conv2d(x, sd[p + ".3.weight"], sd[p + ".3.bias"], padding=1))
        x = self.resnet_block("mid_block1", x, temb, g)
        x = self.q("TRUNK", self.full_attention("mid_attn", x) + x)
        x = self.resnet_block("mid_block2", x, temb, g)
        x = self.resnet_block("conv_fusion", torch.cat([x, cond_feat], 1), None, g)
        for j in range(ns):
            p = f"ups.{j}"
            x = self.resnet_block(p + ".0", torch.cat([x, skips.pop()], 1), temb, g)
            x = self.resnet_block(p + ".1", torch.cat([x, skips.pop()], 1), temb, g)
            x = self.q("TRUNK", attn(p + ".2", x, fa[ns - 1 - j]) + x)
            if j < ns - 1:
                x = self.q("TRUNK", unet_ref.upsample_conv(sd, p + ".3", x))
            else:
                x = self.q("TRUNK", F.conv2d(x, sd[p + ".3.weight"], sd[p + ".3.bias"], padding=1))
        x = self.resnet_block("final_res_block", torch.cat([x, r], 1), temb, g)
        return F.conv2d(x, sd["final_conv.weight"], sd["final_conv.bias"])       # fp32 out of a 16-bit x: final_conv reads TRUNK


def run(on, tdt, until, golden):
    cfg = weights.UnetConfig(mode="mri")
    sd = {k: torch.from_numpy(v) for k, v in weights.procedural_state_dict(cfg, 0).items()}
    emu = Emu(sd, cfg, on, tdt)
    H, T = 128, 1000
    cond = torch.from_numpy(rng.uniform((1, 1, H, H), 5, 1, 0.0, 2.0))
    buf = diffusion_ref.schedule_buffers("sigmoid", T, "pred_x0")
    ns = rng.NoiseStream(10)
    x = torch.from_numpy(ns.next((1, 1, H, H)))
    out = {}
    with torch.no_grad():
        feat = emu.cond_encoder(cond)
        for t in range(T - 1, until - 1, -1):
            x0 = emu.forward(x, feat, torch.full((1,), t, dtype=torch.long)).clamp(0.0, 2.0)
            z = torch.from_numpy(ns.next((1, 1, H, H))) if t > 0 else 0.0
            x = buf["posterior_mean_coef1"][t] * x0 + buf["posterior_mean_coef2"][t] * x + (0.5 * buf["posterior_log_variance_clipped"][t]).exp() * z
            key = f"x_after_t{t}"
            if key in golden.files:
                d = np.abs(x.numpy() - golden[key])
                out[t] = (float(d.max()), float(d.mean()))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--until", type=int, default=100)
    ap.add_argument("--classes", default="none,all,W,RAW,ACT,TRUNK,ATTN")
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    tdt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    golden = np.load(os.path.join(ROOT, "tests", "golden", "g5_cfg2_mri128.npz"))
    print(f"# cfg2 (1x128x128, T=1000) through t={a.until}, roundings to {a.dtype} emulated on the oracle; max-abs / mean-abs vs the golden")
    for name in a.classes.split(","):
        on = CLASSES if name == "all" else (() if name == "none" else tuple(c for c in name.split("+")))
        t0 = time.time()
        res = run(on, tdt, a.until, golden)
        cells = "  ".join(f"t={t}: {mx:.2e} / {mn:.2e}" for t, (mx, mn) in sorted(res.items(), reverse=True))
        print(f"{name:12s} {cells}   [{time.time() - t0:.0f} s]", flush=True)


if __name__ == "__main__":
    main()

"""Parameter inventory and procedural weights for the conditional denoiser.

The parameter names and shapes reproduce the ``state_dict`` of the reference ``Unet``
(/root/reference/ddpm.py:286-398) including its ``ResUnet`` conditioning encoder
(/root/reference/unet_model.py:91-116), so that real checkpoints written by the reference's
``Trainer.save`` (/root/reference/ddpm.py:1495-1507) load by name.  No checkpoint ships with the
reference, so tests/bench use *procedural* weights: every tensor is a pure function of its
parameter name and a seed (see ``procedural_state_dict``), regenerated identically in the build
container (golden generation) and on the GPU box.
"""
from collections import OrderedDict
from dataclasses import dataclass
from typing import Tuple

import numpy as np

from . import rng


@dataclass(frozen=True)
class UnetConfig:
    """Constructor arguments of the reference Unet that change the parameter set
    (/root/reference/ddpm.py:287-307)."""
    dim: int = 32
    init_dim: int = 32
    out_dim: int = 1
    dim_mults: Tuple[int, ...] = (1, 2, 4, 8)
    channels: int = 1
    resnet_block_groups: int = 8
    attn_dim_head: int = 32
    attn_heads: int = 4
    full_attn: Tuple[bool, ...] = (False, False, False, True)
    mode: str = "mri"
    learned_sinusoidal_dim: int = 0       # > 0: RandomOrLearnedSinusoidalPosEmb of that dim instead of SinusoidalPosEmb (ddpm.py:330-337)

    @property
    def dims(self):
        return [self.init_dim] + [self.dim * m for m in self.dim_mults]

    @property
    def in_out(self):
        d = self.dims
        return list(zip(d[:-1], d[1:]))

    @property
    def time_dim(self):
        return self.dim * 4

    @property
    def fourier_dim(self):
        # ddpm.py:332-337: learned / random Fourier features are [t, sin, cos] (dim + 1), the sinusoidal embedding is `dim` wide
        return self.learned_sinusoidal_dim + 1 if self.learned_sinusoidal_dim else self.dim

    @property
    def hidden(self):
        return self.attn_dim_head * self.attn_heads

    @property
    def cond_in_channels(self):
        # /root/reference/unet_model.py:94-99
        if "mvtecGray" in self.mode:
            return 1
        if "mvtec" in self.mode:
            return 3
        return 1

    @property
    def cond_has_mid(self):
        # /root/reference/unet_model.py:113-115 (module exists) ...
        return self.mode in ("mri", "mvtec", "mvtecGray")

    @property
    def cond_early_exit(self):
        # ... /root/reference/unet_model.py:131-132 (forward returns after level 3)
        return self.mode in ("mnist", "mvtecSR")

    @property
    def downsample_factor(self):
        return 2 ** (len(self.dim_mults) - 1)


def _conv(sh, name, cout, cin, k, bias=True):
    sh[name + ".weight"] = (cout, cin, k, k)
    if bias:
        sh[name + ".bias"] = (cout,)


def _gn(sh, name, c):
    sh[name + ".weight"] = (c,)
    sh[name + ".bias"] = (c,)


def _resblock(sh, p, cin, cout, tdim):
    sh[p + ".mlp.1.weight"] = (2 * cout, tdim)
    sh[p + ".mlp.1.bias"] = (2 * cout,)
    _conv(sh, p + ".block1.proj", cout, cin, 3)
    _gn(sh, p + ".block1.norm", cout)
    _conv(sh, p + ".block2.proj", cout, cout, 3)
    _gn(sh, p + ".block2.norm", cout)
    if cin != cout:
        _conv(sh, p + ".res_conv", cout, cin, 1)


def _attn(sh, p, c, hidden, full):
    sh[p + ".norm.g"] = (1, c, 1, 1)
    sh[p + ".to_qkv.weight"] = (3 * hidden, c, 1, 1)
    if full:
        _conv(sh, p + ".to_out", c, hidden, 1)
    else:
        _conv(sh, p + ".to_out.0", c, hidden, 1)
        sh[p + ".to_out.1.g"] = (1, c, 1, 1)


def _basic_block(sh, p, cin, cmid, cout):
    _conv(sh, p + ".convblock.0", cmid, cin, 3)
    _gn(sh, p + ".convblock.1", cmid)
    _conv(sh, p + ".convblock.3", cout, cmid, 3)
    _gn(sh, p + ".convblock.4", cout)
    if cin != cout:
        _conv(sh, p + ".identity.0", cout, cin, 3)
        _gn(sh, p + ".identity.1", cout)


def unet_param_shapes(cfg: UnetConfig) -> "OrderedDict[str, tuple]":
    """name -> shape, in the reference's registration order."""
    sh = OrderedDict()
    f = [32, 32, 64, 128, 256]                    # unet_model.py:100
    _basic_block(sh, "cond_model.residual_conv1.0", cfg.cond_in_channels, f[0], f[1])
    _basic_block(sh, "cond_model.residual_conv2.0", f[1], f[1], f[2])
    _basic_block(sh, "cond_model.residual_conv3.0", f[2], f[2], f[3])
    if cfg.cond_has_mid:
        _basic_block(sh, "cond_model.mid_conv.0", f[3], f[3], f[4])
    _conv(sh, "init_conv", cfg.init_dim, cfg.channels, 7)
    td = cfg.time_dim
    if cfg.learned_sinusoidal_dim:
        sh["time_mlp.0.weights"] = (cfg.learned_sinusoidal_dim // 2,)            # ddpm.py:158 (requires_grad only differs)
    sh["time_mlp.1.weight"] = (td, cfg.fourier_dim)
    sh["time_mlp.1.bias"] = (td,)
    sh["time_mlp.3.weight"] = (td, td)
    sh["time_mlp.3.bias"] = (td,)
    io = cfg.in_out
    n = len(io)
    for i, (cin, cout) in enumerate(io):
        p = f"downs.{i}"
        _resblock(sh, p + ".0", cin, cin, td)
        _resblock(sh, p + ".1", cin, cin, td)
        _attn(sh, p + ".2", cin, cfg.hidden, cfg.full_attn[i])
        if i < n - 1:
            _conv(sh, p + ".3.1", cout, cin * 4, 1)
        else:
            _conv(sh, p + ".3", cout, cin, 3)
    for j, ((cin, cout), full) in enumerate(zip(reversed(io), reversed(cfg.full_attn))):
        p = f"ups.{j}"
        _resblock(sh, p + ".0", cout + cin, cout, td)
        _resblock(sh, p + ".1", cout + cin, cout, td)
        _attn(sh, p + ".2", cout, cfg.hidden, full)
        if j < n - 1:
            _conv(sh, p + ".3.1", cin, cout, 3)
        else:
            _conv(sh, p + ".3", cin, cout, 3)
    mid = cfg.dims[-1]
    _resblock(sh, "mid_block1", mid, mid, td)
    _attn(sh, "mid_attn", mid, cfg.hidden, True)
    _resblock(sh, "mid_block2", mid, mid, td)
    _resblock(sh, "conv_fusion", 2 * mid, mid, td)
    _resblock(sh, "final_res_block", 2 * cfg.dim, cfg.dim, td)
    _conv(sh, "final_conv", cfg.out_dim, cfg.dim, 1)
    return sh


def _fan_in(shape):
    if len(shape) >= 2:
        return int(np.prod(shape[1:]))
    return None


def procedural_tensor(name, shape, seed=0):
    """One parameter tensor as a pure function of (name, seed).

    conv / linear weights and biases: uniform(+-1/sqrt(fan_in)) (bias uses its layer's fan_in);
    GroupNorm gamma = 1 + 0.1 u, beta = 0.1 u;  RMSNorm g = 1 + 0.1 u  (u uniform in [-1, 1)).
    ``final_conv`` is re-centred (bias ~ 1, 3x weights) so that the predicted x0 lives inside the
    [0, 2] clamp window used by the reference's callers (test.py:30-37) instead of saturating.
    """
    key = rng.fnv1a64(name)
    u = rng.uniform(shape, seed, key, -1.0, 1.0)
    if name.endswith(".g"):
        return (1.0 + 0.1 * u).astype(np.float32)
    return u


def procedural_state_dict(cfg: UnetConfig, seed=0, final_gain=3.0):
    """``final_gain``: scale of ``final_conv.weight`` over its fan-in bound.  3.0 (the default every fixture but G16 uses)
    makes a random-init denoiser whose prediction swings over the whole [0, 2] range and whose last 100 reverse steps
    amplify any perturbation x40-70; a small gain makes the chain CONTRACTIVE there, like a trained denoiser whose
    prediction at small t stays near x_t (golden G16: the end-to-end pin of the 16-bit storage modes)."""
    shapes = unet_param_shapes(cfg)
    out = OrderedDict()
    for name, shape in shapes.items():
        u = procedural_tensor(name, shape, seed)
        is_norm = (".norm." in name and not name.endswith(".g")) or \
                  ("convblock.1." in name) or ("convblock.4." in name) or ("identity.1." in name)
        if name.endswith(".g") or name.endswith(".weights"):      # (.weights: the Fourier frequencies, any O(1) values)
            t = u
        elif is_norm:
            t = (1.0 + 0.1 * u) if name.endswith(".weight") else (0.1 * u)
        else:
            if name.endswith(".weight"):
                bound = 1.0 / np.sqrt(_fan_in(shape))
            else:
                wshape = shapes[name[:-len(".bias")] + ".weight"]
                bound = 1.0 / np.sqrt(_fan_in(wshape))
            t = u * bound
            if name == "final_conv.weight":
                t = t * float(final_gain)
            if name == "final_conv.bias":
                t = 1.0 + t
        out[name] = np.ascontiguousarray(t, dtype=np.float32)
    return out


def num_params(cfg: UnetConfig):
    return sum(int(np.prod(s)) for s in unet_param_shapes(cfg).values())

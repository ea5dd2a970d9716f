"""ORACLE (test infrastructure, never shipped on the product path).

CPU restatement, in plain functional PyTorch fp32, of the reference's conditional denoiser:
``Unet.forward`` (/root/reference/ddpm.py:404-451) with its blocks (:114-282), the attention
core (/root/reference/attend.py:84-113) and the ``ResUnet`` conditioning encoder
(/root/reference/unet_model.py:8-51, 91-137).

It is pinned against the reference itself: ``tools/make_goldens.py`` imports the real reference
modules in the build container, loads the same procedural weights by name and checks this file's
outputs against them (and writes the fixtures in ``tests/golden/`` that
``tests/test_oracle_golden.py`` re-checks everywhere).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

Weights are a flat ``{name: tensor}`` dict with the reference's ``state_dict`` names; activations
are NCHW float32, as in the reference.
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- small pieces
def rms_norm(x, g):
    """ddpm.py:126-132  F.normalize over channels (eps 1e-12) * g * sqrt(C)."""
    n = x.norm(dim=1, keepdim=True).clamp_min(1e-12)
    return x / n * g * (x.shape[1] ** 0.5)


def time_embedding(sd, time, dim, theta=10000.0):
    """ddpm.py:136-149 (sinusoidal, half=dim/2 frequencies) + :339-344 (Linear-GELU-Linear).  With a ``time_mlp.0.weights``
    entry the first module is RandomOrLearnedSinusoidalPosEmb (ddpm.py:151-165; learned_sinusoidal_cond /
    random_fourier_features): [t, sin(2 pi w t), cos(2 pi w t)]."""
    if "time_mlp.0.weights" in sd:
        x = time[:, None]                                                  # :161 rearrange 'b -> b 1'
        freqs = x * sd["time_mlp.0.weights"][None, :] * 2 * math.pi        # :162
        emb = torch.cat((x, torch.cat((freqs.sin(), freqs.cos()), dim=-1)), dim=-1)   # :163-164
    else:
        half = dim // 2
        step = math.log(theta) / (half - 1)
        freq = torch.exp(torch.arange(half) * -step)
        ang = time[:, None] * freq[None, :]
        emb = torch.cat([ang.sin(), ang.cos()], dim=-1)
    h = F.linear(emb, sd["time_mlp.1.weight"], sd["time_mlp.1.bias"])
    h = F.gelu(h)
    return F.linear(h, sd["time_mlp.3.weight"], sd["time_mlp.3.bias"])


def conv_gn_act(sd, p, x, groups, film=None):
    """ddpm.py:170-186 Block: conv3x3 -> GroupNorm -> optional FiLM -> SiLU."""
    y = F.conv2d(x, sd[p + ".proj.weight"], sd[p + ".proj.bias"], padding=1)
    y = F.group_norm(y, groups, sd[p + ".norm.weight"], sd[p + ".norm.bias"], eps=1e-5)
    if film is not None:
        scale, shift = film
        y = y * (scale + 1) + shift
    return F.silu(y)


def resnet_block(sd, p, x, temb, groups=8):
    """ddpm.py:188-212.  ``temb=None`` reproduces the conv_fusion call (:436)."""
    film = None
    if temb is not None:
        e = F.linear(F.silu(temb), sd[p + ".mlp.1.weight"], sd[p + ".mlp.1.bias"])
        e = e[:, :, None, None]
        film = e.chunk(2, dim=1)
    h = conv_gn_act(sd, p + ".block1", x, groups, film)
    h = conv_gn_act(sd, p + ".block2", h, groups)
    if (p + ".res_conv.weight") in sd:
        x = F.conv2d(x, sd[p + ".res_conv.weight"], sd[p + ".res_conv.bias"])
    return h + x


def linear_attention(sd, p, x, heads=4, dim_head=32):
    """ddpm.py:214-251 (softmax over d for q, over n for k; no 1/(h*w) scaling of v)."""
    b, c, hh, ww = x.shape
    n = hh * ww
    y = rms_norm(x, sd[p + ".norm.g"])
    qkv = F.conv2d(y, sd[p + ".to_qkv.weight"])
    q, k, v = [t.reshape(b, heads, dim_head, n) for t in qkv.chunk(3, dim=1)]
    q = q.softmax(dim=-2) * (dim_head ** -0.5)
    k = k.softmax(dim=-1)
    ctx = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", ctx, q).reshape(b, heads * dim_head, hh, ww)
    out = F.conv2d(out, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])
    return rms_norm(out, sd[p + ".to_out.1.g"])


def full_attention(sd, p, x, heads=4, dim_head=32):
    """ddpm.py:253-282 + attend.py:84-113 (softmax(q k^T / sqrt(d)) v, no dropout)."""
    b, c, hh, ww = x.shape
    n = hh * ww
    y = rms_norm(x, sd[p + ".norm.g"])
    qkv = F.conv2d(y, sd[p + ".to_qkv.weight"])
    q, k, v = [t.reshape(b, heads, dim_head, n).transpose(-1, -2) for t in qkv.chunk(3, dim=1)]
    sim = torch.einsum("bhid,bhjd->bhij", q, k) * (dim_head ** -0.5)
    att = sim.softmax(dim=-1)
    out = torch.einsum("bhij,bhjd->bhid", att, v)
    out = out.transpose(-1, -2).reshape(b, heads * dim_head, hh, ww)
    return F.conv2d(out, sd[p + ".to_out.weight"], sd[p + ".to_out.bias"])


def pixel_unshuffle_conv(sd, p, x):
    """ddpm.py:120-124  'b c (h p1) (w p2) -> b (c p1 p2) h w' then conv1x1."""
    b, c, h, w = x.shape
    y = x.reshape(b, c, h // 2, 2, w // 2, 2).permute(0, 1, 3, 5, 2, 4)
    y = y.reshape(b, c * 4, h // 2, w // 2)
    return F.conv2d(y, sd[p + ".1.weight"], sd[p + ".1.bias"])


def upsample_conv(sd, p, x):
    """ddpm.py:114-118 nearest x2 then conv3x3."""
    y = F.interpolate(x, scale_factor=2, mode="nearest")
    return F.conv2d(y, sd[p + ".1.weight"], sd[p + ".1.bias"], padding=1)


# ----------------------------------------------------------------------------- cond encoder
def basic_block(sd, p, x, groups=16):
    """unet_model.py:8-51: ReLU( GN(conv(ReLU(GN(conv x)))) + GN(conv_id x) )."""
    y = F.conv2d(x, sd[p + ".convblock.0.weight"], sd[p + ".convblock.0.bias"], padding=1)
    y = F.relu(F.group_norm(y, groups, sd[p + ".convblock.1.weight"], sd[p + ".convblock.1.bias"]))
    y = F.conv2d(y, sd[p + ".convblock.3.weight"], sd[p + ".convblock.3.bias"], padding=1)
    y = F.group_norm(y, groups, sd[p + ".convblock.4.weight"], sd[p + ".convblock.4.bias"])
    idn = x
    if (p + ".identity.0.weight") in sd:
        idn = F.conv2d(x, sd[p + ".identity.0.weight"], sd[p + ".identity.0.bias"], padding=1)
        idn = F.group_norm(idn, groups, sd[p + ".identity.1.weight"], sd[p + ".identity.1.bias"])
    return F.relu(y + idn)


def cond_encoder(sd, cond, mode, prefix="cond_model"):
    """unet_model.py:117-137."""
    x = basic_block(sd, prefix + ".residual_conv1.0", cond)
    x = F.max_pool2d(x, 2)
    x = basic_block(sd, prefix + ".residual_conv2.0", x)
    x = F.max_pool2d(x, 2)
    x = basic_block(sd, prefix + ".residual_conv3.0", x)
    if mode in ("mnist", "mvtecSR"):
        return x
    x = F.max_pool2d(x, 2)
    return basic_block(sd, prefix + ".mid_conv.0", x)


# ----------------------------------------------------------------------------- whole network
def unet_forward(sd, cfg, x, cond, time, taps=None):
    """ddpm.py:404-451.  ``cfg`` is a ``weights.UnetConfig``-like object (dim, dim_mults,
    full_attn, attn_heads, attn_dim_head, resnet_block_groups, mode).  ``taps`` (optional dict)
    receives named intermediate activations for per-layer parity tests."""
    def tap(name, t):
        if taps is not None:
            taps[name] = t
        return t

    g = cfg.resnet_block_groups
    hd, dh = cfg.attn_heads, cfg.attn_dim_head
    n_stage = len(cfg.dim_mults)
    fa = tuple(cfg.full_attn)

    def attn(p, t, full):
        fn = full_attention if full else linear_attention
        return fn(sd, p, t, hd, dh)

    assert all(d % (2 ** (n_stage - 1)) == 0 for d in x.shape[-2:])
    x = F.conv2d(x, sd["init_conv.weight"], sd["init_conv.bias"], padding=3)
    tap("init_conv", x)
    r = x
    temb = tap("time_mlp", time_embedding(sd, time, cfg.dim))
    skips = []
    for i in range(n_stage):
        p = f"downs.{i}"
        x = tap(p + ".0", resnet_block(sd, p + ".0", x, temb, g))
        skips.append(x)
        x = tap(p + ".1", resnet_block(sd, p + ".1", x, temb, g))
        x = tap(p + ".2", attn(p + ".2", x, fa[i]) + x)
        skips.append(x)
        if i < n_stage - 1:
            x = pixel_unshuffle_conv(sd, p + ".3", x)
        else:
            x = F.conv2d(x, sd[p + ".3.weight"], sd[p + ".3.bias"], padding=1)
        tap(p + ".3", x)
    x = tap("mid_block1", resnet_block(sd, "mid_block1", x, temb, g))
    x = tap("mid_attn", full_attention(sd, "mid_attn", x, hd, dh) + x)
    x = tap("mid_block2", resnet_block(sd, "mid_block2", x, temb, g))
    feat = tap("cond_model", cond_encoder(sd, cond.to(torch.float32), cfg.mode))
    x = tap("conv_fusion", resnet_block(sd, "conv_fusion", torch.cat([x, feat], 1), None, g))
    for j in range(n_stage):
        p = f"ups.{j}"
        x = tap(p + ".0", resnet_block(sd, p + ".0", torch.cat([x, skips.pop()], 1), temb, g))
        x = tap(p + ".1", resnet_block(sd, p + ".1", torch.cat([x, skips.pop()], 1), temb, g))
        x = tap(p + ".2", attn(p + ".2", x, fa[n_stage - 1 - j]) + x)
        if j < n_stage - 1:
            x = upsample_conv(sd, p + ".3", x)
        else:
            x = F.conv2d(x, sd[p + ".3.weight"], sd[p + ".3.bias"], padding=1)
        tap(p + ".3", x)
    x = tap("final_res_block", resnet_block(sd, "final_res_block", torch.cat([x, r], 1), temb, g))
    return F.conv2d(x, sd["final_conv.weight"], sd["final_conv.bias"])

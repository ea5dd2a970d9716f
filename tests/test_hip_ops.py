"""Per-kernel parity: each C-ABI op against the same op in plain PyTorch fp32 on the CPU
(the building blocks of oracle/unet_ref.py), on seeded inputs, fp32, bf16 and fp16 storage."""
import ctypes as C
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from localdiffusion_hallucination_amd import _cabi as cabi       # noqa: E402
from localdiffusion_hallucination_amd import rng, schedule        # noqa: E402
from oracle import unet_ref                                        # noqa: E402
import hip_helpers as hh                                           # noqa: E402

DTYPES = ["fp32", "bf16", "fp16"]
LOWP = ["bf16", "fp16"]


def _q(x, dtype):
    """Round a CPU fp32 tensor to the storage dtype (so both sides see identical inputs)."""
    return x.to(hh.TDT[dtype]).float()


# ------------------------------------------------------------------------------ conv3x3
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,cin,cout,H,W", [(2, 32, 32, 16, 16), (1, 64, 32, 20, 28), (2, 32, 64, 8, 8),
                                            (1, 128, 64, 7, 7), (1, 32, 32, 40, 48), (1, 256, 256, 8, 8)])
def test_conv3x3_plain_and_stats(dtype, B, cin, cout, H, W):
    x, w, b = _q(hh.rand((B, cin, H, W), 1), dtype), _q(hh.rand((cout, cin, 3, 3), 2, -0.1, 0.1), dtype), hh.rand((cout,), 3)
    ref = F.conv2d(x, w, b, padding=1)
    stats = hh.stats_buffer(B, 8)
    out = hh.conv3x3([hh.make_src(hh.nhwc(x, dtype), cin)], hh.pack(w, dtype, 3), b.to(hh.DEV), B, H, W, cout, dtype,
                     stats=stats, groups=8)
    assert hh.rel_err(hh.nchw(out), ref) < hh.RTOL[dtype]
    sref = hh.gn_stats_ref(ref, 8)
    assert hh.rel_err(stats.sum(1).cpu(), sref) < (1e-5 if dtype == "fp32" else 1e-2)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("B,c1,c2,cout,H,W", [(2, 256, 128, 256, 16, 16), (1, 128, 64, 128, 24, 40), (2, 64, 32, 64, 17, 23), (3, 32, 32, 32, 32, 16)])
def test_conv3x3_with_the_res_conv_as_a_side_output(dtype, B, c1, c2, cout, H, W):
    """ld_conv3x3_args.side_*: a ResnetBlock's res_conv (a 1x1 convolution of the same concatenated input, ddpm.py:198, 212) as a
    second output of its block1 convolution's launch.  Both outputs against fp32 F.conv2d; the 3x3 output and its statistics
    bit-equal to the launch without the side output where the tile variant is the same; ragged tiles, 32- and 64-channel tiles."""
    x1, x2 = _q(hh.rand((B, c1, H, W), 40), dtype), _q(hh.rand((B, c2, H, W), 41), dtype)
    w, b = _q(hh.rand((cout, c1 + c2, 3, 3), 42, -0.1, 0.1), dtype), hh.rand((cout,), 43)
    wr, br = _q(hh.rand((cout, c1 + c2, 1, 1), 44, -0.1, 0.1), dtype), hh.rand((cout,), 45)
    xc = torch.cat([x1, x2], 1)
    ref, ref_side = F.conv2d(xc, w, b, padding=1), F.conv2d(xc, wr, br)
    srcs = lambda: [hh.make_src(hh.nhwc(x1, dtype), c1), hh.make_src(hh.nhwc(x2, dtype), c2)]
    stats, stats0 = hh.stats_buffer(B, 8), hh.stats_buffer(B, 8)
    out, side = hh.conv3x3(srcs(), hh.pack(w, dtype, 3), b.to(hh.DEV), B, H, W, cout, dtype, stats=stats, groups=8,
                           side=(hh.pack(wr, dtype, 1), br.to(hh.DEV)))
    out0 = hh.conv3x3(srcs(), hh.pack(w, dtype, 3), b.to(hh.DEV), B, H, W, cout, dtype, stats=stats0, groups=8)
    assert hh.rel_err(hh.nchw(out), ref) < hh.RTOL[dtype] and hh.rel_err(hh.nchw(side), ref_side) < hh.RTOL[dtype]
    assert hh.rel_err(stats.sum(1).cpu(), hh.gn_stats_ref(ref, 8)) < 1e-2
    # the taps arrive in the same order per accumulator whatever the tile: the 3x3 output does not change
    assert torch.equal(out, out0)
    # ... and the side output is the 1x1 kernel's result for the same operands (same chunk order, same MFMA)
    side1 = hh.conv1x1(srcs(), hh.pack(wr, dtype, 1), B, H, W, cout, dtype, bias=br.to(hh.DEV))
    assert torch.equal(side, side1)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv3x3_concat_upsample(dtype):
    B, c1, c2, cout, H, W = 2, 64, 32, 32, 12, 12
    x1, x2 = _q(hh.rand((B, c1, H // 2, W // 2), 4), dtype), _q(hh.rand((B, c2, H, W), 5), dtype)
    w, b = _q(hh.rand((cout, c1 + c2, 3, 3), 6, -0.1, 0.1), dtype), hh.rand((cout,), 7)
    ref = F.conv2d(torch.cat([F.interpolate(x1, scale_factor=2, mode="nearest"), x2], 1), w, b, padding=1)
    out = hh.conv3x3([hh.make_src(hh.nhwc(x1, dtype), c1, ups=1), hh.make_src(hh.nhwc(x2, dtype), c2)],
                     hh.pack(w, dtype, 3), b.to(hh.DEV), B, H, W, cout, dtype)
    assert hh.rel_err(hh.nchw(out), ref) < hh.RTOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("act,groups,use_film", [(cabi.ACT_SILU, 8, True), (cabi.ACT_RELU, 16, False)])
def test_conv3x3_gn_film_prologue(dtype, act, groups, use_film):
    """conv(act(GN(x)*(scale+1)+shift)) with GN statistics taken from x itself."""
    B, cin, cout, H, W = 2, 64, 32, 10, 14
    x = _q(hh.rand((B, cin, H, W), 8, -2.0, 3.0), dtype)
    gamma, beta = hh.rand((cin,), 9, 0.5, 1.5), hh.rand((cin,), 10, -0.3, 0.3)
    film = hh.rand((B, 2 * cin), 11, -0.5, 0.5)
    w, b = _q(hh.rand((cout, cin, 3, 3), 12, -0.1, 0.1), dtype), hh.rand((cout,), 13)
    y = F.group_norm(x, groups, gamma, beta, eps=1e-5)
    if use_film:
        y = y * (film[:, :cin, None, None] + 1) + film[:, cin:, None, None]
    y = F.silu(y) if act == cabi.ACT_SILU else F.relu(y)
    ref = F.conv2d(y, w, b, padding=1)
    stats = hh.stats_striped(x, groups)
    g_d, b_d, f_d = gamma.to(hh.DEV), beta.to(hh.DEV), film.to(hh.DEV)
    src = hh.make_src(hh.nhwc(x, dtype), cin, gn=(stats, g_d, b_d, groups), act=act,
                      film=f_d if use_film else None, film_b=2 * cin)
    out = hh.conv3x3([src], hh.pack(w, dtype, 3), b.to(hh.DEV), B, H, W, cout, dtype)
    assert hh.rel_err(hh.nchw(out), ref) < hh.RTOL[dtype] * 2


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("case", ["plain32", "one64", "two32", "pro_film", "pro_nofilm", "pro_perbatch"])
def test_conv3x3_s32_lean_kernel(dtype, case):
    """The lean Cout = 32 kernel of the ResBlock conv path (conv3x3_s32.hip: four workgroups per CU, GroupNorm coefficients
    in registers, statistics as per-wave atomics): against the fp32 reference AND against the generic kernel on the same
    inputs (routing switched through the tuning table and asserted with the launch counters); 32 and 64 input channels,
    two concatenated sources, the GroupNorm + FiLM + SiLU prologue with a shared and a per-sample FiLM row."""
    from localdiffusion_hallucination_amd.tuning import kernel_table
    B, cout, H, W = 2, 32, 48, 32
    cin = 32 if case in ("plain32", "pro_film", "pro_nofilm", "pro_perbatch") else 64
    x = _q(hh.rand((B, cin, H, W), 80, -2.0, 3.0), dtype)
    w, b = _q(hh.rand((cout, cin, 3, 3), 81, -0.1, 0.1), dtype), hh.rand((cout,), 82)
    y, srcs = x, None
    xd = hh.nhwc(x, dtype)
    if case.startswith("pro"):
        gamma, beta = hh.rand((cin,), 83, 0.5, 1.5), hh.rand((cin,), 84, -0.3, 0.3)
        y = F.group_norm(x, 8, gamma, beta, eps=1e-5)
        film = None
        if case == "pro_film":            # the sampler's table mode: one row for the whole batch
            film = hh.rand((1, 2 * cin), 85, -0.5, 0.5)
            y = y * (film[:, :cin, None, None] + 1) + film[:, cin:, None, None]
        elif case == "pro_perbatch":      # Unet.forward: a row per batch element
            film = hh.rand((B, 2 * cin), 85, -0.5, 0.5)
            y = y * (film[:, :cin, None, None] + 1) + film[:, cin:, None, None]
        y = F.silu(y)
        f_d = None if film is None else film.to(hh.DEV)
        # the producer's statistics spread over all 16 stripes (as the convolutions' epilogues leave them)
        wgt = torch.linspace(1.0, 2.0, cabi.STAT_STRIPES, dtype=torch.float64)
        wgt = (wgt / wgt.sum()).to(hh.DEV)
        sx = hh.stats_buffer(B, 8)
        sx[:] = hh.gn_stats_ref(x, 8).to(hh.DEV)[:, None] * wgt[None, :, None, None]
        srcs = [hh.make_src(xd, cin, gn=(sx, gamma.to(hh.DEV), beta.to(hh.DEV), 8), act=cabi.ACT_SILU,
                            film=f_d, film_b=2 * cin if case == "pro_perbatch" else 0)]
    elif case == "two32":
        x2 = hh.nhwc(x[:, 32:].contiguous(), dtype)
        srcs = [hh.make_src(xd, 32, stride=64), hh.make_src(x2, 32)]       # first source: a channel slice of a wider tensor
    else:
        srcs = [hh.make_src(xd, cin)]
    ref = F.conv2d(y, w, b, padding=1)
    wp, bd = hh.pack(w, dtype, 3), b.to(hh.DEV)
    lib = cabi.lib()
    keep = kernel_table(lib)
    outs, stats = {}, {}
    try:
        cabi.check(lib.ld_tuning_set(b"conv_s32_min_tiles", 1), "tuning_set")
        for name, on in (("lean", 1), ("generic", 0)):
            cabi.check(lib.ld_tuning_set(b"conv_s32", 7 * on), "tuning_set")        # (a bit mask: every variant / none)
            n0 = lib.ld_counter(cabi.COUNTER_CONV3X3_S32)
            st = hh.stats_buffer(B, 8)
            out = hh.conv3x3(srcs, wp, bd, B, H, W, cout, dtype, stats=st, groups=8)
            torch.cuda.synchronize()
            assert lib.ld_counter(cabi.COUNTER_CONV3X3_S32) - n0 == on, name
            outs[name], stats[name] = hh.nchw(out), st.sum(1).cpu()
    finally:
        for kname, val in keep.items():
            cabi.check(lib.ld_tuning_set(kname.encode(), val), "tuning_set")
    tol = hh.RTOL[dtype] * (2 if case.startswith("pro") else 1)
    assert hh.rel_err(outs["lean"], ref) < tol
    assert hh.rel_err(stats["lean"], hh.gn_stats_ref(ref, 8)) < 1e-2
    if not case.startswith("pro"):        # same MFMA order, same bias add, same rounding: the raw variants are bit-equal
        assert torch.equal(outs["lean"], outs["generic"])
    else:                                 # (the coefficient sums meet in another order: a storage ulp here and there)
        assert hh.rel_err(outs["lean"], outs["generic"]) < hh.RTOL[dtype] / 2
    assert hh.rel_err(stats["lean"], stats["generic"]) < 1e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", ["plain", "pro", "ragged", "two_src"])
def test_conv3x3_big64_tile_equals_the_small_tile(dtype, case):
    """The 64-channel x 16-row tile of the throughput regime (`<4,4>`, routing entry conv_big4_min): the same accumulation
    order per output element as the 64 x 8-row tile, so the outputs are bit-equal; statistics meet in another order."""
    from localdiffusion_hallucination_amd.tuning import kernel_table
    B, cin, cout = 2, 64, 128
    H, W = (40, 24) if case == "ragged" else (32, 48)
    x = _q(hh.rand((B, cin, H, W), 90, -2.0, 3.0), dtype)
    w, b = _q(hh.rand((cout, cin, 3, 3), 91, -0.1, 0.1), dtype), hh.rand((cout,), 92)
    y, xd = x, hh.nhwc(x, dtype)
    if case == "pro":
        gamma, beta = hh.rand((cin,), 93, 0.5, 1.5), hh.rand((cin,), 94, -0.3, 0.3)
        film = hh.rand((1, 2 * cin), 95, -0.5, 0.5)
        y = F.group_norm(x, 8, gamma, beta, eps=1e-5)
        y = F.silu(y * (film[:, :cin, None, None] + 1) + film[:, cin:, None, None])
        sx = hh.stats_buffer(B, 8)
        sx[:, 0] = hh.gn_stats_ref(x, 8).to(hh.DEV)
        srcs = [hh.make_src(xd, cin, gn=(sx, gamma.to(hh.DEV), beta.to(hh.DEV), 8), act=cabi.ACT_SILU, film=film.to(hh.DEV))]
    elif case == "two_src":
        srcs = [hh.make_src(xd, 32, stride=64), hh.make_src(hh.nhwc(x[:, 32:].contiguous(), dtype), 32)]
    else:
        srcs = [hh.make_src(xd, cin)]
    ref = F.conv2d(y, w, b, padding=1)
    wp, bd = hh.pack(w, dtype, 3), b.to(hh.DEV)
    lib = cabi.lib()
    keep = kernel_table(lib)
    outs, stats = {}, {}
    try:
        cabi.check(lib.ld_tuning_set(b"conv_mt4_min_wgs", 1), "tuning_set")
        for name, big_min in (("rows16", 1), ("rows8", 1 << 40)):
            cabi.check(lib.ld_tuning_set(b"conv_big4_min", big_min), "tuning_set")
            st = hh.stats_buffer(B, 8)
            out = hh.conv3x3(srcs, wp, bd, B, H, W, cout, dtype, stats=st, groups=8)
            torch.cuda.synchronize()
            outs[name], stats[name] = hh.nchw(out), st.sum(1).cpu()
    finally:
        for kname, val in keep.items():
            cabi.check(lib.ld_tuning_set(kname.encode(), val), "tuning_set")
    assert hh.rel_err(outs["rows16"], ref) < hh.RTOL[dtype] * (2 if case == "pro" else 1)
    assert hh.rel_err(stats["rows16"], hh.gn_stats_ref(ref, 8)) < 1e-2
    assert torch.equal(outs["rows16"], outs["rows8"])
    assert hh.rel_err(stats["rows16"], stats["rows8"]) < 1e-5


# ------------------------------------------------------------------------------ conv1x1
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,cin,cout,H,W", [(2, 64, 32, 16, 16), (1, 384, 256, 8, 8), (1, 96, 64, 14, 14),
                                            (1, 32, 384, 28, 28), (2, 128, 128, 7, 7)])
def test_conv1x1_plain(dtype, B, cin, cout, H, W):
    x, w, b = _q(hh.rand((B, cin, H, W), 20), dtype), _q(hh.rand((cout, cin, 1, 1), 21, -0.2, 0.2), dtype), hh.rand((cout,), 22)
    ref = F.conv2d(x, w, b)
    out = hh.conv1x1([hh.make_src(hh.nhwc(x, dtype), cin)], hh.pack(w, dtype, 1), B, H, W, cout, dtype, bias=b.to(hh.DEV))
    assert hh.rel_err(hh.nchw(out), ref) < hh.RTOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv1x1_concat_and_unshuffle(dtype):
    B, c1, c2, cout, H, W = 1, 64, 32, 64, 12, 10
    x1, x2 = _q(hh.rand((B, c1, H, W), 23), dtype), _q(hh.rand((B, c2, H, W), 24), dtype)
    w, b = _q(hh.rand((cout, c1 + c2, 1, 1), 25, -0.2, 0.2), dtype), hh.rand((cout,), 26)
    ref = F.conv2d(torch.cat([x1, x2], 1), w, b)
    out = hh.conv1x1([hh.make_src(hh.nhwc(x1, dtype), c1), hh.make_src(hh.nhwc(x2, dtype), c2)], hh.pack(w, dtype, 1),
                     B, H, W, cout, dtype, bias=b.to(hh.DEV))
    assert hh.rel_err(hh.nchw(out), ref) < hh.RTOL[dtype]
    # Downsample: pixel-unshuffle + conv1x1 (ddpm.py:120-124)
    c, cout = 32, 64
    x = _q(hh.rand((B, c, 2 * H, 2 * W), 27), dtype)
    w, b = _q(hh.rand((cout, 4 * c, 1, 1), 28, -0.2, 0.2), dtype), hh.rand((cout,), 29)
    sd = {"d.1.weight": w, "d.1.bias": b}
    ref = unet_ref.pixel_unshuffle_conv(sd, "d", x)
    out = hh.conv1x1([hh.make_src(hh.nhwc(x, dtype), c)], hh.pack(w, dtype, 1, unshuffle=1), B, H, W, cout, dtype,
                     bias=b.to(hh.DEV), unshuffle=1)
    assert hh.rel_err(hh.nchw(out), ref) < hh.RTOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("c,H,W", [(32, 16, 16), (64, 14, 14), (128, 7, 7)])
def test_qkv_epilogues(dtype, c, H, W):
    """RMSNorm -> to_qkv -> (q softmax over d, * scale) as in LinearAttention; and the full-attn form."""
    B, hid = 2, 128
    x = _q(hh.rand((B, c, H, W), 30, -2, 2), dtype)
    g = hh.rand((1, c, 1, 1), 31, 0.5, 1.5)
    w = _q(hh.rand((3 * hid, c, 1, 1), 32, -0.3, 0.3), dtype)
    qkv = F.conv2d(unet_ref.rms_norm(x, g), w)
    q, k, v = qkv.chunk(3, dim=1)
    qs = q.reshape(B, 4, 32, H * W).softmax(dim=-2).reshape(B, hid, H, W) * 32 ** -0.5
    wp = hh.pack(w, dtype, 1, scale_in=g.flatten() * math.sqrt(c))
    out = hh.conv1x1([hh.make_src(hh.nhwc(x, dtype), c)], wp, B, H, W, 3 * hid, dtype, epi=cabi.EPI_QKV_LINEAR, rms_in=1)
    tol = hh.RTOL[dtype] * (3 if dtype != "fp32" else 5)
    assert hh.rel_err(hh.nchw(out), torch.cat([qs, k, v], 1)) < tol
    out = hh.conv1x1([hh.make_src(hh.nhwc(x, dtype), c)], wp, B, H, W, 3 * hid, dtype, epi=cabi.EPI_QKV_FULL, rms_in=1)
    assert hh.rel_err(hh.nchw(out), torch.cat([q * 32 ** -0.5, k, v], 1)) < tol


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("c", [32, 64, 128])
def test_conv1x1_rms_res_epilogue_and_batched_weights(dtype, c):
    B, hid, H, W = 2, 128, 9, 11
    x = _q(hh.rand((B, hid, H, W), 33), dtype)
    res = _q(hh.rand((B, c, H, W), 34), dtype)
    w = _q(hh.rand((B, c, hid, 1, 1), 35, -0.3, 0.3), dtype)          # a different weight per batch element
    b, g = hh.rand((c,), 36), hh.rand((1, c, 1, 1), 37, 0.5, 1.5)
    ref = torch.cat([unet_ref.rms_norm(F.conv2d(x[i:i + 1], w[i], b), g) for i in range(B)]) + res
    wp = torch.stack([hh.pack(w[i], dtype, 1) for i in range(B)]).contiguous()
    es = 4 if dtype == "fp32" else 2
    out = hh.conv1x1([hh.make_src(hh.nhwc(x, dtype), hid)], wp, B, H, W, c, dtype, bias=b.to(hh.DEV),
                     epi=cabi.EPI_RMS_RES, bstride=c * hid * es, g2=(g.flatten() * math.sqrt(c)).to(hh.DEV),
                     residual=hh.nhwc(res, dtype))
    assert hh.rel_err(hh.nchw(out), ref) < hh.RTOL[dtype] * 3
    ref = F.conv2d(x, w[0], b) + res
    out = hh.conv1x1([hh.make_src(hh.nhwc(x, dtype), hid)], wp, B, H, W, c, dtype, bias=b.to(hh.DEV),
                     epi=cabi.EPI_RES, residual=hh.nhwc(res, dtype))
    assert hh.rel_err(hh.nchw(out), ref) < hh.RTOL[dtype] * 2


# ------------------------------------------------------------------------------ image convs
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,ks,H,W", [(1, 7, 28, 28), (3, 7, 32, 48), (1, 3, 16, 16), (3, 3, 20, 12)])
def test_conv_image(dtype, cin, ks, H, W):
    B = 2
    x, w, b = hh.rand((B, cin, H, W), 40), hh.rand((32, cin, ks, ks), 41, -0.2, 0.2), hh.rand((32,), 42)
    ref = F.conv2d(x, w, b, padding=ks // 2)
    out = torch.empty(B, H, W, 32, dtype=hh.TDT[dtype], device=hh.DEV)
    stats = hh.stats_buffer(B, 16)
    xd, wd, bd = x.to(hh.DEV), w.to(hh.DEV), b.to(hh.DEV)
    cabi.check(cabi.lib().ld_conv_image(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(), stats.data_ptr(),
                                        16, B, cin, H, W, ks, cabi.dtype_code(dtype), hh.st()), "conv_image")
    assert hh.rel_err(hh.nchw(out), ref) < hh.RTOL[dtype]
    assert hh.rel_err(stats.sum(1).cpu(), hh.gn_stats_ref(ref, 16)) < 1e-5


@pytest.mark.parametrize("dtype", LOWP)
@pytest.mark.parametrize("cin,H,W", [(3, 64, 48), (1, 28, 28), (3, 40, 33), (3, 256, 256)])
def test_conv_stem_mfma_16bit(dtype, cin, H, W):
    """init_conv 7x7 for 16-bit storage as an im2col MFMA GEMM (bf16 hi/lo split of image and weights): equal to the
    fp32 reference up to the rounding of the stored output, and to the fp32-FMA kernel's 16-bit output within one ulp."""
    ulp = 2.0 ** -8 if dtype == "bf16" else 2.0 ** -11        # half an ulp, relative
    B = 2
    x, w, b = hh.rand((B, cin, H, W), 43, -1.5, 1.5), hh.rand((32, cin, 7, 7), 44, -0.2, 0.2), hh.rand((32,), 45)
    ref = F.conv2d(x, w, b, padding=3)
    lib = cabi.lib()
    xd, wd, bd = x.to(hh.DEV), w.to(hh.DEV).contiguous(), b.to(hh.DEV)
    wp = torch.empty(int(lib.ld_stem_packed_bytes()), dtype=torch.uint8, device=hh.DEV)
    cabi.check(lib.ld_pack_stem_weight(wd.data_ptr(), wp.data_ptr(), cin, hh.st()), "pack_stem")
    out = torch.empty(B, H, W, 32, dtype=hh.TDT[dtype], device=hh.DEV)
    cabi.check(lib.ld_conv_stem(xd.data_ptr(), wp.data_ptr(), bd.data_ptr(), out.data_ptr(), B, cin, H, W,
                                cabi.dtype_code(dtype), hh.st()), "conv_stem")
    got = hh.nchw(out)
    assert hh.rel_err(got, ref) < hh.RTOL[dtype]
    # the split products keep ~2^-16 relative accuracy: what is left is the final rounding of the stored value
    # (half an ulp) plus, for fp16, that 2^-16 of the largest product
    assert float(((got - ref).abs() / ref.abs().clamp_min(0.25)).max()) < ulp * 1.02 + (0 if dtype == "bf16" else 2.0 ** -13)
    out2 = torch.empty_like(out)
    cabi.check(lib.ld_conv_image(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out2.data_ptr(), None, 0, B, cin, H, W, 7,
                                 cabi.dtype_code(dtype), hh.st()), "conv_image")
    assert float((got - hh.nchw(out2)).abs().max()) <= 2 * ulp * float(ref.abs().max())


# ------------------------------------------------------------------------------ gn_apply
@pytest.mark.parametrize("dtype", DTYPES)
def test_gn_apply_resblock_tail_and_basicblock_tail(dtype):
    B, c, H, W = 2, 64, 12, 10
    a, r = _q(hh.rand((B, c, H, W), 50, -2, 3), dtype), _q(hh.rand((B, c, H, W), 51), dtype)
    ga, ba = hh.rand((c,), 52, 0.5, 1.5), hh.rand((c,), 53, -0.3, 0.3)
    ref = F.silu(F.group_norm(a, 8, ga, ba)) + r
    A = cabi.GnApplyArgs()
    ad, rd = hh.nhwc(a, dtype), hh.nhwc(r, dtype)
    sa = hh.stats_striped(a, 8)
    gad, bad = ga.to(hh.DEV), ba.to(hh.DEV)
    A.a = hh.make_src(ad, c, gn=(sa, gad, bad, 8), act=cabi.ACT_SILU)
    A.b = hh.make_src(rd, c)
    out = torch.empty(B, H, W, c, dtype=hh.TDT[dtype], device=hh.DEV)
    A.out, A.B, A.H, A.W, A.dtype = out.data_ptr(), B, H, W, cabi.dtype_code(dtype)
    cabi.check(cabi.lib().ld_gn_apply(C.byref(A), hh.st()), "gn_apply")
    assert hh.rel_err(hh.nchw(out), ref) < hh.RTOL[dtype]
    # BasicBlock tail with pooling: maxpool(relu(GN16(a) + GN16(r)))
    gr, br = hh.rand((c,), 54, 0.5, 1.5), hh.rand((c,), 55, -0.3, 0.3)
    ref = F.max_pool2d(F.relu(F.group_norm(a, 16, ga, ba) + F.group_norm(r, 16, gr, br)), 2)
    sa, sr = hh.stats_striped(a, 16), hh.stats_striped(r, 16)
    grd, brd = gr.to(hh.DEV), br.to(hh.DEV)
    A = cabi.GnApplyArgs()
    A.a = hh.make_src(ad, c, gn=(sa, gad, bad, 16))
    A.b = hh.make_src(rd, c, gn=(sr, grd, brd, 16))
    A.final_act, A.pool = cabi.ACT_RELU, 1
    out = torch.empty(B, H // 2, W // 2, c, dtype=hh.TDT[dtype], device=hh.DEV)
    A.out, A.B, A.H, A.W, A.dtype = out.data_ptr(), B, H, W, cabi.dtype_code(dtype)
    cabi.check(cabi.lib().ld_gn_apply(C.byref(A), hh.st()), "gn_apply")
    assert hh.rel_err(hh.nchw(out), ref) < hh.RTOL[dtype]


# ------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n_hw", [(7, 7), (16, 16), (32, 32), (64, 64)])
def test_full_attention(dtype, n_hw):
    B, hid = 2, 128
    H, W = n_hw
    n = H * W
    qkv = _q(hh.rand((B, 3 * hid, H, W), 60, -1.5, 1.5), dtype)
    q, k, v = [t.reshape(B, 4, 32, n).transpose(-1, -2) for t in qkv.chunk(3, dim=1)]
    att = (q @ k.transpose(-1, -2)).softmax(dim=-1) @ v            # q is taken as pre-scaled
    ref = att.transpose(-1, -2).reshape(B, hid, H, W)
    out = torch.empty(B, H, W, hid, dtype=hh.TDT[dtype], device=hh.DEV)
    qd = hh.nhwc(qkv, dtype)
    cabi.check(cabi.lib().ld_attention(qd.data_ptr(), out.data_ptr(), B, n, 4, 32, cabi.dtype_code(dtype), hh.st()), "attention")
    assert hh.rel_err(hh.nchw(out), ref) < hh.RTOL[dtype]


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("B,n_hw", [(1, (64, 64)), (2, (18, 18)), (2, (20, 20)), (4, (32, 32)), (1, (48, 48))])
def test_full_attention_two_key_groups(dtype, B, n_hw):
    """Launches of at most attn_split_max_wgs workgroups and at least attn_split_min_n keys run as two key groups per workgroup (attention.hip, KS = 2):
    against the fp32 reference, and against the one-group kernel (routing switched through the tuning table) -- full
    tiles, a ragged last tile, and a second group whose last tile has no key in range (n = 324)."""
    from localdiffusion_hallucination_amd.tuning import kernel_table
    hid = 128
    H, W = n_hw
    n = H * W
    qkv = _q(hh.rand((B, 3 * hid, H, W), 61, -1.5, 1.5), dtype)
    q, k, v = [t.reshape(B, 4, 32, n).transpose(-1, -2) for t in qkv.chunk(3, dim=1)]
    ref = ((q @ k.transpose(-1, -2)).softmax(dim=-1) @ v).transpose(-1, -2).reshape(B, hid, H, W)
    qd = hh.nhwc(qkv, dtype)
    lib = cabi.lib()
    keep = kernel_table(lib)
    outs = {}
    try:
        cabi.check(lib.ld_tuning_set(b"attn_split_min_n", 256), "tuning_set")
        for name, wgs in (("split", 1 << 30), ("one", 0)):
            cabi.check(lib.ld_tuning_set(b"attn_split_max_wgs", wgs), "tuning_set")
            out = torch.empty(B, H, W, hid, dtype=hh.TDT[dtype], device=hh.DEV)
            cabi.check(lib.ld_attention(qd.data_ptr(), out.data_ptr(), B, n, 4, 32, cabi.dtype_code(dtype), hh.st()), "attention")
            outs[name] = hh.nchw(out)
            assert hh.rel_err(outs[name], ref) < hh.RTOL[dtype], name
    finally:
        for kname, val in keep.items():
            cabi.check(lib.ld_tuning_set(kname.encode(), val), "tuning_set")
    # the two differ only in where the running maximum is taken (P is rounded to storage against it)
    assert hh.rel_err(outs["split"], outs["one"]) < hh.RTOL[dtype] / 2


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("n_hw", [(8, 8), (7, 7), (8, 16)])
def test_full_attention_split_thresholds_never_leave_a_key_group_empty(dtype, n_hw):
    """with attn_split_min_n lowered, a sequence that fits ONE key tile (n <= 128) must not take the two-group
    kernel -- its second group would own no key and every query would come out NaN.  Finite and equal to the reference."""
    from localdiffusion_hallucination_amd.tuning import kernel_table
    hid, B = 128, 2
    H, W = n_hw
    n = H * W
    qkv = _q(hh.rand((B, 3 * hid, H, W), 62, -1.5, 1.5), dtype)
    q, k, v = [t.reshape(B, 4, 32, n).transpose(-1, -2) for t in qkv.chunk(3, dim=1)]
    ref = ((q @ k.transpose(-1, -2)).softmax(dim=-1) @ v).transpose(-1, -2).reshape(B, hid, H, W)
    qd = hh.nhwc(qkv, dtype)
    lib = cabi.lib()
    keep = kernel_table(lib)
    try:
        cabi.check(lib.ld_tuning_set(b"attn_split_min_n", 0), "tuning_set")
        cabi.check(lib.ld_tuning_set(b"attn_split_max_wgs", 1 << 30), "tuning_set")
        out = torch.empty(B, H, W, hid, dtype=hh.TDT[dtype], device=hh.DEV)
        cabi.check(lib.ld_attention(qd.data_ptr(), out.data_ptr(), B, n, 4, 32, cabi.dtype_code(dtype), hh.st()), "attention")
        got = hh.nchw(out)
    finally:
        for kname, val in keep.items():
            cabi.check(lib.ld_tuning_set(kname.encode(), val), "tuning_set")
    assert torch.isfinite(got).all()
    assert hh.rel_err(got, ref) < hh.RTOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("c,H,W", [(32, 28, 28), (64, 14, 14), (32, 64, 96)])
def test_linear_attention_block(dtype, c, H, W):
    """RMSNorm -> qkv -> k softmax over n, ctx, fold, to_out, RMSNorm, + x  vs the oracle block."""
    B, hid, n = 2, 128, H * W
    x = _q(hh.rand((B, c, H, W), 70, -2, 2), dtype)
    sd = {"a.norm.g": hh.rand((1, c, 1, 1), 71, 0.5, 1.5),
          "a.to_qkv.weight": _q(hh.rand((3 * hid, c, 1, 1), 72, -0.3, 0.3), dtype),
          "a.to_out.0.weight": hh.rand((c, hid, 1, 1), 73, -0.3, 0.3),
          "a.to_out.0.bias": hh.rand((c,), 74),
          "a.to_out.1.g": hh.rand((1, c, 1, 1), 75, 0.5, 1.5)}
    ref = unet_ref.linear_attention(sd, "a", x) + x
    lib, dt = cabi.lib(), cabi.dtype_code(dtype)
    xd = hh.nhwc(x, dtype)
    wq = hh.pack(sd["a.to_qkv.weight"], dtype, 1, scale_in=sd["a.norm.g"].flatten() * math.sqrt(c))
    kmax = torch.zeros(B, cabi.STAT_STRIPES, hid, dtype=torch.int32, device=hh.DEV)   # fused into the qkv epilogue
    qkv = hh.conv1x1([hh.make_src(xd, c)], wq, B, H, W, 3 * hid, dtype, epi=cabi.EPI_QKV_LINEAR, rms_in=1, kmax_out=kmax)
    kmax2 = torch.zeros(B, cabi.STAT_STRIPES, hid, dtype=torch.int32, device=hh.DEV)  # stand-alone kernel, same result
    cabi.check(lib.ld_linattn_kmax(qkv.data_ptr(), kmax2.data_ptr(), B, n, 4, 32, dt, hh.st()), "kmax")
    nchunks = max(1, min(32, n // 256))
    ctxn = torch.empty(B, 4, 32, 32, device=hh.DEV)
    ctx = torch.empty(int(lib.ld_linattn_ctx_part_floats(B, 4, 32, nchunks)), device=hh.DEV)
    wfold = torch.empty(B, c * hid, dtype=hh.TDT[dtype], device=hh.DEV)
    wout = sd["a.to_out.0.weight"].reshape(c, hid).contiguous().to(hh.DEV)
    cabi.check(lib.ld_linattn_ctx(qkv.data_ptr(), kmax.data_ptr(), ctx.data_ptr(), B, n, 4, 32, nchunks, dt, hh.st()), "ctx")
    cabi.check(lib.ld_linattn_ctx_reduce(ctx.data_ptr(), nchunks, ctxn.data_ptr(), B, 4, 32, hh.st()), "reduce")
    cabi.check(lib.ld_linattn_fold(ctxn.data_ptr(), wout.data_ptr(), wfold.data_ptr(), B, c, 4, 32, 0, dt, hh.st()), "fold")
    wfold2 = torch.zeros_like(wfold)                  # reduce + fold in one launch: same packed weights
    cabi.check(lib.ld_linattn_ctxfold(ctx.data_ptr(), nchunks, wout.data_ptr(), wfold2.data_ptr(), B, c, 4, 32, 0, dt, hh.st()), "ctxfold")
    assert hh.rel_err(wfold2.float().cpu(), wfold.float().cpu()) < (1e-5 if dtype == "fp32" else 1e-2)
    # intermediate check: normalised context = softmax_n(k) . v^T
    q_, k_, v_ = [t.reshape(B, 4, 32, n) for t in F.conv2d(unet_ref.rms_norm(x, sd["a.norm.g"]), sd["a.to_qkv.weight"]).chunk(3, dim=1)]
    cref = torch.einsum("bhdn,bhen->bhde", k_.softmax(dim=-1), v_)
    assert hh.rel_err(ctxn.cpu(), cref) < hh.RTOL[dtype] * 3
    # intermediate check: kmax
    kref = F.conv2d(unet_ref.rms_norm(x, sd["a.norm.g"]), sd["a.to_qkv.weight"]).chunk(3, dim=1)[1].reshape(B, hid, n).amax(-1)
    assert hh.rel_err(hh.dec_max(kmax).amax(1), kref) < hh.RTOL[dtype] * 3
    assert hh.rel_err(hh.dec_max(kmax2).amax(1), kref) < hh.RTOL[dtype] * 3
    es = 4 if dtype == "fp32" else 2
    out = hh.conv1x1([hh.make_src(qkv, hid, stride=3 * hid)], wfold, B, H, W, c, dtype, bias=sd["a.to_out.0.bias"].to(hh.DEV),
                     epi=cabi.EPI_RMS_RES, bstride=c * hid * es, g2=(sd["a.to_out.1.g"].flatten() * math.sqrt(c)).to(hh.DEV),
                     residual=xd)
    assert hh.rel_err(hh.nchw(out), ref) < hh.RTOL[dtype] * (5 if dtype == "fp32" else 2)


# ------------------------------------------------------------------------------ time embedding
def test_time_mlp_and_film():
    dim, td, n, two_c = 32, 128, 7, 64
    sd = {"time_mlp.1.weight": hh.rand((td, dim), 80, -0.2, 0.2), "time_mlp.1.bias": hh.rand((td,), 81),
          "time_mlp.3.weight": hh.rand((td, td), 82, -0.1, 0.1), "time_mlp.3.bias": hh.rand((td,), 83)}
    times = torch.tensor([0, 1, 5, 99, 500, 998, 999])
    ref = unet_ref.time_embedding(sd, times, dim)
    half = dim // 2
    freqs = torch.exp(torch.arange(half) * -(math.log(10000.0) / (half - 1))).to(hh.DEV)
    td_ = {k: v.to(hh.DEV) for k, v in sd.items()}
    temb = torch.empty(n, td, device=hh.DEV)
    tdev = times.to(hh.DEV, torch.int32)
    lib = cabi.lib()
    cabi.check(lib.ld_time_mlp(tdev.data_ptr(), n, freqs.data_ptr(), dim, td_["time_mlp.1.weight"].data_ptr(),
                               td_["time_mlp.1.bias"].data_ptr(), td_["time_mlp.3.weight"].data_ptr(),
                               td_["time_mlp.3.bias"].data_ptr(), td, temb.data_ptr(), hh.st()), "time_mlp")
    assert hh.rel_err(temb.cpu(), ref) < 2e-5
    w, b = hh.rand((two_c, td), 84, -0.1, 0.1), hh.rand((two_c,), 85)
    fref = F.linear(F.silu(ref), w, b)
    film = torch.empty(n, two_c, device=hh.DEV)
    wd, bd = w.to(hh.DEV), b.to(hh.DEV)
    cabi.check(lib.ld_film(temb.data_ptr(), n, td, wd.data_ptr(), bd.data_ptr(), two_c, film.data_ptr(), hh.st()), "film")
    assert hh.rel_err(film.cpu(), fref) < 2e-5


# ------------------------------------------------------------------------------ pointwise / sampler kernels
def test_randn_matches_host_stream():
    n = 1 << 16
    out = torch.empty(n, device=hh.DEV)
    tdev = torch.tensor([7], dtype=torch.int32, device=hh.DEV)
    cabi.check(cabi.lib().ld_randn(out.data_ptr(), n, 10, 3, 0, None, hh.st()), "randn")
    ref = torch.from_numpy(rng.randn((n,), 10, 3))
    assert float((out.cpu() - ref).abs().max()) < 2e-5           # fp32 Box-Muller vs fp64 on the host
    cabi.check(cabi.lib().ld_randn(out.data_ptr(), n, 10, 100, -1, tdev.data_ptr(), hh.st()), "randn")
    assert float((out.cpu() - torch.from_numpy(rng.randn((n,), 10, 93))).abs().max()) < 2e-5


def test_randn_at_is_a_slice_of_the_draw():
    """ld_randn_at(first, n) == ld_randn[first : first + n] bit for bit (sub-batches draw their part of a batch's noise)."""
    n, first, m = 1 << 14, 5000, 3000
    full, part = torch.empty(n, device=hh.DEV), torch.empty(m, device=hh.DEV)
    tdev = torch.tensor([7], dtype=torch.int32, device=hh.DEV)
    cabi.check(cabi.lib().ld_randn(full.data_ptr(), n, 10, 100, -1, tdev.data_ptr(), hh.st()), "randn")
    cabi.check(cabi.lib().ld_randn_at(part.data_ptr(), m, first, 10, 100, -1, tdev.data_ptr(), hh.st()), "randn_at")
    assert torch.equal(part, full[first:first + m])


@pytest.mark.parametrize("dtype", DTYPES)
def test_final_step_at_matches_the_batch_call(dtype):
    """Two ld_final_step_at calls on the halves of a batch (noise index from the device step counter, element
    offset of the half) == one ld_final_step on the whole batch, bit for bit; t == 0 adds no noise."""
    B, cin, cout, H, W, T = 4, 32, 3, 16, 16, 20
    lib = cabi.lib()
    x = hh.nhwc(_q(hh.rand((B, cin, H, W), 110), dtype), dtype)
    w, b = hh.rand((cout, cin), 111, -0.3, 0.3).to(hh.DEV), hh.rand((cout,), 112).to(hh.DEV)
    sched = hh.rand((T, 8), 113, 0.1, 0.9).to(hh.DEV)
    xt = hh.rand((B, cout, H, W), 114, -1, 1).to(hh.DEV)
    for t, base in ((7, 13 + 7), (0, 20)):
        tdev = torch.tensor([t], dtype=torch.int32, device=hh.DEV)
        mo1, x1, x01 = torch.empty_like(xt), xt.clone(), torch.empty_like(xt)
        cabi.check(lib.ld_final_step(x.data_ptr(), w.data_ptr(), b.data_ptr(), mo1.data_ptr(), x1.data_ptr(), x01.data_ptr(),
                                     sched.data_ptr(), tdev.data_ptr(), 0.0, 2.0, 0, 10, base - t if t > 0 else 0,
                                     B, H, W, cin, cout, cabi.dtype_code(dtype), hh.st()), "final_step")
        mo2, x2, x02 = torch.empty_like(xt), xt.clone(), torch.empty_like(xt)
        h = B // 2
        for i in range(2):
            cabi.check(lib.ld_final_step_at(x[i * h:].data_ptr(), w.data_ptr(), b.data_ptr(), mo2[i * h:].data_ptr(),
                                            x2[i * h:].data_ptr(), x02[i * h:].data_ptr(), sched.data_ptr(), tdev.data_ptr(),
                                            0.0, 2.0, 0, 10, base, -1, i * h * cout * H * W, None,
                                            h, H, W, cin, cout, cabi.dtype_code(dtype), hh.st()), "final_step_at")
        assert torch.equal(mo1, mo2) and torch.equal(x1, x2) and torch.equal(x01, x02)
        assert bool(torch.isfinite(x1).all())


@pytest.mark.parametrize("objective", ["pred_x0", "pred_noise", "pred_v"])
def test_ddpm_step(objective):
    T, shape = 50, (2, 3, 8, 8)
    buf = schedule.make_buffers(T, "sigmoid", objective)
    table = torch.stack([buf["posterior_mean_coef1"], buf["posterior_mean_coef2"],
                         (0.5 * buf["posterior_log_variance_clipped"]).exp(), buf["sqrt_recip_alphas_cumprod"],
                         buf["sqrt_recipm1_alphas_cumprod"], buf["sqrt_alphas_cumprod"],
                         buf["sqrt_one_minus_alphas_cumprod"], buf["alphas_cumprod"]], 1).contiguous().to(hh.DEV)
    x, mo, z = hh.rand(shape, 90, -2, 2), hh.rand(shape, 91, -1, 3), hh.rand(shape, 92, -2, 2)
    for t in (0, 1, 17, 49):
        if objective == "pred_x0":
            x0 = mo
        elif objective == "pred_noise":
            x0 = buf["sqrt_recip_alphas_cumprod"][t] * x - buf["sqrt_recipm1_alphas_cumprod"][t] * mo
        else:
            x0 = buf["sqrt_alphas_cumprod"][t] * x - buf["sqrt_one_minus_alphas_cumprod"][t] * mo
        x0 = x0.clamp(0.0, 2.0)
        ref = buf["posterior_mean_coef1"][t] * x0 + buf["posterior_mean_coef2"][t] * x
        if t > 0:
            ref = ref + (0.5 * buf["posterior_log_variance_clipped"][t]).exp() * z
        xd, md, zd = x.to(hh.DEV), mo.to(hh.DEV), z.to(hh.DEV)
        out, x0o = torch.empty_like(xd), torch.empty_like(xd)
        tdev = torch.tensor([t], dtype=torch.int32, device=hh.DEV)
        cabi.check(cabi.lib().ld_ddpm_step(xd.data_ptr(), md.data_ptr(), zd.data_ptr(), out.data_ptr(), x0o.data_ptr(),
                                           table.data_ptr(), tdev.data_ptr(), 0.0, 2.0, cabi.OBJ[objective], xd.numel(),
                                           hh.st()), "ddpm_step")
        assert float((out.cpu() - ref).abs().max()) < 1e-6
        assert float((x0o.cpu() - x0).abs().max()) < 1e-6


def test_branch_and_fusion_kernels():
    B, Cc, H = 2, 3, 8
    HW = H * H
    lib = cabi.lib()
    cond, mask = hh.rand((B, Cc, H, H), 93, 0, 2), torch.zeros(B, 1, H, H)
    mask[:, :, :, :3] = 1.0
    mask[:, :, 0, 5] = 0.7
    binary = (mask >= 1).float()
    cd, md = cond.to(hh.DEV), mask.to(hh.DEV)
    co, ci = torch.empty_like(cd), torch.empty_like(cd)
    cabi.check(lib.ld_branch_conditions(cd.data_ptr(), md.data_ptr(), co.data_ptr(), ci.data_ptr(), 0.95, B, Cc, HW, hh.st()), "bc")
    assert torch.equal(co.cpu(), cond * binary) and torch.equal(ci.cpu(), cond * torch.clip(1 - binary, 0.95, 1.0))
    mo = hh.rand((B, Cc, H, H), 94, -1, 3)
    mod = mo.to(hh.DEV)
    cabi.check(lib.ld_mask_out(mod.data_ptr(), md.data_ptr(), 0.25, B, Cc, HW, hh.st()), "mask_out")
    assert torch.equal(mod.cpu(), torch.where(binary == 0, torch.tensor(0.25), mo * binary))
    xo, xi, p0o, p0i = [hh.rand((B, Cc, H, H), k, -1, 3) for k in (95, 96, 97, 98)]
    x0 = (p0i.clamp(0, 2) * (1 - binary) + p0o.clamp(0, 2)).clamp(0, 2)
    a, b_ = xo * binary, xi * (1 - binary)
    xref = torch.where(a == 0, b_, a)
    d = [t.to(hh.DEV) for t in (xo, xi, p0o, p0i)]
    xd, x0d = torch.empty_like(d[0]), torch.empty_like(d[0])
    cabi.check(lib.ld_fuse_ddpm(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), md.data_ptr(),
                                xd.data_ptr(), x0d.data_ptr(), 0.0, 2.0, B, Cc, HW, hh.st()), "fuse")
    assert torch.equal(xd.cpu(), xref) and float((x0d.cpu() - x0).abs().max()) < 1e-6
    # K-mask recomposition
    K = 4
    masks = torch.zeros(K, HW)
    for k in range(K):
        masks[k].view(H, H)[:, 2 * k:2 * k + 2] = 1.0
    patches = hh.rand((B, K, Cc, HW), 99)
    ref = (patches * masks[None, :, None, :]).sum(1).reshape(B, Cc, H, H)
    pd, mk, od = patches.to(hh.DEV), masks.to(hh.DEV), torch.empty(B, Cc, H, H, device=hh.DEV)
    cabi.check(lib.ld_recompose(pd.data_ptr(), mk.data_ptr(), od.data_ptr(), B, K, Cc, HW, hh.st()), "recompose")
    assert float((od.cpu() - ref).abs().max()) < 1e-6


@pytest.mark.parametrize("dtype", DTYPES)
def test_final_conv(dtype):
    B, cin, cout, H, W = 2, 32, 3, 9, 13
    x, w, b = _q(hh.rand((B, cin, H, W), 100), dtype), hh.rand((cout, cin, 1, 1), 101, -0.3, 0.3), hh.rand((cout,), 102)
    ref = F.conv2d(x, w, b)
    out = torch.empty(B, cout, H, W, device=hh.DEV)
    xd, wd, bd = hh.nhwc(x, dtype), w.reshape(cout, cin).contiguous().to(hh.DEV), b.to(hh.DEV)
    cabi.check(cabi.lib().ld_final_conv(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(), B, H, W, cin, cout,
                                        cabi.dtype_code(dtype), hh.st()), "final_conv")
    assert hh.rel_err(out.cpu(), ref) < 1e-5


@pytest.mark.parametrize("dtype", LOWP)
@pytest.mark.parametrize("single_sweep", [False, True])
@pytest.mark.parametrize("c,H,W", [(32, 28, 28), (64, 14, 14), (32, 64, 96), (128, 32, 32)])
def test_linear_attention_fused_16bit(dtype, c, H, W, single_sweep):
    """The fused 16-bit path (q/k/v recomputed from x, never stored) against the oracle block; with the exact
    two-sweep k maximum and with the single-sweep Cauchy-Schwarz shift + the fused reduce/fold launch."""
    f = 1.0 if dtype == "bf16" else 0.25                    # fp16 carries 3 more mantissa bits; fp32 parts (Z, softmax) do not shrink
    B, hid, n = 2, 128, H * W
    x = _q(hh.rand((B, c, H, W), 170, -2, 2), dtype)
    sd = {"a.norm.g": hh.rand((1, c, 1, 1), 171, 0.5, 1.5),
          "a.to_qkv.weight": _q(hh.rand((3 * hid, c, 1, 1), 172, -0.3, 0.3), dtype),
          "a.to_out.0.weight": hh.rand((c, hid, 1, 1), 173, -0.3, 0.3),
          "a.to_out.0.bias": hh.rand((c,), 174),
          "a.to_out.1.g": hh.rand((1, c, 1, 1), 175, 0.5, 1.5)}
    ref = unet_ref.linear_attention(sd, "a", x) + x
    lib, dt = cabi.lib(), cabi.dtype_code(dtype)
    xd = hh.nhwc(x, dtype)
    scale = sd["a.norm.g"].flatten() * math.sqrt(c)
    w = sd["a.to_qkv.weight"]
    wq = hh.pack(w[:hid].contiguous(), dtype, 1, scale_in=scale)
    wkv = torch.cat([hh.pack(torch.cat([w[hid + 32 * h: hid + 32 * h + 32], w[2 * hid + 32 * h: 2 * hid + 32 * h + 32]], 0).contiguous(),
                             dtype, 1, scale_in=scale) for h in range(4)]).contiguous()
    nchunks = max(1, min(32, n // 256))
    ctx = torch.empty(int(lib.ld_linattn_ctx_part_floats(B, 4, 32, nchunks)), device=hh.DEV)
    ctxn = torch.empty(B, 4, 32, 32, device=hh.DEV)
    wfold = torch.empty(B, c * hid, dtype=hh.TDT[dtype], device=hh.DEV)
    wout = sd["a.to_out.0.weight"].reshape(c, hid).contiguous().to(hh.DEV)
    kshift = None
    if single_sweep:   # |k_d| <= ||W_k[d] * g * sqrt(C)||_2 because the RMS-normalised pixel has unit 2-norm
        kshift = (w[hid:2 * hid, :, 0, 0] * scale[None, :]).norm(dim=1).contiguous().to(hh.DEV)
        nchunks = max(1, min(128, n // 512))
        ctx = torch.empty(int(lib.ld_linattn_ctx_part_floats(B, 4, 32, nchunks)), device=hh.DEV)
    cabi.check(lib.ld_linattn_kvctx(xd.data_ptr(), wkv.data_ptr(), None if kshift is None else kshift.data_ptr(), ctx.data_ptr(),
                                    B, n, c, 4, 32, nchunks, dt, hh.st()), "kvctx")
    cabi.check(lib.ld_linattn_ctx_reduce(ctx.data_ptr(), nchunks, ctxn.data_ptr(), B, 4, 32, hh.st()), "reduce")
    q_, k_, v_ = [t.reshape(B, 4, 32, n) for t in F.conv2d(unet_ref.rms_norm(x, sd["a.norm.g"]), w).chunk(3, dim=1)]
    if single_sweep:
        assert float((k_.reshape(B, hid, n).amax(-1) - kshift.cpu()[None]).max()) <= 1e-3      # the bound holds
    cref = torch.einsum("bhdn,bhen->bhde", k_.softmax(dim=-1), v_)
    assert hh.rel_err(ctxn.cpu(), cref) < 2e-2 * f
    cabi.check(lib.ld_linattn_fold(ctxn.data_ptr(), wout.data_ptr(), wfold.data_ptr(), B, c, 4, 32, 1, dt, hh.st()), "fold")
    if single_sweep:   # one launch for reduce + fold: identical packed M_b
        wfold2 = torch.zeros_like(wfold)
        cabi.check(lib.ld_linattn_ctxfold(ctx.data_ptr(), nchunks, wout.data_ptr(), wfold2.data_ptr(), B, c, 4, 32, 1, dt, hh.st()), "ctxfold")
        assert hh.rel_err(wfold2.float().cpu(), wfold.float().cpu()) < 1e-2 * f
        wfold = wfold2
    out = torch.empty(B, H, W, c, dtype=hh.TDT[dtype], device=hh.DEV)
    bias, g2 = sd["a.to_out.0.bias"].to(hh.DEV), (sd["a.to_out.1.g"].flatten() * math.sqrt(c)).to(hh.DEV)
    qshift = None
    if single_sweep:
        qshift = (w[:hid, :, 0, 0] * scale[None, :]).norm(dim=1).reshape(4, 32).amax(dim=1).contiguous().to(hh.DEV)
    cabi.check(lib.ld_linattn_out(xd.data_ptr(), wq.data_ptr(), None if qshift is None else qshift.data_ptr(), wfold.data_ptr(),
                                  bias.data_ptr(), g2.data_ptr(), out.data_ptr(), B, n, c, 32 ** -0.5, dt, hh.st()), "linattn_out")
    assert hh.rel_err(hh.nchw(out), ref) < 6e-2 * f


# ------------------------------------------------------------------------------ persistent C=32 conv (conv3x3_c32.hip)
@pytest.mark.parametrize("dtype", DTYPES)
def test_conv3x3_c32_persistent_path(dtype):
    """32 -> 32 at 256^2 (>= 2048 tiles, one K-chunk, H and W multiples of 16) takes the persistent deep-ring
    LDS-DMA kernel: plain + statistics, GroupNorm+FiLM+SiLU prologue; the 64 -> 32 concat / nearest-x2 cases
    (two K-chunks, also a ragged width) take the register-staged kernel at the same size."""
    B, H, W, cout = 8, 256, 256, 32
    cin = 32
    x, w, b = _q(hh.rand((B, cin, H, W), 201), dtype), _q(hh.rand((cout, cin, 3, 3), 202, -0.1, 0.1), dtype), hh.rand((cout,), 203)
    ref = F.conv2d(x, w, b, padding=1)
    stats = hh.stats_buffer(B, 8)
    out = hh.conv3x3([hh.make_src(hh.nhwc(x, dtype), cin)], hh.pack(w, dtype, 3), b.to(hh.DEV), B, H, W, cout, dtype,
                     stats=stats, groups=8)
    assert hh.rel_err(hh.nchw(out), ref) < hh.RTOL[dtype]
    assert hh.rel_err(stats.sum(1).cpu(), hh.gn_stats_ref(ref, 8)) < (1e-5 if dtype == "fp32" else 1e-2)
    # GroupNorm + FiLM + SiLU prologue (statistics of x itself)
    gamma, beta = hh.rand((cin,), 204, 0.5, 1.5), hh.rand((cin,), 205, -0.3, 0.3)
    film = hh.rand((B, 2 * cin), 206, -0.5, 0.5)
    y = F.silu(F.group_norm(x, 8, gamma, beta, eps=1e-5) * (film[:, :cin, None, None] + 1) + film[:, cin:, None, None])
    ref = F.conv2d(y, w, b, padding=1)
    g_d, b_d, f_d = gamma.to(hh.DEV), beta.to(hh.DEV), film.to(hh.DEV)
    src = hh.make_src(hh.nhwc(x, dtype), cin, gn=(hh.stats_striped(x, 8), g_d, b_d, 8), act=cabi.ACT_SILU, film=f_d, film_b=2 * cin)
    out = hh.conv3x3([src], hh.pack(w, dtype, 3), b.to(hh.DEV), B, H, W, cout, dtype)
    assert hh.rel_err(hh.nchw(out), ref) < hh.RTOL[dtype] * 2
    # 64 -> 32 in two K-chunks: upsampled 32 ch ++ 32 ch, statistics + prologue on the second source, on a
    # multiple-of-16 width and a ragged one
    for Wr in ((256, 250) if dtype != "fp32" else ()):
        x1, x2 = _q(hh.rand((B, 32, H // 2, Wr // 2), 207), dtype), _q(hh.rand((B, 32, H, Wr), 208), dtype)
        w2 = _q(hh.rand((cout, 64, 3, 3), 209, -0.1, 0.1), dtype)
        y2 = F.silu(F.group_norm(x2, 8, gamma, beta, eps=1e-5))
        ref = F.conv2d(torch.cat([F.interpolate(x1, scale_factor=2, mode="nearest"), y2], 1), w2, b, padding=1)
        stats = hh.stats_buffer(B, 8)
        src2 = hh.make_src(hh.nhwc(x2, dtype), 32, gn=(hh.stats_striped(x2, 8), g_d, b_d, 8), act=cabi.ACT_SILU)
        out = hh.conv3x3([hh.make_src(hh.nhwc(x1, dtype), 32, ups=1), src2],
                         hh.pack(w2, dtype, 3), b.to(hh.DEV), B, H, Wr, cout, dtype, stats=stats, groups=8)
        assert hh.rel_err(hh.nchw(out), ref) < hh.RTOL[dtype] * 2
        assert hh.rel_err(stats.sum(1).cpu(), hh.gn_stats_ref(ref, 8)) < 1e-2


def test_conv3x3_split_k_halves_variant():
    """The opt-in 512-thread variant (LD_CONV_SK=2: two halves of a workgroup own alternate K-chunks in opposite
    phase, partial sums joined through LDS) against F.conv2d: plain, with statistics, with the GroupNorm prologue,
    ragged size, odd chunk count.  The switch is read once per process, hence the child process."""
    import os, subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, torch, torch.nn.functional as F
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import hip_helpers as hh
        from localdiffusion_hallucination_amd import _cabi as cabi
        dtype = "bf16"
        q = lambda t: t.to(torch.bfloat16).float()
        for (B, cin, cout, H, W) in [(2, 256, 64, 16, 16), (1, 96, 32, 13, 18), (2, 64, 64, 8, 24)]:
            x, w, b = q(hh.rand((B, cin, H, W), 301)), q(hh.rand((cout, cin, 3, 3), 302, -0.1, 0.1)), hh.rand((cout,), 303)
            ref = F.conv2d(x, w, b, padding=1)
            stats = hh.stats_buffer(B, 8)
            out = hh.conv3x3([hh.make_src(hh.nhwc(x, dtype), cin)], hh.pack(w, dtype, 3), b.to(hh.DEV), B, H, W, cout, dtype,
                             stats=stats, groups=8)
            assert hh.rel_err(hh.nchw(out), ref) < hh.RTOL[dtype], ("plain", cin, cout)
            assert hh.rel_err(stats.sum(1).cpu(), hh.gn_stats_ref(ref, 8)) < 1e-2
            gamma, beta = hh.rand((cin,), 304, 0.5, 1.5), hh.rand((cin,), 305, -0.3, 0.3)
            y = F.silu(F.group_norm(x, 8, gamma, beta, eps=1e-5))
            ref = F.conv2d(y, w, b, padding=1)
            src = hh.make_src(hh.nhwc(x, dtype), cin, gn=(hh.stats_striped(x, 8), gamma.to(hh.DEV), beta.to(hh.DEV), 8), act=cabi.ACT_SILU)
            out = hh.conv3x3([src], hh.pack(w, dtype, 3), b.to(hh.DEV), B, H, W, cout, dtype)
            assert hh.rel_err(hh.nchw(out), ref) < hh.RTOL[dtype] * 2, ("prologue", cin, cout)
        print("SK-OK")
    """) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LD_CONV_SK="2")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "SK-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv1x1_gn_tail_epilogue(dtype):
    """ResnetBlock tail fused into res_conv: out = conv1x1(cat(x1,x2)) + SiLU(GN(raw2))."""
    B, c1, c2, cout, H, W = 2, 64, 32, 64, 12, 10
    x1, x2 = _q(hh.rand((B, c1, H, W), 223), dtype), _q(hh.rand((B, c2, H, W), 224), dtype)
    raw2 = _q(hh.rand((B, cout, H, W), 225, -2, 3), dtype)
    w, b = _q(hh.rand((cout, c1 + c2, 1, 1), 226, -0.2, 0.2), dtype), hh.rand((cout,), 227)
    gamma, beta = hh.rand((cout,), 228, 0.5, 1.5), hh.rand((cout,), 229, -0.3, 0.3)
    ref = F.conv2d(torch.cat([x1, x2], 1), w, b) + F.silu(F.group_norm(raw2, 8, gamma, beta))
    gd, bd = gamma.to(hh.DEV), beta.to(hh.DEV)
    tail = hh.make_src(hh.nhwc(raw2, dtype), cout, gn=(hh.stats_striped(raw2, 8), gd, bd, 8), act=cabi.ACT_SILU)
    out = hh.conv1x1([hh.make_src(hh.nhwc(x1, dtype), c1), hh.make_src(hh.nhwc(x2, dtype), c2)], hh.pack(w, dtype, 1),
                     B, H, W, cout, dtype, bias=b.to(hh.DEV), epi=cabi.EPI_GN_TAIL, gn_tail=tail)
    assert hh.rel_err(hh.nchw(out), ref) < hh.RTOL[dtype] * 2


def _c1_group_probe(path):
    """Child process of test_conv1x1_grouped_staging_is_bitwise_the_plain_k_loop: every epilogue of ld_conv1x1 at K
    extents that exercise full groups and the tail group (nch = 2, 3, 5, 13 chunks of 32 channels), results to ``path``."""
    out = {}
    B, H, W = 2, 16, 16
    for nch in (2, 3, 5, 13):
        cin = 32 * nch
        for dtype in ("bf16", "fp32"):
            x = hh.rand((B, cin, H, W), 900 + nch)
            xs = hh.make_src(hh.nhwc(x, dtype), cin)
            for epi, cout in ((cabi.EPI_PLAIN, 64), (cabi.EPI_RES, 64), (cabi.EPI_QKV_FULL, 384), (cabi.EPI_QKV_LINEAR, 384),
                              (cabi.EPI_RMS_RES, 64), (cabi.EPI_GN_TAIL, 64)):
                w = hh.rand((cout, cin, 1, 1), 910 + nch + epi, -0.2, 0.2)
                kw = {}
                if epi in (cabi.EPI_RES, cabi.EPI_RMS_RES):
                    kw["residual"] = hh.nhwc(hh.rand((B, cout, H, W), 920 + nch), dtype)
                if epi == cabi.EPI_RMS_RES:
                    kw["g2"] = hh.rand((cout,), 921, 0.5, 1.5).to(hh.DEV)
                if epi in (cabi.EPI_QKV_FULL, cabi.EPI_QKV_LINEAR):
                    kw["rms_in"] = 1
                if epi == cabi.EPI_QKV_LINEAR:
                    kw["kmax_out"] = torch.zeros(B, cabi.STAT_STRIPES, 128, dtype=torch.int32, device=hh.DEV)
                if epi == cabi.EPI_GN_TAIL:
                    h = hh.rand((B, cout, H, W), 930 + nch)
                    kw["gn_tail"] = hh.make_src(hh.nhwc(h, dtype), cout, gn=(hh.stats_striped(h, 8), hh.rand((cout,), 931, 0.5, 1.5).to(hh.DEV),
                                                                      hh.rand((cout,), 932).to(hh.DEV), 8), act=cabi.ACT_SILU)
                bias = None if epi in (cabi.EPI_QKV_FULL, cabi.EPI_QKV_LINEAR) else hh.rand((cout,), 940).to(hh.DEV)
                y = hh.conv1x1([xs], hh.pack(w, dtype, 1), B, H, W, cout, dtype, bias=bias, epi=epi, **kw)
                out[f"{dtype}_{nch}_{epi}"] = y.float().cpu().numpy()
    torch.cuda.synchronize()
    np.savez(path, **out)


@pytest.mark.gpu
def test_conv1x1_grouped_staging_is_bitwise_the_plain_k_loop(tmp_path):
    """the grouped K staging (four chunks per barrier pair, pairs on mid-size maps) takes most small-map
    launches by default, the plain loop only the large ones.  The same calls in two processes, LD_C1_GROUP=1 and =0, for
    every epilogue and for K extents with a partial last group: the arithmetic order is unchanged, so bit-equal."""
    import subprocess
    import sys
    res = {}
    for flag in ("1", "0"):
        path = str(tmp_path / f"c1_{flag}.npz")
        code = (f"import sys; sys.path.insert(0, {os.path.dirname(os.path.abspath(__file__))!r}); "
                f"sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r}); "
                f"import test_hip_ops as t; t._c1_group_probe({path!r})")
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LD_C1_GROUP=flag), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        res[flag] = np.load(path)
    assert len(res["1"].files) == 4 * 2 * 6
    for k in res["1"].files:
        assert np.isfinite(res["1"][k]).all(), k
        assert np.array_equal(res["1"][k], res["0"][k]), (k, float(np.abs(res["1"][k] - res["0"][k]).max()))


def _c32_probe(path):
    """Child process of test_persistent_conv_geometries_agree: Cout = 32 single-chunk 3x3 convolutions (plain, with a
    GroupNorm + SiLU prologue, with output statistics), results to ``path``."""
    out = {}
    for dtype in ("bf16", "fp16"):
        for (B, H, W) in ((2, 32, 48), (3, 64, 64)):
            x = hh.rand((B, 32, H, W), 700 + H)
            w, b = hh.rand((32, 32, 3, 3), 701, -0.1, 0.1), hh.rand((32,), 702).to(hh.DEV)
            wp = hh.pack(w, dtype, 3)
            gn = (hh.stats_striped(x, 8), hh.rand((32,), 703, 0.5, 1.5).to(hh.DEV), hh.rand((32,), 704).to(hh.DEV), 8)
            for name, src in (("plain", hh.make_src(hh.nhwc(x, dtype), 32)), ("pro", hh.make_src(hh.nhwc(x, dtype), 32, gn=gn, act=cabi.ACT_SILU))):
                st = hh.stats_buffer(B, 8)
                y = hh.conv3x3([src], wp, b, B, H, W, 32, dtype, stats=st, groups=8)
                out[f"{dtype}_{H}_{name}"] = y.float().cpu().numpy()
                out[f"{dtype}_{H}_{name}_stats"] = st.sum(1).cpu().numpy()
    torch.cuda.synchronize()
    np.savez(path, **out)


@pytest.mark.gpu
def test_persistent_conv_geometries_agree(tmp_path):
    """The C = 32 convolution has two code paths: the generic kernel (tuning table: conv_c32 = 0) and the persistent ring
    kernel (512 threads, 16 x 16 tiles; conv_c32_min_tiles = 1 puts every eligible launch on it).  Same MFMA order in
    both: outputs bit-equal, statistics equal up to the length of their fp32 partial sums (a persistent workgroup keeps
    its sums in registers over all its tiles).  The routing is switched through ld_tuning_set, the launch counters
    confirm which kernel ran."""
    from localdiffusion_hallucination_amd.tuning import kernel_table
    lib = cabi.lib()
    keep = kernel_table(lib)
    res = {}
    try:
        for name, sets in (("generic", {"conv_c32": 0}), ("ring", {"conv_c32": 1, "conv_c32_min_tiles": 1})):
            for k, v in sets.items():
                cabi.check(lib.ld_tuning_set(k.encode(), v), "tuning_set")
            c0 = lib.ld_counter(cabi.COUNTER_CONV3X3_C32)
            path = str(tmp_path / f"c32_{name}.npz")
            _c32_probe(path)
            ran = lib.ld_counter(cabi.COUNTER_CONV3X3_C32) - c0
            assert (ran > 0) == (name == "ring"), (name, ran)
            res[name] = np.load(path)
    finally:
        for k in ("conv_c32", "conv_c32_min_tiles"):
            lib.ld_tuning_set(k.encode(), keep[k])
    assert len(res["generic"].files) == 2 * 2 * 2 * 2
    for k in res["generic"].files:
        a, b = res["generic"][k], res["ring"][k]
        assert np.isfinite(b).all(), k
        if k.endswith("_stats"):
            assert float(np.abs(a - b).max()) <= 2e-6 * float(np.abs(a).max()), (k, float(np.abs(a - b).max() / np.abs(a).max()))
        else:
            assert np.array_equal(a, b), (k, float(np.abs(a - b).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", LOWP)
def test_two_term_weights_remove_the_weight_rounding(dtype):
    """ld_pack_conv_weight_terms(..., 2) + weight_terms = 2 (W = hi + lo, x*hi + x*lo): against the fp32-weight reference on
    inputs that are exact in the storage dtype, what is left is the rounding of the OUTPUT to the storage dtype, i.e.
    the result equals the storage-rounded exact result almost everywhere -- and the one-term result is measurably
    further away.  3x3 (with concat + upsample + GroupNorm prologue) and 1x1 (concat; GN-tail epilogue)."""
    B, c1, c2, cout, H, W = 2, 64, 32, 64, 16, 16
    lib = cabi.lib()

    def packed(w, k, terms):
        w = w.to(hh.DEV, torch.float32).contiguous()
        out = torch.empty(terms * w.numel(), dtype=hh.TDT[dtype], device=hh.DEV)
        cabi.check(lib.ld_pack_conv_weight_terms(w.data_ptr(), None, out.data_ptr(), w.shape[0], w.shape[1], k, 0,
                                                 cabi.dtype_code(dtype), terms, hh.st()), "pack")
        return out
    x1, x2 = _q(hh.rand((B, c1, H // 2, W // 2), 4), dtype), _q(hh.rand((B, c2, H, W), 5), dtype)
    w, b = hh.rand((cout, c1 + c2, 3, 3), 6, -0.1, 0.1), hh.rand((cout,), 7)            # fp32 weights: NOT exact in 16 bits
    ref = F.conv2d(torch.cat([F.interpolate(x1, scale_factor=2, mode="nearest"), x2], 1), w, b, padding=1)
    errs = {}
    for terms in (1, 2):
        a = cabi.Conv3x3Args()
        s1, s2 = hh.make_src(hh.nhwc(x1, dtype), c1, ups=1), hh.make_src(hh.nhwc(x2, dtype), c2)
        a.src[0], a.src[1], a.nsrc = s1, s2, 2
        wp, bd = packed(w, 3, terms), b.to(hh.DEV)
        out = torch.empty(B, H, W, cout, dtype=hh.TDT[dtype], device=hh.DEV)
        a.weight, a.bias, a.out, a.weight_terms = wp.data_ptr(), bd.data_ptr(), out.data_ptr(), terms
        a.B, a.H, a.W, a.Cout, a.dtype = B, H, W, cout, cabi.dtype_code(dtype)
        cabi.check(lib.ld_conv3x3(C.byref(a), hh.st()), "conv3x3")
        errs[terms] = float((hh.nchw(out) - ref).abs().mean())
        if terms == 2:
            exact = (hh.nchw(out) == _q(ref, dtype)).float().mean()
            assert float(exact) > 0.97, float(exact)            # = the storage-rounded exact result (ties / fp32 summation order aside)
    print(f"conv3x3 {dtype}: mean-abs vs fp32-weight reference: one term {errs[1]:.3e}, two terms {errs[2]:.3e}")
    assert errs[2] < 0.8 * errs[1]
    # 1x1 over a concatenation with the ResnetBlock tail epilogue
    y1, y2 = _q(hh.rand((B, c1, H, W), 8), dtype), _q(hh.rand((B, c2, H, W), 9), dtype)
    w1, b1 = hh.rand((cout, c1 + c2, 1, 1), 10, -0.3, 0.3), hh.rand((cout,), 11)
    h = _q(hh.rand((B, cout, H, W), 12, -2, 2), dtype)
    gamma, beta = hh.rand((cout,), 13, 0.5, 1.5), hh.rand((cout,), 14, -0.3, 0.3)
    ref1 = F.conv2d(torch.cat([y1, y2], 1), w1, b1) + F.silu(F.group_norm(h, 8, gamma, beta, eps=1e-5))
    for terms in (1, 2):
        tail = hh.make_src(hh.nhwc(h, dtype), cout, gn=(hh.stats_striped(h, 8), gamma.to(hh.DEV), beta.to(hh.DEV), 8), act=cabi.ACT_SILU)
        a = cabi.Conv1x1Args()
        s1, s2 = hh.make_src(hh.nhwc(y1, dtype), c1), hh.make_src(hh.nhwc(y2, dtype), c2)
        a.src[0], a.src[1], a.nsrc = s1, s2, 2
        wp, bd = packed(w1, 1, terms), b1.to(hh.DEV)
        out = torch.empty(B, H, W, cout, dtype=hh.TDT[dtype], device=hh.DEV)
        a.weight, a.bias, a.out, a.weight_terms = wp.data_ptr(), bd.data_ptr(), out.data_ptr(), terms
        a.epilogue, a.hidden, a.q_scale, a.gn_tail = cabi.EPI_GN_TAIL, 128, 32 ** -0.5, tail
        a.B, a.H, a.W, a.Cout, a.dtype = B, H, W, cout, cabi.dtype_code(dtype)
        cabi.check(lib.ld_conv1x1(C.byref(a), hh.st()), "conv1x1")
        errs[terms] = float((hh.nchw(out) - ref1).abs().mean())
    print(f"conv1x1 {dtype}: mean-abs vs fp32-weight reference: one term {errs[1]:.3e}, two terms {errs[2]:.3e}")
    assert errs[2] < 0.8 * errs[1]

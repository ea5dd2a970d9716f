"""Thin wrappers that call the C-ABI ops directly on small tensors (GPU tests only)."""
import ctypes as C
import math

import numpy as np
import torch

from localdiffusion_hallucination_amd import _cabi as cabi
from localdiffusion_hallucination_amd import rng

DEV = "cuda"
TDT = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}
# max-abs tolerance relative to the reference tensor's max-abs, per storage dtype
RTOL = {"fp32": 2e-5, "bf16": 3e-2, "fp16": 4e-3}      # 16-bit: ~8 half-ulps (2^-8 / 2^-11) of the largest value


def st():
    return torch.cuda.current_stream().cuda_stream


def rand(shape, key, lo=-1.0, hi=1.0):
    return torch.from_numpy(rng.uniform(shape, 1234, key, lo, hi))


def nhwc(x, dtype):
    """NCHW fp32 (cpu) -> NHWC storage dtype on device."""
    return x.permute(0, 2, 3, 1).contiguous().to(DEV, TDT[dtype])


def nchw(x):
    """NHWC device tensor -> NCHW fp32 cpu."""
    return x.float().permute(0, 3, 1, 2).contiguous().cpu()


def rel_err(got, ref):
    return float((got - ref).abs().max()) / max(1e-12, float(ref.abs().max()))


def pack(w, dtype, ksize, scale_in=None, unshuffle=0):
    w = w.to(DEV, torch.float32).contiguous()
    out = torch.empty(w.numel(), dtype=TDT[dtype], device=DEV)
    si = None if scale_in is None else scale_in.to(DEV, torch.float32).contiguous()
    cabi.check(cabi.lib().ld_pack_conv_weight(w.data_ptr(), cabi.ptr(si), out.data_ptr(), w.shape[0], w.shape[1],
                                              ksize, unshuffle, cabi.dtype_code(dtype), st()), "pack")
    return out


def make_src(t, c, stride=0, ups=0, gn=None, act=0, film=None, film_b=0, film_t=0):
    s = cabi.Src()
    s.data, s.C, s.pix_stride, s.upsample = t.data_ptr(), c, stride, ups
    if gn is not None:
        stats, gamma, beta, groups = gn
        s.gn_stats, s.gn_gamma, s.gn_beta, s.gn_groups = stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), groups
    s.act = act
    if film is not None:
        s.film, s.film_bstride, s.film_tstride = film.data_ptr(), film_b, film_t
    s._keep = (t, gn, film)        # the struct only holds raw pointers: keep the tensors alive with it
    return s


def conv3x3(srcs, wpacked, bias, B, H, W, cout, dtype, stats=None, groups=8, t_ptr=None, side=None):
    """``side`` = (packed 1x1 weight, bias): returns (out, side_out)."""
    a = cabi.Conv3x3Args()
    for i, s in enumerate(srcs):
        a.src[i] = s
    a.nsrc = len(srcs)
    a.weight, a.bias = wpacked.data_ptr(), bias.data_ptr()
    out = torch.empty(B, H, W, cout, dtype=TDT[dtype], device=DEV)
    a.out = out.data_ptr()
    if stats is not None:
        a.out_stats, a.out_groups = stats.data_ptr(), groups
    a.B, a.H, a.W, a.Cout, a.dtype = B, H, W, cout, cabi.dtype_code(dtype)
    a.t_ptr = cabi.ptr(t_ptr)
    side_out = None
    if side is not None:
        side_out = torch.empty(B, H, W, cout, dtype=TDT[dtype], device=DEV)
        a.side_weight, a.side_bias, a.side_out = side[0].data_ptr(), side[1].data_ptr(), side_out.data_ptr()
    cabi.check(cabi.lib().ld_conv3x3(C.byref(a), st()), "conv3x3")
    return out if side is None else (out, side_out)


def conv1x1(srcs, wpacked, B, H, W, cout, dtype, bias=None, epi=0, unshuffle=0, rms_in=0, bstride=0, g2=None,
            residual=None, hidden=128, kmax_out=None, gn_tail=None):
    a = cabi.Conv1x1Args()
    for i, s in enumerate(srcs):
        a.src[i] = s
    a.nsrc, a.unshuffle, a.rms_in = len(srcs), unshuffle, rms_in
    a.weight, a.weight_bstride, a.bias = wpacked.data_ptr(), bstride, cabi.ptr(bias)
    a.epilogue, a.hidden, a.q_scale = epi, hidden, 32 ** -0.5
    a.g2, a.residual = cabi.ptr(g2), cabi.ptr(residual)
    a.kmax_out = cabi.ptr(kmax_out)
    if gn_tail is not None:
        a.gn_tail = gn_tail
    out = torch.empty(B, H, W, cout, dtype=TDT[dtype], device=DEV)
    a.out = out.data_ptr()
    a.B, a.H, a.W, a.Cout, a.dtype = B, H, W, cout, cabi.dtype_code(dtype)
    cabi.check(cabi.lib().ld_conv1x1(C.byref(a), st()), "conv1x1")
    return out


def gn_stats_ref(y, groups):
    """[B, groups, 2] fp64 (sum, sumsq) of an NCHW fp32 tensor."""
    B, Cc = y.shape[:2]
    g = y.double().reshape(B, groups, -1)
    return torch.stack([g.sum(-1), (g * g).sum(-1)], dim=-1)


def stats_buffer(B, groups):
    """Zeroed device statistics buffer in the kernels' striped layout [B, stripes, groups, 2]."""
    return torch.zeros(B, cabi.STAT_STRIPES, groups, 2, dtype=torch.float64, device=DEV)


def stats_striped(y, groups):
    """Reference statistics of y placed in stripe 0 of the striped layout (device)."""
    s = stats_buffer(y.shape[0], groups)
    s[:, 0] = gn_stats_ref(y, groups).to(DEV)
    return s


def dec_max(u):
    """Decode the kernels' order-preserving uint32 max code (int32 tensor) -> float32 (0 -> -inf-ish)."""
    shape = u.shape
    u = u.cpu().to(torch.int64).flatten() & 0xFFFFFFFF
    neg = (u & 0x80000000) == 0
    bits = torch.where(neg, (~u) & 0xFFFFFFFF, u & 0x7FFFFFFF)
    f = torch.from_numpy(bits.numpy().astype(np.uint32).view(np.float32).copy())
    f = torch.where(u == 0, torch.tensor(-3.0e38), f)          # untouched stripes
    return f.reshape(shape)

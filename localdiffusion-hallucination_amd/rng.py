"""Portable counter-based random numbers (host side).

The reference draws x_T with ``torch.manual_seed(10); torch.randn(shape)`` and each step's z
with ``torch.randn_like`` (/root/reference/ddpm.py:934-935, :852, :857, :989, :1020, :1064).
torch's CPU and GPU generators produce different streams, so golden vectors made on a CPU cannot
be reproduced on the GPU box from a torch seed.  This module defines a tiny counter-based
generator that is a pure function of ``(seed, stream, element index)``:

    base = mix64(seed ^ (stream * 0xD1B54A32D192ED03))
    h    = mix64(base + (i + 1) * 0x9E3779B97F4A7C15)
    u1   = ((h >> 40) + 1) * 2**-24          in (0, 1]
    u2   = ((h >> 16) & 0xFFFFFF) * 2**-24   in [0, 1)
    z    = sqrt(-2 ln u1) * cos(2 pi u2)     (Box-Muller, one normal per element)

``mix64`` is the splitmix64 finaliser.  ``stream`` is the index of the draw inside one sampling
run (0 = x_T, k = the k-th ``randn_like``), ``i`` the flat element index in NCHW order.
The same function is implemented on the device in ``csrc/pointwise.hip`` (``ld_randn``) in fp32
arithmetic; the two agree to fp32 rounding of the Box-Muller transform (the integer part is
bit-identical).  Golden fixtures and the parity tests use THIS host version for both sides.
"""
import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_STREAM = np.uint64(0xD1B54A32D192ED03)
_MASK64 = (1 << 64) - 1


def mix64(z):
    """splitmix64 finaliser on uint64 numpy arrays (wrap-around arithmetic)."""
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def _mix64_int(z):
    z &= _MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK64
    return z ^ (z >> 31)


def stream_base(seed, stream):
    """The per-draw 64-bit key (python int); also what the device kernel derives."""
    return _mix64_int((int(seed) & _MASK64) ^ ((int(stream) * 0xD1B54A32D192ED03) & _MASK64))


def _hash(seed, stream, n, offset=0):
    base = np.uint64(stream_base(seed, stream))
    idx = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return mix64(base + idx * _GOLD)


def randn(shape, seed, stream, offset=0):
    """Standard normals, float32, C-order over ``shape``: elements [offset, offset + prod(shape)) of the draw's
    sequence (a rank that owns a slice of a batch generates exactly the slice's values)."""
    n = int(np.prod(shape))
    h = _hash(seed, stream, n, int(offset))
    u1 = ((h >> np.uint64(40)).astype(np.float64) + 1.0) * (2.0 ** -24)
    u2 = ((h >> np.uint64(16)) & np.uint64(0xFFFFFF)).astype(np.float64) * (2.0 ** -24)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return z.astype(np.float32).reshape(shape)


def uniform(shape, seed, stream, lo=0.0, hi=1.0):
    """Uniform [lo, hi) float32 (53-bit mantissa draw, then scaled in float64)."""
    n = int(np.prod(shape))
    h = _hash(seed, stream, n)
    u = (h >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def fnv1a64(text):
    """FNV-1a hash of a parameter name -> 64-bit stream id for procedural weights."""
    h = 0xCBF29CE484222325
    for b in text.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & _MASK64
    return h


class NoiseStream:
    """Stateful view used by samplers: draw 0 is x_T, draw k the k-th noise tensor."""

    def __init__(self, seed=10):
        self.seed = int(seed)
        self.count = 0

    def reset(self):
        self.count = 0

    def next(self, shape):
        out = randn(shape, self.seed, self.count)
        self.count += 1
        return out

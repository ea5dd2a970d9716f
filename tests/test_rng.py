"""Portable RNG: known answers (so the device kernel and future ports can be checked against the
same integers) and distribution sanity."""
import numpy as np

from localdiffusion_hallucination_amd import rng


def test_mix64_known_answers():
    # splitmix64 finaliser of 0 is 0 by construction; the next two pin the constants.
    assert int(rng.mix64(np.uint64(0))) == 0
    assert rng._mix64_int(1) == int(rng.mix64(np.uint64(1)))
    assert rng.stream_base(10, 0) == rng._mix64_int(10)
    assert rng.stream_base(10, 3) == rng._mix64_int(10 ^ ((3 * 0xD1B54A32D192ED03) & ((1 << 64) - 1)))


def test_randn_is_pure_function_of_key():
    a = rng.randn((2, 3, 8, 8), 10, 5)
    b = rng.randn((2, 3, 8, 8), 10, 5)
    c = rng.randn((2, 3, 8, 8), 10, 6)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert a.dtype == np.float32
    # prefix property: element i does not depend on the tensor's size
    assert np.array_equal(rng.randn((10,), 1, 2), rng.randn((100,), 1, 2)[:10])


def test_randn_moments():
    z = rng.randn((1 << 18,), 123, 0).astype(np.float64)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01
    assert abs((z ** 3).mean()) < 0.03 and abs((z ** 4).mean() - 3.0) < 0.1
    assert np.isfinite(z).all()


def test_uniform_range():
    u = rng.uniform((1 << 16,), 7, 1, 0.0, 2.0)
    assert u.min() >= 0.0 and u.max() < 2.0 and abs(u.mean() - 1.0) < 0.02

"""The activation pool's placement (localdiffusion_hallucination_amd.pool): pure host logic, checked without a GPU.
Buffers whose closed liveness intervals meet never share an address; on UNet-shaped lifetimes (short-lived pairs beside
skip tensors that live across the U) the largest-first placement reaches the peak live set."""
import random

from localdiffusion_hallucination_amd.pool import peak_live, place_intervals


def _check(size, first, last, offset, top):
    n = len(size)
    for a in range(n):
        assert offset[a] >= 0 and offset[a] + size[a] <= top
        for b in range(a + 1, n):
            if first[a] <= last[b] and first[b] <= last[a]:                       # live together (closed intervals)
                assert offset[a] + size[a] <= offset[b] or offset[b] + size[b] <= offset[a], (a, b)


def _unet_like(levels=4, unit=16):
    """Lifetimes of a U: per level two blocks of (h1, h2, out) with out feeding the next block, one skip per block kept
    until the mirrored up-path block, sizes halving per level."""
    size, first, last, t = [], [], [], 0
    skips = []
    def buf(sz, f, l):
        size.append(sz); first.append(f); last.append(l)
        return len(size) - 1
    for lv in range(levels):
        sz = unit >> lv if unit >> lv else 1
        for _ in range(2):
            buf(sz, t, t + 1); buf(sz, t + 1, t + 2)
            skips.append((buf(sz, t + 2, t + 3), sz))
            t += 3
    for lv in reversed(range(levels)):
        sz = unit >> lv if unit >> lv else 1
        for _ in range(2):
            k, _sz = skips.pop()
            last[k] = t + 2                                                       # the skip is read by the up-path block
            buf(sz, t, t + 1); buf(sz, t + 1, t + 2); buf(sz, t + 2, t + 3)
            t += 3
    return size, first, last


def test_no_two_live_buffers_overlap_random():
    rnd = random.Random(7)
    for _ in range(50):
        n = rnd.randint(1, 40)
        size = [rnd.choice([1, 2, 4, 8, 16]) * 256 for _ in range(n)]
        first = [rnd.randint(0, 30) for _ in range(n)]
        last = [f + rnd.randint(0, 12) for f in first]
        for by_size in (True, False):
            offset, top = place_intervals(size, first, last, by_size)
            _check(size, first, last, offset, top)
            assert top >= peak_live(size, first, last)
            assert (offset, top) == place_intervals(size, first, last, by_size)    # deterministic


def test_unet_lifetimes_reach_the_peak_live_set_largest_first():
    size, first, last = _unet_like()
    off_s, top_s = place_intervals(size, first, last, True)
    off_f, top_f = place_intervals(size, first, last, False)
    _check(size, first, last, off_s, top_s)
    _check(size, first, last, off_f, top_f)
    peak = peak_live(size, first, last)
    assert top_s == peak, (top_s, peak)
    assert top_f >= top_s
    assert top_s < 0.5 * sum(size)


def test_inputs_and_outputs_of_one_launch_never_alias():
    # a writes at launch 3 what b (dead after launch 3) feeds: closed intervals keep them apart
    offset, top = place_intervals([8, 8], [0, 3], [3, 5], True)
    assert top == 16 and offset[0] != offset[1]
    offset, top = place_intervals([8, 8], [0, 4], [3, 5], True)                  # disjoint: shared
    assert top == 8 and offset[0] == offset[1] == 0


def test_empty_and_single():
    assert place_intervals([], [], []) == ({}, 0) and peak_live([], [], []) == 0
    assert place_intervals([5], [2], [2]) == ({0: 0}, 5)


def test_declarations_and_the_reflection_scan_must_agree():
    """pool.check_declared: the builders' reads / writes and what the scan of the launch list finds are compared launch by
    launch; either direction of disagreement names the launch and the buffer."""
    import pytest
    from localdiffusion_hallucination_amd.pool import check_declared, intervals_from_declared
    ops, bufs = ["conv a", "gn_apply", "conv b"], ["#0", "#1", "#2"]
    check_declared([{0}, {0, 1}, {1, 2}], [{0}, {0, 1}, {1, 2}], ops, bufs)
    with pytest.raises(RuntimeError, match=r"launch 1 \(gn_apply\).*#2.*does not\s+declare"):
        check_declared([{0}, {0, 1}, {1, 2}], [{0}, {0, 1, 2}], ops, bufs)          # an undeclared pointer into buffer 2
    with pytest.raises(RuntimeError, match=r"launch 2 \(conv b\).*#1.*no patchable reference"):
        check_declared([{0}, {0, 1}, {1, 2}], [{0}, {0, 1}, {2}], ops, bufs)          # declared, but held as something the scan cannot patch
    first, last = intervals_from_declared([{0}, {0, 1}, {1, 2}], 4, live_out={2})
    assert first == [0, 1, 2, -1] and last == [1, 2, 3, 3]                            # live_out -> n_ops; undeclared buffer 3 is kept apart

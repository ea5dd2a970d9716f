"""Placement of a plan's buffers in one pool by liveness (pure host logic; ``unet._Plan._pool_buffers`` feeds it).

A buffer k is live over the CLOSED interval [first[k], last[k]] of launch indices (an op's inputs and outputs are live
together, so they never alias); two buffers may share addresses iff their intervals are disjoint.  The reference has
no counterpart: PyTorch's caching allocator does this dynamically for ddpm.py's eager tensors."""
from typing import Dict, List, Sequence, Tuple


def place_intervals(size: Sequence[int], first: Sequence[int], last: Sequence[int], by_size: bool = True) -> Tuple[Dict[int, int], int]:
    """Offsets for every buffer and the pool size.  ``by_size``: the largest buffers first, each at the lowest offset free of
    every placed buffer whose interval meets its own (the pool then equals the peak live set on the UNet's plans); otherwise
    in order of first use (9 % more pool at the bench shape, docs/findings.md 108)."""
    n = len(size)
    order = sorted(range(n), key=(lambda k: (-size[k], first[k], k)) if by_size else (lambda k: (first[k], -size[k], k)))
    offset: Dict[int, int] = {}
    placed: List[Tuple[int, int, int, int]] = []        # (offset, size, first, last)
    top = 0
    for k in order:
        busy = sorted((o, sz) for o, sz, f0, l0 in placed if f0 <= last[k] and first[k] <= l0)
        pos = 0
        for off, sz in busy:
            if off - pos >= size[k]:
                break
            pos = max(pos, off + sz)
        offset[k] = pos
        placed.append((pos, size[k], first[k], last[k]))
        top = max(top, pos + size[k])
    return offset, top


def peak_live(size: Sequence[int], first: Sequence[int], last: Sequence[int]) -> int:
    """The largest sum of sizes live at one launch: the lower bound of any placement."""
    events = sorted({f for f in first} | {l for l in last})
    return max((sum(size[k] for k in range(len(size)) if first[k] <= i <= last[k]) for i in events), default=0)

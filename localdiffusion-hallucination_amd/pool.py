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


def check_declared(declared: Sequence[set], scanned: Sequence[set], op_names: Sequence[str], buf_names: Sequence[str]) -> None:
    """A plan's launches DECLARE the pooled buffers they read and write (``_Plan.decl``); the reflection scan of the same
    launches (pointers in ctypes argument blocks, tensors captured by raw launches) is the cross-check -- and the only
    thing that can PATCH a pointer.  They must agree launch by launch:
      * a buffer the scan finds but the launch does not declare would get no liveness from the declarations: its memory
        would be re-used while the launch still touches it;
      * a buffer a launch declares but the scan cannot find is held in a form that cannot be re-pointed at the pool (a
        plain integer, a pointer computed past a tracked range, a tensor inside an object): the launch would keep reading
        or writing the freed allocation.
    Either is a RuntimeError that names the launch and the buffer."""
    for i, (d, f) in enumerate(zip(declared, scanned)):
        extra, missing = sorted(f - d), sorted(d - f)
        if extra:
            raise RuntimeError(f"pool: launch {i} ({op_names[i]}) touches pooled buffer(s) {[buf_names[k] for k in extra]} that it does not "
                               "declare in reads= / writes=: their liveness would be wrong")
        if missing:
            raise RuntimeError(f"pool: launch {i} ({op_names[i]}) declares pooled buffer(s) {[buf_names[k] for k in missing]} but holds no "
                               "patchable reference to them (a ctypes pointer field or a captured tensor): it cannot be re-pointed at the pool")


def intervals_from_declared(declared: Sequence[set], n_bufs: int, live_out: set) -> Tuple[List[int], List[int]]:
    """first / last launch index of every buffer from the per-launch declarations.  ``live_out``: buffers something reads
    AFTER the launch list (the sampler's fused final step): live from their first launch to the end, n_ops.  A buffer no launch
    declares is kept apart for the whole evaluation ([-1, n_ops])."""
    n_ops = len(declared)
    first, last = [None] * n_bufs, [None] * n_bufs
    for i, ks in enumerate(declared):
        for k in ks:
            if first[k] is None:
                first[k] = i
            last[k] = i
    for k in range(n_bufs):
        if first[k] is None:
            first[k], last[k] = -1, n_ops
        elif k in live_out:
            last[k] = n_ops
    return first, last

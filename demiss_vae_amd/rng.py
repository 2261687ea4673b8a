"""State of the on-device Philox stream used when no explicit noise is passed.

A draw of n elements consumes the counter range [offset, offset + n) of the stream named by
`seed`; kernels index the counter by element, so results do not depend on launch geometry.
Under data parallelism each rank offsets its seed by its rank (set_seed(seed + rank)).

Recording a step in a HIP graph bakes the kernel arguments in, (seed, offset) included.  `CaptureStream` keeps such
a recording random: while it is active, draws come from a stream of their own (the seed mixed with the recording's
id) with offsets counted from zero, and every Philox kernel also receives the address of a device-resident counter
whose value it adds to its offset.  Leaving the scope (still inside the recording) records `counter += draws of one
step`, so replay k uses the counters [k * step, (k + 1) * step) of that stream: fresh noise on every replay,
disjoint from the eager stream and from other recordings."""
import itertools

import torch

_MASK = 0xFFFFFFFFFFFFFFFF
_seed = None
_offset = 0
_capture = None
_ids = itertools.count(1)


def set_seed(seed: int):
    global _seed, _offset
    _seed = int(seed) & _MASK
    _offset = 0


def get_state():
    return _seed, _offset


def set_state(seed, offset):
    global _seed, _offset
    _seed, _offset = seed, offset


def next_offset(n: int):
    """Reserve n counter values; returns (seed, first counter)."""
    global _seed, _offset
    if _seed is None:
        set_seed(torch.initial_seed())
    n4 = (int(n) + 3) & ~3               # ranges start on a multiple of 4: one generator call serves 4 elements
    if _capture is not None:
        off = _capture.used
        _capture.used += n4
        # (a kernel of the recording may already have advanced the device counter by `bumped`: see advance_in_kernel)
        return _capture.seed(_seed), off - _capture.bumped
    off = _offset
    _offset += n4
    return _seed, off


def base_ptr():
    """address of the active recording's device counter (None outside recordings): the offset_base argument of the
    Philox-consuming entry points"""
    return _capture.counter.data_ptr() if _capture is not None else None


def advance_in_kernel():
    """For a kernel of the recording that can add to the device counter on the side (the optimiser's update kernel):
    (address of the counter, amount) covering the draws recorded so far, or None outside recordings / when nothing
    is owed.  The caller must make exactly one thread add the amount; draws recorded later are handed offsets relative
    to the advanced counter, and leaving the recording adds only what is still owed - usually nothing, which saves the
    step a launch."""
    c = _capture
    if c is None or c.used == c.bumped:
        return None
    owed, c.bumped = c.used - c.bumped, c.used
    return c.counter.data_ptr(), owed


class CaptureStream:
    def __init__(self, device):
        self.counter = torch.zeros(1, dtype=torch.int64, device=device)
        self.id = next(_ids)
        self.used = 0
        self.bumped = 0

    def seed(self, base_seed):
        return (base_seed ^ (0xD1B54A32D192ED03 * self.id)) & _MASK

    def __enter__(self):
        global _capture
        assert _capture is None, "recordings do not nest"
        _capture, self.used, self.bumped = self, 0, 0
        return self

    def __exit__(self, *exc):
        global _capture
        _capture = None
        if self.used > self.bumped and exc[0] is None:
            self.counter.add_(self.used - self.bumped)     # recorded with the step: advances on every replay
        return False

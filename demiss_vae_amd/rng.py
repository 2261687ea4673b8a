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
        return _capture.seed(_seed), off
    off = _offset
    _offset += n4
    return _seed, off


def base_ptr():
    """address of the active recording's device counter (None outside recordings): the offset_base argument of the
    Philox-consuming entry points"""
    return _capture.counter.data_ptr() if _capture is not None else None


class CaptureStream:
    def __init__(self, device):
        self.counter = torch.zeros(1, dtype=torch.int64, device=device)
        self.id = next(_ids)
        self.used = 0

    def seed(self, base_seed):
        return (base_seed ^ (0xD1B54A32D192ED03 * self.id)) & _MASK

    def __enter__(self):
        global _capture
        assert _capture is None, "recordings do not nest"
        _capture, self.used = self, 0
        return self

    def __exit__(self, *exc):
        global _capture
        _capture = None
        if self.used and exc[0] is None:
            self.counter.add_(self.used)     # recorded with the step: advances on every replay
        return False

"""State of the on-device Philox stream used when no explicit noise is passed.

A draw of n elements consumes the counter range [offset, offset + n) of the stream named by
`seed`; kernels index the counter by element, so results do not depend on launch geometry.
Under data parallelism each rank offsets its seed by its rank (set_seed(seed + rank))."""
import torch

_seed = None
_offset = 0


def set_seed(seed: int):
    global _seed, _offset
    _seed = int(seed) & 0xFFFFFFFFFFFFFFFF
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
    off = _offset
    _offset += (int(n) + 3) & ~3     # ranges start on a multiple of 4: one generator call serves 4 elements
    return _seed, off

"""Device-resident batches.

The reference feeds the step from a torch DataLoader (up to 8 worker processes, per-sample __getitem__, numpy
collation, a host-to-device copy per batch: vgiwae/data/missing_data_module.py:289-312).  At a few hundred
microseconds per step that loader is the bottleneck.  Here a split is moved to the device once (all its arrays:
data, mask, optional targets, indices) and a batch is an index selection on the device; shuffling is a device
randperm per epoch.  Batches have the same tuple layout and the same last-short-batch behaviour as the reference's
loader.

With `shard=(rank, world)` the loader keeps only rows rank, rank+world, ... of the split (the same number on every
rank; up to world-1 trailing rows are left out), so under data parallelism a rank meets the same datapoints every
epoch - what the per-rank imputation buffers of DeMissVAE need - and all ranks run the same number of batches."""
import numpy as np
import torch


class DeviceLoader:
    def __init__(self, dataset, batch_size: int, shuffle: bool, device, generator: torch.Generator = None,
                 drop_last: bool = False, shard=None):
        items = dataset[:]
        items = items if isinstance(items, tuple) else (items,)
        if shard is not None and shard[1] > 1:
            r, w = shard
            per = len(items[0]) // w
            items = tuple(np.asarray(a)[r:per * w:w] for a in items)
        self.arrays = tuple(torch.as_tensor(np.ascontiguousarray(a)).to(device) for a in items)
        self.n = self.arrays[0].shape[0]
        self.batch_size, self.shuffle, self.device, self.generator = batch_size, shuffle, device, generator
        self.drop_last = drop_last
        self.dataset = dataset
        # True: full-size batches are written into the same tensors every time (marked `_dmv_persistent`), so a
        # recorded step can take them as its inputs without a copy; a consumer must be done with a batch before
        # asking for the next
        self.reuse_buffers = False
        self._bufs = None
        self._row_bytes = None

    def _gather(self, sel, out):
        """rows `sel` of every array of the split into `out`: one launch for the whole batch tuple on the GPU"""
        if not self.arrays[0].is_cuda or len(self.arrays) > 4:
            for a, b in zip(self.arrays, out):
                torch.index_select(a, 0, sel, out=b)
            return
        import ctypes as C
        from . import _lib as L
        if self._row_bytes is None:
            self._row_bytes = (C.c_int64 * len(self.arrays))(*[a[0].numel() * a.element_size() for a in self.arrays])
            self._src = L.ptr_array(self.arrays)
        L.check(L.lib().dmv_rows_gather_multi(len(self.arrays), self._src, L.ptr_array(out), self._row_bytes, L.ptr(sel),
                                              sel.shape[0], L.stream_ptr()))

    def __len__(self):
        return self.n // self.batch_size if self.drop_last else (self.n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        if self.shuffle:
            order = torch.randperm(self.n, device=self.device, generator=self.generator)
        else:
            order = torch.arange(self.n, device=self.device)
        for i in range(len(self)):
            sel = order[i * self.batch_size:(i + 1) * self.batch_size]
            if self.reuse_buffers and sel.shape[0] == self.batch_size:
                if self._bufs is None:
                    self._bufs = tuple(a.new_empty((self.batch_size,) + tuple(a.shape[1:])) for a in self.arrays)
                    for b in self._bufs:
                        b._dmv_persistent = True
                self._gather(sel, self._bufs)
                yield self._bufs
            else:
                out = tuple(a.new_empty((sel.shape[0],) + tuple(a.shape[1:])) for a in self.arrays)
                self._gather(sel, out)
                yield out

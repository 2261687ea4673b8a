"""Appends the datapoint index to every item (reference: vgiwae/data/dataset_with_indices.py:8-45); the index is
what addresses the imputation buffer."""
import numbers

import numpy as np
import torch
from torch.utils.data import Dataset


class DatasetWithIndices(Dataset):
    def __init__(self, dataset: Dataset):
        self.dataset = dataset
        first = dataset[0]
        self.index_idx = len(first) if isinstance(first, tuple) else 1

    def _as_index(self, item):
        if isinstance(item, slice):
            return np.arange(item.start or 0, item.stop or len(self), item.step or 1, dtype=np.int64)
        if isinstance(item, (numbers.Integral, np.ndarray, torch.Tensor)):
            return item
        raise TypeError(f"{type(self.dataset).__name__} indices must be integers, arrays, tensors or slices, "
                        f"not {type(item).__name__}.")

    def __getitem__(self, idx):
        data = self.dataset[idx]
        data = data if isinstance(data, tuple) else (data,)
        return data + (self._as_index(idx),)

    def __setitem__(self, key, value):
        self.dataset[key] = value

    def __len__(self):
        return len(self.dataset)

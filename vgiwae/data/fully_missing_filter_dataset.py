"""View of a dataset without the rows whose mask has no observed entry
(reference behaviour: vgiwae/data/fully_missing_filter_dataset.py:6-31)."""
import numpy as np
import torch
from torch.utils.data import Dataset


def _rows_with_observations(mask):
    """indices of the rows with at least one observed value, in the array library of the mask"""
    if torch.is_tensor(mask):
        keep = mask.sum(dim=1) != 0
        return torch.nonzero(keep, as_tuple=False).squeeze()
    if isinstance(mask, np.ndarray):
        keep = mask.sum(axis=1) != 0
        return np.flatnonzero(keep)
    raise TypeError("Invalid missingness mask type!")


class FullyMissingDataFilter(Dataset):
    """Indexing goes through `not_fully_missing_idx` (the kept rows of the wrapped dataset, ascending); writes reach
    the wrapped dataset at the same rows.  `miss_mask_idx`: position of the mask in the wrapped dataset's tuples."""

    def __init__(self, dataset: Dataset, miss_mask_idx: int = 1):
        everything = dataset[:]
        self.dataset = dataset
        self.not_fully_missing_idx = _rows_with_observations(everything[miss_mask_idx])

    def __len__(self):
        return len(self.not_fully_missing_idx)

    def _inner(self, key):
        return self.not_fully_missing_idx[key]

    def __getitem__(self, idx):
        return self.dataset[self._inner(idx)]

    def __setitem__(self, key, value):
        self.dataset[self._inner(key)] = value

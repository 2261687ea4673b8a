"""Drops datapoints with no observed value (reference: vgiwae/data/fully_missing_filter_dataset.py:6-31)."""
import numpy as np
import torch
from torch.utils.data import Dataset


class FullyMissingDataFilter(Dataset):
    def __init__(self, dataset: Dataset, miss_mask_idx: int = 1):
        self.dataset = dataset
        M = dataset[:][miss_mask_idx]
        if isinstance(M, torch.Tensor):
            self.not_fully_missing_idx = torch.nonzero(M.sum(dim=1) != 0, as_tuple=False).squeeze()
        elif isinstance(M, np.ndarray):
            self.not_fully_missing_idx = np.flatnonzero(M.sum(axis=1) != 0)
        else:
            raise TypeError("Invalid missingness mask type!")

    def __getitem__(self, idx):
        return self.dataset[self.not_fully_missing_idx[idx]]

    def __setitem__(self, key, value):
        self.dataset[self.not_fully_missing_idx[key]] = value

    def __len__(self):
        return len(self.not_fully_missing_idx)

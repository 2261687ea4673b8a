"""Uniform (MCAR) missingness (reference: vgiwae/data/uniform_missingness_provider.py:8-76)."""
import math

import numpy as np
import torch
from torch.utils.data import Dataset


def generate_uniform_mask(shape, total_miss: float, rng: torch.Generator = None) -> np.ndarray:
    """True = observed.  Exactly floor(total_miss * numel) entries are missing; their positions come from one
    torch.randperm of the flattened array drawn from `rng` (so the mask depends only on shape, fraction, seed)."""
    total = math.prod(shape)
    flat = torch.ones(total, dtype=torch.bool)
    flat[: int(total_miss * total)] = False
    return flat[torch.randperm(total, generator=rng)].numpy().reshape(shape)


class _MaskedDataset(Dataset):
    """Inserts the mask right after the tensor it belongs to in every item tuple."""

    def __init__(self, dataset, target_idx, miss_mask):
        self.dataset, self.target_idx, self.miss_mask = dataset, target_idx, miss_mask

    def _target(self):
        data = self.dataset[:]
        return data[self.target_idx] if isinstance(data, tuple) else data

    def __getitem__(self, idx):
        item, m = self.dataset[idx], self.miss_mask[idx]
        if not isinstance(item, tuple):
            return (item, m)
        return item[: self.target_idx + 1] + (m,) + item[self.target_idx + 1:]

    def __setitem__(self, key, value):
        self.dataset[key] = value

    def __len__(self):
        return len(self.dataset)


class UniformMissingnessDataset(_MaskedDataset):
    def __init__(self, dataset: Dataset, target_idx: int = 0, total_miss: float = 0.0, rng: torch.Generator = None):
        super().__init__(dataset, target_idx, None)
        self.total_miss, self.rng = total_miss, rng
        self.miss_mask = generate_uniform_mask(self._target().shape, total_miss, rng)

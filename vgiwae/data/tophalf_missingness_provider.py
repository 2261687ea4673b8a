"""Block missingness: the first int(D * total_miss) dimensions of every datapoint are missing - the top rows of a
flattened image (reference: vgiwae/data/tophalf_missingness_provider.py:5-75)."""
import numpy as np
import torch
from torch.utils.data import Dataset

from .uniform_missingness_provider import _MaskedDataset


def generate_tophalf_mask(shape, total_miss: float) -> np.ndarray:
    mask = np.ones(shape, dtype=bool)
    mask[:, : int(shape[-1] * total_miss)] = False
    return mask


class TopHalfMissingnessDataset(_MaskedDataset):
    def __init__(self, dataset: Dataset, target_idx: int = 0, total_miss: float = 0.0, rng: torch.Generator = None):
        super().__init__(dataset, target_idx, None)
        self.total_miss, self.rng = total_miss, rng
        self.miss_mask = generate_tophalf_mask(self._target().shape, total_miss)

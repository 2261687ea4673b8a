"""Image-quadrant missingness (reference: vgiwae/data/quadrant_missingness_provider.py:8-95)."""
from typing import List

import numpy as np
import torch
from torch.utils.data import Dataset

from .uniform_missingness_provider import _MaskedDataset


def generate_quadrant_mask(n: int, img_shape, total_miss: float, rng: torch.Generator = None) -> np.ndarray:
    """True = observed.  int(4 * total_miss) of the four quadrants of every image are missing, drawn without
    replacement with torch.multinomial from `rng`.  Quadrant order: top-left, top-right, bottom-left,
    bottom-right; the row split uses img_shape[1] like the reference (identical for square images)."""
    H, W = int(img_shape[0]), int(img_shape[1])
    mask = torch.ones(n, H, W, dtype=torch.bool)
    if total_miss != 0.:
        q = torch.multinomial(torch.ones(n, 4), num_samples=int(total_miss * 4), replacement=False, generator=rng)
        top, bottom = slice(0, H // 2), slice(W // 2, None)
        left, right = slice(0, W // 2), slice(W // 2, None)
        for k, (rows, cols) in enumerate(((top, left), (top, right), (bottom, left), (bottom, right))):
            hit = (q == k).any(-1)
            mask[hit, rows, cols] = False
    return mask.reshape(n, -1).numpy()


class QuadrantMissingnessDataset(_MaskedDataset):
    def __init__(self, dataset: Dataset, target_idx: int = 0, total_miss: float = 0.00, img_dims: List[int] = None,
                 rng: torch.Generator = None):
        assert total_miss in (0., 0.25, 0.5, 0.75), "Missingness fraction can only be 0., 0.25, 0.5, or 0.75!"
        super().__init__(dataset, target_idx, None)
        self.total_miss, self.rng, self.img_shape = total_miss, rng, img_dims
        self.miss_mask = generate_quadrant_mask(self._target().shape[0], img_dims, total_miss, rng)

"""Batch contract of the hot path: (X float32 [B,D], M bool [B,D], index int64 [B]).  Mask generation, the
fully-missing filter and the index wrapper reproduce the reference bit for bit (integer / boolean work)."""
from .uniform_missingness_provider import UniformMissingnessDataset, generate_uniform_mask  # noqa: F401
from .quadrant_missingness_provider import QuadrantMissingnessDataset, generate_quadrant_mask  # noqa: F401
from .fully_missing_filter_dataset import FullyMissingDataFilter  # noqa: F401
from .dataset_with_indices import DatasetWithIndices  # noqa: F401
from .missing_data_module import MissingDataModule, DATASET, MISSINGNESS, PRE_IMPUTATION  # noqa: F401,E402

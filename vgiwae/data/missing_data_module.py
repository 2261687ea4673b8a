"""Data module of the configs' `data:` section (reference: vgiwae/data/missing_data_module.py:24-312): loads the
complete data, adds synthetic missingness (bit-exact masks), drops fully-missing rows, appends indices, optionally
pre-imputes.  The loaders keep every split resident on the device and yield (X, M, [target,] index) batches by
index selection there: no worker processes, no per-sample collation, no host-to-device copies inside an epoch."""
from enum import Enum, auto
from typing import List, Optional

import numpy as np
import torch

from demiss_vae_amd.loader import DeviceLoader
from demiss_vae_amd.pl_compat import LightningDataModule
from .datasets import MNIST, Omniglot, ToyDataset, UCITable
from .dataset_with_indices import DatasetWithIndices
from .fully_missing_filter_dataset import FullyMissingDataFilter
from .quadrant_missingness_provider import QuadrantMissingnessDataset
from .tophalf_missingness_provider import TopHalfMissingnessDataset
from .uniform_missingness_provider import UniformMissingnessDataset


class DATASET(Enum):
    toy_mog = auto()
    toy_vae = auto()
    toy_mog2 = auto()
    toy_mog2_large = auto()
    toy_mog3_large = auto()
    mnist = auto()
    mnist_bin = auto()
    mnist_14x14 = auto()
    mnist_5x5 = auto()
    mnist_7x7 = auto()
    uci_gas = auto()
    uci_power = auto()
    uci_hepmass = auto()
    uci_miniboone = auto()
    omniglot_bin = auto()
    omniglot_28x28_bin = auto()

    def __call__(self, *args, **kwargs):
        n = self.name
        if n.startswith("toy_"):
            return ToyDataset(*args, filename="data_" + n[4:], **kwargs)
        if n.startswith("uci_"):
            return UCITable(*args, folder=n[4:], **kwargs)
        if n == "mnist":
            return MNIST(*args, **kwargs)
        if n == "mnist_bin":
            return MNIST(*args, binarise=True, **kwargs)
        if n.startswith("mnist_"):                                   # mnist_14x14, mnist_7x7, mnist_5x5
            h, w = (int(v) for v in n[6:].split("x"))
            return MNIST(*args, resize=(h, w), **kwargs)
        if n == "omniglot_bin":
            return Omniglot(*args, binarise_fixed=True, **kwargs)
        if n == "omniglot_28x28_bin":
            return Omniglot(*args, resize=(28, 28), binarise_fixed=True, **kwargs)
        raise NotImplementedError(f"dataset {n}")


class MISSINGNESS(Enum):
    uniform = "uniform"
    top_half = "top_half"
    quadrants = "quadrants"


class PRE_IMPUTATION(Enum):
    oracle = "oracle"
    mean = "mean"
    zero = "zero"
    empirical = "empirical"


def _add_missingness(dataset, kind: MISSINGNESS, total_miss, img_dims, rng):
    if kind is MISSINGNESS.uniform:
        return UniformMissingnessDataset(dataset, target_idx=0, total_miss=total_miss, rng=rng)
    if kind is MISSINGNESS.top_half:
        return TopHalfMissingnessDataset(dataset, target_idx=0, total_miss=total_miss, rng=rng)
    if kind is MISSINGNESS.quadrants:
        return QuadrantMissingnessDataset(dataset, target_idx=0, img_dims=img_dims, total_miss=total_miss, rng=rng)
    raise NotImplementedError(f"missingness {kind.value}")


def impute_with_mean(dataset, rng=None):
    X, M = dataset[:][:2]
    means = (X * M).sum(axis=0) / M.sum(axis=0)
    dataset[:] = X * M + means * ~M
    return means


def impute_with_zero(dataset, rng=None):
    X, M = dataset[:][:2]
    dataset[:] = X * M


def impute_with_empirical_distribution_sample(dataset, rng=None):
    """every missing entry <- a value drawn (torch.multinomial over the observed rows of that column, `rng`) from the
    column's observed values (missing_data_module.py:134-155)"""
    X, M = dataset[:][:2]
    X = np.array(X, copy=True)
    Mt = torch.as_tensor(M)
    for i in range(M.shape[-1]):
        n_miss = int((~M[:, i]).sum())
        if n_miss == 0:
            continue
        picks = torch.multinomial(Mt[:, i].float(), n_miss, generator=rng).numpy()
        X[~M[:, i], i] = X[:, i][picks]
    dataset[:] = X


_PRE_IMPUTE = {"oracle": lambda dataset, rng: None, "mean": impute_with_mean, "zero": impute_with_zero,
               "empirical": impute_with_empirical_distribution_sample}


class MissingDataModule(LightningDataModule):
    def __init__(self, dataset: DATASET, batch_size: int, missingness: MISSINGNESS, total_miss_train: float,
                 total_miss_val: float, pre_imputation: PRE_IMPUTATION, pre_impute_val: bool,
                 filter_fully_missing_train: bool = True, filter_fully_missing_val: bool = True,
                 use_test_instead_val: bool = False, img_dims: List[int] = None, data_root: str = "./data",
                 setup_seed: int = None, num_workers: int = None, total_miss_test: float = 0.,
                 use_test_instead_train: bool = False, filter_fully_missing_test: bool = True, test_batch_size: int = None):
        super().__init__()
        self.save_hyperparameters()
        self.device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")

    def _prepare(self, split, total_miss, filter_fully_missing, rng, rng_dataset):
        h = self.hparams
        core = h.dataset(h.data_root, split=split, rng=rng_dataset)
        data = _add_missingness(core, h.missingness, total_miss, h.img_dims, rng)
        if filter_fully_missing:
            data = FullyMissingDataFilter(data, miss_mask_idx=1)
        return core, DatasetWithIndices(data)

    def setup(self, stage: Optional[str] = None):
        h = self.hparams
        rng = torch.Generator().manual_seed(h.setup_seed)
        rng_dataset = torch.Generator().manual_seed(int(h.setup_seed * 2.33 / 2))
        if stage == "fit":
            self.train_data_core, self.train_data = self._prepare("train", h.total_miss_train, h.filter_fully_missing_train,
                                                                  rng, rng_dataset)
            self.val_data_core, self.val_data = self._prepare("test" if h.use_test_instead_val else "val", h.total_miss_val,
                                                              h.filter_fully_missing_val, rng, rng_dataset)
            pre = _PRE_IMPUTE[h.pre_imputation.value]
            pre(self.train_data, rng=rng)
            if h.pre_impute_val:
                pre(self.val_data, rng=rng)
            print("Train data size:", len(self.train_data))
            print("Validation data size:", len(self.val_data))
            if h.use_test_instead_train:
                self.setup(stage="test")
                # train_data_core stays the training split's, as in the reference (its assignment at
                # missing_data_module.py:265 goes to a misspelt attribute); only "dataset" clipping reads it
                self.train_data = self.test_data
        elif stage == "test":
            self.test_data_core, self.test_data = self._prepare("test", h.total_miss_test, h.filter_fully_missing_test, rng,
                                                                rng_dataset)
            print("Test data size:", len(self.test_data))
        else:
            raise NotImplementedError(f"stage={stage} is not implemented.")

    def train_dataloader(self):
        """Under torch.distributed every rank trains on its own fixed part of the split with batch_size / world rows a
        step (the global batch stays the configured one); validation and test loaders are whole on every rank."""
        from demiss_vae_amd import parallel as P
        w = P.world()
        return DeviceLoader(self.train_data, -(-self.hparams.batch_size // w), shuffle=True, device=self.device,
                            shard=(P.rank(), w))

    def val_dataloader(self):
        return DeviceLoader(self.val_data, self.hparams.batch_size, shuffle=False, device=self.device)

    def test_dataloader(self):
        bs = self.hparams.test_batch_size if self.hparams.test_batch_size is not None else self.hparams.batch_size
        return DeviceLoader(self.test_data, bs, shuffle=False, device=self.device)

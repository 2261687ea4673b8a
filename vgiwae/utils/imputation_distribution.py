"""ImputationDistribution with the reference's interface (vgiwae/utils/imputation_distribution.py:18-187),
backed by the device-resident demiss_vae_amd.ImputationBuffer: the K imputations of every datapoint stay
in HBM, `get_imputed_datapoints` / `set_imputed_datapoints` are row gathers / scatters through libdmv."""
from enum import Enum
from typing import Tuple

import numpy as np
import torch

from demiss_vae_amd.imputation import ImputationBuffer


class IMPUTATION_INIT(Enum):
    oracle = "oracle"
    mean = "mean"
    zero = "zero"
    empirical = "empirical"
    empirical_parallel = "empirical_parallel"


class ImputationDistribution(torch.nn.Module):
    def __init__(self, dataset, target_idx: int, mis_idx: int, index_idx: int, num_imputations: int,
                 imputation_init: IMPUTATION_INIT, device=None):
        super().__init__()
        all_data = dataset if isinstance(dataset, tuple) else dataset[:]
        self.dataset, self.all_data = dataset, all_data
        self.target_idx, self.mis_idx, self.index_idx = target_idx, mis_idx, index_idx
        self.num_imputations, self.imputation_init = num_imputations, imputation_init
        data, index = all_data[target_idx], all_data[index_idx]
        index = np.asarray(index) if not isinstance(index, torch.Tensor) else index.cpu().numpy()
        assert index.dtype == np.int64, "The dataset index is not of type long!"
        data = np.asarray(data) if not isinstance(data, torch.Tensor) else data.cpu().numpy()
        data = data[index]      # sort by index in case the dataset comes shuffled
        self.device = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
        self.buffer = ImputationBuffer(torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32)),
                                       num_imputations, self.device)
        self.imp_idx = len(all_data)     # the imputation index is appended at the end of the batch tuple

    # reference attribute names
    @property
    def data_imp(self):
        return self.buffer.data_imp

    @property
    def idx_imp(self):
        return self.buffer.idx_imp

    def init_imputations(self):
        kind = self.imputation_init.value if isinstance(self.imputation_init, IMPUTATION_INIT) else str(self.imputation_init)
        data = np.asarray(self.all_data[self.target_idx])
        mask = np.asarray(self.all_data[self.mis_idx]).astype(bool)
        if kind == "oracle":
            return
        if kind in ("empirical", "empirical_parallel"):
            self.buffer.init_empirical(data, mask)
        elif kind == "zero":
            self.buffer.init_zero(mask)
        elif kind == "mean":
            self.buffer.init_mean(data, mask)
        else:
            raise NotImplementedError(kind)

    def get_imputed_datapoints(self, batch: Tuple[torch.Tensor, ...], num_imputations: int = None):
        """batch (X [B,D], M [B,D], ..., index [B]) -> (X_imp [B,K,D], M [B,K,D], ..., index [B,K], idx_imp [B,K])"""
        index = torch.as_tensor(batch[self.index_idx])
        B = index.shape[0]
        K = self.num_imputations if num_imputations is None else num_imputations
        sampled = self.buffer.sample_subset(B, num_imputations) if num_imputations is not None else None
        data_imp, idx_imp = self.buffer.gather(index, sampled)
        out = []
        for i, t in enumerate(batch):
            if i == self.target_idx:
                out.append(data_imp)
                continue
            t = torch.as_tensor(t).to(self.device)
            out.append(t.unsqueeze(1).expand(B, K, *t.shape[1:]).contiguous())
        out.insert(self.imp_idx, idx_imp)
        return tuple(out)

    def set_imputed_datapoints(self, batch: Tuple[torch.Tensor, ...]):
        self.buffer.scatter(batch[self.imp_idx].reshape(-1), batch[self.target_idx].reshape(-1, self.buffer.D))

    def save_imputations(self, path):
        arrays = {f"dataset_{i}": np.asarray(t) for i, t in enumerate(self.all_data)}
        np.savez_compressed(path, **arrays, target_idx=self.target_idx, mis_idx=self.mis_idx, index_idx=self.index_idx,
                            num_imputations=self.num_imputations, imputation_init=str(self.imputation_init),
                            data_imps=self.buffer.data_imp.cpu().numpy(), imp_idx=self.imp_idx)

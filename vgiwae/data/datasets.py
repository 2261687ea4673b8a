"""Complete-data sets named by the configs' `data.dataset` (reference: vgiwae/data/toy.py:170-214,
uci_{gas,power,hepmass,miniboone}.py:74-110, mnist.py:46-117).  Each holds a float32 array `data` [N, D] that can be
read and overwritten through indexing (pre-imputation writes into it), plus per-dimension `data_min` / `data_max`
(used to clip imputations).  Files live under `<data_root>/<folder>/`; nothing is downloaded."""
import os

import numpy as np
import torch
from torch.utils.data import Dataset


class _ArrayDataset(Dataset):
    data: np.ndarray

    def _finish(self):
        self.data_min = np.min(self.data, axis=0)
        self.data_max = np.max(self.data, axis=0)

    def __getitem__(self, index):
        return self.data[index]

    def __setitem__(self, key, value):
        self.data[key] = value

    def __len__(self):
        return self.data.shape[0]


class ToyDataset(_ArrayDataset):
    """`<root>/toy/<filename>.mat` with one array per split ('train', 'val', 'test')."""

    def __init__(self, root: str, filename="data_mog", split: str = "train", rng: torch.Generator = None):
        import scipy.io as sio
        self.data_file = sio.loadmat(os.path.join(os.path.expanduser(root), "toy", f"{filename}.mat"))
        self.data = np.ascontiguousarray(self.data_file[split], dtype=np.float32)
        self._finish()


class UCITable(_ArrayDataset):
    """`<root>/<folder>/<split>.npz` holding the standardised table under 'arr_0' (the MAF splits)."""

    def __init__(self, root: str, folder: str, split: str = "train", rng: torch.Generator = None):
        path = os.path.join(os.path.expanduser(root), folder, f"{split}.npz")
        self.data = np.load(path, allow_pickle=False)["arr_0"].astype(np.float32)
        self._finish()


class MNIST(_ArrayDataset):
    """`<root>/MNIST/<split>.npz` ('data' uint8 [N, 28, 28], 'targets').  Pixels are scaled to [0, 1] and flattened;
    binarise: x = 2 * (u <= p) - 1 with u ~ U[0, 1) from `rng`; otherwise dequantised ((255 p + u) / 256).
    Items are (image, target) pairs like the reference's."""

    def __init__(self, root: str, split: str = "train", transform=None, binarise: bool = False, rng: torch.Generator = None):
        if transform is not None:
            raise NotImplementedError("resized MNIST variants need torchvision, which this build does not use")
        f = np.load(os.path.join(os.path.expanduser(root), "MNIST", f"{split}.npz"), allow_pickle=False)
        self.targets = f["targets"]
        p = (f["data"].astype(np.float32) / 255.0).reshape(len(f["data"]), -1)     # torchvision's ToTensor scaling
        u = torch.rand(*p.shape, generator=rng).numpy()
        if binarise:
            self.data = (u <= p).astype(np.float32) * 2 - 1
        else:
            self.data = (p * 255 + u.astype(np.float32)) / 256
        self._finish()

    def __getitem__(self, index):
        return self.data[index], self.targets[index]

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


def _resized(images, size, mode):
    """every image through PIL's bilinear resize - what torchvision.transforms.Resize does to a PIL image - as float32
    [N, h*w]; mode 'L' (uint8, then scaled to [0, 1] like ToTensor) or 'F' (float, values kept)"""
    from PIL import Image
    out = np.empty((len(images), size[0] * size[1]), dtype=np.float32)
    for i, img in enumerate(images):
        im = Image.fromarray(img, mode=mode).resize((size[1], size[0]), Image.BILINEAR)
        a = np.asarray(im, dtype=np.float32)
        out[i] = (a / 255.0 if mode == "L" else a).reshape(-1)
    return out


class MNIST(_ArrayDataset):
    """`<root>/MNIST/<split>.npz` ('data' uint8 [N, 28, 28], 'targets').  Pixels are scaled to [0, 1] and flattened
    (after an optional resize); binarise: x = 2 * (u <= p) - 1 with u ~ U[0, 1) from `rng`; otherwise dequantised
    ((255 p + u) / 256).  Items are (image, target) pairs like the reference's."""

    def __init__(self, root: str, split: str = "train", resize=None, binarise: bool = False, rng: torch.Generator = None):
        f = np.load(os.path.join(os.path.expanduser(root), "MNIST", f"{split}.npz"), allow_pickle=False)
        self.targets = f["targets"]
        if resize is not None:
            p = _resized(f["data"], resize, "L")
        else:
            p = (f["data"].astype(np.float32) / 255.0).reshape(len(f["data"]), -1)     # torchvision's ToTensor scaling
        u = torch.rand(*p.shape, generator=rng).numpy()
        if binarise:
            self.data = (u <= p).astype(np.float32) * 2 - 1
        else:
            self.data = (p * 255 + u.astype(np.float32)) / 256
        self._finish()

    def __getitem__(self, index):
        return self.data[index], self.targets[index]


class Omniglot(_ArrayDataset):
    """`<root>/omniglot/<split>.npz` ('data' float32 [N, 105, 105] in [0, 1], 'targets'; reference:
    vgiwae/data/omniglot.py:66-160).  Optional bilinear resize, then binarise_fixed: x = 2 * (u <= p) - 1 with
    u ~ U[0, 1) from `rng`, or dequantisation (255 p + u) / 256.  Items are (image, target) pairs."""

    def __init__(self, root: str, split: str = "train", resize=None, binarise_fixed: bool = False,
                 binarise_stochastic: bool = False, rng: torch.Generator = None):
        assert not (binarise_fixed and binarise_stochastic), "You can only use fixed binarisation or stochastic not both."
        if binarise_stochastic:
            raise NotImplementedError("per-batch binarisation is used by no config")
        f = np.load(os.path.join(os.path.expanduser(root), "omniglot", f"{split}.npz"), allow_pickle=False)
        self.targets = f["targets"]
        imgs = f["data"].astype(np.float32)
        self.image_shape = tuple(resize) if resize is not None else imgs.shape[-2:]
        p = _resized(imgs, resize, "F") if resize is not None else imgs.reshape(len(imgs), -1)
        u = torch.rand(*p.shape, generator=rng).numpy()
        if binarise_fixed:
            self.data = (u <= p).astype(np.float32) * 2 - 1
        else:
            self.data = (p * 255 + u.astype(np.float32)) / 256
        self._finish()

    @property
    def img_shape(self):
        return self.image_shape

    def __getitem__(self, index):
        return self.data[index], self.targets[index]

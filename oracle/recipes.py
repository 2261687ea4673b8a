"""Seeded recipes for network parameters and synthetic batches (test infrastructure).

The recipes use numpy's PCG64 bit generator so that a (spec, seed) pair names the same
tensors in the build container (where golden vectors are generated from the reference)
and on the GPU box (where the HIP path is checked against those vectors).  Full-width
golden cases store only their seed and outputs and rebuild inputs from these recipes.
"""
from collections import OrderedDict
from dataclasses import dataclass

import numpy as np
import torch


@dataclass(frozen=True)
class NetSpec:
    """Shape of a residual fully-connected network.

    Mirrors the constructor arguments of the reference's ResidualFCNetwork
    (vgiwae/shared/neural_nets.py:92-99).
    """
    in_dim: int
    out_dim: int
    nb: int
    hidden: int
    act: str = "relu"
    dropout: float = 0.0

    def layer_names(self):
        names = ["initial_layer"]
        for b in range(self.nb):
            names += [f"blocks.{b}.linear_layers.0", f"blocks.{b}.linear_layers.1"]
        names.append("final_layer")
        return names

    def layer_shapes(self):
        shp = [(self.hidden, self.in_dim)]
        shp += [(self.hidden, self.hidden)] * (2 * self.nb)
        shp.append((self.out_dim, self.hidden))
        return shp

    def num_params(self):
        return sum(o * i + o for o, i in self.layer_shapes())


def make_params(spec: NetSpec, seed: int, dtype=torch.float32) -> "OrderedDict[str, torch.Tensor]":
    """Parameters named as in the reference module's state_dict, U(-1/sqrt(in), 1/sqrt(in))."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = OrderedDict()
    for name, (o, i) in zip(spec.layer_names(), spec.layer_shapes()):
        bound = 1.0 / np.sqrt(i)
        out[name + ".weight"] = torch.from_numpy(
            rng.uniform(-bound, bound, size=(o, i)).astype(np.float32)).to(dtype)
        out[name + ".bias"] = torch.from_numpy(
            rng.uniform(-bound, bound, size=(o,)).astype(np.float32)).to(dtype)
    return out


def make_mask(rng, B, D, miss=0.5):
    """Uniform missingness mask (True = observed) with no fully-missing row."""
    M = rng.random((B, D)) >= miss
    empty = ~M.any(axis=1)
    if empty.any():
        cols = rng.integers(0, D, size=int(empty.sum()))
        M[np.nonzero(empty)[0], cols] = True
    return M


def make_batch(seed: int, B: int, D: int, miss=0.5, binary=False):
    """Synthetic (X float32 [B,D], M bool [B,D]) batch.

    binary=True gives X in {-1,+1} with P(+1)=0.13 (binarised-image like).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    if binary:
        X = np.where(rng.random((B, D)) < 0.13, 1.0, -1.0).astype(np.float32)
    else:
        X = rng.standard_normal((B, D)).astype(np.float32)
    M = make_mask(rng, B, D, miss)
    return torch.from_numpy(X), torch.from_numpy(M)


def make_noise(seed: int, *shape):
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(rng.standard_normal(shape).astype(np.float32))


def make_uniform(seed: int, *shape):
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(rng.random(shape).astype(np.float32))

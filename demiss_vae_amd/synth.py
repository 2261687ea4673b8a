"""Synthetic workloads for the benchmarks: network shapes, seeded parameters and batches of the shapes the reference's
configs use (its large training files are not redistributable; UCI tables are standardised, so standard normals stand
in for them).  numpy's PCG64 keeps a (shape, seed) pair the same tensors on every machine."""
from dataclasses import dataclass

import numpy as np
import torch


@dataclass(frozen=True)
class NetShape:
    """Constructor arguments of ResidualFCNetwork (vgiwae/shared/neural_nets.py:92-99)."""
    in_dim: int
    out_dim: int
    nb: int
    hidden: int
    act: str = "relu"
    dropout: float = 0.0

    def layer_shapes(self):
        return [(self.hidden, self.in_dim)] + [(self.hidden, self.hidden)] * (2 * self.nb) + [(self.out_dim, self.hidden)]

    def flops_per_row(self):
        return 2 * sum(o * i for o, i in self.layer_shapes())


def packed_net(shape: NetShape, seed: int, device, requires_grad=True):
    """A PackedNet of this shape with nn.Linear-style uniform(-1/sqrt(in), 1/sqrt(in)) parameters."""
    from .mlp import NetDims, PackedNet
    gen = np.random.Generator(np.random.PCG64(seed))
    ws, bs = [], []
    for o, i in shape.layer_shapes():
        lim = 1.0 / np.sqrt(i)
        ws.append(torch.from_numpy(gen.uniform(-lim, lim, size=(o, i)).astype(np.float32)).to(device).requires_grad_(requires_grad))
        bs.append(torch.from_numpy(gen.uniform(-lim, lim, size=(o,)).astype(np.float32)).to(device).requires_grad_(requires_grad))
    return PackedNet(NetDims(shape.in_dim, shape.out_dim, shape.nb, shape.hidden, shape.act, shape.dropout), ws, bs)


def batch(seed: int, B: int, D: int, miss=0.5):
    """X float32 [B, D] standard normal, M bool [B, D] observed with probability 1 - miss, no row fully missing."""
    gen = np.random.Generator(np.random.PCG64(seed))
    X = gen.standard_normal((B, D)).astype(np.float32)
    M = gen.random((B, D)) >= miss
    dead = np.nonzero(~M.any(axis=1))[0]
    M[dead, gen.integers(0, D, size=dead.size)] = True
    return torch.from_numpy(X), torch.from_numpy(M)

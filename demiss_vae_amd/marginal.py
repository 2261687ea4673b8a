"""Marginal log-probability of (in)complete datapoints by numerical integration over a 1-d / 2-d latent grid: the
evaluation metric of the toy configs (reference: vgiwae/shared/vae_marginal_logprob.py:8-51 create_Z / create_2D_grid,
:66-90 get_marginal_logprob, :194-307).  The decoder runs once over the grid (inference forward kernel), one more
kernel pair integrates every datapoint against it."""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib as L
from .mlp import PackedNet, mlp_forward
from .objectives import ModelDims, _u8


def latent_grid(latent_dim: int, device, grid_min=None, grid_max=None, grid_steps=None):
    """(Z [Nz, latent_dim] on `device`, cell volume) with the reference's conventions: 1-d grids include both end
    points (torch.linspace) yet use (max - min) / steps as the cell; 2-d grids hold the lower-left corners
    min + i (max - min) / steps (numpy.mgrid with a step)."""
    if latent_dim == 1:
        lo, hi, steps = -15, 15, 4000
    elif latent_dim == 2:
        lo, hi, steps = -10, 10, 500
    else:
        raise NotImplementedError("Function not implemented for latent dimensionality greater than 2.")
    lo = grid_min if grid_min is not None else lo
    hi = grid_max if grid_max is not None else hi
    steps = grid_steps if grid_steps is not None else steps
    if latent_dim == 1:
        return torch.linspace(lo, hi, steps=steps).reshape(-1, 1).to(device), (hi - lo) / steps
    step = (hi - lo) / steps
    z0, z1 = np.mgrid[lo:hi:step, lo:hi:step]
    Z = torch.tensor(np.stack([z0, z1], -1)).float().reshape(-1, 2)
    return Z.to(device), step ** 2


@torch.no_grad()
def grid_marginal_logprob(dims: ModelDims, dec: PackedNet, X, M, *, compute_complete=False, grid_min=None, grid_max=None,
                          grid_steps=None):
    """log p(x_o) [B] (and log p(x) [B] when compute_complete) for X, M [B, D]."""
    lib, st = L.lib(), L.stream_ptr()
    dev = X.device
    Z, dz = latent_grid(dims.Zd, dev, grid_min, grid_max, grid_steps)
    Nz = Z.shape[0]
    dec_out, _ = mlp_forward(dec, Z.contiguous(), dims.Zd, Nz, save=False)
    logw = (-0.5 * Z * Z - 0.5 * math.log(2 * math.pi)).sum(-1) + math.log(dz)
    Xc, Mu = X.contiguous().float(), _u8(M)
    B = Xc.shape[0]
    ws = torch.empty(lib.dmv_grid_marginal_ws_floats(Nz, dims.D), dtype=torch.float32, device=dev)
    obs = torch.empty(B, dtype=torch.float32, device=dev)
    full = torch.empty(B, dtype=torch.float32, device=dev) if compute_complete else None
    L.check(lib.dmv_grid_marginal(L.GEN[dims.gen_dist], dims.D, Nz, L.ptr(dec_out), dec_out.shape[1], L.ptr(logw.contiguous()), B,
                                  L.ptr(Xc), L.ptr(Mu), L.ptr(ws), L.ptr(obs), L.ptr(full), st))
    return (obs, full) if compute_complete else obs
